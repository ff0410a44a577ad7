"""Soak of K3's two-array instantiation (A2): random maps of two arrays in front of a resampler -- operation, rate pair,
channel count, lengths, windows, paddings -- against the materialised path (SIGOPS_NO_ARR2=1: bit-equal wherever A2 runs) and
the oracle.  python3 tools/soak_two_arrays.py [cases] [seed] -> one JSON line.
FILT=1: a `Filt` between the map and the resampler and SIGOPS_RSOS_MINGROUPS=1 -- the fused resampler + IIR kernel's two-array
form (k_rsos.hip, rsos_loader's A2; round 6): bit-equal to the materialised sum wherever it runs, the oracle's to 1e-8."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sigops_amd as so
from oracle_bridge import oracle_semantics, oracle_sink, relerr

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
RATES = [8.0, 11.025, 16.0, 22.05, 24.0, 32.0, 44.1, 48.0, 88.2, 96.0]
FILT = os.environ.get("FILT") == "1"
if FILT:
    os.environ["SIGOPS_RSOS_MINGROUPS"] = "1"
OPS = ["mix", "amp", "sub", "rsub"]


def steps_of(x):
    n, nch = so.nframes(x), so.nchannels(x)
    p = so.Plan(so.ToChannels(x, nch), (n, nch), np.float32 if so.sampletype(x) == np.float32 else np.float64, (1, n), False)
    names = [s["name"] for s in p.steps()]
    p.close()
    return names


bad, fused, worst, worst32 = [], 0, 0.0, 0.0
for case in range(ncases):
    fi, fo = rng.choice(RATES, 2, replace=False)
    if rng.random() < 0.6:  # (the 14-k-step family the instantiation takes, up-sampling: the lazy map is resampled as a whole)
        fi, fo = [(44.1, 48.0), (22.05, 24.0), (32.0, 48.0), (8.0, 11.025), (11.025, 16.0)][int(rng.integers(0, 5))]
    nch = int(rng.choice([8, 16, 24, 2, 4, 3, 6, 2]) if FILT else rng.choice([1, 2, 3, 4, 8, 16]))
    nx = int(rng.integers(1, 60000)) if not FILT else int(rng.integers(20000, 300000))
    ny = nx if rng.random() < (0.85 if FILT else 0.5) else int(rng.integers(1, 60000))
    dt = np.float32 if rng.random() < (0.1 if FILT else 0.35) else np.float64  # (both operands: a Float32 signal all the way, or Float64)
    x = np.asfortranarray(rng.standard_normal((nx, nch)).astype(dt))
    y = np.asfortranarray(rng.standard_normal((ny, nch)).astype(dt))
    X, Y = so.Signal(x, fi * so.kHz), so.Signal(y, fi * so.kHz)
    if rng.random() < 0.3:
        X = so.After(X, int(rng.integers(0, max(1, nx // 2))) * so.frames)
    if rng.random() < 0.3:
        Y = so.After(Y, int(rng.integers(0, max(1, ny // 2))) * so.frames)
    op = OPS[int(rng.integers(0, 4))]
    t = so.Mix(X, Y) if op == "mix" else so.Amplify(X, Y) if op == "amp" else so.OperateOn("-", X, Y) if op == "sub" else so.OperateOn("-", Y, X)
    if rng.random() < 0.3:
        t = so.Until(t, int(rng.integers(1, 50000)) * so.frames)
    if rng.random() < 0.2:
        t = so.Pad(t, so.zero) | so.Until(int(rng.integers(1, 80000)) * so.frames)
    if FILT:
        kind = int(rng.integers(0, 3))
        t = t | (so.Filt(so.Lowpass, 0.1 * fi * so.kHz) if kind == 0 else so.Filt(so.Bandstop, 0.02 * fi * so.kHz, 0.05 * fi * so.kHz) if kind == 1
                 else so.Filt(so.Highpass, 0.05 * fi * so.kHz, order=int(rng.integers(1, 9))))
    t = t | so.ToFramerate(fo * so.kHz)
    try:
        os.environ.pop("SIGOPS_NO_ARR2", None)
        names = steps_of(t)
        a = so.sink(t)[0]
        os.environ["SIGOPS_NO_ARR2"] = "1"
        b = so.sink(t)[0]
        os.environ.pop("SIGOPS_NO_ARR2", None)
        with oracle_semantics("intended"):  # (a resampled Mix operand ends after its frames: reference quirk C-7, HISTORY.md)
            w = oracle_sink(t)
    except so.ErrorException:
        continue
    one = names == (["k_rsos"] if FILT else ["k_resample_periodic"])
    fused += one
    e = float(relerr(a, w)) if a.size else 0.0
    if dt == np.float64:
        worst = max(worst, e)
    else:
        worst32 = max(worst32, e)
    # (FILT: the two runs are the same kernel -- and bit-equal -- only where the fused form took both arrays)
    if (not np.array_equal(a, b) and (one or not FILT)) or not (e <= (1e-6 if dt == np.float32 else 1e-8)):
        bad.append({"case": case, "fi": fi, "fo": fo, "nch": nch, "nx": nx, "ny": ny, "op": op, "dtype": np.dtype(dt).name, "names": names, "relerr": e,
                    "equal": bool(np.array_equal(a, b))})
print(json.dumps({"cases": ncases, "seed": seed, "one_launch": int(fused), "worst_relerr": worst, "worst_relerr_f32": worst32, "bad": bad[:10], "n_bad": len(bad)}))
