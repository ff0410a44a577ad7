"""Soak of the fused kernel's Float32 ring (RsSos::ring32): random Float32 signals -- arrays, windows, appended pieces,
paddings -- under a random IIR, optionally resampled, with SIGOPS_RSOS_MINGROUPS=1 (the one-pass / fused forms whenever they
apply) against the widening loader (SIGOPS_RSOS_NO_RING32=1: bit-equal), the three-pass / two-kernel forms and the oracle
(1e-6).  python3 tools/soak_f32_ring.py [cases] [seed] -> one JSON line."""
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


def steps_of(x):
    n, nch = so.nframes(x), so.nchannels(x)
    p = so.Plan(so.ToChannels(x, nch), (n, nch), np.float32, (1, n), False)
    names = [s["name"] for s in p.steps()]
    p.close()
    return names


def sink_with(tree, **env):
    old = {k: os.environ.get(k) for k in env}
    for k, v in env.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    try:
        return so.sink(tree)[0]
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


bad, ring, worst = [], 0, 0.0
for case in range(ncases):
    fi = float(rng.choice(RATES))
    nch = int(rng.choice([1, 2, 3, 4, 8, 16]))
    n = int(rng.integers(1, 300_000))
    X = so.Signal(np.asfortranarray(rng.standard_normal((n, nch)).astype(np.float32)), fi * so.kHz)
    t = X
    r = rng.random()
    if r < 0.2:
        t = so.After(t, int(rng.integers(0, max(1, n // 2))) * so.frames)
    elif r < 0.4:
        m = int(rng.integers(1, 100_000))
        t = so.Append(t, so.Signal(np.asfortranarray(rng.standard_normal((m, nch)).astype(np.float32)), fi * so.kHz))
    elif r < 0.5:
        t = so.Pad(t, so.zero) | so.Until(int(rng.integers(1, 400_000)) * so.frames)
    nyq = fi / 2
    kind = int(rng.integers(0, 4))
    lo = float(rng.uniform(0.02, 0.4)) * nyq
    hi = lo + float(rng.uniform(0.05, 0.5)) * (nyq - lo)
    order = int(rng.integers(1, 6))
    f = (so.Filt(so.Lowpass, lo * so.kHz, order=order) if kind == 0 else so.Filt(so.Highpass, lo * so.kHz, order=order) if kind == 1
         else so.Filt(so.Bandpass, lo * so.kHz, hi * so.kHz, order=min(order, 3)) if kind == 2 else so.Filt(so.Bandstop, lo * so.kHz, hi * so.kHz, order=min(order, 3)))
    t = t | f
    if rng.random() < 0.5:
        fo = float(rng.choice([r_ for r_ in RATES if r_ != fi]))
        t = t | so.ToFramerate(fo * so.kHz)
    try:
        os.environ["SIGOPS_RSOS_MINGROUPS"] = "1"
        names = steps_of(t)
        a = sink_with(t, SIGOPS_RSOS_MINGROUPS=1, SIGOPS_RSOS_NO_RING32=None)
        b = sink_with(t, SIGOPS_RSOS_MINGROUPS=1, SIGOPS_RSOS_NO_RING32=1)
        c = sink_with(t, SIGOPS_RSOS_MINGROUPS=None, SIGOPS_NO_RSOS=1)
        os.environ.pop("SIGOPS_RSOS_MINGROUPS", None)
        with oracle_semantics("intended"):
            w = oracle_sink(t)
    except so.ErrorException:
        os.environ.pop("SIGOPS_RSOS_MINGROUPS", None)
        continue
    ring += "k_rsos" in names
    e = float(relerr(a, w)) if a.size else 0.0
    ec = float(relerr(a, c)) if a.size else 0.0
    worst = max(worst, e)
    # (round 6: resampled Float32 signals run the Float32 MFMA inside the fused kernel -- another rounding than the widening loader's
    #  Float64 products; bit-equality with it holds, and is checked, under SIGOPS_RSOS_NO_F32MFMA=1: profiles/r06/soak_r06_others.txt)
    same = np.array_equal(a, b) if os.environ.get("SIGOPS_RSOS_NO_F32MFMA") else float(relerr(a, b)) <= 3e-7
    if a.dtype != np.float32 or not same or not (e <= 1e-6) or not (ec <= 3e-7):
        bad.append({"case": case, "fi": fi, "nch": nch, "n": n, "names": names, "relerr": e, "vs_unfused": ec, "equal": bool(np.array_equal(a, b))})
print(json.dumps({"cases": ncases, "seed": seed, "with_k_rsos": int(ring), "worst_relerr": worst, "bad": bad[:8], "n_bad": len(bad)}))
