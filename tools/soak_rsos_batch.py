"""Soak of the batched one-pass launch (k_rsos_batch, round 6): random `Append`s of filtered scenes -- scene count, channel
count, lengths, filter kind and order per scene, an optional mixed-in sine, ramps, Float32 -- with the batch forced
(SIGOPS_RSOS_BATCH=1) against the scenes' own oracle results (the reference's Append never leaves a filtered child longer than
one filter block, quirk C-7: the expected value is the concatenation) and against the three-pass batch (=0).
python3 tools/soak_rsos_batch.py [cases] [seed] -> one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sigops_amd as so
from oracle_bridge import oracle_sink, relerr

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)


def steps_of(x, dt):
    n, nch = so.nframes(x), so.nchannels(x)
    p = so.Plan(so.ToChannels(x, nch), (n, nch), dt, (1, n), False)
    names = [s["name"] for s in p.steps()]
    p.close()
    return names


bad, batched, worst, worst32 = [], 0, 0.0, 0.0
for case in range(ncases):
    nsc = int(rng.integers(2, 24))
    nch = int(rng.choice([1, 2, 2, 4, 8, 3]))
    dt = np.float32 if rng.random() < 0.25 else np.float64
    same_filter = rng.random() < 0.5
    kind0, ord0 = int(rng.integers(0, 4)), int(rng.integers(1, 9))
    mix = rng.random() < 0.4 and dt == np.float64
    ramp = rng.random() < 0.4
    kids = []
    for k in range(nsc):
        n = int(np.exp(rng.uniform(np.log(3000), np.log(150000))))
        x = so.Signal(np.asfortranarray((rng.standard_normal((n, nch)) * rng.uniform(0.1, 2)).astype(dt)), 44.1 * so.kHz)
        if mix:
            x = so.Mix(so.Signal(so.sin, 44.1 * so.kHz, ω=float(rng.uniform(100, 5000)) * so.Hz) | so.Until(n * so.frames), x)
        kind, order = (kind0, ord0) if same_filter else (int(rng.integers(0, 4)), int(rng.integers(1, 9)))
        f = (so.Filt(so.Lowpass, 3 * so.kHz, order=order) if kind == 0 else so.Filt(so.Highpass, 0.8 * so.kHz, order=order) if kind == 1
             else so.Filt(so.Bandpass, 1 * so.kHz, 4 * so.kHz, order=min(order, 6)) if kind == 2 else so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz, order=min(order, 6)))
        x = x | f
        if ramp:
            x = x | so.Ramp(5 * so.ms)
        kids.append(x)
    tree = so.Append(*kids)
    try:
        os.environ["SIGOPS_RSOS_BATCH"] = "1"
        names = steps_of(tree, dt)
        a = so.sink(tree, so.Array)
        os.environ["SIGOPS_RSOS_BATCH"] = "0"
        b = so.sink(tree, so.Array)
        w = np.concatenate([oracle_sink(k) for k in kids])
    except so.ErrorException:
        continue
    finally:
        os.environ.pop("SIGOPS_RSOS_BATCH", None)
    batched += "k_rsos_batch" in names
    e, e2 = float(relerr(a, w)), float(relerr(a, b))
    if dt == np.float64:
        worst = max(worst, e)
    else:
        worst32 = max(worst32, e)
    tol = 2e-6 if dt == np.float32 else 1e-8
    if not (e <= tol and e2 <= tol) or not np.isfinite(a).all():
        bad.append({"case": case, "nsc": nsc, "nch": nch, "dtype": np.dtype(dt).name, "mix": bool(mix), "ramp": bool(ramp), "names": names, "relerr": e, "vs_three": e2})
print(json.dumps({"cases": ncases, "seed": seed, "batched": int(batched), "worst_relerr": worst, "worst_relerr_f32": worst32, "bad": bad[:10], "n_bad": len(bad)}))
