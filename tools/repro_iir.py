import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, sigops_amd as so
from oracle_bridge import oracle_sink, relerr
seed = 10102
rng = np.random.default_rng(seed)
for it in range(14):
    fs = float(rng.choice([8000, 16000, 44100, 48000]))
    nch = int(rng.choice([1, 2, 3, 5, 8, 17]))
    n = int(rng.choice([50, 700, 5000, 70000, 300000]))
    dt = np.float64 if rng.random() < 0.7 else np.float32
    x = np.asfortranarray(rng.standard_normal((n, nch)).astype(dt))
    sig = so.Signal(x, fs * so.Hz)
    typ, order = int(rng.integers(0, 4)), int(rng.integers(1, 9))
    lo, hi = sorted(rng.uniform(0.02, 0.45, 2) * fs)
    if hi - lo < 0.02 * fs:
        hi = lo + 0.03 * fs
    meth = so.Butterworth(order) if rng.random() < 0.6 else so.Chebyshev1(order, 1.0)
    if typ == 0: tree = so.Filt(sig, so.Lowpass, lo * so.Hz, method=meth)
    elif typ == 1: tree = so.Filt(sig, so.Highpass, lo * so.Hz, method=meth)
    elif typ == 2: tree = so.Filt(sig, so.Bandpass, lo * so.Hz, hi * so.Hz, method=meth)
    else: tree = so.Filt(sig, so.Bandstop, lo * so.Hz, hi * so.Hz, method=meth)
    a = rng.random() < 0.3 and n > 100
    if a: tree = tree | so.After(37 * so.frames)
    m = rng.random() < 0.3
    if m: tree = so.Mix(tree, so.Signal(so.sin, ω=100 * so.Hz)) | so.Until((n // 2) * so.frames)
    if (fs, nch, n, dt.__name__, typ, order) != (48000.0, 5, 70000, 'float32', 0, 1): continue
    want = oracle_sink(tree)
    for env in ({}, {"SIGOPS_SOS_NOALIGN": "1"}, {"SIGOPS_SOS_CHUNK": "64"}, {"SIGOPS_SOS_CHUNK": "128"}, {"SIGOPS_SOS_CHUNK": "2048"}):
        for k in ("SIGOPS_SOS_NOALIGN", "SIGOPS_SOS_CHUNK"): os.environ.pop(k, None)
        os.environ.update(env)
        got = so.sink(tree)[0]
        d = np.abs(got.astype(np.float64) - want)
        print(it, "lo", lo, "after", a, "mix", m, env, "relerr", relerr(got, want), "max abs diff", d.max(), "at", np.unravel_index(d.argmax(), d.shape), "ulps(f32) max", (d / np.maximum(np.abs(want), 1e-30)).max(), flush=True)
