"""Soak: LONG signals (1-2 M frames: every persistent workgroup wraps its rings many times) through the
fused-source resampler paths -- Float32 / Float64 leaves, sine / ramp / constant gains, Mix -- and the IIR,
one-shot, against the oracle.  python tools/soak_long_fused.py SEED0 SEED1"""
import sys, time, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import sigops_amd as so
from oracle_bridge import oracle_sink, relerr
RATES = [(44100, 48000), (48000, 44100), (44100, 16000), (32000, 48000), (22050, 44100)]
bad = 0; n = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(60000 + seed)
    fi, fo = RATES[int(rng.integers(0, len(RATES)))]
    nch = int(rng.choice([1, 2, 4, 8])); dt = np.float32 if rng.random() < 0.6 else np.float64
    N = int(rng.integers(900_000, 2_000_000)) // (nch if nch > 2 else 1) * (2 if nch > 2 else 1)
    x = so.Signal(np.asfortranarray(rng.standard_normal((N, nch)).astype(dt)), fi * so.Hz)
    k = int(rng.integers(0, 6))
    if k == 0: t = x | so.Amplify(so.Signal(so.sin, ω=float(rng.uniform(1, 50)) * so.Hz)) | so.Until(N * so.frames) | so.ToFramerate(fo * so.Hz)
    elif k == 1: t = x | so.Ramp(0.5 * so.s) | so.ToFramerate(fo * so.Hz)
    elif k == 2: t = so.Mix(so.Signal(so.sin, ω=440 * so.Hz), x) | so.Until(N * so.frames) | so.ToFramerate(fo * so.Hz)
    elif k == 3: t = x | so.Amplify(0.37) | so.ToFramerate(fo * so.Hz) | so.Filt(so.Lowpass, 0.2 * min(fi, fo) * so.Hz)
    elif k == 4: t = x | so.Amplify(so.Signal(so.sin, ω=5 * so.Hz)) | so.Until(N * so.frames) | so.Filt(so.Highpass, 0.01 * fi * so.Hz) | so.ToFramerate(fo * so.Hz)
    else: t = x | so.ToEltype(np.float64) | so.Amplify(so.Signal(so.cos, ω=2 * so.Hz)) | so.Until(N * so.frames) | so.ToFramerate(fo * so.Hz)
    try:
        want = oracle_sink(t)
    except so.ErrorException:
        continue
    errs = [relerr(so.sink(t, so.Array), want) for _ in range(2)]
    n += 1
    tol = 2e-6 if want.dtype == np.float32 else 1e-8
    flag = '' if max(errs) <= tol else '  <-- BAD'
    print(seed, fi, fo, nch, dt.__name__, N, 'case', k, want.shape, ['%.3g' % e for e in errs], flag, flush=True)
    if flag: bad += 1
print('cases', n, 'bad', bad)
