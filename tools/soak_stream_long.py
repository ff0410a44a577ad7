"""Soak: so.stream over long trees (1-4 M frames), random block sizes, against the one-shot sink.
python tools/soak_stream_long.py SEED0 SEED1"""
import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import sigops_amd as so
from oracle_bridge import relerr
bad = 0; n = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(70000 + seed)
    nch = int(rng.choice([1, 2, 8])); dt = np.float32 if rng.random() < 0.4 else np.float64
    N = int(rng.integers(1_000_000, 4_000_000)) // (4 if nch == 8 else 1)
    x = so.Signal(np.asfortranarray(rng.standard_normal((N, nch)).astype(dt)), 44.1 * so.kHz)
    k = int(rng.integers(0, 4))
    if k == 0: t = so.Mix(so.Signal(so.sin, ω=1 * so.kHz), x) | so.Until(N * so.frames) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.ToFramerate(48 * so.kHz)
    elif k == 1: t = x | so.Amplify(so.Signal(so.sin, ω=5 * so.Hz)) | so.Until(N * so.frames) | so.ToFramerate(48 * so.kHz)
    elif k == 2: t = so.Append(x | so.Filt(so.Lowpass, 4 * so.kHz) | so.Ramp(10 * so.ms), x | so.Until((N // 3) * so.frames) | so.Amplify(0.5), x | so.After((N // 2) * so.frames) | so.Filt(so.Highpass, 300 * so.Hz))
    else: t = x | so.Filt(so.Lowpass, 4 * so.kHz) | so.ToFramerate(16 * so.kHz) | so.Pad(so.zero) | so.Until((N // 2) * so.frames)
    whole = so.sink(t, so.Array)
    bsz = int(rng.choice([48000, 100_003, 441_000, 1_000_000]))
    got = np.concatenate(list(so.stream(t, bsz, so.Array)), axis=0); n += 1
    e = relerr(got, whole) if got.shape == whole.shape else float('inf')
    ok = e <= (1e-6 if dt == np.float32 else 1e-11)
    print(seed, nch, dt.__name__, N, k, bsz, '%.3g' % e, '' if ok else '  <-- BAD', flush=True); bad += not ok
print('streams', n, 'bad', bad)
