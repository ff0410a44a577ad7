"""Soak: sizes and window offsets around the planner's switch points (2048: periodic resampler; 4096: reference
block / one-pass; 8192: warm starts; chunk and tile multiples) for every channel count, against the oracle.
python tools/soak_thresholds.py"""
import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import sigops_amd as so
from oracle_bridge import oracle_sink, relerr
bad = 0; n = 0
rng = np.random.default_rng(99)
sizes = [s + d for s in (2048, 4096, 8192, 16384, 640 * 16, 147 * 64) for d in (-2, -1, 0, 1, 2)]
for nch in (1, 2, 3, 4, 8):
    for N in sizes:
        for dt in (np.float64, np.float32):
            x = so.Signal(np.asfortranarray(rng.standard_normal((N + 9000, nch)).astype(dt)), 44.1 * so.kHz)
            trees = {
                'resample out=N': x | so.ToFramerate(48 * so.kHz) | so.Until(N * so.frames),
                'resample in=N': x | so.Until(N * so.frames) | so.ToFramerate(48 * so.kHz),
                'filt N': x | so.Until(N * so.frames) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz),
                'fused N': x | so.Amplify(so.Signal(so.sin, ω=5 * so.Hz)) | so.Until(N * so.frames) | so.ToFramerate(48 * so.kHz),
                'window at N': x | so.Filt(so.Lowpass, 3 * so.kHz) | so.ToFramerate(48 * so.kHz) | so.After(N * so.frames) | so.Until(700 * so.frames),
                'filt window at N': x | so.Filt(so.Lowpass, 3 * so.kHz) | so.After(N * so.frames),
            }
            for name, t in trees.items():
                want = oracle_sink(t); got = so.sink(t, so.Array); n += 1
                e = relerr(got, want) if got.shape == want.shape else float('inf')
                if not e <= (2e-6 if dt == np.float32 else 1e-9):
                    print('BAD', nch, N, dt.__name__, name, got.shape, want.shape, '%.3g' % e, flush=True); bad += 1
print('checks', n, 'bad', bad)
