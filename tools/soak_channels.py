"""Soak: every channel count 1..12, 16, 24, 33, 64, 65 through the stateful paths at 120 000 frames (the resampler
picks 8-, 4-, 2- or 1-channel tiles, the IIR packs chunks x channels into waves), against the oracle.
python tools/soak_channels.py"""
import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import sigops_amd as so
from oracle_bridge import oracle_sink, relerr
bad = 0; n = 0
rng = np.random.default_rng(7)
N = 120_000
for nch in list(range(1, 13)) + [16, 24, 33, 64, 65]:
    for dt in (np.float64, np.float32):
        x = so.Signal(np.asfortranarray(rng.standard_normal((N, nch)).astype(dt)), 44.1 * so.kHz)
        trees = {
            'resample': x | so.ToFramerate(48 * so.kHz),
            'down': x | so.ToFramerate(16 * so.kHz),
            'filt': x | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz),
            'fused': x | so.Amplify(so.Signal(so.sin, ω=5 * so.Hz)) | so.Until(N * so.frames) | so.ToFramerate(48 * so.kHz),
            'pipeline': so.Mix(so.Signal(so.sin, ω=1 * so.kHz), x) | so.Until(N * so.frames) | so.Filt(so.Lowpass, 4 * so.kHz) | so.ToFramerate(48 * so.kHz),
            'window': x | so.Filt(so.Lowpass, 3 * so.kHz) | so.ToFramerate(48 * so.kHz) | so.After(100_001 * so.frames),
            'normpower': x | so.Normpower | so.Ramp(5 * so.ms),
        }
        for name, t in trees.items():
            want = oracle_sink(t); got = so.sink(t, so.Array); n += 1
            e = relerr(got, want) if got.shape == want.shape else float('inf')
            if not e <= (2e-6 if dt == np.float32 else 1e-9):
                print('BAD', nch, dt.__name__, name, got.shape, want.shape, '%.3g' % e, flush=True); bad += 1
print('checks', n, 'bad', bad)
