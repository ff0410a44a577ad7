"""Soak: one long signal cut along time into W ranges (sharding.shard_time), each evaluated on this GPU from its
warm start, reassembled against the one-shot sink.  python tools/soak_time_shards.py SEED0 SEED1"""
import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import sigops_amd as so
from sigops_amd import sharding
from oracle_bridge import relerr
bad = 0; n = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(10000 + seed)
    nch = int(rng.choice([1, 2, 8])); dt = np.float32 if rng.random() < 0.4 else np.float64
    N = int(rng.integers(1_000_000, 4_000_000)); W = int(rng.choice([2, 3, 5, 8]))
    x = so.Signal(np.asfortranarray(rng.standard_normal((N, nch)).astype(dt)), 44.1 * so.kHz)
    k = int(rng.integers(0, 3))
    if k == 0: t = so.Mix(so.Signal(so.sin, ω=1 * so.kHz), x) | so.Until(N * so.frames) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.ToFramerate(48 * so.kHz)
    elif k == 1: t = x | so.Amplify(so.Signal(so.sin, ω=5 * so.Hz)) | so.Until(N * so.frames) | so.ToFramerate(48 * so.kHz)
    else: t = x | so.Filt(so.Lowpass, 4 * so.kHz) | so.ToFramerate(16 * so.kHz) | so.Filt(so.Highpass, 50 * so.Hz)
    whole = so.sink(t, so.Array)
    parts = []
    align = int(rng.choice([1, 160, 2560]))
    for r in range(W):
        slab, start = sharding.sink_time_sharded(t, rank=r, world=W, gather=False, align=align)
        parts.append(slab.cpu().numpy())
    got = np.concatenate(parts); n += 1
    e = relerr(got, whole) if got.shape == whole.shape else float('inf')
    ok = e <= (1e-6 if dt == np.float32 else 1e-11)
    print(seed, nch, dt.__name__, N, W, k, '%.3g' % e, '' if ok else '  <-- BAD', flush=True); bad += not ok
print('signals', n, 'bad', bad)
