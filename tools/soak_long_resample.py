"""Soak: long resamplers (>= 2 M outputs, so the threaded accumulator replay runs) at random integer
frame rates against the oracle.  python tools/soak_long_resample.py SEED0 SEED1"""
import sys, time, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import sigops_amd as so
from oracle_bridge import oracle_sink, relerr
RATES = [8000, 11025, 12000, 16000, 22050, 24000, 32000, 44100, 48000, 88200, 96000]
bad = 0; n = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(40000 + seed)
    fi, fo = (int(v) for v in rng.choice(RATES, 2, replace=False))
    nch = int(rng.choice([1, 2]))
    n_out = int(rng.integers(2_200_000, 3_500_000))
    n_in = int(n_out * fi / fo) + 10
    x = np.asfortranarray(rng.standard_normal((n_in, nch)))
    tree = so.Signal(x, fi * so.Hz) | so.ToFramerate(fo * so.Hz)
    t0 = time.perf_counter(); want = oracle_sink(tree); t1 = time.perf_counter()
    got = so.sink(tree, so.Array); t2 = time.perf_counter()
    n += 1
    e = relerr(got, want) if got.shape == want.shape else float('inf')
    print(seed, fi, fo, nch, got.shape, 'relerr %.3g' % e, 'oracle %.1fs engine(incl. plan) %.3fs' % (t1 - t0, t2 - t1), flush=True)
    if not e <= 1e-8: bad += 1
print('cases', n, 'bad', bad)
