"""Soak: every pair of the eleven audio rates of `rate_matrix.py` on Float32 signals of 4 and 8 channels (the shapes K3's
16-row and 32-row Float32 tiles serve) and 1 - 3 channels, against the oracle (1e-6), with the kernel each one ran.
python tools/soak_rates_f32.py [SEED]"""
import collections, sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import sigops_amd as so
from oracle_bridge import oracle_sink, relerr
rates = [8.0, 11.025, 16.0, 22.05, 24.0, 32.0, 44.1, 48.0, 88.2, 96.0, 192.0]
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(52000 + seed)
bad = 0; n = 0; kernels = collections.Counter(); worst = 0.0
for fi in rates:
    for fo in rates:
        if fi == fo: continue
        for nch in (8, 4, int(rng.integers(1, 4))):
            N = int(rng.integers(60_000, 120_000))
            x = np.asfortranarray((rng.standard_normal((N, nch)) * 0.5).astype(np.float32))
            tree = so.Signal(x, fi * so.kHz) | so.ToFramerate(fo * so.kHz)
            want = oracle_sink(tree)
            nout = so.nframes(tree)
            p = so.Plan(so.ToChannels(tree, nch), (nout, nch), np.float32, (1, nout), False)
            names = "+".join(s["name"] for s in p.steps()); p.close()
            kernels[names] += 1
            got = so.sink(tree)[0]
            n += 1
            e = relerr(got, want) if got.shape == want.shape and got.dtype == want.dtype else float('inf')
            worst = max(worst, e)
            if not e <= 1e-6: print('BAD', fi, fo, nch, N, names, '%.3g' % e, flush=True); bad += 1
print('cases', n, 'bad', bad, 'worst %.3g' % worst, dict(kernels))
