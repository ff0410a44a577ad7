"""Soak: random multi-block trees with filtered / resampled children under Append / Pad / Mix / After /
Ramp (test_gpu_fuzz._multirate_tree) against the oracle's intended-semantics mode, with and without
window aliasing.  Run on a GPU box from the repo root: python tools/tree_soak_multiblock.py SEED0 SEED1"""
import os, sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import sigops_amd as so
from oracle_bridge import oracle_semantics, oracle_sink, relerr
import test_gpu_fuzz as t

bad = 0; n = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(90000 + seed)
    for i in range(10):
        nch = int(rng.choice([1, 2, 3, 8]))
        info = {}
        try:
            tree = t._multirate_tree(rng, nch, info)
            with oracle_semantics("intended"):
                want = oracle_sink(tree)
        except so.ErrorException:  # e.g. a filter band beyond the new Nyquist rate
            continue
        n += 1
        try:
            got = so.sink(tree)[0]
            os.environ["SIGOPS_NO_WINDOW_ALIAS"] = "1"
            ref = so.sink(tree)[0]
            os.environ.pop("SIGOPS_NO_WINDOW_ALIAS")
        except Exception as e:
            os.environ.pop("SIGOPS_NO_WINDOW_ALIAS", None)
            print('ENGINE ERROR', seed, i, str(e)[:200]); bad += 1; continue
        if got.shape != want.shape or got.dtype != want.dtype:
            print('SHAPE', seed, i, got.shape, want.shape); bad += 1; continue
        if not np.array_equal(got, ref):
            print('ALIAS', seed, i, relerr(got, ref), repr(tree)[:300]); bad += 1
        tol = 2e-6 if (info.get('f32') or got.dtype == np.float32) else 1e-8
        e = relerr(got, want)
        if not e <= tol:
            print('VALUE', seed, i, '%.3g' % e, got.shape, repr(tree)[:400]); bad += 1
print('trees', n, 'bad', bad)
