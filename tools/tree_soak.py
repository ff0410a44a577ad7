"""Soak: random operator trees (tests/test_gpu_fuzz.py::_random_tree) through the HIP engine against
the oracle, many seeds.  Run on a GPU box from the repo root: python tools/tree_soak.py SEED0 SEED1"""
import sys, numpy as np, traceback
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import sigops_amd as so
from oracle_bridge import oracle_sink, relerr
import test_gpu_fuzz as t
bad = 0; n = 0; nerr = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(1000 + seed)
    for i in range(25):
        nch = int(rng.choice([1, 2, 3])); fs = float(rng.choice([50, 100, 8000])) * so.Hz
        info = {}
        tree = t._random_tree(rng, nch, fs, int(rng.integers(1, 6)), info)
        if len(sys.argv) > 3: print('tree', seed, i, flush=True)
        if so.nframes(tree) == 0: continue  # (empty sinks of After trees can loop forever in the reference's block loop, and in the oracle with it)
        try:
            want = oracle_sink(tree)
        except Exception as e:
            nerr += 1
            try:
                so.sink(tree); print('ENGINE ACCEPTED what oracle rejected', seed, i, str(e)[:100]); bad += 1
            except Exception:
                pass
            continue
        n += 1
        try:
            got = so.sink(tree)[0]
        except Exception as e:
            print('ENGINE ERROR', seed, i, str(e)[:200]); bad += 1; continue
        if got.shape != want.shape or got.dtype != want.dtype:
            print('SHAPE', seed, i, got.shape, want.shape, got.dtype, want.dtype); bad += 1; continue
        if want.size:
            if np.isfinite(want).all():
                e = relerr(got, want); tol = 2e-6 if info.get('f32') else 1e-9
                if not e <= tol:
                    d = np.abs(got - want); d = np.where(np.isnan(d), np.inf, d)
                    bf = np.argwhere(~(d <= 1e-6 * max(1.0, float(np.abs(want).max()))))
                    print('VALUE', seed, i, e, info, 'bad frames', bf[:3].tolist(), '..', bf[-2:].tolist(), len(bf), 'nonfinite', int((~np.isfinite(got)).sum()))
                    bad += 1
            elif not np.array_equal(np.isfinite(got), np.isfinite(want)):
                print('NONFINITE', seed, i); bad += 1
print('trees', n, 'oracle-rejected', nerr, 'bad', bad)
