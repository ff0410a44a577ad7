"""Soak: random multi-rate trees (ToFramerate / mixed frame rates / channel ops on top of
test_gpu_fuzz._random_tree) against the oracle; trees of the shape of divergence C-7 are
skipped.  Run on a GPU box from the repo root: python tools/tree_soak_multirate.py SEED0 SEED1"""
import sys, numpy as np, traceback
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import sigops_amd as so
from oracle_bridge import oracle_sink, relerr
import test_gpu_fuzz as t
RATES = [4000.0, 6000.0, 8000.0, 12000.0]

def tree2(rng, nch, depth, info):
    fs = float(rng.choice(RATES)) * so.Hz
    x = t._random_tree(rng, nch, fs, depth, info)
    k = int(rng.choice([0, 3, 4, 5, 6, 0, 5]))
    n = so.nframes(x)
    if k == 0:
        return x | so.ToFramerate(float(rng.choice(RATES)) * so.Hz)
    if k == 1:
        y = t._random_tree(rng, nch, float(rng.choice(RATES)) * so.Hz, int(rng.integers(0, 3)), info)
        return so.Mix(x, y)
    if k == 2:
        y = t._random_tree(rng, nch, float(rng.choice(RATES)) * so.Hz, int(rng.integers(0, 3)), info)
        return so.Append(x, y)
    if k == 3:
        return x | so.ToChannels(1)
    if k == 4 and x.nch == 1:
        return x | so.ToChannels(int(rng.integers(2, 4)))
    if k == 5:
        return x | so.ToFramerate(float(rng.choice(RATES)) * so.Hz) | so.After(7 * so.frames) | so.Ramp(5 * so.frames)
    return x | so.ToEltype(np.float32 if rng.random() < 0.5 else np.float64)

bad = 0; n = 0; nerr = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(5000 + seed)
    for i in range(20):
        nch = int(rng.choice([1, 2, 3]))
        info = {}
        try:
            tree = tree2(rng, nch, int(rng.integers(0, 4)), info)
        except Exception as e:
            print('GEN', seed, i, str(e)[:120]); continue
        try:
            if so.nframes(tree) == 0: continue
        except Exception:
            pass
        from sigops_amd import lowering as _lw
        try:
            _L = _lw.lower(tree)
            _rs = [k for k, nd in enumerate(_L.nodes) if nd.kind in (9, 10)]
            _c7 = False
            for k, nd in enumerate(_L.nodes):
                if _rs and k > min(_rs) and (nd.kind in (5, 6) or (nd.kind == 8 and nd.n_children >= 2 and sum(_L.nodes[nd.children[j]].kind != 1 for j in range(nd.n_children)) >= 2)):
                    _c7 = True
            if _c7:
                continue
        except Exception:
            pass
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
        if want.size and np.isfinite(want).all():
            e = relerr(got, want); tol = 2e-6 if (info.get('f32') or got.dtype == np.float32) else 1e-8
            if not e <= tol:
                from sigops_amd import lowering
                KN = ["ARRAY","CONST","FUNC","UNTIL","AFTER","PAD","APPEND","RAMP","MAP","FILT","RESAMPLE","NORM"]
                L = lowering.lower(tree)
                desc = ' '.join(f"{k}:{KN[nd.kind]}({','.join(str(nd.children[j]) for j in range(nd.n_children))})n{nd.nframes}" + (f"@{nd.fs:g}" if nd.kind in (0,2,10) else '') for k, nd in enumerate(L.nodes))
                d = np.abs(got - want); j = np.unravel_index(np.argmax(d), d.shape)
                print('VALUE', seed, i, '%.3g' % e, got.shape, 'maxdiff at', j, '|', desc[:700]); bad += 1
print('trees', n, 'oracle-rejected', nerr, 'bad', bad)
