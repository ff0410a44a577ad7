"""Re-create one tree of tools/tree_soak.py (kind a) or tools/tree_soak_multirate.py (kind b) and show
both sides.  python tools/soak_repro.py a SEED I [a SEED I ...]"""
import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import sigops_amd as so
from sigops_amd import lowering
from oracle_bridge import oracle_sink, relerr
import test_gpu_fuzz as t
KN = ["ARRAY", "CONST", "FUNC", "UNTIL", "AFTER", "PAD", "APPEND", "RAMP", "MAP", "FILT", "RESAMPLE", "NORM"]


def gen(kind, seed, i):
    if kind == 'a':
        rng = np.random.default_rng(1000 + seed)
        for k in range(i + 1):
            nch = int(rng.choice([1, 2, 3])); fs = float(rng.choice([50, 100, 8000])) * so.Hz
            info = {}
            tree = t._random_tree(rng, nch, fs, int(rng.integers(1, 6)), info)
        return tree, info
    import importlib.util
    src = open('tools/tree_soak_multirate.py').read().split("bad = 0; n = 0; nerr = 0")[0]
    ns = {}
    exec(compile(src, 'soakb', 'exec'), ns)
    rng = np.random.default_rng(5000 + seed)
    for k in range(i + 1):
        nch = int(rng.choice([1, 2, 3]))
        info = {}
        try:
            tree = ns['tree2'](rng, nch, int(rng.integers(0, 4)), info)
        except Exception as e:
            tree = None
    return tree, info


args = [a for a in sys.argv[1:] if not a.startswith('--')]
for j in range(0, len(args), 3):
    kind, seed, i = args[j], int(args[j + 1]), int(args[j + 2])
    tree, info = gen(kind, seed, i)
    print('=====', kind, seed, i, info)
    try:
        L = lowering.lower(tree)
        for k, nd in enumerate(L.nodes):
            print('  %2d %-8s kids=%s n=%s nch=%s dt=%s fs=%g i0=%s i1=%s i2=%s l0=%s d0=%g' % (k, KN[nd.kind], [nd.children[q] for q in range(nd.n_children)],
                  nd.nframes, nd.nch, nd.dtype, nd.fs, nd.i0, nd.i1, nd.i2, nd.l0, nd.d0))
    except Exception as e:
        print('  lowering failed:', e)
    want = got = None
    if '--no-engine' in sys.argv:
        so.sink = lambda *a, **k: (_ for _ in ()).throw(RuntimeError('engine not run'))
    try:
        want = oracle_sink(tree); print('  oracle:', want.shape, want.dtype)
    except Exception as e:
        print('  oracle error:', str(e)[:200])
    try:
        got = so.sink(tree)[0]; print('  engine:', got.shape, got.dtype)
    except Exception as e:
        print('  engine error:', str(e)[:200])
    if want is not None and got is not None and want.shape == got.shape and want.size:
        d = np.abs(got.astype(np.float64) - want.astype(np.float64))
        d = np.where(np.isnan(d), np.inf, d)
        jx = np.unravel_index(np.argmax(d), d.shape)
        print('  relerr', relerr(got, want), 'max diff at', jx, got[jx], want[jx], 'nonfinite got/want', (~np.isfinite(got)).sum(), (~np.isfinite(want)).sum())
        bad = np.argwhere(~(d <= 1e-6 * max(1.0, np.nanmax(np.abs(want)))))
        print('  differing frames:', bad[:5].tolist(), '...', bad[-3:].tolist(), len(bad))
