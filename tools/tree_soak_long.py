"""Soak: the random operator trees of tests/test_gpu_fuzz.py::_random_tree with leaves 1000 x longer (5 000 ...
400 000 frames instead of 5 ... 400): the same shapes, but every kernel runs many workgroups and wraps its rings.
Filtered children are now longer than the reference's 4096-frame block, so the oracle runs in its
intended-semantics mode (a filtered child ends after nframes(x) frames; quirk C-7).
python tools/tree_soak_long.py SEED0 SEED1 [SCALE [multirate]]"""
import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import sigops_amd as so
from oracle_bridge import oracle_semantics, oracle_sink, relerr
import test_gpu_fuzz as t
SCALE = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
MULTIRATE = len(sys.argv) > 4 and sys.argv[4] == 'multirate'


class LongRng:
    """default_rng whose leaf-length draws (integers(5, 400)) are scaled"""
    def __init__(self, seed): self.g = np.random.default_rng(seed)
    def integers(self, lo, hi=None, *a, **k):
        if (lo, hi) == (5, 400): return self.g.integers(5 * SCALE, 400 * SCALE)
        return self.g.integers(lo, hi, *a, **k)
    def __getattr__(self, name): return getattr(self.g, name)


bad = 0; n = 0; nerr = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = LongRng(3000 + seed)
    for i in range(5):
        nch = int(rng.choice([1, 2, 3, 8])); fsv = float(rng.choice([8000, 44100])); fs = fsv * so.Hz
        info = {}
        tree = t._random_tree(rng, nch, fs, int(rng.integers(1, 5)), info)
        if MULTIRATE:  # a resampler (and more) on top, as in tools/tree_soak_multirate.py
            fo = float(rng.choice([r for r in (8000, 16000, 44100, 48000) if r != fsv]))
            k = int(rng.integers(0, 4))
            if so.nframes(tree) in (None,) or so.signals.isknowninf(so.nframes(tree)):
                tree = tree | so.Until(150000 * so.frames)
            if k == 0: tree = tree | so.ToFramerate(fo * so.Hz)
            elif k == 1: tree = tree | so.ToFramerate(fo * so.Hz) | so.After(7 * so.frames) | so.Ramp(5 * so.frames)
            elif k == 2: tree = tree | so.ToFramerate(fo * so.Hz) | so.Filt(so.Lowpass, 0.2 * min(fo, fsv) * so.Hz)
            else: tree = tree | so.ToEltype(np.float32 if rng.random() < 0.5 else np.float64) | so.ToFramerate(fo * so.Hz)
            info['f32'] = info.get('f32', False) or tree.dtype == np.float32
        N = so.nframes(tree)
        if N == 0 or N > 6_000_000: continue
        try:
            with oracle_semantics("intended"):
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
            print('SHAPE', seed, i, got.shape, want.shape); bad += 1; continue
        if want.size and np.isfinite(want).all():
            e = relerr(got, want); tol = 2e-6 if (info.get('f32') or got.dtype == np.float32) else (1e-8 if MULTIRATE else 1e-9)
            if not e <= tol:
                d = np.abs(got.astype(float) - want.astype(float)); bf = np.argwhere(d.max(axis=1) > 1e-6 * max(1.0, float(np.abs(want).max()))).ravel()
                print('VALUE', seed, i, '%.3g' % e, info, got.shape, 'bad frames', (int(bf[0]), int(bf[-1]), len(bf)) if len(bf) else None); bad += 1
print('trees', n, 'oracle-rejected', nerr, 'bad', bad)
