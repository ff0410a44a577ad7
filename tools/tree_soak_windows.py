"""Soak: windows (`After(a) |> Until(m)`) of random trees with long filtered / resampled children -- the
planner's warm starts and windowed aliasing -- against the same window of the oracle's result and of the
engine's own one-shot result.  python tools/tree_soak_windows.py SEED0 SEED1"""
import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import sigops_amd as so
from oracle_bridge import oracle_semantics, oracle_sink, relerr

RATES = [8000.0, 12000.0, 16000.0, 44100.0, 48000.0]


def tree(rng, nch, info):
    def leaf(fs, lo=20000, hi=70000):
        n = int(rng.integers(lo, hi))
        dt = np.float64 if rng.random() < 0.8 else np.float32
        info["f32"] = info.get("f32", False) or dt == np.float32
        return so.Signal(np.asfortranarray(rng.standard_normal((n, nch)).astype(dt)), fs * so.Hz)

    def filt(x, fs):
        k = int(rng.integers(0, 4))
        if k == 0:
            return x | so.Filt(so.Lowpass, float(rng.uniform(0.05, 0.4)) * fs * so.Hz)
        if k == 1:
            return x | so.Filt(so.Highpass, float(rng.uniform(0.02, 0.3)) * fs * so.Hz)
        if k == 2:
            return x | so.Filt(so.Bandstop, 0.05 * fs * so.Hz, 0.2 * fs * so.Hz)
        return x | so.Filt(so.Bandpass, 0.05 * fs * so.Hz, 0.2 * fs * so.Hz, order=int(rng.integers(3, 11)))

    fs = float(rng.choice(RATES))
    op = int(rng.integers(0, 7))
    if op == 0:
        return filt(leaf(fs), fs)
    fi = float(rng.choice([r for r in RATES if r != fs]))
    if op == 1:
        return leaf(fi) | so.ToFramerate(fs * so.Hz)
    if op == 2:
        return filt(leaf(fi), fi) | so.ToFramerate(fs * so.Hz)
    if op == 3:
        return so.Mix(so.Signal(so.sin, ω=0.01 * fs * so.Hz), filt(leaf(fs), fs)) | so.Ramp(100 * so.frames)
    if op == 4:
        return so.Append(filt(leaf(fs), fs), leaf(fi) | so.ToFramerate(fs * so.Hz))
    if op == 5:
        return filt(filt(leaf(fs), fs) | so.Amplify(0.7), fs)
    x0 = leaf(fi)
    x = x0 | so.Amplify(so.Signal(so.sin, ω=3 * so.Hz))
    return filt(x | so.Until(so.nframes(x0) * so.frames) | so.ToFramerate(fs * so.Hz), fs)


bad = 0; n = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(70000 + seed)
    for i in range(6):
        nch = int(rng.choice([1, 2, 3, 8]))
        info = {}
        try:
            t = tree(rng, nch, info)
            N = so.nframes(t)
            with oracle_semantics("intended"):
                want = oracle_sink(t)
        except so.ErrorException as e:
            continue
        whole = so.sink(t, so.Array)
        tol = 2e-6 if (info.get('f32') or whole.dtype == np.float32) else 1e-8
        for j in range(3):
            a = int(rng.integers(N // 4, N - 10))
            m = int(rng.integers(1, N - a + 1)) if rng.random() < 0.5 else N - a
            w = t | so.After(a * so.frames) | so.Until(m * so.frames)
            n += 1
            try:
                got = so.sink(w, so.Array)
            except Exception as e:
                print('ENGINE ERROR', seed, i, j, str(e)[:200]); bad += 1; continue
            if got.shape != (m, want.shape[1]):
                print('SHAPE', seed, i, j, got.shape, m); bad += 1; continue
            e1, e2 = relerr(got, want[a:a + m]), relerr(got, whole[a:a + m])
            if not (e1 <= tol and e2 <= (1e-6 if got.dtype == np.float32 else 1e-10)):
                print('VALUE', seed, i, j, 'vs oracle %.3g vs one-shot %.3g' % (e1, e2), (a, m, N), repr(t)[:300]); bad += 1
print('windows', n, 'bad', bad)
