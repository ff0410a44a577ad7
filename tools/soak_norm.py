"""Soak: `Normpower` over host arrays, device tensors (planar rows with pitches and offsets of their own, interleaved), windows
of them, and the outputs of filters and resamplers -- row lengths from one frame to a few million, 1 ... 9 channels -- against
the oracle.  Float32 signals over arrays: bit-equal (the reduction follows the oracle's order: K4); others
1e-6 / 1e-12 (behind a filter or resampler 1e-9).  Also that the copying path (`SIGOPS_NORM_COPY=1`) gives the same bits.   python tools/soak_norm.py SEED0 SEED1"""
import os, sys, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import sigops_amd as so
from oracle_bridge import oracle_sink, relerr
bad = 0; n = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(81000 + seed)
    nch = int(rng.integers(1, 10))
    N = int(np.exp(rng.uniform(0, np.log(3e6 / nch)))) + int(rng.integers(0, 3))
    f32 = rng.random() < 0.6
    ndt, tdt = (np.float32, torch.float32) if f32 else (np.float64, torch.float64)
    host = np.asfortranarray((rng.standard_normal((N, nch)) * rng.uniform(0.01, 3)).astype(ndt))
    base = torch.zeros((nch + 1, N + 13), dtype=tdt, device="cuda")
    off = int(rng.integers(0, 9))
    base[1:, off:off + N] = torch.from_numpy(np.ascontiguousarray(host.T)).cuda()
    leaves = {'host': host, 'planar pitch': base[1:, off:off + N].t(), 'interleaved': torch.from_numpy(np.ascontiguousarray(host)).cuda()}
    a = int(rng.integers(0, max(1, N // 3))); b = int(rng.integers(a + 1, N + 1))
    kind = int(rng.integers(0, 6))
    def pipe(s):
        if kind == 0: return s | so.Normpower
        if kind == 1: return s | so.After(a * so.frames) | so.Until((b - a) * so.frames) | so.Normpower
        if kind == 2: return s | so.Normpower | so.Amplify(-6 * so.dB) | so.After(a * so.frames)
        if kind == 3: return s | so.Filt(so.Lowpass, 2 * so.kHz) | so.Normpower
        if kind == 4: return s | so.ToFramerate(48 * so.kHz) | so.Normpower
        return so.Mix(s | so.Normpower, s | so.Until(b * so.frames) | so.Normpower | so.Amplify(0.5))
    try:
        want = oracle_sink(pipe(so.Signal(host, 44.1 * so.kHz)))
    except so.ErrorException:
        continue
    exact = f32 and kind in (0, 1) and want.dtype == np.float32
    for name, leaf in leaves.items():
        tree = pipe(so.Signal(leaf, 44.1 * so.kHz))
        got = so.sink(tree)[0] if name == 'host' else None
        if got is None:
            out = torch.full((want.shape[1], want.shape[0] + 5), float("nan"), dtype=torch.float32 if want.dtype == np.float32 else torch.float64, device="cuda")
            so.sink_into(out.t()[:want.shape[0]], tree)
            got = out[:, :want.shape[0]].t().cpu().numpy()
        os.environ["SIGOPS_NORM_COPY"] = "1"
        try:
            cp = so.sink(pipe(so.Signal(host, 44.1 * so.kHz)))[0] if name == 'host' else None
        finally:
            del os.environ["SIGOPS_NORM_COPY"]
        n += 1
        e = relerr(got, want) if want.size else 0.0
        ok = got.dtype == want.dtype and got.shape == want.shape and (np.array_equal(got, want) if exact else e <= (1e-6 if want.dtype == np.float32 else (1e-9 if kind in (3, 4) else 1e-12)))
        if cp is not None and not np.array_equal(cp, got): ok = False
        if not ok: print('BAD', seed, name, kind, nch, N, a, b, ndt.__name__, '%.3g' % e, flush=True); bad += 1
print('checks', n, 'bad', bad)
