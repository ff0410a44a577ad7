"""A seeded, time-boxed slice of the round-2 soak, inside the suite the driver runs (VERDICT r2 item 6:
"~155 000 soak trees are claims, not records").  Five families, every case against the oracle:

* 400 random operator trees (generators of test_gpu_fuzz.py, other seeds; errors compared too),
* 100 multi-rate multi-block trees with filtered / resampled children under Append / Pad / Mix / After
  (oracle in intended-semantics mode, each also bit-equal with window aliasing off),
* 60 windows `After(a) |> Until(m)` of long stateful trees (warm starts),
* 40 filter designs across the IIR geometry choices (orders 1-12, Butterworth / Chebyshev I, cut-offs
  0.0005-0.49 fs, FIR, cascades; includes ill-conditioned ones that take the exact-order kernel),
* 16 long (> 512 tiles per workgroup ring wrap) fused-source resamplers in Float32 and Float64,
* 84 checks over channel counts 1 ... 65 and 540 at sizes around the planner's switch points.

These replace tools/tree_soak.py, tree_soak_multiblock.py, tree_soak_windows.py, soak_filters.py,
soak_long_fused.py, soak_channels.py and soak_thresholds.py (reference behaviour under test: the whole of SURVEY.md section 8(a))."""
import os

import numpy as np
import pytest

import sigops_amd as so
import test_gpu_fuzz as fz
from oracle_bridge import oracle_semantics, oracle_sink, relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(16))
def test_soak_operator_trees(seed):
    rng = np.random.default_rng(31000 + seed)
    for i in range(25):
        nch = int(rng.choice([1, 2, 3]))
        fs = float(rng.choice([50, 100, 8000])) * so.Hz
        info = {}
        tree = fz._random_tree(rng, nch, fs, int(rng.integers(1, 6)), info)
        if so.nframes(tree) == 0:
            continue  # (empty sinks under After: tests/test_gpu_fences.py)
        try:
            want = oracle_sink(tree)
        except Exception:
            with pytest.raises(Exception):
                so.sink(tree)
            continue
        got = so.sink(tree)[0]
        assert got.shape == want.shape and got.dtype == want.dtype, (seed, i, repr(tree)[:300])
        if not want.size:
            continue
        if np.isfinite(want).all():
            assert relerr(got, want) <= (1e-6 if info.get("f32") else 1e-9), (seed, i, repr(tree)[:400])
        else:
            assert np.array_equal(np.isfinite(got), np.isfinite(want)), (seed, i, repr(tree)[:400])


@pytest.mark.parametrize("seed", range(10))
def test_soak_multirate_multiblock_trees(seed):
    rng = np.random.default_rng(93000 + seed)
    for i in range(10):
        nch = int(rng.choice([1, 2, 3, 8]))
        info = {}
        try:
            tree = fz._multirate_tree(rng, nch, info)
            with oracle_semantics("intended"):
                want = oracle_sink(tree)
        except so.ErrorException:  # e.g. a filter band beyond the new Nyquist rate
            continue
        got = so.sink(tree)[0]
        os.environ["SIGOPS_NO_WINDOW_ALIAS"] = "1"
        try:
            ref = so.sink(tree)[0]
        finally:
            os.environ.pop("SIGOPS_NO_WINDOW_ALIAS", None)
        assert got.shape == want.shape and got.dtype == want.dtype, (seed, i)
        assert np.array_equal(got, ref), (seed, i, "window aliasing changed the result")
        assert relerr(got, want) <= (1e-6 if (info.get("f32") or got.dtype == np.float32) else 1e-8), (seed, i, repr(tree)[:400])


RATES = [8000.0, 12000.0, 16000.0, 44100.0, 48000.0]


def _stateful_tree(rng, nch, info):
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


@pytest.mark.parametrize("seed", range(10))
def test_soak_windows_of_stateful_trees(seed):
    rng = np.random.default_rng(73000 + seed)
    for i in range(2):
        nch = int(rng.choice([1, 2, 3, 8]))
        info = {}
        try:
            t = _stateful_tree(rng, nch, info)
            N = so.nframes(t)
            with oracle_semantics("intended"):
                want = oracle_sink(t)
        except so.ErrorException:
            continue
        whole = so.sink(t, so.Array)
        tol = 1e-6 if (info.get("f32") or whole.dtype == np.float32) else 1e-8
        for j in range(3):
            a = int(rng.integers(N // 4, N - 10))
            m = int(rng.integers(1, N - a + 1)) if rng.random() < 0.5 else N - a
            got = so.sink(t | so.After(a * so.frames) | so.Until(m * so.frames), so.Array)
            assert got.shape == (m, want.shape[1]), (seed, i, j)
            assert relerr(got, want[a:a + m]) <= tol, (seed, i, j, a, m, N, repr(t)[:300])
            assert relerr(got, whole[a:a + m]) <= (1e-6 if got.dtype == np.float32 else 1e-10), (seed, i, j, a, m, N)


@pytest.mark.parametrize("seed", range(40))
def test_soak_filter_designs(seed):
    rng = np.random.default_rng(16000 + seed)
    nch = int(rng.choice([1, 2, 3, 8]))
    dt = np.float32 if rng.random() < 0.3 else np.float64
    fs = float(rng.choice([8000, 44100, 96000]))
    N = int(rng.integers(60_000, 300_000)) // (2 if nch == 8 else 1)
    x = so.Signal(np.asfortranarray(rng.standard_normal((N, nch)).astype(dt)), fs * so.Hz)
    order = int(rng.integers(1, 13))
    method = so.Butterworth(order) if rng.random() < 0.6 else so.Chebyshev1(order, float(rng.uniform(0.1, 3.0)))
    f1 = float(10 ** rng.uniform(np.log10(0.0005), np.log10(0.2))) * fs
    f2 = min(0.49 * fs, f1 * float(rng.uniform(1.2, 8.0)))
    k = int(rng.integers(0, 6))
    try:
        if k == 0:
            t = x | so.Filt(so.Lowpass, f1 * so.Hz, method=method)
        elif k == 1:
            t = x | so.Filt(so.Highpass, f1 * so.Hz, method=method)
        elif k == 2:
            t = x | so.Filt(so.Bandpass, f1 * so.Hz, f2 * so.Hz, method=method)
        elif k == 3:
            t = x | so.Filt(so.Bandstop, f1 * so.Hz, f2 * so.Hz, method=method)
        elif k == 4:
            h = rng.standard_normal(int(rng.integers(3, 300)))
            t = so.Filt(x, h / np.abs(h).sum())
        else:
            t = x | so.Filt(so.Lowpass, f2 * so.Hz, method=method) | so.Filt(so.Highpass, f1 * so.Hz, method=method)
        want = oracle_sink(t)
    except so.ErrorException:
        pytest.skip("the design is rejected by the host layer (same for the oracle)")
    if not np.isfinite(want).all() or np.abs(want).max() > 1e6:
        pytest.skip("an unstable design: nothing to compare")
    tol = 1e-6 if dt == np.float32 else 1e-8  # (north_star: 1e-6 for Float32; observed: profiles/r04/relerr_maxima.json)
    got = so.sink(t, so.Array)
    assert relerr(got, want) <= tol, (k, order, method, f1 / fs)
    a = int(rng.integers(N // 2, N - 1000))
    m = int(rng.integers(500, N - a))
    w = so.sink(t | so.After(a * so.frames) | so.Until(m * so.frames), so.Array)
    if np.abs(want[a:a + m]).max() > 0:
        assert relerr(w, want[a:a + m]) <= tol, (k, order, method, f1 / fs, a, m)


LONG_RATES = [(44100, 48000), (48000, 44100), (44100, 16000), (32000, 48000), (22050, 44100)]


@pytest.mark.parametrize("seed", range(16))
def test_soak_long_fused_resamplers(seed):
    """signals long enough for every persistent workgroup to wrap its tile and gain rings many times (the
    round-2 GA race needed > 512 tiles and was invisible to every shorter test)"""
    rng = np.random.default_rng(61000 + seed)
    fi, fo = LONG_RATES[int(rng.integers(0, len(LONG_RATES)))]
    nch = int(rng.choice([4, 8]))
    dt = np.float32 if seed % 2 == 0 else np.float64
    N = int(rng.integers(400_000, 600_000))
    x = so.Signal(np.asfortranarray(rng.standard_normal((N, nch)).astype(dt)), fi * so.Hz)
    k = seed % 5
    if k == 0:
        t = x | so.Amplify(so.Signal(so.sin, ω=float(rng.uniform(1, 50)) * so.Hz)) | so.Until(N * so.frames) | so.ToFramerate(fo * so.Hz)
    elif k == 1:
        t = x | so.Ramp(0.5 * so.s) | so.ToFramerate(fo * so.Hz)
    elif k == 2:
        t = so.Mix(so.Signal(so.sin, ω=440 * so.Hz), x) | so.Until(N * so.frames) | so.ToFramerate(fo * so.Hz)
    elif k == 3:
        t = x | so.Amplify(0.37) | so.ToFramerate(fo * so.Hz) | so.Filt(so.Lowpass, 0.2 * min(fi, fo) * so.Hz)
    else:
        t = x | so.ToEltype(np.float64) | so.Amplify(so.Signal(so.cos, ω=2 * so.Hz)) | so.Until(N * so.frames) | so.ToFramerate(fo * so.Hz)
    want = oracle_sink(t)
    tol = 1e-6 if want.dtype == np.float32 else 1e-8
    for _ in range(2):  # (twice: the ring protocols are timing dependent)
        assert relerr(so.sink(t, so.Array), want) <= tol, (fi, fo, nch, dt.__name__, N, k)


@pytest.mark.parametrize("nch", [1, 2, 3, 5, 6, 7, 9, 12, 16, 24, 33, 65])
def test_soak_channel_counts(nch):
    """every stateful path at channel counts on both sides of the tile widths (the resampler picks 8-, 4-, 2- or
    1-channel tiles, the IIR packs chunks x channels into waves); replaces tools/soak_channels.py"""
    rng = np.random.default_rng(700 + nch)
    N = 30_000
    dt = np.float32 if nch % 3 == 0 else np.float64
    x = so.Signal(np.asfortranarray(rng.standard_normal((N, nch)).astype(dt)), 44.1 * so.kHz)
    trees = {
        "resample": x | so.ToFramerate(48 * so.kHz),
        "down": x | so.ToFramerate(16 * so.kHz),
        "filt": x | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz),
        "fused": x | so.Amplify(so.Signal(so.sin, ω=5 * so.Hz)) | so.Until(N * so.frames) | so.ToFramerate(48 * so.kHz),
        "pipeline": so.Mix(so.Signal(so.sin, ω=1 * so.kHz), x) | so.Until(N * so.frames) | so.Filt(so.Lowpass, 4 * so.kHz) | so.ToFramerate(48 * so.kHz),
        "window": x | so.Filt(so.Lowpass, 3 * so.kHz) | so.ToFramerate(48 * so.kHz) | so.After(25_001 * so.frames),
        "normpower": x | so.Normpower | so.Ramp(5 * so.ms),
    }
    for name, t in trees.items():
        want = oracle_sink(t)
        got = so.sink(t, so.Array)
        assert got.shape == want.shape, (nch, name)
        assert relerr(got, want) <= (1e-6 if dt == np.float32 else 1e-9), (nch, dt.__name__, name)


@pytest.mark.parametrize("nch", [1, 2, 8])
@pytest.mark.parametrize("base", [2048, 4096, 8192, 16384, 640 * 16, 147 * 64])
def test_soak_sizes_around_the_planner_switch_points(nch, base):
    """sizes and window offsets around the planner's switch points (2048: periodic resampler; 4096: the reference's
    block; 8192: warm starts; chunk and tile multiples), +-2 frames; replaces tools/soak_thresholds.py"""
    rng = np.random.default_rng(99 + nch + base)
    for d in (-2, -1, 0, 1, 2):
        N = base + d
        dt = np.float32 if (d == 1) else np.float64
        x = so.Signal(np.asfortranarray(rng.standard_normal((N + 9000, nch)).astype(dt)), 44.1 * so.kHz)
        trees = {
            "resample out=N": x | so.ToFramerate(48 * so.kHz) | so.Until(N * so.frames),
            "resample in=N": x | so.Until(N * so.frames) | so.ToFramerate(48 * so.kHz),
            "filt N": x | so.Until(N * so.frames) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz),
            "fused N": x | so.Amplify(so.Signal(so.sin, ω=5 * so.Hz)) | so.Until(N * so.frames) | so.ToFramerate(48 * so.kHz),
            "window at N": x | so.Filt(so.Lowpass, 3 * so.kHz) | so.ToFramerate(48 * so.kHz) | so.After(N * so.frames) | so.Until(700 * so.frames),
            "filt window at N": x | so.Filt(so.Lowpass, 3 * so.kHz) | so.After(N * so.frames),
        }
        for name, t in trees.items():
            want = oracle_sink(t)
            got = so.sink(t, so.Array)
            assert got.shape == want.shape, (nch, N, name)
            assert relerr(got, want) <= (1e-6 if dt == np.float32 else 1e-9), (nch, N, dt.__name__, name)


@pytest.mark.parametrize("block", range(4))
def test_soak_round3_paths(block):
    """tools/soak_round3.py, ten seeds per block: batches of independent filters into device results at random row
    offsets and strides, rates without a period at every channel-group width through windows, big maps over several
    arrays before and after their background specialisation (3 700 checks of it ran clean on the final round-3 code)"""
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "soak_round3.py")
    spec = importlib.util.spec_from_file_location("soak_round3", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    checks, bad = mod.run(2000 + 10 * block, 2010 + 10 * block)
    assert checks >= 20 and bad == 0


def test_soak_round4_kernels():
    """tools/soak_arb.py, a slice: the persistent kernel of rates without a period against the tiled one and the oracle
    over random rates / channel counts / lengths / windows, Float32 array sources of the fused resampler + IIR kernel
    against K3-GA + K2 and the oracle (600 + 75 cases ran clean on the final round-4 code: profiles/r04/soak_arb.txt)"""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_arb.py"), "4040", "48"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "'on_k_resample_arb':" in r.stdout
