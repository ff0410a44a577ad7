"""Seeded random resampler geometries (rate pairs, channel counts, lengths, sample types, fused or
filtered sources, cut outputs) through the HIP path against the oracle: the three resampler
kernels (K3 MFMA ring, K3r row-tiled, thread-per-output fallback) are chosen by geometry, so a
sweep over geometry is what exercises their selection and edge handling."""
import numpy as np
import pytest

import sigops_amd as so
from oracle_bridge import oracle_sink, relerr

pytestmark = pytest.mark.gpu

RATES = [8000, 11025, 12000, 16000, 22050, 24000, 32000, 44100, 48000, 88200, 96000]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_resampler_geometries(seed):
    rng = np.random.default_rng(seed)
    for _ in range(12):
        fi, fo = rng.choice(RATES, 2, replace=False)
        nch = int(rng.choice([1, 2, 3, 4, 6, 8, 16]))
        n = int(rng.integers(3000, 30000))
        dt = np.float64 if rng.random() < 0.7 else np.float32
        x = np.asfortranarray(rng.standard_normal((n, nch)).astype(dt))
        kind = int(rng.integers(0, 4))
        sig = so.Signal(x, float(fi) * so.Hz)
        if kind == 1:
            sig = sig | so.Amplify(so.Signal(so.sin, ω=7 * so.Hz)) | so.Until(n * so.frames)
        elif kind == 2:
            sig = sig | so.Ramp(10 * so.ms)
        elif kind == 3:
            sig = sig | so.Filt(so.Lowpass, float(min(fi, fo)) * 0.2 * so.Hz)
        tree = sig | so.ToFramerate(float(fo) * so.Hz)
        if rng.random() < 0.3:
            tree = tree | so.Until(int(rng.integers(2100, 2600)) * so.frames)
        want = oracle_sink(tree)
        got = so.sink(tree)[0]
        assert got.shape == want.shape and got.dtype == want.dtype
        # (a Float32 leaf keeps the tolerance of Float32 even where the result is Float64: `Amplify(x32, sin)` is
        #  resampled as Float32 (reference rewrite, src/mapsignal.jl:46-64; out eltype float(T), src/filters.jl:105) and
        #  multiplied afterwards, and where the resampler's Float64 value sits within 1e-12 -- the drift of DSP.jl's
        #  accumulated alpha against the closed form -- of a Float32 rounding boundary, engine and oracle round apart:
        #  4-9 in 10 000 elements, each 6e-8 off (tools/soak_kernels.py seed 10317: 1.04e-9 norm-wise))
        tol = 1e-6 if dt == np.float32 else 1e-9
        assert relerr(got, want) <= tol, (fi, fo, nch, n, dt.__name__, kind)


@pytest.mark.parametrize("seed", [1, 2])
def test_random_iir_geometries(seed):
    """K2 (chunked exact IIR): filter type / order / design method, channel count, lengths from one
    block to many chunks, sample type, `After` (state carried through the cut) and a `Mix` on top"""
    rng = np.random.default_rng(seed)
    for _ in range(14):
        fs = float(rng.choice([8000, 16000, 44100, 48000]))
        nch = int(rng.choice([1, 2, 3, 5, 8, 17]))
        n = int(rng.choice([50, 700, 5000, 70000, 300000]))
        dt = np.float64 if rng.random() < 0.7 else np.float32
        x = np.asfortranarray(rng.standard_normal((n, nch)).astype(dt))
        sig = so.Signal(x, fs * so.Hz)
        typ, order = int(rng.integers(0, 4)), int(rng.integers(1, 9))
        lo, hi = sorted(rng.uniform(0.02, 0.45, 2) * fs)
        if hi - lo < 0.02 * fs:
            hi = lo + 0.03 * fs
        meth = so.Butterworth(order) if rng.random() < 0.6 else so.Chebyshev1(order, 1.0)
        if typ == 0:
            tree = so.Filt(sig, so.Lowpass, lo * so.Hz, method=meth)
        elif typ == 1:
            tree = so.Filt(sig, so.Highpass, lo * so.Hz, method=meth)
        elif typ == 2:
            tree = so.Filt(sig, so.Bandpass, lo * so.Hz, hi * so.Hz, method=meth)
        else:
            tree = so.Filt(sig, so.Bandstop, lo * so.Hz, hi * so.Hz, method=meth)
        if rng.random() < 0.3 and n > 100:
            tree = tree | so.After(37 * so.frames)
        if rng.random() < 0.3:
            tree = so.Mix(tree, so.Signal(so.sin, ω=100 * so.Hz)) | so.Until((n // 2) * so.frames)
        want = oracle_sink(tree)
        got = so.sink(tree)[0]
        assert got.shape == want.shape
        tol = 1e-6 if got.dtype == np.float32 else 1e-8
        assert relerr(got, want) <= tol, (fs, nch, n, dt.__name__, typ, order)


@pytest.mark.parametrize("seed", [11, 12])
def test_random_fused_sine_gains(seed):
    """The resampler's fused `Amplify(x, Signal(sin, ...))` source (gain ring + two-level sine
    evaluation in the TWO kernel variant): random rates, channel groupings (8-, 4-channel tiles),
    frequencies, phases, source offsets and cuts — including tiles at the edges and sources too
    short for a single fast tile, which fall back to the general staging path."""
    rng = np.random.default_rng(seed)
    for _ in range(14):
        fi, fo = rng.choice(RATES, 2, replace=False)
        nch = int(rng.choice([1, 2, 4, 8, 12, 16]))
        n = int(rng.choice([300, 2500, 9000, 40000]))
        x = np.asfortranarray(rng.standard_normal((n + 64, nch)))
        gen = {}
        if rng.random() < 0.8:
            gen["omega"] = float(rng.choice([0.5, 5.0, 440.0, 3000.0])) * so.Hz
        if rng.random() < 0.5:
            gen["phi"] = float(rng.uniform(0, 6.0))
        sig = so.Signal(x, float(fi) * so.Hz)
        off = int(rng.choice([0, 1, 17, 64]))
        if off:
            sig = sig | so.After(off * so.frames)
        sig = sig | so.Amplify(so.Signal(so.sin, **gen)) | so.Until(n * so.frames)
        tree = sig | so.ToFramerate(float(fo) * so.Hz)
        if rng.random() < 0.3:
            tree = tree | so.After(11 * so.frames)
        want = oracle_sink(tree)
        got = so.sink(tree)[0]
        assert got.shape == want.shape and got.dtype == want.dtype
        assert relerr(got, want) <= 1e-9, (fi, fo, nch, n, gen, off)


def _random_tree(rng, nch, fs, depth, info):
    """A random finite signal over the structural / pointwise operators (the planner's piece
    algebra + K1), with an occasional stateful stage (Filt, Normpower) underneath."""
    def leaf():
        k = int(rng.integers(0, 4))
        if k <= 1:
            n = int(rng.integers(5, 400))
            c = nch if rng.random() < 0.7 else 1
            dt = np.float64 if rng.random() < 0.75 else np.float32
            info["f32"] = info.get("f32", False) or dt == np.float32
            return so.Signal(np.asfortranarray(rng.standard_normal((n, c)).astype(dt)), fs)
        fn = so.sin if k == 2 else so.cos
        return so.Signal(fn, fs, ω=float(rng.uniform(0.5, 40)) * so.Hz) | so.Until(int(rng.integers(5, 400)) * so.frames)

    if depth == 0:
        return leaf()
    x = _random_tree(rng, nch, fs, depth - 1, info)
    n = so.nframes(x)
    if n is None or n < 0:  # length not known before evaluation: pin it
        x = x | so.Until(150 * so.frames)
        n = so.nframes(x)
    op = int(rng.integers(0, 13))
    if op == 0:
        return x | so.Until(int(rng.integers(0, n + 1)) * so.frames)
    if op == 1:
        return x | so.After(int(rng.integers(0, n + 1)) * so.frames)
    if op == 2:
        pad = [so.zero, so.one, so.lastframe, 2.5, so.zero, so.one, so.lastframe, -0.5, so.cycle, so.mirror][int(rng.integers(0, 10))]
        if n == 0 and pad in (so.lastframe, so.cycle, so.mirror):
            pad = so.zero
        return so.Pad(x, pad) | so.Until((n + int(rng.integers(1, 3 * n + 5))) * so.frames)
    if op == 3 and n >= 4:
        return so.Ramp(x, int(rng.integers(1, n // 2 + 1)) * so.frames)
    if op == 4 and n >= 2:
        return (so.RampOn if rng.random() < 0.5 else so.RampOff)(x, int(rng.integers(1, n + 1)) * so.frames)
    if op == 5:
        return so.Amplify(x, float(rng.uniform(-12, 6)) * so.dB)
    if op == 6:
        return so.Mix(x, float(rng.uniform(-1, 1)))
    y = _random_tree(rng, nch, fs, int(rng.integers(0, depth)), info)
    if op == 7:
        return so.Mix(x, y)
    if op == 8:
        return so.Amplify(x, y)
    if op == 9:
        return so.Append(x, y)
    if op == 10 and n >= 2:
        return so.FadeTo(x, y, int(rng.integers(1, max(1, min(n, so.nframes(y) or 1)) + 1)) * so.frames)
    if op == 11 and n >= 30:
        return so.Filt(x, so.Lowpass, float(rng.uniform(0.05, 0.4)) * fs)
    if op == 12 and n >= 1:
        return so.Normpower(x)
    return so.Append(y, x)


@pytest.mark.parametrize("seed", range(6))
def test_random_operator_trees(seed):
    """Random trees of Until / After / Pad / Ramp / Amplify / Mix / Append / FadeTo (+ Filt,
    Normpower) over arrays, generators and numbers against the oracle: lengths and dtypes equal,
    values within the parity bound; a tree the oracle rejects must be rejected by the engine too."""
    rng = np.random.default_rng(1000 + seed)
    for _ in range(25):
        nch = int(rng.choice([1, 2, 3]))
        fs = float(rng.choice([50, 100, 8000])) * so.Hz
        info = {}
        tree = _random_tree(rng, nch, fs, int(rng.integers(1, 6)), info)
        try:
            want = oracle_sink(tree)
        except Exception:
            with pytest.raises(Exception):
                so.sink(tree)
            continue
        got = so.sink(tree)[0]
        assert got.shape == want.shape and got.dtype == want.dtype, repr(tree)[:300]
        if want.size:
            ok = np.isfinite(want).all()
            # Float32 anywhere in the tree (even under a Float64 result): Float32 bound
            tol = 1e-6 if info.get("f32") else 1e-9
            if ok:
                assert relerr(got, want) <= tol, repr(tree)[:400]
            else:
                assert np.array_equal(np.isfinite(got), np.isfinite(want)), repr(tree)[:400]


def _multirate_tree(rng, nch, info):
    """Multi-block (> 4096 frames) filtered / resampled children directly under Append / Pad / Mix /
    After / Ramp -- the shapes where the reference never leaves a filtered child (quirk C-7) and the
    engine implements the documented meaning instead."""
    rates = [4000.0, 6000.0, 8000.0, 12000.0]

    def leaf(fs, lo=3000, hi=12000):
        n = int(rng.integers(lo, hi))
        dt = np.float64 if rng.random() < 0.8 else np.float32
        info["f32"] = info.get("f32", False) or dt == np.float32
        return so.Signal(np.asfortranarray(rng.standard_normal((n, nch)).astype(dt)), fs * so.Hz)

    def stateful(fs):
        """a child that ends in a stateful stage, at rate fs"""
        k = int(rng.integers(0, 3))
        if k == 0:
            fi = float(rng.choice([r for r in rates if r != fs]))
            return leaf(fi) | so.ToFramerate(fs * so.Hz)
        if k == 1:
            return leaf(fs) | so.Filt(so.Lowpass, float(rng.uniform(0.05, 0.4)) * fs * so.Hz)
        fi = float(rng.choice([r for r in rates if r != fs]))
        return leaf(fi) | so.Filt(so.Highpass, 0.1 * fi * so.Hz) | so.ToFramerate(fs * so.Hz)

    fs = float(rng.choice(rates))
    x = stateful(fs)
    n = so.nframes(x)
    op = int(rng.integers(0, 7))
    if op == 0:
        return so.Append(x, stateful(fs))
    if op == 1:
        return so.Append(x, leaf(fs, 500, 6000), stateful(fs))
    if op == 2:
        pad = [so.zero, so.one, so.lastframe, 2.5][int(rng.integers(0, 4))]
        return so.Pad(x, pad) | so.Until((n + int(rng.integers(100, 6000))) * so.frames)
    if op == 3:
        return so.Mix(x, stateful(fs))  # different lengths: the shorter one is padded
    if op == 4:
        return x | so.After(int(rng.integers(1, n // 2)) * so.frames) | so.Ramp(50 * so.frames)
    if op == 5:
        app = so.Append(x, stateful(fs))
        return (so.Amplify(app, so.Signal(so.sin, ω=3 * so.Hz)) | so.Until(so.nframes(app) * so.frames)
                | so.ToFramerate(float(rng.choice(rates)) * so.Hz))
    return so.Append(stateful(fs) | so.Normpower, x)


@pytest.mark.parametrize("seed", range(4))
def test_random_multirate_multiblock_trees(seed):
    """automated parity for the multi-rate, multi-block shapes of divergence C-7, against
    the oracle's intended-semantics mode (a filtered child ends after nframes(x) frames)."""
    from oracle_bridge import oracle_semantics

    rng = np.random.default_rng(7000 + seed)
    for _ in range(10):
        nch = int(rng.choice([1, 2, 3]))
        info = {}
        tree = _multirate_tree(rng, nch, info)
        with oracle_semantics("intended"):
            want = oracle_sink(tree)
        got = so.sink(tree)[0]
        assert got.shape == want.shape and got.dtype == want.dtype, repr(tree)[:300]
        tol = 1e-6 if (info.get("f32") or got.dtype == np.float32) else 1e-8
        assert relerr(got, want) <= tol, repr(tree)[:400]
