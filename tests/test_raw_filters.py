"""Raw filter objects, `Filt(x, h)` (reference src/filters.jl:89-97: RawFilterFn, resolve_filter =
DF2TFilter(h); test/runtests.jl:365-368): ZeroPoleGain from `digitalfilter`, SecondOrderSections,
Biquad, small PolynomialRatio and FIR coefficient vectors.  CPU part: the oracle against SciPy and
the reference's own equality; GPU part: the engine against the oracle."""
import numpy as np
import pytest
from scipy import signal as sps

import sigops_amd as so
from sigops_amd import Signal, Filt, Highpass, Lowpass, Bandpass, Chebyshev1, Butterworth, Hz, digitalfilter
from cases import F, rng
from oracle_bridge import oracle_sink, relerr


def _trees():
    x = F(rng(3).standard_normal((9000, 3)))
    sig = Signal(x, 100 * Hz)
    named = sig | Filt(Highpass, 8 * Hz, method=Chebyshev1(5, 1))
    zpk = digitalfilter(Highpass(8, fs=100), Chebyshev1(5, 1))
    raw = sig | Filt(zpk)
    return x, sig, named, raw, zpk


def test_digitalfilter_matches_scipy():
    zpk = digitalfilter(Highpass(8, fs=100), Chebyshev1(5, 1))
    z, p, k = sps.cheby1(5, 1, 8, "highpass", fs=100, output="zpk")
    assert np.allclose(np.sort_complex(zpk.p), np.sort_complex(p), atol=1e-12)
    assert np.allclose(np.sort_complex(zpk.z), np.sort_complex(z), atol=1e-12) and abs(zpk.k - k) < 1e-12 * abs(k)
    bp = digitalfilter(Bandpass(5, 20, fs=100), Butterworth(3))
    z, p, k = sps.butter(3, [5, 20], "bandpass", fs=100, output="zpk")
    assert np.allclose(np.sort_complex(bp.p), np.sort_complex(p), atol=1e-12) and abs(bp.k - k) < 1e-12 * abs(k)


def test_custom_filter_interface_equals_the_named_form():
    """runtests.jl:365-368: `Array(high) == Array(high4)` -- exact equality"""
    x, sig, named, raw, zpk = _trees()
    assert np.array_equal(oracle_sink(named), oracle_sink(raw))


def test_raw_objects_against_scipy():
    x, sig, named, raw, zpk = _trees()
    want = sps.sosfilt(sps.zpk2sos(zpk.z, zpk.p, zpk.k), x, axis=0)
    assert relerr(oracle_sink(raw), want) < 1e-10
    sos = sps.butter(4, 10, "lowpass", fs=100, output="sos")
    y = oracle_sink(sig | Filt(so.SecondOrderSections(sos, 1.0)))
    assert relerr(y, sps.sosfilt(sos, x, axis=0)) < 1e-12
    b, a = sps.butter(2, 10, "lowpass", fs=100)
    y = oracle_sink(sig | Filt(so.PolynomialRatio(b, a)))
    assert relerr(y, sps.lfilter(b, a, x, axis=0)) < 1e-11
    y = oracle_sink(sig | Filt(so.Biquad(b[0], b[1], b[2], a[1], a[2])))
    assert relerr(y, sps.lfilter(b, a, x, axis=0)) < 1e-11


@pytest.mark.parametrize("ntaps", [1, 2, 31, 64, 301])
def test_fir_coefficients(ntaps):
    x, sig = _trees()[:2]
    h = sps.firwin(ntaps, 0.3) if ntaps > 2 else np.array([0.5, -0.25][:ntaps])
    y = oracle_sink(Filt(sig, h))
    assert y.shape == x.shape
    assert relerr(y, sps.lfilter(h, [1.0], x, axis=0)) < 1e-12
    # curried form and an explicit PolynomialRatio(h, [1]) are the same filter
    assert np.array_equal(oracle_sink(sig | Filt(h)), y)
    assert np.array_equal(oracle_sink(sig | Filt(so.PolynomialRatio(h, [1.0]))), y)


def test_fir_blocksize_invariance_and_after_state():
    """the FIR's history crosses block boundaries like any filter state (runtests.jl:353-362)"""
    x, sig = _trees()[:2]
    h = sps.firwin(45, 0.2)
    a = oracle_sink(Filt(sig, h), blocksize=4096)
    b = oracle_sink(Filt(sig, h), blocksize=64)
    assert np.array_equal(a, b)
    c = oracle_sink(Filt(sig, h) | so.After(100 * so.frames))
    assert np.array_equal(c, a[100:])


# ------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_gpu_custom_filter_interface():
    x, sig, named, raw, zpk = _trees()
    a, b = so.sink(named)[0], so.sink(raw)[0]
    assert np.array_equal(a, b)  # runtests.jl:368
    assert relerr(b, oracle_sink(raw)) < 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("ntaps,n,nch,dt", [(31, 9000, 3, np.float64), (301, 50000, 2, np.float64), (1, 5000, 1, np.float64),
                                           (64, 300000, 8, np.float64), (45, 20000, 4, np.float32), (2, 100, 2, np.float64)])
def test_gpu_fir_coefficients(ntaps, n, nch, dt):
    x = F(rng(5).standard_normal((n, nch)).astype(dt))
    h = sps.firwin(ntaps, 0.3) if ntaps > 2 else np.array([0.5, -0.25][:ntaps])
    tree = Signal(x, 8000 * Hz) | Filt(h)
    got = so.sink(tree)[0]
    want = oracle_sink(tree)
    assert got.shape == want.shape and got.dtype == want.dtype == dt
    assert relerr(got, want) < (1e-11 if dt == np.float64 else 2e-7)
    assert relerr(got, sps.lfilter(h, [1.0], x.astype(np.float64), axis=0)) < (1e-11 if dt == np.float64 else 2e-7)


@pytest.mark.gpu
def test_gpu_fir_then_resample_and_objects():
    x = F(rng(6).standard_normal((30000, 2)))
    h = sps.firwin(63, 0.25)
    sos = sps.butter(4, 1000, "lowpass", fs=8000, output="sos")
    tree = (Signal(x, 8000 * Hz) | Filt(h) | Filt(so.SecondOrderSections(sos)) | so.ToFramerate(12000 * Hz)
            | Filt(so.Biquad(0.2, 0.4, 0.2, -0.3, 0.1)))
    assert relerr(so.sink(tree)[0], oracle_sink(tree)) < 1e-10
