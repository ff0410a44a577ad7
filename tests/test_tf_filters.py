"""`filt(b, a, x[, si])` and `Filt(x, PolynomialRatio(b, a))` of any order (reference src/filters.jl:68-95 hand DSP.jl's
direct-form recurrence the coefficients): the engine factors the two polynomials into second-order sections
(`so_tf_to_sos`, csrc/design.cpp) and runs them as a `Filt` node; an initial state enters through the direct form's
zero-input response (`so_tf_zero_input`).  CPU part: the oracle's restatement of the direct form against
scipy.signal.lfilter (the same recurrence), the factoring against it, and the gate for ill-conditioned polynomials.
GPU part: the engine against the oracle."""
import numpy as np
import pytest
from scipy import signal as sps

import sigops_amd as so
from sigops_amd import signals as S
from sigops_amd import Signal, Hz
from cases import F, rng
from oracle_bridge import oracle_filt, relerr

WELL = {
    "butter3": lambda: sps.butter(3, 0.2),
    "butter5": lambda: sps.butter(5, 0.3),
    "butter8": lambda: sps.butter(8, 0.25),          # eight zeros at -1: a cluster the factoring has to find
    "butter4_bp": lambda: sps.butter(4, [0.2, 0.45], "bandpass"),
    "cheby7_hp": lambda: sps.cheby1(7, 1, 0.3, "highpass"),
    "ellip6": lambda: sps.ellip(6, 1, 60, 0.25),
    "delay3": lambda: (np.array([0, 0, 0, 1.0, 0.5]), np.array([1, -0.9])),
    "fir_heavy": lambda: (np.array([1, 2, 3, 4, 5, 6.0]), np.array([1, -0.3])),
    "a0_is_2": lambda: (np.array([1.0, 0, 0, 0.5]), np.array([2.0, 0, 0, -1.0])),
    "double_pole": lambda: (np.array([1.0, 0.3]), np.convolve([1, -1.2, 0.52], [1, -1.2, 0.52])),
    "integrator": lambda: (np.array([1.0, 1.0, 0.0, 0.25]), np.array([1.0, -1.0, 0.0, 0.0])),
}


@pytest.mark.parametrize("name", sorted(WELL))
def test_oracle_direct_form_is_lfilter(name):
    b, a = WELL[name]()
    x = F(rng(3).standard_normal((4000, 2)))
    assert relerr(oracle_filt(b, a, x), sps.lfilter(b, a, x, axis=0)) < 1e-13
    ord_ = max(len(a), len(b)) - 1
    zi = rng(4).standard_normal((ord_, 2))
    want, _ = sps.lfilter(b, a, x, axis=0, zi=zi)
    assert relerr(oracle_filt(b, a, x, zi), want) < 1e-13


@pytest.mark.parametrize("name", sorted(WELL))
def test_factored_cascade_is_the_direct_form(name):
    b, a = WELL[name]()
    sos, gain, resid = S.tf_to_sos(b, a)
    assert resid < 1e-9
    x = rng(5).standard_normal(20000)
    want = oracle_filt(b, a, F(x[:, None]))[:, 0]
    got = gain * sps.sosfilt(sos, x)
    assert relerr(got, want) < 1e-9
    assert len(sos) <= (max(len(a), len(b)) - 1 + 1) // 2 + 1


def test_ill_conditioned_polynomials_are_reported():
    """a 12th-order low-pass at 1 % of Nyquist written as one polynomial: the direct form itself has lost the filter, no
    factoring agrees with it, and the residual says so (the engine then refuses `Filt` and `filt` takes the host path)"""
    b, a = sps.butter(12, 0.02)
    assert S.tf_to_sos(b, a)[2] > 1e-3
    with pytest.raises(S.ErrorException, match="ill-conditioned"):
        S._raw_filter(S.PolynomialRatio(b, a))


def test_zero_input_response_is_cut_where_it_has_decayed():
    b, a = sps.butter(4, 0.2)
    zi = sps.lfilter_zi(b, a)
    z = S.tf_zero_input(b, a, zi, 100000)
    want, _ = sps.lfilter(b, a, np.zeros(1000), zi=zi)
    assert 100 < len(z) < 1000 and relerr(z, want[: len(z)]) < 1e-14 and np.abs(want[len(z):]).max() < 1e-22
    assert len(S.tf_zero_input([1.0, 1.0], [1.0, -1.0], [0.5], 5000)) == 5000  # an integrator never forgets
    with pytest.raises(S.ErrorException, match="initial state"):
        S.tf_zero_input(b, a, zi[:2], 10)
    with pytest.raises(S.ErrorException, match="nonzero"):
        S.tf_to_sos([1.0], [0.0, 1.0])


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(WELL))
def test_gpu_filt_of_any_order(name):
    b, a = WELL[name]()
    x = F(rng(6).standard_normal((30000, 3)))
    sig = Signal(x, 1000 * Hz) | so.Amplify(0.5)
    got, fs = so.filt(b, a, sig)
    assert fs == 1000.0 and got.dtype == np.float64
    tol = 1e-9
    assert relerr(got, oracle_filt(b, a, sig)) < tol
    ord_ = max(len(a), len(b)) - 1
    zi = rng(7).standard_normal(ord_)
    assert relerr(so.filt(b, a, sig, zi)[0], oracle_filt(b, a, sig, zi)) < tol      # one state for every channel
    zi2 = rng(8).standard_normal((ord_, 3))
    assert relerr(so.filt(b, a, sig, zi2)[0], oracle_filt(b, a, sig, zi2)) < tol    # a state per channel
    y = so.sink(so.Filt(Signal(x, 1000 * Hz), so.PolynomialRatio(b, a)))[0]
    assert relerr(y, oracle_filt(b, a, x)) < tol


@pytest.mark.gpu
def test_gpu_filt_keeps_the_plan_on_the_device():
    """one plan, IIR kernels only: nothing of a high-order `filt` runs on the host any more"""
    b, a = sps.butter(6, 0.3)
    x = F(rng(9).standard_normal((200000, 2)))
    tree = so.Filt(Signal(x, 1000 * Hz), so.PolynomialRatio(b, a))
    p = so.Plan(so.ToChannels(tree, 2), (200000, 2), np.float64, (1, 200000), False)
    names = [s["name"] for s in p.steps()]
    p.close()
    assert names and all(n.startswith("k_sos") for n in names), names


@pytest.mark.gpu
def test_gpu_filt_float32_signal_and_ill_conditioned_fallback():
    x32 = F(rng(10).standard_normal((20000, 2)).astype(np.float32))
    b, a = sps.butter(5, 0.3)
    got = so.filt(b, a, Signal(x32, 1000 * Hz))[0]
    assert got.dtype == np.float64  # promote_type(Float64 coefficients, Float32 samples), reference src/filters.jl:73
    assert relerr(got, oracle_filt(b, a, x32.astype(np.float64))) < 1e-9
    got32 = so.filt(b.astype(np.float32), a.astype(np.float32), Signal(x32, 1000 * Hz))[0]
    assert got32.dtype == np.float32
    assert relerr(got32, oracle_filt(b.astype(np.float32), a.astype(np.float32), x32.astype(np.float64))) < 1e-6
    bi, ai = sps.butter(12, 0.02)   # refused by the gate: the reference's own sequence (engine sink, host recurrence)
    x = F(rng(11).standard_normal((5000, 1)))
    assert relerr(so.filt(bi, ai, Signal(x, 1000 * Hz))[0], oracle_filt(bi, ai, x)) < 1e-9
