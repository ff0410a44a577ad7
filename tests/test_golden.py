"""Committed golden fixtures (tests/golden/, generator: tests/golden/make_golden.py).
design.npz comes from SciPy and pins the C-ABI design entry points; cases.npz pins the oracle
(CPU) and gives the HIP path (gpu) expected values that were computed in the build container."""
import os

import numpy as np
import pytest

import sigops_amd as so
from cases import CASES
from oracle_bridge import oracle_sink, relerr

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DESIGN = np.load(os.path.join(G, "design.npz"))
GOLD = np.load(os.path.join(G, "cases.npz"))


def _check(name, y, tol):
    assert tuple(GOLD[name + "__shape"]) == y.shape and str(GOLD[name + "__dtype"]) == str(y.dtype)
    if name + "__full" in GOLD:
        want = GOLD[name + "__full"]
        if want.size == 0:
            return
        assert relerr(y, want) <= tol, name
    else:
        assert relerr(y[:64], GOLD[name + "__head"]) <= tol and relerr(y[-64:], GOLD[name + "__tail"]) <= tol, name
        assert abs(np.linalg.norm(y.astype(np.float64)) - float(GOLD[name + "__norm"])) <= tol * float(GOLD[name + "__norm"]), name


def _same_roots(a, b, tol):
    """the same multiset of complex roots (order-free: equal real parts sort unpredictably)"""
    a, b = list(np.asarray(a)), list(np.asarray(b))
    if len(a) != len(b):
        return False
    for x in a:
        k = int(np.argmin([abs(x - y) for y in b]))
        if abs(x - b[k]) > tol:
            return False
        b.pop(k)
    return True


def test_design_matches_scipy_fixtures():
    for name, (design, args, fs) in {"bandstop_500_2000_44100": ("bandstop", (500.0, 2000.0), 44100.0),
                                     "bandstop_500_2000_48000": ("bandstop", (500.0, 2000.0), 48000.0),
                                     "lowpass_4000_16000": ("lowpass", (4000.0,), 16000.0)}.items():
        resp = {"bandstop": so.Bandstop, "lowpass": so.Lowpass}[design](*args, fs=fs)
        zpk = so.digitalfilter(resp, so.Butterworth(5))
        assert _same_roots(zpk.p, DESIGN[name + "_p"], 1e-12), name
        assert _same_roots(zpk.z, DESIGN[name + "_z"], 1e-9), name
        assert abs(zpk.k - float(DESIGN[name + "_k"])) <= 1e-12 * abs(float(DESIGN[name + "_k"])), name
    zpk = so.digitalfilter(so.Highpass(8, fs=100), so.Chebyshev1(5, 1))
    assert _same_roots(zpk.p, DESIGN["cheby1_hp_8_100_p"], 1e-12)
    for key, ratio in (("taps_48000_44100", 48000 / 44100), ("taps_16000_44100", 16000 / 44100), ("taps_2_1", (2, 1))):
        h = so.design_resample(ratio)
        assert h.shape == DESIGN[key].shape and np.allclose(h, DESIGN[key], rtol=0, atol=1e-13 * np.abs(DESIGN[key]).max()), key


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_golden(name):
    _check(name, np.asarray(oracle_sink(CASES[name]())), 1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_engine_matches_golden(name):
    y = so.sink(CASES[name](), so.Array)
    _check(name, np.asarray(y), 1e-6 if y.dtype == np.float32 else 1e-9)
