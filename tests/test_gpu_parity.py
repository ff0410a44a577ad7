"""GPU parity: the HIP engine (through the C-ABI, libsigops.so) against the CPU
oracle on the same seeded trees.  Tolerances (BASELINE.json north_star):
  - structural / index work (Until, After, Pad, Append, channel maps): bit-exact
  - floating point: norm-wise relative error <= 1e-6 (Julia isapprox semantics)
"""
import numpy as np
import pytest

import sigops_amd as so
from cases import CASES
from oracle_bridge import oracle_sink, relerr

pytestmark = pytest.mark.gpu

# cases whose values are pure selections / exact arithmetic on both sides
EXACT = {"array_plus_one", "cut_after_until", "pad_zero_after", "pad_cycle", "pad_mirror",
         "pad_lastframe_array", "mix_65_channels", "reverse_channels", "padded_mix", "padded_amplify",
         "addchannel_extend", "select_channel", "offset_append_sum", "float32_append_pad", "sub_div",
         "negate", "strided_array", "gain_db", "empty_signal"}
TOL = 1e-6


@pytest.mark.parametrize("name", sorted(CASES))
def test_case_matches_oracle(name):
    x = CASES[name]()
    want = oracle_sink(x)
    got = so.sink(x, so.Array)
    assert got.shape == want.shape
    assert got.dtype == want.dtype
    if name in EXACT:
        assert np.array_equal(got, want), f"{name}: not bit-exact"
    else:
        err = relerr(got, want)
        assert err <= TOL, f"{name}: rel err {err:.3e}"
        # in practice both sides are fp64 throughout: keep a much tighter watch too
        assert err <= (1e-6 if got.dtype == np.float32 else 1e-9), f"{name}: rel err {err:.3e}"


def test_sink_into_device_tensor():
    torch = pytest.importorskip("torch")
    x = CASES["resample_441_48"]()
    want = oracle_sink(x)
    res, fs = so.sink(x, "torch")
    assert fs == 48000.0 and res.is_cuda
    assert relerr(res.cpu().numpy(), want) < 1e-11


def test_device_resident_leaf():
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(5)
    host = np.asfortranarray(rng.standard_normal((30000, 4)))
    dev = torch.from_numpy(np.ascontiguousarray(host.T)).cuda().t()  # column-major on device
    x_dev = so.Signal(dev, 44.1 * so.kHz) | so.Filt(so.Lowpass, 4 * so.kHz) | so.ToFramerate(16 * so.kHz)
    x_host = so.Signal(host, 44.1 * so.kHz) | so.Filt(so.Lowpass, 4 * so.kHz) | so.ToFramerate(16 * so.kHz)
    got, _ = so.sink(x_dev)
    want = oracle_sink(x_host)
    assert relerr(got, want) < 1e-11


def test_errors_match_reference():
    with pytest.raises(so.ErrorException):  # runtests.jl:62
        so.sink_into(np.ones((10, 2), order="F"), np.ones((5, 2)))
    with pytest.raises(so.ErrorException):  # runtests.jl:138
        so.sink(so.Signal(np.arange(1.0, 11.0), 5 * so.Hz) | so.After(3 * so.s))
    with pytest.raises(so.ErrorException):  # runtests.jl:572-573
        so.sink(so.Signal(np.sin, 200 * so.Hz) | so.ToChannels(2))


def test_sink_into_wider_buffer():  # runtests.jl:306-310
    rng = np.random.default_rng(3)
    x = np.asfortranarray(rng.random((10, 2)))
    y = np.asfortranarray(rng.random((5, 2)))
    z = np.ones((10, 4), order="F")
    so.sink_into(z, so.Signal(x, 10 * so.Hz) | so.AddChannel(y))
    assert np.all(z[5:, 2:] == 0) and np.array_equal(z[:, :2], x)


def test_tile_invariance_linearity():
    """size-independent property at a larger size: the engine is linear in its leaves"""
    rng = np.random.default_rng(7)
    a = np.asfortranarray(rng.standard_normal((300000, 2)))
    b = np.asfortranarray(rng.standard_normal((300000, 2)))

    def pipe(z):
        return so.Signal(z, 44.1 * so.kHz) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) \
            | so.ToFramerate(48 * so.kHz)

    ya, _ = so.sink(pipe(a))
    yb, _ = so.sink(pipe(b))
    yab, _ = so.sink(pipe(np.asfortranarray(a + 2.0 * b)))
    assert relerr(yab, ya + 2.0 * yb) < 1e-11
