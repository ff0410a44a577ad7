"""`Mix(x, y) |> ToFramerate` and `Amplify(x, y) |> ToFramerate` over TWO array operands: the reference
resamples the lazy map block by block (`/root/reference/src/mapsignal.jl:54-57` inside the resampler's pull,
`src/filters.jl:240-244`); K1 used to materialise the sum.  K3's A2 instantiation stages both arrays by LDS-DMA and applies
the one operation in place, so its values are the materialised path's BIT FOR BIT (the same IEEE operation on the same
samples, then the same products) -- and the oracle's within the resampler's tolerance.  Shapes the instantiation does not
take (Float32, other rates' window lengths, extra steps) stay with K1 and stay right."""
import contextlib
import os

import numpy as np
import pytest

import sigops_amd as so
from oracle_bridge import oracle_sink, relerr

pytestmark = pytest.mark.gpu


@contextlib.contextmanager
def env(**kw):
    old = {k: os.environ.get(k) for k in kw}
    for k, v in kw.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def steps_of(x, dtype=np.float64):
    n, nch = so.nframes(x), so.nchannels(x)
    p = so.Plan(so.ToChannels(x, nch), (n, nch), dtype, (1, n), False)
    names = [s["name"] for s in p.steps()]
    p.close()
    return names


def arrays(n, nch, seed, dtype=np.float64):
    rng = np.random.default_rng(seed)
    return (np.asfortranarray(rng.standard_normal((n, nch)).astype(dtype)),
            np.asfortranarray(rng.standard_normal((n, nch)).astype(dtype)))


def both(tree):
    with env(SIGOPS_NO_ARR2=None):
        names = steps_of(tree)
        a = so.sink(tree)[0]
    with env(SIGOPS_NO_ARR2=1):
        names0 = steps_of(tree)
        b = so.sink(tree)[0]
    return a, b, names, names0


OPS = {
    "Mix": lambda X, Y: so.Mix(X, Y),
    "Amplify": lambda X, Y: so.Amplify(X, Y),
    "x - y": lambda X, Y: so.OperateOn("-", X, Y),
    "y - x": lambda X, Y: so.OperateOn("-", Y, X),
}


@pytest.mark.parametrize("nch", [8, 4, 16])
@pytest.mark.parametrize("op", sorted(OPS))
def test_two_arrays_in_front_of_the_resampler_one_launch_bit_equal_to_the_materialised_path(op, nch):
    n = 123_457
    x, y = arrays(n, nch, 11)
    tree = OPS[op](so.Signal(x, 44.1 * so.kHz), so.Signal(y, 44.1 * so.kHz)) | so.ToFramerate(48 * so.kHz)
    # (Float64 groups of four channels: the two-array form is slower there than K1's sum + the one-array kernel -- 1.81 against
    #  0.87 ms for 25 M x 4 --, so the planner leaves them to K1; SIGOPS_RS_ARR2_CT4 keeps the instantiation under test)
    with env(SIGOPS_RS_ARR2_CT4=1 if nch == 4 else None):
        a, b, names, names0 = both(tree)
    if nch == 4:
        with env(SIGOPS_RS_ARR2_CT4=None):
            assert len(steps_of(tree)) == 2 and np.array_equal(so.sink(tree)[0], b)
    assert names == ["k_resample_periodic"], names            # one launch: no materialised operand
    assert len(names0) == 2 and names0[1] == "k_resample_periodic" and "pointwise" in names0[0], names0
    assert np.array_equal(a, b)
    assert relerr(a, oracle_sink(tree)) <= 1e-9


@pytest.mark.parametrize("nch", [8, 4])
@pytest.mark.parametrize("op", sorted(OPS))
def test_two_float32_arrays_of_a_float32_signal(op, nch):
    """Float32 operands: the step rounds to Float32 (Julia's Float32 arithmetic, what K1's map computes) in the Float32
    tile, the products run on the Float32 MFMA as for one Float32 array -- bit-equal to the materialised path again"""
    n = 98_765
    x, y = arrays(n, nch, 21, np.float32)
    tree = OPS[op](so.Signal(x, 44.1 * so.kHz), so.Signal(y, 44.1 * so.kHz)) | so.ToFramerate(48 * so.kHz)
    with env(SIGOPS_NO_ARR2=None):
        names = steps_of(tree, np.float32)
        a = so.sink(tree)[0]
    with env(SIGOPS_NO_ARR2=1):
        names0 = steps_of(tree, np.float32)
        b = so.sink(tree)[0]
    assert a.dtype == np.float32 and names == ["k_resample_periodic"] and len(names0) == 2, (names, names0)
    assert np.array_equal(a, b)
    assert relerr(a, oracle_sink(tree)) <= 1e-6
    with env(SIGOPS_RS_NO_F32MFMA=1):  # (without the Float32 MFMA there is no such instantiation: K1 materialises)
        assert len(steps_of(tree, np.float32)) == 2
        assert relerr(so.sink(tree)[0], oracle_sink(tree)) <= 1e-6


def test_rates_and_lengths():
    """other rate pairs of the 14-k-step family, lengths around tile and period boundaries, a one-frame signal"""
    for (fi, fo) in [(44.1, 48.0), (48.0, 44.1), (32.0, 48.0), (22.05, 24.0)]:
        for n in (1, 159, 160, 4097, 50_001):
            x, y = arrays(n, 8, n)
            tree = so.Mix(so.Signal(x, fi * so.kHz), so.Signal(y, fi * so.kHz)) | so.ToFramerate(fo * so.kHz)
            a, b, names, names0 = both(tree)
            assert np.array_equal(a, b), (fi, fo, n, names, names0)
            assert relerr(a, oracle_sink(tree)) <= 1e-9, (fi, fo, n)


def test_operands_of_different_lengths_and_offsets():
    """the shorter operand is zero-padded by Mix (one-padded by Amplify): the sum's pieces are carriers of their own, and a
    window (After / Until) moves both arrays' offsets"""
    x, _ = arrays(90_000, 8, 3)
    _, y = arrays(61_234, 8, 4)
    X, Y = so.Signal(x, 44.1 * so.kHz), so.Signal(y, 44.1 * so.kHz)
    for tree in (so.Mix(X, Y) | so.ToFramerate(48 * so.kHz),
                 so.Amplify(Y, X) | so.ToFramerate(48 * so.kHz),
                 so.Mix(X, Y) | so.After(1234 * so.frames) | so.Until(50_000 * so.frames) | so.ToFramerate(48 * so.kHz),
                 so.Mix(so.After(X, 777 * so.frames), Y) | so.ToFramerate(48 * so.kHz)):
        a, b, names, names0 = both(tree)
        assert np.array_equal(a, b), (names, names0)
        assert relerr(a, oracle_sink(tree)) <= 1e-9


def test_what_the_instantiation_does_not_take_stays_right():
    n = 40_000
    x, y = arrays(n, 8, 5)
    x32, y32 = x.astype(np.float32), y.astype(np.float32)
    X, Y = so.Signal(x, 44.1 * so.kHz), so.Signal(y, 44.1 * so.kHz)
    trees = {
        "Float32 and Float64 operands": so.Mix(so.Signal(x32, 44.1 * so.kHz), Y) | so.ToFramerate(48 * so.kHz),
        "three operands": so.Mix(X, Y, so.Signal(so.sin, ω=1 * so.kHz)) | so.Until(n * so.frames) | so.ToFramerate(48 * so.kHz),
        "a gain on top": so.Amplify(so.Mix(X, Y), 0.5) | so.ToFramerate(48 * so.kHz),
        "another rate's window": so.Mix(X, Y) | so.ToFramerate(16 * so.kHz),
        "two channels": so.Mix(so.Signal(x[:, :2].copy(order="F"), 44.1 * so.kHz), so.Signal(y[:, :2].copy(order="F"), 44.1 * so.kHz)) | so.ToFramerate(48 * so.kHz),
        "in front of the fused filter": so.Mix(X, Y) | so.Filt(so.Lowpass, 3 * so.kHz) | so.ToFramerate(48 * so.kHz),
    }
    for name, tree in trees.items():
        got = so.sink(tree)[0]
        tol = 1e-6 if got.dtype == np.float32 else 1e-8
        assert relerr(got, oracle_sink(tree)) <= tol, name


def test_device_arrays_replaced_between_executes():
    """so_plan_set_array on either operand: the carrier's second base is patched like its first"""
    torch = pytest.importorskip("torch")
    n, nch = 60_000, 8
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    mk = lambda: torch.randn((nch, n), dtype=torch.float64, device=dev, generator=g).t()
    x1, y1, x2, y2 = mk(), mk(), mk(), mk()
    X, Y = so.Signal(x1, 44.1 * so.kHz), so.Signal(y1, 44.1 * so.kHz)
    tree = so.Mix(X, Y) | so.ToFramerate(48 * so.kHz)
    n_out = so.nframes(tree)
    out = torch.empty((nch, n_out), dtype=torch.float64, device=dev).t()
    plan = so.Plan(so.ToChannels(tree, nch), (n_out, nch), np.float64, (out.stride(0), out.stride(1)), True, device=0)
    assert [s["name"] for s in plan.steps()] == ["k_resample_periodic"]
    stream = torch.cuda.current_stream().cuda_stream

    def run():
        plan.execute(out.data_ptr(), stream)
        torch.cuda.synchronize()
        return out.cpu().numpy().copy()

    def want(xa, ya):
        t = so.Mix(so.Signal(xa.cpu().numpy(), 44.1 * so.kHz), so.Signal(ya.cpu().numpy(), 44.1 * so.kHz)) | so.ToFramerate(48 * so.kHz)
        return oracle_sink(t)

    r11 = run()
    assert relerr(r11, want(x1, y1)) <= 1e-9
    plan.set_array(1, y2)  # (array leaves in depth-first order: x, y)
    assert relerr(run(), want(x1, y2)) <= 1e-9
    plan.set_array(0, x2)
    assert relerr(run(), want(x2, y2)) <= 1e-9
    plan.close()
