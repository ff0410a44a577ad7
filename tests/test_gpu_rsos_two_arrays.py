"""`Mix(x, y) |> Filt |> ToFramerate` over TWO arrays in ONE launch (round 6): the fused resampler + IIR
kernel takes the second array too (k_rsos.hip, rsos_loader's A2: the first array by LDS-DMA into the ring, the second one's
samples through the registers of the sixteen-wave geometry's two step waves -- three chunks in flight each --, added to the
landed chunk by the LDS's own adder).  The
reference evaluates the lazy map block by block inside the resampler's pull (`src/mapsignal.jl:54-57`, `src/filters.jl:240-244`)
and filters what the resampler yields (`src/filters.jl:143-148`).  Asserted: one launch; the values of the materialised path
(K1's sum, then the same kernel: the same IEEE operation on the same samples) BIT FOR BIT; the oracle's to 1e-9; edges of the
signal, windows, operands that start inside their arrays; the shapes the loader does not take stay with the earlier forms and
stay right; non-finite samples in either array give the reference's set."""
import numpy as np
import pytest

import sigops_amd as so
from oracle_bridge import oracle_sink, relerr
from test_gpu_two_arrays import OPS, arrays, env, steps_of

pytestmark = pytest.mark.gpu

FILT = lambda: so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz)


def three(tree):
    """fused with two arrays / the sum materialised by K1, then the fused kernel / resampler and filter apart"""
    with env(SIGOPS_RSOS_MINGROUPS=1, SIGOPS_RSOS_NO_ARR2=None, SIGOPS_NO_ARR2=None):
        names = steps_of(tree)
        a = so.sink(tree)[0]
    with env(SIGOPS_RSOS_MINGROUPS=1, SIGOPS_NO_ARR2=1):
        names1 = steps_of(tree)
        b = so.sink(tree)[0]
    with env(SIGOPS_RSOS_MINGROUPS=1, SIGOPS_RSOS_NO_ARR2=1, SIGOPS_NO_ARR2=None):
        names2 = steps_of(tree)
        c = so.sink(tree)[0]
    return a, b, c, names, names1, names2


@pytest.mark.parametrize("nch", [8, 16, 24, 4, 12, 2, 6])
@pytest.mark.parametrize("op", sorted(OPS))
def test_one_launch_bit_equal_to_the_materialised_sum(op, nch):
    n = 400_003
    x, y = arrays(n, nch, 21)
    tree = OPS[op](so.Signal(x, 44.1 * so.kHz), so.Signal(y, 44.1 * so.kHz)) | FILT() | so.ToFramerate(48 * so.kHz)
    a, b, c, names, names1, names2 = three(tree)
    assert names == ["k_rsos"], names
    assert len(names1) == 2 and names1[0].startswith("k_pointwise") and names1[1] == "k_rsos", names1
    # (groups of two or four channels: the resampler alone takes no second Float64 array -- K1's sum, then the fused kernel)
    assert "k_rsos" not in names2[:1] and ("k_resample_periodic" in names2 or nch % 8 != 0), names2
    assert np.array_equal(a, b), float(np.abs(a - b).max())
    want = oracle_sink(tree)
    assert relerr(a, want) <= 1e-9 and relerr(c, want) <= 1e-9
    assert relerr(a, c) <= 1e-10


@pytest.mark.parametrize("rates", [(44.1, 48.0), (22.05, 24.0), (32.0, 48.0), (24.0, 48.0), (48.0, 44.1)])
def test_rate_pairs(rates):
    fi, fo = rates
    n = int(300_000 * fi / 44.1) + 7
    x, y = arrays(n, 8, 22)
    tree = so.Mix(so.Signal(x, fi * so.kHz), so.Signal(y, fi * so.kHz)) | so.Filt(so.Lowpass, 3 * so.kHz) | so.ToFramerate(fo * so.kHz)
    a, b, c, names, names1, names2 = three(tree)
    if names == ["k_rsos"]:
        assert np.array_equal(a, b)
    want = oracle_sink(tree)
    assert relerr(a, want) <= 1e-9 and relerr(b, want) <= 1e-9


def test_operands_that_start_inside_their_arrays_and_a_window_of_the_result():
    """`After` on either operand (different first frames: different alignments of the two rows), `Until`, and a window behind"""
    x, y = arrays(500_000, 8, 23)
    for dx, dy in ((0, 0), (2, 4), (6, 2), (3, 3), (1, 2)):
        X = so.Signal(x, 44.1 * so.kHz) | so.After(dx * so.frames) | so.Until(400_000 * so.frames)
        Y = so.Signal(y, 44.1 * so.kHz) | so.After(dy * so.frames) | so.Until(400_000 * so.frames)
        tree = so.Mix(X, Y) | FILT() | so.ToFramerate(48 * so.kHz) | so.After(0.5 * so.s) | so.Until(3 * so.s)
        a, b, c, names, names1, names2 = three(tree)
        want = oracle_sink(tree)
        assert relerr(a, want) <= 1e-9, (dx, dy, names)
        assert relerr(b, want) <= 1e-9 and relerr(c, want) <= 1e-9, (dx, dy)


def test_operands_of_different_lengths():
    """the shorter operand is padded with zeros by the reference's Mix (src/mapsignal.jl:141-160): pieces of one array beside the
    piece of two"""
    x, _ = arrays(400_000, 8, 24)
    _, y = arrays(250_000, 8, 25)
    tree = so.Mix(so.Signal(x, 44.1 * so.kHz), so.Signal(y, 44.1 * so.kHz)) | FILT() | so.ToFramerate(48 * so.kHz)
    with env(SIGOPS_RSOS_MINGROUPS=1):
        got = so.sink(tree)[0]
    assert relerr(got, oracle_sink(tree)) <= 1e-9


def test_odd_channel_counts_and_float32_keep_the_earlier_forms():
    """(three channels: groups of one channel have sixteen units, which the step waves do not take; Float32: K3's Float32
    two-array form + the filter)"""
    x, y = arrays(300_000, 3, 26)
    tree = so.Mix(so.Signal(x, 44.1 * so.kHz), so.Signal(y, 44.1 * so.kHz)) | FILT() | so.ToFramerate(48 * so.kHz)
    with env(SIGOPS_RSOS_MINGROUPS=1):
        names = steps_of(tree)
        got = so.sink(tree)[0]
    assert names != ["k_rsos"]
    assert relerr(got, oracle_sink(tree)) <= 1e-9
    x32, y32 = arrays(300_000, 8, 27, np.float32)
    tree = so.Mix(so.Signal(x32, 44.1 * so.kHz), so.Signal(y32, 44.1 * so.kHz)) | FILT() | so.ToFramerate(48 * so.kHz)
    with env(SIGOPS_RSOS_MINGROUPS=1):
        got = so.sink(tree)[0]
    assert got.dtype == np.float32 and relerr(got, oracle_sink(tree)) <= 1e-6


@pytest.mark.parametrize("which", ["x", "y"])
def test_non_finite_samples_in_either_array(which):
    x, y = arrays(400_000, 8, 28)
    d = x if which == "x" else y
    d[123_456, 1] = np.nan
    d[300_001, 6] = np.inf
    tree = so.Mix(so.Signal(x, 44.1 * so.kHz), so.Signal(y, 44.1 * so.kHz)) | FILT() | so.ToFramerate(48 * so.kHz)
    with env(SIGOPS_RSOS_MINGROUPS=1):
        assert steps_of(tree) == ["k_rsos"]
        got = so.sink(tree)[0]
    want = oracle_sink(tree)
    assert np.array_equal(np.isfinite(got), np.isfinite(want))
    ok = np.isfinite(want)
    assert (~ok).any() and relerr(got[ok], want[ok]) <= 1e-9


def test_device_arrays_graph_replay_and_a_replaced_operand():
    import torch
    rng = np.random.default_rng(29)
    n, nch = 400_000, 8
    xs = [torch.from_numpy(np.ascontiguousarray(rng.standard_normal((nch, n)))).cuda() for _ in range(3)]
    tree = so.Mix(so.Signal(xs[0].t(), 44.1 * so.kHz), so.Signal(xs[1].t(), 44.1 * so.kHz)) | FILT() | so.ToFramerate(48 * so.kHz)
    m = so.nframes(tree)
    host = lambda a, b: so.Mix(so.Signal(np.asfortranarray(a.t().cpu().numpy()), 44.1 * so.kHz),
                               so.Signal(np.asfortranarray(b.t().cpu().numpy()), 44.1 * so.kHz)) | FILT() | so.ToFramerate(48 * so.kHz)
    with env(SIGOPS_RSOS_MINGROUPS=1):
        p = so.Plan(so.ToChannels(tree, nch), (m, nch), np.float64, (1, m), True)
    assert [s["name"] for s in p.steps()] == ["k_rsos"]
    out = torch.empty((nch, m), dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    want = oracle_sink(host(xs[0], xs[1]))
    for _ in range(4):
        out.zero_()
        p.execute(out.data_ptr(), st)
        torch.cuda.synchronize()
        assert relerr(np.asfortranarray(out.t().cpu().numpy()), want) <= 1e-9
    p.set_array(1, xs[2].t())
    want2 = oracle_sink(host(xs[0], xs[2]))
    for _ in range(3):
        out.zero_()
        p.execute(out.data_ptr(), st)
        torch.cuda.synchronize()
        assert relerr(np.asfortranarray(out.t().cpu().numpy()), want2) <= 1e-9
    p.check()
    p.close()
