"""`Normpower` of a plain array (reference src/filters.jl:296-309: `vals` is filled from the child, its rms taken, every
block divided).  The engine takes the sum of squares over the array where it lies and lets whoever reads the normed
signal divide the array's own frames: no copy of the child into a `vals` buffer (planner.cpp, `Stage::norm_direct`;
`SIGOPS_NORM_COPY=1` at plan creation keeps the copy).  Same reduction order, same division: results are bit-equal to the
copying path; against the oracle bit-equal for Float32, within 1e-15 for Float64 (whose sum of squares is a fixed tree of
Float64 partial sums on the device, not the reference's pairwise order)."""
import os

import numpy as np
import pytest

import sigops_amd as so
from oracle_bridge import oracle_sink, relerr

pytestmark = pytest.mark.gpu


def F(a):
    return np.asfortranarray(a)


class env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def close64(got, want):
    """Float64 sums of squares: a fixed tree of Float64 partial sums on the device, not the reference's pairwise order --
    the rms agrees to an ulp or two (1.6e-16 on these lengths, with the copy as without it)"""
    return got.dtype == want.dtype and got.shape == want.shape and np.max(np.abs(got - want)) <= 1e-15 * np.max(np.abs(want))


def step_names(tree):
    n, c = so.nframes(tree), so.nchannels(tree)
    p = so.Plan(tree, (n, c), np.float64, (1, n), False)
    names = [s["name"].replace("_rtc", "") for s in p.steps()]
    p.close()
    return names


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("nch", [1, 2, 8])
def test_normpower_reads_the_array_in_place(dtype, nch):
    rng = np.random.default_rng(400 + nch)
    n = 300_007
    x = F((rng.standard_normal((n, nch)) * 0.4).astype(dtype))
    sig = so.Signal(x, 10 * so.kHz)
    trees = {
        "plain": sig | so.Normpower,
        "until": sig | so.Until(200_001 * so.frames) | so.Normpower,
        "after": sig | so.After(1_003 * so.frames) | so.Normpower,
        "window": sig | so.After(77 * so.frames) | so.Until(250_000 * so.frames) | so.Normpower | so.Amplify(0.25),
        "read twice": so.Mix(sig | so.Normpower, sig | so.Normpower | so.Amplify(-0.5)),
        "window of it": sig | so.Normpower | so.After(5_000 * so.frames) | so.Until(100_000 * so.frames),
        "under a filter": sig | so.Normpower | so.Filt(so.Lowpass, 2 * so.kHz),
    }
    for name, tree in trees.items():
        got = so.sink(tree)[0]
        want = oracle_sink(tree)
        assert got.dtype == want.dtype, name
        if name == "under a filter":
            tol = 1e-6 if dtype == np.float32 else 1e-10
            assert np.max(np.abs(got - want)) <= tol * np.max(np.abs(want)), name
        elif dtype == np.float32 and got.dtype == np.float32:
            assert np.array_equal(got, want), name  # (the Float32 reduction follows the oracle's order bit for bit)
        else:
            assert close64(got, want), name
        with env(SIGOPS_NORM_COPY=1):
            copied = so.sink(tree)[0]
        assert np.array_equal(got, copied), name


def test_no_copy_step():
    rng = np.random.default_rng(9)
    x = F(rng.standard_normal((100_000, 4)))
    tree = so.Signal(x, 10 * so.kHz) | so.Normpower
    names = step_names(tree)
    assert names == ["k_sumsq", "k_pointwise"], names
    with env(SIGOPS_NORM_COPY=1):
        assert step_names(tree) == ["k_pointwise", "k_sumsq", "k_pointwise"]


def test_arrays_the_copy_stays_for():
    """an interleaved array (frame stride = channel count) and a Float32 array under a Float64 `Normpower` keep the copy"""
    rng = np.random.default_rng(10)
    x = np.ascontiguousarray(rng.standard_normal((50_000, 3)))  # row-major: interleaved
    tree = so.Signal(x, 10 * so.kHz) | so.Normpower
    assert close64(so.sink(tree)[0], oracle_sink(tree))
    x32 = F(rng.standard_normal((50_000, 3)).astype(np.float32))
    tree = so.Signal(x32, 10 * so.kHz) | so.ToEltype(np.float64) | so.Normpower
    assert close64(so.sink(tree)[0], oracle_sink(tree))


def test_device_leaf_rebound_between_executes():
    """the array's address is taken at execute time (`so_plan_set_array`), for the rms pass as for the readers"""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(11)
    n = 120_000
    a = rng.standard_normal((n, 2))
    b = rng.standard_normal((n, 2)) * 3.0
    for x in (a, b):
        xd = torch.from_numpy(np.ascontiguousarray(x.T)).cuda()  # (nch, n) row-major = planar
        tree = so.Signal(xd.T, 10 * so.kHz) | so.Normpower
        out = torch.empty((2, n), dtype=torch.float64, device="cuda")
        so.sink_into(out.T, tree)
        want = oracle_sink(so.Signal(F(x), 10 * so.kHz) | so.Normpower)
        assert close64(out.T.cpu().numpy(), want)


@pytest.mark.parametrize("nch", [1, 2, 5])
def test_float32_sum_of_squares_edges(nch):
    """the Float32 reduction (blocks of 1024 front to back, neighbours folded level by level -- oracle/sigops_oracle.c,
    NORMPOWER) over row lengths around its chunk (64), block (1024) and wave (64 blocks) sizes: channel boundaries inside a
    block, rows shorter than a chunk, a last wave with one block"""
    rng = np.random.default_rng(500 + nch)
    for n in (1, 3, 17, 63, 64, 65, 200, 1023, 1024, 1025, 4097, 65_535, 65_536, 65_537, 131_073, 262_145):
        x = F((rng.standard_normal((n, nch)) * 0.7).astype(np.float32))
        tree = so.Signal(x, 10 * so.kHz) | so.Normpower
        got, want = so.sink(tree)[0], oracle_sink(tree)
        assert got.dtype == np.float32 and np.array_equal(got, want), (n, nch)
        if n > 64:  # a copy with a pitch of its own (the filter's output buffer) and a window of the array
            t2 = so.Signal(x, 10 * so.kHz) | so.After(7 * so.frames) | so.Normpower
            assert np.array_equal(so.sink(t2)[0], oracle_sink(t2)), (n, nch, "after")


def test_normpower_over_views_the_in_place_form_does_not_take():
    """interleaved and strided device tensors, a channel subset: `vals` is a copy there (or the in-place form falls back to
    one, stages.cpp) -- never a refused plan"""
    import torch

    rng = np.random.default_rng(77)
    base = rng.standard_normal((50000, 6))
    il = torch.tensor(base, device="cuda")                      # [frames x channels] row-major: interleaved
    planar = torch.tensor(np.ascontiguousarray(base.T), device="cuda").t()
    strided = torch.tensor(np.ascontiguousarray(np.repeat(base.T, 2, axis=1)), device="cuda").t()[::2]   # every second frame of a longer tensor
    for leaf, ref in ((il, base), (planar, base), (strided, base)):
        x = so.Signal(leaf, 10 * so.kHz) | so.Normpower
        got = so.sink(x)[0]
        want = ref / np.sqrt(np.mean(ref ** 2))
        assert relerr(got, want) < 1e-12
        y = so.Signal(leaf, 10 * so.kHz) | so.After(100 * so.frames) | so.Until(40000 * so.frames) | so.Normpower
        goty = so.sink(y)[0]
        w = ref[100:40100]
        assert relerr(goty, w / np.sqrt(np.mean(w ** 2))) < 1e-12
