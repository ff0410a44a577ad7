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


def test_plan_reuse_graph_replay_and_set_array():
    """One plan executed repeatedly: plain launches, HIP-graph capture, replay; then the array
    leaves are swapped with so_plan_set_array (which must invalidate the capture).  The tree has
    two independently filtered operands (separate streams) feeding a resampler."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(17)
    fs = 8 * so.kHz

    def host(n, nch):
        return np.asfortranarray(rng.standard_normal((n, nch)))

    def dev(a):
        return torch.from_numpy(np.ascontiguousarray(a.T)).cuda().t()

    def tree(a, b):
        return (so.Mix(so.Signal(a, fs) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz),
                       so.Signal(b, fs) | so.Filt(so.Lowpass, 1 * so.kHz))
                | so.Ramp(5 * so.ms) | so.ToFramerate(12 * so.kHz))

    A = [host(6000, 2) for _ in range(2)]
    B = [host(6000, 2) for _ in range(2)]
    dA, dB = [dev(a) for a in A], [dev(b) for b in B]
    x = tree(*dA)
    n = so.nframes(x)
    out_t = torch.empty((2, n), dtype=torch.float64, device="cuda")
    out = out_t.t()
    plan = so.Plan(so.ToChannels(x, 2), (n, 2), np.float64, (out.stride(0), out.stride(1)), True)
    stream = torch.cuda.current_stream().cuda_stream
    want_a, want_b = oracle_sink(tree(*A)), oracle_sink(tree(*B))
    try:
        assert plan.stats()["n_stages"] >= 3
        for _ in range(4):  # direct, capture + launch, replay, replay
            out_t.zero_()
            plan.execute(out.data_ptr(), stream)
            torch.cuda.synchronize()
            assert relerr(out.cpu().numpy(), want_a) < 1e-10
        for k in range(2):
            plan.set_array(k, dB[k])
        for _ in range(3):
            out_t.zero_()
            plan.execute(out.data_ptr(), stream)
            torch.cuda.synchronize()
            assert relerr(out.cpu().numpy(), want_b) < 1e-10
    finally:
        plan.close()


def test_set_array_on_fused_resampler_carrier():
    """so_plan_set_array must also re-point the carriers of a fused resampler source"""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(23)
    fs = 44.1 * so.kHz

    def dev(a):
        return torch.from_numpy(np.ascontiguousarray(a.T)).cuda().t()

    def tree(a):
        return so.Signal(a, fs) | so.Amplify(so.Signal(so.sin, ω=5 * so.Hz)) | so.Until(20000 * so.frames) | so.ToFramerate(48 * so.kHz)

    a, b = (np.asfortranarray(rng.standard_normal((20000, 8))) for _ in range(2))
    da, db = dev(a), dev(b)
    x = tree(da)
    n = so.nframes(x)
    out_t = torch.empty((8, n), dtype=torch.float64, device="cuda")
    out = out_t.t()
    plan = so.Plan(so.ToChannels(x, 8), (n, 8), np.float64, (out.stride(0), out.stride(1)), True)
    stream = torch.cuda.current_stream().cuda_stream
    try:
        plan.execute(out.data_ptr(), stream)
        torch.cuda.synchronize()
        assert relerr(out.cpu().numpy(), oracle_sink(tree(a))) < 1e-10
        plan.set_array(0, db)
        plan.execute(out.data_ptr(), stream)
        torch.cuda.synchronize()
        assert relerr(out.cpu().numpy(), oracle_sink(tree(b))) < 1e-10
    finally:
        plan.close()


def test_append_after_long_filtered_child_is_a_concatenation():
    """Documented divergence: the reference's FilteredSignal end test compares
    a buffer-local index with the global length (src/filters.jl:224-227), so a filtered child
    longer than one block never reports its end and `Append` keeps pulling its zero-padded tail
    instead of moving on (the oracle reproduces that: tests/test_oracle_dsp.py).  The engine
    implements the documented meaning of Append: each child's own frames, one after the other."""
    rng = np.random.default_rng(29)
    fs = 8 * so.kHz
    a, b = (np.asfortranarray(rng.standard_normal((6000, 2))) for _ in range(2))
    ra, rb = (so.Signal(v, fs) | so.ToFramerate(12 * so.kHz) for v in (a, b))
    got, _ = so.sink(so.Append(ra, rb))
    want = np.concatenate([oracle_sink(ra), oracle_sink(rb)])
    assert relerr(got, want) < 1e-11


@pytest.mark.parametrize("fused", [True, False])
def test_resampler_ring_is_deterministic(fused):
    """The persistent resampler synchronises loader and compute waves with counted vmcnt waits,
    raw barriers and LDS rings written across barrier intervals: repeated executes over a NaN-
    prefilled result must be bit-identical (a race shows up as a sporadic difference or a NaN)."""
    torch = pytest.importorskip("torch")
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    n_in = 44100 * 20
    noise = torch.randn((8, n_in), dtype=torch.float64, device="cuda", generator=g).t()
    x = so.Signal(noise, 44.1 * so.kHz)
    if fused:
        x = x | so.Amplify(so.Signal(so.sin, ω=5 * so.Hz)) | so.Until(n_in * so.frames)
    x = x | so.ToFramerate(48 * so.kHz)
    n = so.nframes(x)
    out_t = torch.empty((8, n), dtype=torch.float64, device="cuda")
    out = out_t.t()
    plan = so.Plan(so.ToChannels(x, 8), (n, 8), np.float64, (out.stride(0), out.stride(1)), True)
    stream = torch.cuda.current_stream().cuda_stream
    try:
        plan.execute(out.data_ptr(), stream)
        torch.cuda.synchronize()
        ref = out_t.clone()
        assert not torch.isnan(ref).any()
        for _ in range(40):
            out_t.fill_(float("nan"))
            plan.execute(out.data_ptr(), stream)
            torch.cuda.synchronize()
            assert torch.equal(out_t, ref)
    finally:
        plan.close()


@pytest.mark.parametrize("nch,gen", [(8, dict(ω=5 * so.Hz)), (4, dict(ω=440 * so.Hz, ϕ=1.25)), (8, dict()),
                                     (2, dict(ω=50 * so.Hz, ϕ=0.5)), (1, dict(ω=5 * so.Hz))])
def test_resampler_two_level_sine_gain(nch, gen, monkeypatch):
    """A fused `Amplify(x, Signal(sin, ...))` in front of the periodic resampler is evaluated in
    two levels (share bases by one wave + one fma per frame, kernel variant TWO).  Same result as
    the general per-frame evaluation (SIGOPS_RS_NOTWO) far beyond the parity bound, deep into a
    long signal (large phases) and with an offset source, and within the bound of the oracle."""
    rng = np.random.default_rng(31)
    fs = 44.1 * so.kHz
    n_in = 1_500_000
    a = np.asfortranarray(rng.standard_normal((n_in + 1000, nch)))

    def tree(arr):
        return (so.Signal(arr, fs) | so.After(1000 * so.frames) | so.Amplify(so.Signal(so.sin, **gen))
                | so.Until(n_in * so.frames) | so.ToFramerate(48 * so.kHz))

    got, _ = so.sink(tree(a))
    monkeypatch.setenv("SIGOPS_RS_NOTWO", "1")
    general, _ = so.sink(tree(a))
    assert got.shape == general.shape
    # the two differ by a few ulp of the PHASE (up to 1.5e4 cycles here: ulp = 1.8e-12 cycles)
    assert relerr(got, general) < 5e-11
    assert relerr(got[-50000:], general[-50000:]) < 5e-11  # where the phase is largest
    m = 200_000  # the oracle on a prefix (seconds of CPU)
    want = oracle_sink(so.Signal(a[:m + 1000], fs) | so.After(1000 * so.frames) | so.Amplify(so.Signal(so.sin, **gen))
                       | so.Until(m * so.frames) | so.ToFramerate(48 * so.kHz))
    k = want.shape[0] - 200  # (the prefix ends where the long signal goes on)
    assert relerr(got[:k], want[:k]) < 1e-10


def test_set_array_reaches_materialised_sub_expressions():
    """A tree beyond the fused interpreter's stack depth is split by the planner into scratch
    buffers written by extra pointwise steps (planner.cpp legalise): plan reuse, graph replay and
    so_plan_set_array must reach the array leaves inside those sub-expressions too."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(61)
    fs = 100 * so.Hz

    def dev(a):
        return torch.from_numpy(np.ascontiguousarray(a.T)).cuda().t()

    def tree(xs):
        e = so.Signal(xs[6], fs)
        for k in (5, 4, 3, 2, 1, 0):
            e = so.Mix(so.Signal(xs[k], fs), e) if k % 2 else so.Amplify(so.Signal(xs[k], fs), e)
        return e

    A = [np.asfortranarray(rng.standard_normal((4000, 2))) for _ in range(7)]
    B = [np.asfortranarray(rng.standard_normal((4000, 2))) for _ in range(7)]
    dA, dB = [dev(a) for a in A], [dev(b) for b in B]
    x = tree(dA)
    n = so.nframes(x)
    out_t = torch.empty((2, n), dtype=torch.float64, device="cuda")
    out = out_t.t()
    plan = so.Plan(so.ToChannels(x, 2), (n, 2), np.float64, (out.stride(0), out.stride(1)), True)
    stream = torch.cuda.current_stream().cuda_stream
    want_a, want_b = oracle_sink(tree(A)), oracle_sink(tree(B))
    try:
        for _ in range(4):
            out_t.zero_()
            plan.execute(out.data_ptr(), stream)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), want_a)
        assert plan.stats()["n_launches"] >= 2  # at least one materialising step
        for k in range(7):
            plan.set_array(k, dB[k])
        for _ in range(3):
            out_t.zero_()
            plan.execute(out.data_ptr(), stream)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), want_b)
    finally:
        plan.close()


@pytest.mark.parametrize("nch,leaf_dtype,fused", [(8, np.float64, True), (4, np.float32, True), (8, np.float64, False)])
def test_float64_signal_into_float32_result(nch, leaf_dtype, fused):
    """`sink!` of a Float64 signal into a Float32 buffer converts on write (reference
    src/sink.jl:262-266).  When the root is the periodic resampler its fp64 kernel rounds in its
    own store (variant with a Float32 result pointer) instead of a separate conversion pass:
    same values as the oracle's Float64 result rounded once."""
    rng = np.random.default_rng(67)
    n = 20000
    x = np.asfortranarray(rng.standard_normal((n, nch)).astype(leaf_dtype))
    sig = so.Signal(x, 44.1 * so.kHz)
    if fused:
        sig = sig | so.Amplify(so.Signal(so.sin, ω=5 * so.Hz)) | so.Until(n * so.frames)
    else:
        sig = sig | so.ToEltype(np.float64)
    tree = sig | so.ToFramerate(48 * so.kHz)
    want = oracle_sink(tree)
    assert want.dtype == np.float64
    res = np.full((so.nframes(tree), nch), np.nan, dtype=np.float32, order="F")
    so.sink_into(res, tree)
    w32 = want.astype(np.float32)
    # (the engine's Float64 values differ from the oracle's by ~1e-15: a different Float32 only
    #  where that crosses a rounding boundary)
    assert relerr(res.astype(np.float64), w32.astype(np.float64)) < 1e-7
    assert np.mean(res != w32) < 1e-3


@pytest.mark.parametrize("nch", [1, 2, 8])
@pytest.mark.parametrize("skip", [0, 30001])
def test_float64_filter_into_float32_result(nch, skip):
    """... and when the root is an IIR its pass 3 rounds in its own store (also for a window of it: the
    warm-up frames in front are left out)"""
    rng = np.random.default_rng(68)
    n = 60000
    x = np.asfortranarray(rng.standard_normal((n, nch)))
    tree = so.Mix(so.Signal(so.sin, ω=1 * so.kHz), so.Signal(x, 44.1 * so.kHz)) | so.Until(n * so.frames) \
        | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz)
    if skip:
        tree = tree | so.After(skip * so.frames)
    want = oracle_sink(tree)
    assert want.dtype == np.float64
    res = np.full((so.nframes(tree), nch), np.nan, dtype=np.float32, order="F")
    so.sink_into(res, tree)
    w32 = want.astype(np.float32)
    assert relerr(res.astype(np.float64), w32.astype(np.float64)) < 1e-7
    assert np.mean(res != w32) < 1e-3
    torch = pytest.importorskip("torch")
    dev = torch.full((nch, so.nframes(tree) + 3), float("nan"), dtype=torch.float32, device="cuda")
    so.sink_into(dev.t()[:so.nframes(tree)], tree)
    assert np.array_equal(dev[:, :so.nframes(tree)].t().cpu().numpy(), res)
    assert bool(torch.isnan(dev[:, so.nframes(tree):]).all())


@pytest.mark.parametrize("nch,n,res_dt,gen", [(8, 300_000, np.float64, dict(ω=5 * so.Hz)), (4, 120_000, np.float64, dict(ω=440 * so.Hz, ϕ=0.3)),
                                             (8, 50_000, np.float32, dict(ω=5 * so.Hz)), (8, 9_000, np.float64, dict()),
                                             (4, 200_000, np.float32, dict(ω=50 * so.Hz))])
def test_float32_array_times_float64_gain_is_fused(nch, n, res_dt, gen, monkeypatch):
    """`Amplify(x::Float32 array, Signal(sin))` is a Float64 signal (Julia promotion).  The periodic
    resampler's GA instantiation keeps the raw Float32 tile in LDS and multiplies at the MFMA's A
    operand, so the product never touches HBM (it used to be materialised by a K1 pass: 1.53 ms on
    config 3 with a Float32 leaf).  Same values as the materialising path and as the oracle."""
    rng = np.random.default_rng(71)
    x = np.asfortranarray(rng.standard_normal((n + 50, nch)).astype(np.float32))
    tree = (so.Signal(x, 44.1 * so.kHz) | so.After(50 * so.frames) | so.Amplify(so.Signal(so.sin, **gen)) | so.Until(n * so.frames)
            | so.ToFramerate(48 * so.kHz))
    want = oracle_sink(tree)
    assert want.dtype == np.float64
    res = np.full((so.nframes(tree), nch), np.nan, dtype=res_dt, order="F")
    so.sink_into(res, tree)
    if res_dt == np.float64:
        assert relerr(res, want) < 1e-9
    else:
        assert relerr(res.astype(np.float64), want.astype(np.float32).astype(np.float64)) < 1e-7
    monkeypatch.setenv("SIGOPS_RS_NOGA", "1")
    ref = np.full_like(res, np.nan)
    so.sink_into(ref, tree)
    assert relerr(res.astype(np.float64), ref.astype(np.float64)) < (1e-10 if res_dt == np.float64 else 1e-7)


@pytest.mark.parametrize("nch,n,res_dt,gen,tone_len", [(8, 300_000, np.float64, dict(ω=1 * so.kHz), 0), (4, 120_000, np.float64, dict(ω=440 * so.Hz, ϕ=0.3), 0),
                                                      (8, 50_000, np.float32, dict(ω=5 * so.Hz), 0), (8, 470_000, np.float64, dict(ω=1 * so.kHz), 0),
                                                      (8, 60_000, np.float64, dict(ω=100 * so.Hz), 63_000), (4, 30_000, np.float64, dict(ω=100 * so.Hz), 20_000)])
def test_float32_array_plus_float64_generator_is_fused(nch, n, res_dt, gen, tone_len, monkeypatch):
    """`Mix(Signal(sin), x::Float32 array)` is a Float64 signal too: the GA instantiation ADDS the
    generator at the A operand (round 3; the headline pipeline with a Float32 leaf used to materialise
    the Mix with a K1 pass, 0.67 ms).  The zero extension of the resampler's input stays zero (the gain
    ring holds zeros outside the fused pieces), a tone longer than the array continues alone (a
    generated piece, staged as 0.0f), a shorter one ends inside the array (two pieces: K1 path)."""
    rng = np.random.default_rng(72)
    x = np.asfortranarray(rng.standard_normal((n, nch)).astype(np.float32))
    tone = so.Signal(so.sin, **gen)
    total = n
    if tone_len:
        tone = tone | so.Until(tone_len * so.frames)
        total = max(n, tone_len)
    tree = so.Mix(tone, so.Signal(x, 44.1 * so.kHz)) | so.Until(total * so.frames) | so.ToFramerate(48 * so.kHz)
    want = oracle_sink(tree)
    assert want.dtype == np.float64
    res = np.full((so.nframes(tree), nch), np.nan, dtype=res_dt, order="F")
    so.sink_into(res, tree)
    if res_dt == np.float64:
        assert relerr(res, want) < 1e-9
        assert relerr(res[-2000:], want[-2000:]) < 1e-9  # (the end: where the input's zero extension is read)
    else:
        assert relerr(res.astype(np.float64), want.astype(np.float32).astype(np.float64)) < 1e-7
    if tone_len == 0 or tone_len >= n:
        from sigops_amd.engine import Plan
        p = Plan(so.ToChannels(tree, nch), res.shape, res_dt, (1, res.shape[0]), False)
        names = [s_["name"] for s_ in p.steps()]
        p.close()
        assert names == ["k_resample_periodic"], names  # one launch: nothing materialised in front
    monkeypatch.setenv("SIGOPS_RS_NOGA", "1")
    ref = np.full_like(res, np.nan)
    so.sink_into(ref, tree)
    assert relerr(res.astype(np.float64), ref.astype(np.float64)) < (1e-10 if res_dt == np.float64 else 1e-7)


@pytest.mark.parametrize("nch,n,dt", [(8, 70_000, np.float64), (2, 33_333, np.float64), (5, 20_000, np.float32), (12, 9_000, np.float64),
                                     (1, 5_000, np.float64), (8, 600, np.float64)])
def test_interleaved_leaves_and_results(nch, n, dt):
    """frame-interleaved buffers (frame_stride = nch, chan_stride = 1: WAV data, row-major arrays,
    `PermutedDimsArray` views) through k_pointwise's LDS-transposing chain path: bit-exact
    against the planar evaluation"""
    rng = np.random.default_rng(73)
    x = rng.standard_normal((n, nch)).astype(dt)  # C order: interleaved frames
    xp = np.asfortranarray(x)                      # planar copy
    assert x.strides[1] == x.itemsize and (nch == 1 or x.strides[0] == nch * x.itemsize)

    def tree(a):
        return so.Signal(a, 8 * so.kHz) | so.Amplify(so.Signal(so.sin, ω=3 * so.Hz)) | so.Until(n * so.frames) | so.Ramp(20 * so.ms)

    want = oracle_sink(tree(xp))
    got_planar = so.sink(tree(xp))[0]
    got_il_leaf = so.sink(tree(x))[0]
    assert np.array_equal(got_il_leaf, got_planar)
    res = np.full((n, nch), np.nan, dtype=want.dtype, order="C")  # interleaved result
    so.sink_into(res, tree(x))
    assert np.array_equal(res, got_planar)
    assert relerr(res, want) < (1e-12 if dt == np.float64 else 2e-7)


def test_interleaved_device_result():
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(74)
    n, nch = 100_000, 8
    x = rng.standard_normal((n, nch))
    xd = torch.from_numpy(x).cuda()  # row-major on the device: interleaved leaf
    tree = so.Signal(xd, 44.1 * so.kHz) | so.Amplify(0.5)
    out = torch.full((n, nch), float("nan"), dtype=torch.float64, device="cuda")  # interleaved result
    so.sink_into(out, tree)
    assert np.array_equal(out.cpu().numpy(), x * 0.5)


def test_float32_normpower_reduction_order():
    """`Normpower` of a Float32 signal.  The reference reduces
    `mean(x -> float(x)^2, vals)` in Float32, pairwise over 1024-element blocks (Base.mapreduce_impl;
    the inner @simd loops make even the reference's value depend on the host's vector width).  The
    engine reduces in Float32 in the order of the oracle's restatement -- blocks of 1024 front to back,
    neighbours folded level by level (k_sumsq32_blocks / k_sumsq32_fold) -- so the two agree to the
    last bit of the rms; round 1 summed in Float64 and was 2e-6 off on the worst soak shape (a short
    signal ending in a long constant `lastframe` pad, where Float32 partial sums lose the most)."""
    rng = np.random.default_rng(77)
    x = np.asfortranarray((rng.standard_normal((584, 1)) * 0.3).astype(np.float32))
    tree = so.Pad(so.Signal(x, 1 * so.kHz), so.lastframe) | so.Until(60_000 * so.frames) | so.Normpower
    got = so.sink(tree)[0]
    want = oracle_sink(tree)
    assert got.dtype == want.dtype == np.float32
    assert np.array_equal(got, want)
    # ordinary noise, several blocks and channels, an odd number of blocks
    # (the last two caught a fused multiply-add in the block sums: 1 ulp of the rms)
    for shape in ((100_000, 2), (1158, 2), (5000, 3), (1_000_000, 8), (1_105_534, 4)):
        y = np.asfortranarray(rng.standard_normal(shape).astype(np.float32))
        t2 = so.Signal(y, 1 * so.kHz) | so.Normpower
        assert np.array_equal(so.sink(t2)[0], oracle_sink(t2)), shape


def _both_raise(tree):
    with pytest.raises(so.ErrorException) as want:
        oracle_sink(tree)
    with pytest.raises(so.ErrorException) as got:
        so.sink(tree)
    assert str(got.value)[:40] == str(want.value)[:40]


def test_errors_in_frames_nobody_uses():
    """The reference evaluates the frames `After` skips and the whole input blocks a filter reads
    (src/cutting.jl:160-173, src/filters.jl:221-262): errors raised there are raised by the engine too."""
    rng = np.random.default_rng(12)
    tone = so.Signal(so.sin, 8 * so.kHz, ω=16 * so.Hz) | so.Until(120 * so.frames)
    # After skips the mirror-padded region of something that is not an array
    _both_raise(so.Pad(tone, so.mirror) | so.Until(420 * so.frames) | so.After(300 * so.frames))
    # ... also through a Pad / Until / After chain and a channel map on top
    _both_raise(so.Pad(tone, so.mirror) | so.Until(420 * so.frames) | so.Pad(2.5) | so.Until(1056 * so.frames)
                | so.After(729 * so.frames) | so.ToChannels(3))
    # a filter reads its input in blocks of 4096 frames: the padding is evaluated although 13 outputs are used
    x = so.Signal(np.asfortranarray(rng.standard_normal((279, 3))), 100 * so.Hz) | so.Amplify(0.5)
    _both_raise(so.Pad(x, so.cycle) | so.Until(388 * so.frames) | so.Filt(so.Lowpass, 20 * so.Hz) | so.Until(13 * so.frames))
    # the same trees without the offending padding are fine
    ok = so.Pad(x, so.zero) | so.Until(388 * so.frames) | so.Filt(so.Lowpass, 20 * so.Hz) | so.Until(13 * so.frames)
    assert relerr(so.sink(ok, so.Array), oracle_sink(ok)) < 1e-9


def test_after_longer_than_a_child_that_is_never_evaluated():
    """`After` raises from its first block (src/cutting.jl:174-181): inside a Mix whose result is empty
    nobody asks for one; at the root of the tree the sink always does (src/sink.jl:225-226)."""
    short = so.Signal(so.sin, 6 * so.kHz, ω=24 * so.Hz) | so.Until(4 * so.frames) | so.After(7 * so.frames)
    tree = so.Ramp(short, 5 * so.frames)
    want = oracle_sink(tree)
    got = so.sink(tree, so.Array)
    assert got.shape == want.shape == (0, 1)
    _both_raise(short)
    _both_raise(so.Append(short, so.Signal(np.ones(5), 6 * so.kHz)))


def test_lastframe_pad_after_an_append_whose_last_child_is_empty():
    x = np.asfortranarray(np.arange(10.0).reshape(5, 2))
    y = np.asfortranarray(100 + np.arange(20.0).reshape(10, 2))
    t = so.Pad(so.Append(so.Signal(x, 50 * so.Hz), so.Signal(y, 50 * so.Hz) | so.Until(0 * so.frames)), so.lastframe) | so.Until(8 * so.frames)
    assert np.array_equal(so.sink(t, so.Array), oracle_sink(t))


@pytest.mark.parametrize("nch", [8, 4])
def test_fused_float32_gain_on_a_signal_of_many_tiles(nch):
    """Float32 array x Float64 sine gain fused into the periodic resampler (GA instantiation) on a signal
    long enough for every workgroup to wrap its three gain arrays (> 512 tiles).  The gains of a tile are
    indexed from the 128-byte aligned frame it is staged from -- up to 31 frames below its first input for
    Float32 tiles; with 16 reserved they ran into the array the compute waves were reading (found by the
    BlockStream soak: 1.6e-2 off, run to run different)."""
    rng = np.random.default_rng(91)
    n = 454382
    x = np.asfortranarray(rng.standard_normal((n, nch)).astype(np.float32))
    tree = so.Signal(x, 44.1 * so.kHz) | so.Amplify(so.Signal(so.sin, ω=3 * so.Hz)) | so.Until(n * so.frames) | so.ToFramerate(48 * so.kHz)
    want = oracle_sink(tree)
    for _ in range(3):
        assert relerr(so.sink(tree, so.Array), want) <= 1e-9


@pytest.mark.parametrize("nch", [1, 2])
def test_filtering_spectral_inequalities_on_the_engine(nch):
    """runtests.jl:314-350 restated (tests/spectral_checks.py), evaluated by the HIP engine; and the same
    trees agree with the oracle"""
    from spectral_checks import filtering_inequalities

    high = filtering_inequalities(lambda t: so.sink(t, so.Array), nch)
    high_o = filtering_inequalities(oracle_sink, nch)
    assert relerr(high, high_o) < 1e-9


def test_opaque_closures_are_materialised_on_the_host():
    """what the engine cannot lower is materialised on the host and passed as an
    array leaf -- `Signal(fn)` with an arbitrary closure (reference src/functions.jl:53-60) and
    `OperateOn(fn, xs...)` (src/mapsignal.jl:131-145) -- instead of sending the whole tree to the CPU.
    The closure runs in NumPy; its operands and everything above it run on the GPU."""
    rng = np.random.default_rng(81)
    x = np.asfortranarray(rng.standard_normal((30_000, 2)))
    cube = lambda t: np.sin(t) ** 3  # noqa: E731
    tone = so.Signal(cube, 44.1 * so.kHz, ω=440 * so.Hz)
    tree = so.Mix(tone, so.Signal(x, 44.1 * so.kHz)) | so.Until(30_000 * so.frames) | so.Filt(so.Lowpass, 3 * so.kHz) | so.ToFramerate(48 * so.kHz)
    got = so.sink(tree, so.Array)
    # the same tree with the closure's values supplied as data (what the host evaluation must equal)
    # (the tone is infinite: the resampler under the filter reads it a little beyond frame 30 000)
    t = np.arange(1, 30_201) / 44100.0
    vals = cube(2 * np.pi * np.fmod(t * 440.0 + 0.0, 1.0)).reshape(-1, 1)
    ref_tree = (so.Mix(so.Signal(np.asfortranarray(vals), 44.1 * so.kHz), so.Signal(x, 44.1 * so.kHz)) | so.Until(30_000 * so.frames)
                | so.Filt(so.Lowpass, 3 * so.kHz) | so.ToFramerate(48 * so.kHz))
    want = oracle_sink(ref_tree)
    assert got.shape == want.shape and relerr(got, want) < 1e-9
    # a closure without a frequency sees t + phase; first frame t = 1/fs (runtests.jl:564-566 semantics)
    ramp = so.sink(so.Signal(lambda t: 2 * t, 10 * so.Hz) | so.Until(5 * so.frames), so.Array)
    assert np.allclose(ramp[:, 0], 2 * np.arange(1, 6) / 10.0, rtol=0, atol=1e-15)
    # OperateOn with a closure: operands on the GPU (one of them filtered), padded with the map's padding
    a = so.Signal(x, 10 * so.kHz) | so.Filt(so.Highpass, 1 * so.kHz)
    b = so.Signal(np.asfortranarray(rng.standard_normal((20_000, 2))), 10 * so.kHz)
    m = so.OperateOn(lambda u, v: np.maximum(u, v), a, b) | so.Amplify(0.5)
    got_m = so.sink(m, so.Array)
    fa = oracle_sink(a)
    fb = np.vstack([oracle_sink(b), np.zeros((10_000, 2))])  # default padding of an opaque map: zero
    assert got_m.shape == (30_000, 2) and relerr(got_m, 0.5 * np.maximum(fa, fb)) < 1e-9
    # bychannel=false: the closure sees whole frames
    sw = so.sink(so.OperateOn(lambda fr: (fr[1], fr[0] + fr[1]), so.Signal(x, 10 * so.kHz), bychannel=False), so.Array)
    assert np.array_equal(sw, np.column_stack([x[:, 1], x[:, 0] + x[:, 1]]))


@pytest.mark.parametrize("nch,n,build", [
    (2, 300_000, lambda x, n: so.Mix(so.Signal(so.sin, ω=1 * so.kHz) | so.Until(n * so.frames), x)),
    (2, 300_000, lambda x, n: so.Mix(x, so.Signal(so.sin, ω=1 * so.kHz, ϕ=0.25)) | so.Until(n * so.frames)),
    (8, 120_000, lambda x, n: so.Amplify(x, so.Signal(so.sin, ω=5 * so.Hz)) | so.Until(n * so.frames)),
    (1, 70_001, lambda x, n: so.Mix(so.Signal(so.sin), x) | so.Until(n * so.frames)),  # sin without a frequency: a 1 Hz tone
])
def test_filter_forms_mix_with_a_sine_in_its_own_loads(nch, n, build, monkeypatch):
    """`Mix(Signal(sin), x) |> Filt` (BASELINE configs 2 and 4): the IIR adds / multiplies the generator
    while it loads its input (SosGeom::src_op) -- one rotation per frame from the exact value at the start
    of every chunk -- instead of reading a K1-materialised copy.  Same values as the materialising path."""
    from sigops_amd.engine import Plan

    rng = np.random.default_rng(95)
    x = so.Signal(np.asfortranarray(rng.standard_normal((n, nch))), 44.1 * so.kHz)
    tree = build(x, n) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz)
    want = oracle_sink(tree)
    res = np.empty((n, nch), order="F")
    p = Plan(so.ToChannels(tree, nch), res.shape, res.dtype, (1, n), False)
    p.execute(res.ctypes.data)
    names = [s_["name"] for s_ in p.steps()]
    p.close()
    assert names == ["k_sos"], names
    # (the reference rounds the phase (n/fs)*omega per frame: at 7e3 cycles that is +-1e-12 of a cycle of
    #  noise of its own, which a rotation from an exact chunk start does not reproduce)
    assert relerr(res, want) < 1e-10
    # a window far into the signal (warm start: the generator's phase follows the stage's first frame)
    a, m = n - 50_000, 20_000
    w = so.sink(tree | so.After(a * so.frames) | so.Until(m * so.frames), so.Array)
    assert relerr(w, want[a:a + m]) < 1e-10
    monkeypatch.setenv("SIGOPS_SOS_NOSRC", "1")
    ref = so.sink(tree, so.Array)
    assert relerr(res, ref) < 1e-10


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("shape", ["mix2", "mix3", "amp_of_mix", "mixed_types", "gain_and_arrays", "four_arrays"])
def test_maps_over_several_arrays(dt, shape, monkeypatch):
    """`Mix(a, b)`, `Amplify(Mix(a, b), c)` ...: maps over two to four arrays, Float32 arithmetic rounded per operation
    (reference src/mapsignal.jl:249-272 evaluates every child per frame) -- against the oracle, and the specialised
    (hipRTC) form of the same step bit for bit against the interpreter.  (A chain-path form of K1 with array operands
    -- all of a batch's loads in flight together -- was built in round 3 and measured SLOWER than the interpreter:
    Mix(a, b) 3.6-4.0 against 4.65 TB/s at 156-204 VGPRs; removed.)"""
    rng = np.random.default_rng(77)
    n, nch = 70001, 3
    mk = lambda d=dt: so.Signal(np.asfortranarray(rng.standard_normal((n, nch)).astype(d)), 44.1 * so.kHz)
    a, b, c, d = mk(), mk(), mk(), mk()
    tone = so.Signal(so.sin, 44.1 * so.kHz, ω=440 * so.Hz) | so.Until(n * so.frames)
    tree = {"mix2": lambda: so.Mix(a, b),
            "mix3": lambda: so.Mix(a, b, c),
            "amp_of_mix": lambda: so.Amplify(so.Mix(a, b), c),
            "mixed_types": lambda: so.Mix(a, mk(np.float32 if dt == np.float64 else np.float64)),
            "gain_and_arrays": lambda: so.Amplify(so.Mix(a, tone), b) | so.Ramp(20 * so.ms),
            "four_arrays": lambda: so.Mix(a, b, c, d)}[shape]() | so.After(13 * so.frames) | so.Until(60001 * so.frames)
    got = so.sink(tree)[0]
    want = oracle_sink(tree)
    assert got.shape == want.shape and got.dtype == want.dtype
    assert relerr(got, want) <= (1e-12 if got.dtype == np.float64 else 1e-6)
    monkeypatch.setenv("SIGOPS_RTC", "1")
    special = so.sink(tree)[0]
    assert np.array_equal(got, special)
