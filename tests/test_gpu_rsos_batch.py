"""Several plain filters through the one-pass kernel in ONE launch (round 6, k_rsos_batch in csrc/k_rsos.hip): the scenes under
an `Append` (reference src/appending.jl:59-76: every child evaluated on its own, filter state included, src/filters.jl:252-255)
each get `gpm` workgroups of one grid instead of three batched passes (tests/test_gpu_sos_batch.py) -- config 4's 64 scenes:
1.30 ms against 1.63.  The planner's estimate picks the form; `SIGOPS_RSOS_BATCH=1` takes it whenever it fits (short scenes
here), `=0` never.  Asserted: the oracle's values for ragged lengths, channel counts 1-8, Float64 and Float32, fused sines,
members of different filters / instantiations, results that move (graph replay) and arrays that are replaced, and -- member
by member -- the reference's own set of non-finite outputs."""
import numpy as np
import pytest

import sigops_amd as so
from oracle_bridge import oracle_semantics, oracle_sink, relerr
from test_gpu_rsos import env

pytestmark = pytest.mark.gpu


def _noise(rng, n, nch, dt=np.float64):
    return np.asfortranarray(rng.standard_normal((n, nch)).astype(dt))


def _steps(tree, nch, dt=np.float64):
    import torch
    n = so.nframes(tree)
    out = torch.empty((nch, n), dtype=torch.float64 if dt == np.float64 else torch.float32, device="cuda")
    p = so.Plan(so.ToChannels(tree, nch), (n, nch), dt, (1, n), True)
    p.set_profiling(True)
    p.execute(out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    names = [(s["name"], s["launches"]) for s in p.steps()]
    p.close()
    return names


def _check(tree, tol=1e-9):
    with env(SIGOPS_RSOS_BATCH=1):
        got = so.sink(tree, so.Array)
    with env(SIGOPS_RSOS_BATCH=0):
        three = so.sink(tree, so.Array)
    with oracle_semantics("intended"):
        want = oracle_sink(tree)
    assert got.shape == want.shape and got.dtype == three.dtype
    assert relerr(got, want) <= tol
    assert relerr(got, three) <= tol
    return got


@pytest.mark.parametrize("nch", [1, 2, 4, 5, 8])
@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_scenes_of_ragged_lengths(nch, dt):
    """members of many periods next to members too short for the form (less than a period of 160 frames, three frames)"""
    rng = np.random.default_rng(300 + nch)
    kids = [so.Signal(_noise(rng, n, nch, dt), 44.1 * so.kHz) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz)
            for n in (50001, 17, 7777, 123456, 3, 160, 161, 30000)]
    tree = so.Append(*kids)
    _check(tree, tol=2e-6 if dt == np.float32 else 1e-9)
    with env(SIGOPS_RSOS_BATCH=1):
        names = _steps(tree, nch, dt)
    # (the members of fewer than 4096 samples have no carrier: they stay with the three-pass batch)
    assert ("k_rsos_batch", 2) in names and ("k_sos_batch", 4) in names and not any(n in ("k_sos", "k_rsos") for n, _ in names), names


def test_config4_scenes():
    """config 4's scene: Mix(sin, noise) |> Filt |> Ramp -- the sine added by the step waves (16-wave instantiation), the ramps in
    place over the windows the members wrote"""
    rng = np.random.default_rng(31)
    kids = []
    for k in range(7):
        n = 60000 + 1001 * k
        nz = so.Signal(_noise(rng, n, 2), 44.1 * so.kHz)
        tone = so.Signal(so.sin, 44.1 * so.kHz, ω=(500.0 + 25 * k) * so.Hz) | so.Until(n * so.frames)
        kids.append(so.Mix(tone, nz) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.Ramp(10 * so.ms))
    tree = so.Append(*kids)
    _check(tree, tol=1e-10)
    with env(SIGOPS_RSOS_BATCH=1):
        assert ("k_rsos_batch", 2) in _steps(tree, 2)


def test_more_members_than_compute_units():
    """300 members: one workgroup each, the grid in two rounds"""
    rng = np.random.default_rng(32)
    kids = [so.Signal(_noise(rng, 2100 + 7 * k, 2), 44.1 * so.kHz) | so.Filt(so.Lowpass, 3 * so.kHz) for k in range(300)]
    _check(so.Append(*kids))


def test_different_filters_and_orders_in_one_launch():
    """the members' coefficients, sections and warm-ups are their own (RsosItem::g)"""
    rng = np.random.default_rng(33)
    kids = []
    for k, (lo, hi, order) in enumerate([(0.5, 2.0, 3), (1.0, 4.0, 5), (0.2, 0.9, 2), (3.0, 8.0, 6), (2.0, 2.5, 4)]):
        kids.append(so.Signal(_noise(rng, 40000 + 999 * k, 2), 44.1 * so.kHz) | so.Filt(so.Bandpass, lo * so.kHz, hi * so.kHz, order=order))
    tree = so.Append(*kids)
    _check(tree)
    with env(SIGOPS_RSOS_BATCH=1):
        names = [n for n, _ in _steps(tree, 2)]
    assert names.count("k_rsos_batch") == 1 and "k_sos_batch" not in names, names


def test_members_that_read_buffers_and_members_mixed_together():
    """members whose results a later launch reads (no windows of the result)"""
    rng = np.random.default_rng(34)
    a = so.Signal(_noise(rng, 40000, 2), 44.1 * so.kHz) | so.Filt(so.Lowpass, 1 * so.kHz)
    b = so.Signal(_noise(rng, 40000, 2), 44.1 * so.kHz) | so.Filt(so.Highpass, 5 * so.kHz)
    c = so.Signal(_noise(rng, 40000, 2), 44.1 * so.kHz) | so.Filt(so.Bandpass, 2 * so.kHz, 3 * so.kHz, order=5)
    _check(so.Mix(a, b, c))
    _check(so.Amplify(a, b))


def test_members_that_start_inside_their_array():
    rng = np.random.default_rng(35)
    kids = []
    for k in range(4):
        x = so.Signal(_noise(rng, 90000, 2), 44.1 * so.kHz)
        kids.append(x | so.After((100 + 37 * k) * so.frames) | so.Until((50000 + k) * so.frames) | so.Filt(so.Lowpass, 4 * so.kHz))
    _check(so.Append(*kids))


def test_two_instantiations_in_one_plan():
    """two-channel scenes with a fused sine (sixteen waves: the step waves) and plain ones (twelve) are launched apart"""
    rng = np.random.default_rng(36)
    kids = []
    for k in range(3):
        n = 30000 + 10 * k
        nz = so.Signal(_noise(rng, n, 2), 44.1 * so.kHz)
        tone = so.Signal(so.sin, 44.1 * so.kHz, ω=700.0 * so.Hz) | so.Until(n * so.frames)
        kids.append(so.Mix(tone, nz) | so.Filt(so.Lowpass, 2 * so.kHz))
    for k in range(3):
        kids.append(so.Signal(_noise(rng, 31000 + k, 2), 44.1 * so.kHz) | so.Filt(so.Highpass, 1 * so.kHz))
    tree = so.Append(*kids)
    _check(tree)
    with env(SIGOPS_RSOS_BATCH=1):
        names = [n for n, _ in _steps(tree, 2)]
    assert names.count("k_rsos_batch") == 2, names


def test_the_launch_table_follows_result_and_arrays():
    """the table holds result and carrier pointers: executes into two results alternate (direct launches and graph replays),
    then an array is replaced (so_plan_set_array)"""
    import torch
    rng = np.random.default_rng(37)
    n, nsc = 50000, 5
    arrays = [torch.from_numpy(np.ascontiguousarray(_noise(rng, n, 2).T)).cuda() for _ in range(nsc)]
    pipe = lambda s: s | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.Ramp(5 * so.ms)
    tree = so.Append(*[pipe(so.Signal(a.t(), 44.1 * so.kHz)) for a in arrays])
    total = so.nframes(tree)
    with oracle_semantics("intended"):
        want = oracle_sink(so.Append(*[pipe(so.Signal(np.asfortranarray(a.t().cpu().numpy()), 44.1 * so.kHz)) for a in arrays]))
    with env(SIGOPS_RSOS_BATCH=1):
        p = so.Plan(so.ToChannels(tree, 2), (total, 2), np.float64, (1, total), True)
    st = torch.cuda.current_stream().cuda_stream
    A = torch.zeros((2, total), dtype=torch.float64, device="cuda")
    B = torch.zeros((2, total), dtype=torch.float64, device="cuda")
    for dst in (A, A, A, B, A, B, B, B, A):
        dst.zero_()
        p.execute(dst.data_ptr(), st)
        torch.cuda.synchronize()
        assert relerr(np.asfortranarray(dst.t().cpu().numpy()), want) <= 1e-9
    repl = torch.from_numpy(np.ascontiguousarray(_noise(rng, n, 2).T)).cuda()
    p.set_array(2, repl.t())
    hosts = [np.asfortranarray((repl if k == 2 else a).t().cpu().numpy()) for k, a in enumerate(arrays)]
    with oracle_semantics("intended"):
        want2 = oracle_sink(so.Append(*[pipe(so.Signal(h, 44.1 * so.kHz)) for h in hosts]))
    for dst in (A, A, A, B):
        dst.zero_()
        p.execute(dst.data_ptr(), st)
        torch.cuda.synchronize()
        assert relerr(np.asfortranarray(dst.t().cpu().numpy()), want2) <= 1e-9
    p.check()
    p.close()


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_non_finite_samples_member_by_member(dt):
    """a NaN in one member, an Inf in another, twice in a third: each member's non-finite outputs are the reference's set (from the
    first output whose input is non-finite to the member's end: the fix-up launch, one z-slice per member), the others are
    untouched"""
    rng = np.random.default_rng(38)
    datas = [rng.standard_normal((70000 + 13 * k, 2)).astype(dt) for k in range(6)]
    datas[1][12345, 0] = np.nan
    datas[3][60001, 1] = np.inf
    datas[4][160 * 200 + 15, 0] = -np.inf
    datas[4][50000, 0] = np.nan
    datas[4][100, 1] = np.nan
    kids = [so.Signal(np.asfortranarray(d), 44.1 * so.kHz) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) for d in datas]
    with env(SIGOPS_RSOS_BATCH=1):
        got = so.sink(so.Append(*kids), so.Array)
    # (scene by scene: the reference's Append never leaves a filtered child longer than one filter block -- quirk C-7,
    #  tests/test_gpu_configs.py::test_config4_miniature)
    want = np.concatenate([oracle_sink(k) for k in kids])
    bad_g, bad_w = ~np.isfinite(got), ~np.isfinite(want)
    assert bad_w.any() and np.array_equal(bad_g, bad_w)
    assert relerr(got[~bad_g], want[~bad_w]) <= (2e-6 if dt == np.float32 else 1e-9)
    # ... and a clean execute after it reports nothing (the words are reset by every launch)
    clean = so.Append(*[so.Signal(np.asfortranarray(np.nan_to_num(d, nan=0.0, posinf=0.0, neginf=0.0)), 44.1 * so.kHz)
                        | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) for d in datas])
    with env(SIGOPS_RSOS_BATCH=1):
        assert np.isfinite(so.sink(clean, so.Array)).all()


def test_host_result():
    """a host result: the members write windows of the staging buffer"""
    rng = np.random.default_rng(39)
    kids = [so.Signal(_noise(rng, 30000 + k, 2), 44.1 * so.kHz) | so.Filt(so.Lowpass, 3 * so.kHz) | so.Ramp(5 * so.ms) for k in range(4)]
    tree = so.Append(*kids)
    with env(SIGOPS_RSOS_BATCH=1):
        got = so.sink(tree)[0]
    with oracle_semantics("intended"):
        want = oracle_sink(tree)
    assert relerr(got, want) <= 1e-9


def test_the_estimate_takes_the_launch_at_config4_scale_only():
    """by default: short scenes stay in the three-pass batch (a workgroup's warm-up would outweigh them)"""
    rng = np.random.default_rng(40)
    kids = [so.Signal(_noise(rng, 40000 + k, 2), 44.1 * so.kHz) | so.Filt(so.Lowpass, 3 * so.kHz) for k in range(6)]
    with env(SIGOPS_RSOS_BATCH=None):
        names = [n for n, _ in _steps(so.Append(*kids), 2)]
    assert "k_sos_batch" in names and "k_rsos_batch" not in names


def test_config4_full_size():
    """BASELINE.json configs[3] itself: 64 scenes of 60 s x 2 channels.  The default plan is the batched launch; four scenes --
    the first, two inside, the last -- whole against the oracle on host copies of the noise the engine filtered, and every scene's
    ramps and energy through what the oracle cannot reach at this size: the result is finite, starts and ends at zero scene by
    scene, and equals the three-pass form's to rounding"""
    import torch
    import bench
    nsc, nch, n = 64, 2, 2_646_000
    noises = []
    for k in range(nsc):
        g = torch.Generator(device="cuda")
        g.manual_seed(1983 + k)
        noises.append(torch.randn((nch, n), dtype=torch.float64, device="cuda", generator=g))
    tree = so.Append(*[bench.scene(so, nz.t(), k, n) for k, nz in enumerate(noises)])
    total = nsc * n
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for e in (None, 0):
        with env(SIGOPS_RSOS_BATCH=e):
            p = so.Plan(so.ToChannels(tree, nch), (total, nch), np.float64, (1, total), True)
        p.set_profiling(True)
        out = torch.empty((nch, total), dtype=torch.float64, device="cuda")
        p.execute(out.data_ptr(), st)
        torch.cuda.synchronize()
        p.check()
        names = [s["name"] for s in p.steps()]
        assert ("k_rsos_batch" in names) == (e is None), names
        p.close()
        outs.append(out)
    a, b = outs
    assert bool(torch.isfinite(a).all())
    # (the two forms evaluate the scenes' sines at arguments that round differently: 4e-11 of the largest sample at most, and
    #  either is 5e-11 from the oracle's by a scene's end)
    assert float((a - b).abs().max() / b.abs().max()) <= 1e-9
    edges = a.view(nch, nsc, n)
    assert float(edges[:, :, 0].abs().max()) == 0.0 and float(edges[:, :, -1].abs().max()) <= 1e-2  # (the ramps: 0 at a scene's first frame)
    for k in (0, 21, 42, 63):
        host = np.asfortranarray(noises[k].t().cpu().numpy())
        want = oracle_sink(bench.scene(so, host, k, n))
        got = a[:, k * n:(k + 1) * n].t().cpu().numpy()
        assert relerr(got, want) <= 1e-10, k


def test_filters_that_read_a_stages_buffer_keep_launches_of_their_own():
    """the batch's step stands where its first member stood and waits for nothing: a filter whose source is another stage's
    buffer -- here a long-period resampler's (44.1 -> 16 kHz runs the row-tiled kernel, which the filter does not fuse with) --
    could run before its producer, so it is not a member (found by running the whole suite with SIGOPS_RSOS_BATCH=1:
    tests/test_gpu_fuzz.py's multirate trees returned NaN)"""
    rng = np.random.default_rng(41)
    kids = [so.Signal(_noise(rng, 60000 + 17 * k, 2), 44.1 * so.kHz) | so.ToFramerate(16 * so.kHz) | so.Filt(so.Lowpass, 3 * so.kHz) for k in range(3)]
    kids += [so.Signal(_noise(rng, 30000 + k, 2), 16 * so.kHz) | so.Filt(so.Lowpass, 3 * so.kHz) for k in range(3)]
    tree = so.Append(*kids)
    with env(SIGOPS_RSOS_BATCH=1):
        names = [n for n, _ in _steps(tree, 2)]
        got = so.sink(tree, so.Array)
    want = np.concatenate([oracle_sink(k) for k in kids])
    assert np.isfinite(got).all() and relerr(got, want) <= 1e-9
    # (the three filters over arrays are one batch; the three behind resamplers are not in it)
    assert names.count("k_rsos_batch") == 1, names
    i_batch = names.index("k_rsos_batch")
    assert any(n.startswith("k_resample") for n in names), names
    assert sum(1 for n in names if n in ("k_sos", "k_rsos", "k_sos_batch")) >= 1, names
    del i_batch
