"""GPU parity for warm starts: a stateful stage whose first frames nobody reads (`After`, a later
block of `so.stream`) starts from zero state a decay time before the first frame that is read
instead of at frame 0.  The reference runs the skipped frames through the filter
(src/cutting.jl:160-173); the oracle does the same, so these are plain parity tests."""
import os

import numpy as np
import pytest

import sigops_amd as so
from oracle_bridge import oracle_sink, relerr

pytestmark = pytest.mark.gpu


def _noise(rng, n, nch, dt=np.float64):
    return np.asfortranarray(rng.standard_normal((n, nch)).astype(dt))


def _check(tree, tol=1e-9):
    want = oracle_sink(tree)
    got = so.sink(tree, so.Array)
    os.environ["SIGOPS_NO_WARM_START"] = "1"
    try:
        ref = so.sink(tree, so.Array)
    finally:
        os.environ.pop("SIGOPS_NO_WARM_START", None)
    assert got.shape == want.shape and got.dtype == want.dtype
    assert relerr(ref, want) <= tol
    assert relerr(got, want) <= tol
    assert relerr(got, ref) <= 1e-13 if got.dtype == np.float64 else 1e-6
    return got


@pytest.mark.parametrize("nch", [1, 2, 8])
def test_filter_then_after(nch):
    rng = np.random.default_rng(21 + nch)
    x = so.Signal(_noise(rng, 90000, nch), 44.1 * so.kHz)
    _check(x | so.Filt(so.Lowpass, 3 * so.kHz) | so.After(70001 * so.frames))
    _check(x | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.After(65000 * so.frames) | so.Until(9999 * so.frames))


def test_cascaded_groups_and_chained_filters():
    rng = np.random.default_rng(22)
    x = so.Signal(_noise(rng, 120000, 2), 44.1 * so.kHz)
    # order 10 band-pass = 10 sections = two cascaded groups of the IIR kernel
    _check(x | so.Filt(so.Bandpass, 1 * so.kHz, 4 * so.kHz, order=10) | so.After(100000 * so.frames))
    # filter reading a filter (direct stage-buffer source), both warm-started
    _check(x | so.Filt(so.Highpass, 0.3 * so.kHz) | so.Filt(so.Lowpass, 5 * so.kHz) | so.After(90000 * so.frames))
    # pointwise work between them and a gain after
    _check(x | so.Filt(so.Highpass, 0.3 * so.kHz) | so.Amplify(0.5) | so.Filt(so.Lowpass, 5 * so.kHz)
           | so.After(90000 * so.frames) | so.Ramp(100 * so.frames))


def test_float32_and_mixed_consumers():
    rng = np.random.default_rng(23)
    x = so.Signal(_noise(rng, 80000, 2, np.float32), 44.1 * so.kHz)
    _check(x | so.Filt(so.Lowpass, 3 * so.kHz) | so.After(60000 * so.frames), tol=1e-6)
    y = so.Signal(_noise(rng, 80000, 2), 44.1 * so.kHz)
    f = y | so.Filt(so.Lowpass, 3 * so.kHz)
    # the same stage read at two offsets: the earlier one decides where it starts
    _check(so.Mix(f | so.After(60000 * so.frames), f | so.After(30000 * so.frames) | so.Until(20000 * so.frames)))
    # under an Append, next to an unfiltered child
    _check(so.Append(y | so.Until(5000 * so.frames), f | so.After(70000 * so.frames)))


@pytest.mark.parametrize("nch", [1, 2, 8])
@pytest.mark.parametrize("rates", [(44100, 48000), (44100, 16000), (8000, 16000), (12000, 8000)])
def test_resample_then_after(nch, rates):
    fi, fo = rates
    rng = np.random.default_rng(31 + nch)
    x = so.Signal(_noise(rng, 100000, nch), fi * so.Hz)
    n_out = so.nframes(x | so.ToFramerate(fo * so.Hz))
    skip = int(0.8 * n_out) + 7
    _check(x | so.ToFramerate(fo * so.Hz) | so.After(skip * so.frames))
    # a short window in the middle (the thread-per-output kernel)
    _check(x | so.ToFramerate(fo * so.Hz) | so.After((skip // 2) * so.frames) | so.Until(500 * so.frames))


def test_pipeline_window():
    """the north-star pipeline, a window near its end"""
    rng = np.random.default_rng(41)
    n = 300000
    x = so.Signal(_noise(rng, n, 8), 44.1 * so.kHz)
    tree = (so.Mix(so.Signal(so.sin, ω=1 * so.kHz), x) | so.Until(n * so.frames)
            | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.ToFramerate(48 * so.kHz))
    _check(tree | so.After(250000 * so.frames) | so.Until(60000 * so.frames))
    _check(tree | so.After(250001 * so.frames))


def test_fir_and_gain_under_a_window():
    rng = np.random.default_rng(42)
    x = so.Signal(_noise(rng, 70000, 2), 44.1 * so.kHz)
    h = np.hanning(33) / np.hanning(33).sum()
    _check(so.Filt(x, h) | so.After(60000 * so.frames))
    env = so.Signal(so.sin, ω=5 * so.Hz)
    _check(x | so.Amplify(env) | so.Until(70000 * so.frames) | so.ToFramerate(48 * so.kHz) | so.After(65000 * so.frames))


def test_warm_start_really_skips_work():
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(43)
    n = 400000
    x = so.Signal(_noise(rng, n, 2), 44.1 * so.kHz)
    tree = so.ToChannels(x | so.Filt(so.Lowpass, 3 * so.kHz) | so.ToFramerate(48 * so.kHz) | so.After(400000 * so.frames), 2)
    m = so.nframes(tree)
    out = torch.empty((2, m), dtype=torch.float64, device="cuda")

    def work():
        p = so.Plan(tree, (m, 2), np.float64, (1, m), True, device=0)
        p.set_profiling(True)
        p.execute(out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        b = sum(s["algorithmic_bytes"] for s in p.steps())
        p.close()
        return b

    warm = work()
    os.environ["SIGOPS_NO_WARM_START"] = "1"
    try:
        cold = work()
    finally:
        os.environ.pop("SIGOPS_NO_WARM_START", None)
    assert warm < 0.25 * cold


def _stream_matches(tree, blocksize, tol=1e-12):
    whole = so.sink(tree, so.Array)
    blocks = list(so.stream(tree, blocksize, so.Array))
    assert all(b.shape[0] == blocksize for b in blocks[:-1]) and 0 < blocks[-1].shape[0] <= blocksize
    got = np.concatenate(blocks, axis=0)
    assert got.shape == whole.shape and got.dtype == whole.dtype
    assert relerr(got, whole) <= tol
    return got


@pytest.mark.parametrize("blocksize", [10007, 65536])
def test_stream_pipeline_blocks_concatenate_to_the_whole(blocksize):
    rng = np.random.default_rng(51)
    n = 200000
    x = so.Signal(_noise(rng, n, 8), 44.1 * so.kHz)
    tree = (so.Mix(so.Signal(so.sin, ω=1 * so.kHz), x) | so.Until(n * so.frames)
            | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.ToFramerate(48 * so.kHz))
    got = _stream_matches(tree, blocksize)
    assert relerr(got, oracle_sink(tree)) <= 1e-9


def test_stream_structural_trees_are_bit_exact():
    rng = np.random.default_rng(52)
    a = so.Signal(_noise(rng, 7001, 2), 8 * so.kHz)
    b = so.Signal(_noise(rng, 3333, 2), 8 * so.kHz)
    tree = so.Append(a | so.After(11 * so.frames), so.Pad(b, so.mirror) | so.Until(9000 * so.frames)) | so.Ramp(100 * so.frames)
    whole = so.sink(tree, so.Array)
    got = np.concatenate(list(so.stream(tree, 1000, so.Array)), axis=0)
    assert np.array_equal(got, whole)


def test_stream_filtered_children_under_append_and_device_blocks():
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(53)
    kids = [so.Signal(_noise(rng, n, 2), 44.1 * so.kHz) | so.Filt(so.Lowpass, 3 * so.kHz) | so.Ramp(5 * so.ms)
            for n in (30001, 50000, 41111)]
    tree = so.Append(*kids)
    whole = so.sink(tree, so.Array)
    blocks = [b for b, fs in so.stream(tree, 25000, "torch")]
    assert all(b.is_cuda for b in blocks)
    got = np.concatenate([b.cpu().numpy() for b in blocks], axis=0)
    assert relerr(got, whole) <= 1e-12


def test_stream_of_an_infinite_signal():
    import itertools

    tree = so.Signal(so.sin, 44.1 * so.kHz, ω=440 * so.Hz) | so.Filt(so.Highpass, 100 * so.Hz) | so.ToFramerate(48 * so.kHz)
    blocks = list(itertools.islice(so.stream(tree, 48000, so.Array), 4))
    whole = so.sink(tree | so.Until(4 * 48000 * so.frames), so.Array)
    # (a 100 Hz high-pass at 48 kHz: large, slowly decaying states; chunk scan and warm start both cut at 2^-70 of them)
    assert relerr(np.concatenate(blocks, axis=0), whole) <= 1e-10


@pytest.mark.parametrize("last", ["filt"])
def test_window_of_a_last_stage_is_written_by_the_stage_itself(last):
    """a block of a stream whose last stage is an IIR: no copy launch after it (the same for the periodic
    resampler was measured and dropped: the extra store condition cost its kernel 2 %)"""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(61)
    n = 300000
    x = so.Signal(_noise(rng, n, 8), 44.1 * so.kHz)
    tree = x | so.Filt(so.Lowpass, 3 * so.kHz) if last == "filt" else x | so.Amplify(so.Signal(so.sin, ω=5 * so.Hz)) | so.Until(n * so.frames) | so.ToFramerate(48 * so.kHz)
    blk = so.ToChannels(tree | so.After(200003 * so.frames) | so.Until(50000 * so.frames), 8)
    out = torch.full((8, 50000 + 7), float("nan"), dtype=torch.float64, device="cuda")
    p = so.Plan(blk, (50000, 8), np.float64, (1, 50000 + 7), True, device=0)
    p.set_profiling(True)
    p.execute(out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    names = [s["name"] for s in p.steps()]
    p.close()
    assert "k_pointwise" not in names, names
    want = oracle_sink(blk)
    assert relerr(out[:, :50000].t().cpu().numpy(), want) <= 1e-9
    assert bool(torch.isnan(out[:, 50000:]).all())


def _push_all(bs, x, sizes):
    outs, pos = [], 0
    for m in sizes:
        outs.append(bs.push(x[pos:pos + m]).cpu().numpy())
        pos += m
    assert pos == x.shape[0]
    outs.append(bs.finish().cpu().numpy())
    return np.concatenate(outs, axis=0), [o.shape[0] for o in outs]


@pytest.mark.parametrize("name", ["pipeline", "filt", "down", "f32"])
def test_block_stream_of_an_unbounded_input(name):
    """so.BlockStream: input pushed block by block, a bounded tail of it resident; the outputs concatenate
    to the sink of the pipeline over the whole input"""
    rng = np.random.default_rng(71)
    n, nch = 400000, 2
    dt = np.float32 if name == "f32" else np.float64
    x = rng.standard_normal((n, nch)).astype(dt)
    pipes = {
        "pipeline": lambda s: (so.Mix(so.Signal(so.sin, ω=1 * so.kHz), s) | so.Until(so.nframes(s) * so.frames)
                               | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.ToFramerate(48 * so.kHz)),
        "filt": lambda s: s | so.Filt(so.Lowpass, 3 * so.kHz) | so.Amplify(0.5),
        "down": lambda s: s | so.Filt(so.Lowpass, 4 * so.kHz) | so.ToFramerate(16 * so.kHz),
        "f32": lambda s: s | so.ToFramerate(48 * so.kHz),
    }
    pipe = pipes[name]
    whole = so.sink(pipe(so.Signal(np.asfortranarray(x), 44.1 * so.kHz)), so.Array)
    sizes = [30000, 1, 70001, 44100, 9, 100000]
    sizes.append(n - sum(sizes))
    bs = so.BlockStream(pipe, 44.1 * so.kHz, nch=nch, dtype=dt, history=40000)
    got, counts = _push_all(bs, x, sizes)
    assert got.shape == whole.shape and got.dtype == whole.dtype
    assert relerr(got, whole) <= (1e-6 if dt == np.float32 else 1e-11)
    assert counts[0] > 0 and counts[-1] < 200  # outputs leave as their inputs arrive; only the look-ahead waits for finish()
    assert bs.cap <= 2 * (40000 + max(sizes))  # the resident tail stays bounded


def test_block_stream_refuses_to_read_what_is_gone():
    rng = np.random.default_rng(72)
    x = rng.standard_normal((200000, 1))
    # a band-stop this narrow needs a long warm start; 2000 frames of history are not enough for it
    bs = so.BlockStream(lambda s: s | so.Filt(so.Bandstop, 0.99 * so.kHz, 1.01 * so.kHz), 44.1 * so.kHz, nch=1, history=2000)
    bs.push(x[:60000])
    with pytest.raises(so.ErrorException, match="no longer resident"):
        for k in range(60000, 200000, 20000):
            bs.push(x[k:k + 20000])


@pytest.mark.parametrize("name,pipeline", [
    ("Normpower", lambda x: x | so.Filt(so.Lowpass, 3 * so.kHz) | so.Normpower),
    ("Ramp", lambda x: x | so.Ramp(10 * so.ms)),
    ("RampOff", lambda x: so.Mix(x | so.RampOff(10 * so.ms), x)),
    ("lastframe", lambda x: x | so.Pad(so.lastframe) | so.Until(10 * so.s)),
])
def test_block_stream_refuses_pipelines_that_need_the_total_length(name, pipeline):
    """a ramp at the end, `Normpower` or an end-indexing pad used to be applied to the input
    received so far at every push -- wrong frames, marked final.  Now an error at the first push."""
    bs = so.BlockStream(pipeline, fs=44.1 * so.kHz, nch=2)
    with pytest.raises(so.ErrorException, match="not streamable"):
        bs.push(np.zeros((5000, 2)))


def test_block_stream_with_a_per_channel_padding_vector():
    """`x.pad in (lastframe, cycle, mirror)` compared a NumPy padding vector with `==` and raised
    "truth value of an array is ambiguous" at the first push; identity tests now, like the lowering's"""
    rng = np.random.default_rng(73)
    x = rng.standard_normal((30000, 2))
    pipe = lambda s: s | so.Pad(np.array([1.0, 2.0])) | so.Until(40000 * so.frames) | so.Amplify(0.5)  # noqa: E731
    bs = so.BlockStream(pipe, 44.1 * so.kHz, nch=2)
    out = bs.push(x[:20000]).cpu().numpy()
    assert out.shape[1] == 2 and np.array_equal(out, 0.5 * x[:out.shape[0]])


# ---- rates without a period: warm start (round 4) ----------------------------------------------------------------
IRR = 44100 * np.pi / 3


def test_window_of_an_irrational_rate_starts_at_the_window():
    """`After` over a resampler between non-integer frame rates: the stage starts at the window's first output (g.m0)
    and stages its input from taps + 2 frames before it (g.j0) instead of evaluating everything from output 0
    (reference src/filters.jl:221-262: the outputs are the same whatever block they are asked in)"""
    rng = np.random.default_rng(81)
    x = np.asfortranarray(rng.standard_normal((600000, 2)))
    tree = so.Signal(x, 44100 * so.Hz) | so.ToFramerate(IRR * so.Hz)
    whole = so.sink(tree)[0]
    import torch

    xd = torch.from_numpy(np.ascontiguousarray(x.T)).cuda().t()   # (a device leaf: scratch is what the stages allocate)
    dtree = so.Signal(xd, 44100 * so.Hz) | so.ToFramerate(IRR * so.Hz)
    for a, n in ((8192, 5000), (100000, 70000), (400001, 12345), (600000, 20000)):
        part = so.sink(tree | so.After(a * so.frames) | so.Until(n * so.frames))[0]
        # (short windows run the tiled kernel, long signals the persistent one: two associations of the same sums)
        assert relerr(part, whole[a:a + n]) < 1e-14 and np.abs(part - whole[a:a + n]).max() < 1e-13, (a, n)
        p = so.Plan(so.ToChannels(dtree | so.After(a * so.frames) | so.Until(n * so.frames), 2), (n, 2), np.float64, (1, n), False)
        st = p.stats()
        p.close()
        assert st["scratch_bytes"] < 8 * 2 * (2 * n + 50000)  # (nothing of the frames before the window is evaluated)
    from oracle_bridge import oracle_sink

    ref = oracle_sink(tree)
    assert relerr(whole, ref) < 1e-9


def test_stream_of_an_irrational_rate_costs_the_same_everywhere():
    """`so.stream` at x pi/3: a block at position 10^8 stages and evaluates what a block at position 0 does (the
    accumulator's deviations come from the checkpoint the previous block left)"""
    import time

    import torch

    tone = so.Signal(so.sin, 44100 * so.Hz, ω=440 * so.Hz) | so.ToFramerate(IRR * so.Hz)   # infinite: any position exists
    nblk = 1 << 20

    def block_at(pos, reps=3):
        blk = tone | so.After(pos * so.frames) | so.Until(nblk * so.frames)
        out = torch.empty((1, nblk), dtype=torch.float64, device="cuda").t()
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            so.sink_into(out, blk)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return out.cpu().numpy(), best

    y0, t0 = block_at(0)
    block_at(10 ** 8 - nblk, reps=1)            # (the block before: leaves the checkpoint a stream would have)
    y1, t1 = block_at(10 ** 8)
    assert t1 < 3 * t0 + 0.02, (t0, t1)
    # values: the sine at the resampled positions (amplitude error of the interpolation filter << 1e-3); frame i (from 1)
    # of a signal is at time i / fs on both sides of the resampler
    def err(y, pos):
        t = (pos + np.arange(2000, 3000) + 1) / float(IRR)
        return np.abs(y[2000:3000, 0] - np.sin(2 * np.pi * 440 * t)).max()

    assert err(y0, 0) < 1e-3 and err(y1, 10 ** 8) < 1e-3, (err(y0, 0), err(y1, 10 ** 8))
    from oracle_bridge import oracle_sink

    ref = oracle_sink(tone | so.Until(20000 * so.frames))
    assert relerr(y0[:20000], ref) < 1e-9


def test_block_stream_at_an_irrational_rate():
    """BlockStream no longer refuses a resampler between non-integer frame rates: bounded history, same frames"""
    rng = np.random.default_rng(82)
    x = rng.standard_normal((300000, 2))
    pipe = lambda s: s | so.ToFramerate(IRR * so.Hz)  # noqa: E731
    whole = so.sink(so.Signal(np.asfortranarray(x), 44100 * so.Hz) | so.ToFramerate(IRR * so.Hz))[0]
    bs = so.BlockStream(pipe, 44100 * so.Hz, nch=2, history=20000)
    outs = [bs.push(x[k:k + 30000]).cpu().numpy() for k in range(0, 300000, 30000)]
    outs.append(bs.finish().cpu().numpy())
    got = np.concatenate(outs)
    assert got.shape == whole.shape and relerr(got, whole) < 1e-12
    assert bs.cap <= 2 * (20000 + 30000)
