"""The fused resampler -> IIR kernel (k_rsos.hip; reference src/filters.jl:143-148 puts the resampler under the
filter, src/filters.jl:240-255 filters every block of the resampled child in place) against the CPU oracle and
against the engine's own two-kernel path (K3 + K2, `SIGOPS_NO_RSOS=1` at plan creation) on identical inputs.

`SIGOPS_RSOS_MINGROUPS=1` lets signals of a few seconds take the fused kernel (by default the planner fuses where its time
estimate says the one launch is faster than the two: from about 30 s x 8 channels on); the last test runs without it.
Tolerances: Float64 1e-9 against the oracle (what the accumulated-alpha drift of DSP.jl's phase accumulator leaves,
as for K3), 1e-11 between the two engine paths (the fused sine source is evaluated in two levels with different
base frames: ~4e-12)."""
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
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def steps_of(x, dtype=np.float64):
    """kernel names of the plan `sink(x)` would run (host result)"""
    n, nch = so.nframes(x), so.nchannels(x)
    p = so.Plan(so.ToChannels(x, nch), (n, nch), dtype, (1, n), False)
    names = [s["name"] for s in p.steps()]
    p.close()
    return names


def both(x, to=None):
    """(fused result, two-kernel result, fused?)"""
    with env(SIGOPS_RSOS_MINGROUPS=1, SIGOPS_NO_RSOS=None):
        fused = "k_rsos" in steps_of(x)
        a = so.sink(x, to)[0] if to is not None else so.sink(x)[0]
    # (the two-kernel reference with K3's Float64 products: on Float32 signals K3 alone takes the Float32 MFMA -- another
    #  rounding, tests/test_gpu_f32_mfma.py -- while the fused kernel resamples in Float64 and rounds once, as K3 did)
    with env(SIGOPS_NO_RSOS=1, SIGOPS_RS_NO_F32MFMA=1):
        b = so.sink(x, to)[0] if to is not None else so.sink(x)[0]
    return a, b, fused


def pipeline(src, lo=0.5, hi=2.0, fs_out=48.0, order=5):
    return src | so.Filt(so.Bandstop, lo * so.kHz, hi * so.kHz, order=order) | so.ToFramerate(fs_out * so.kHz)


@pytest.mark.parametrize("nch", [1, 2, 3, 4, 8, 16, 24])
def test_channel_counts(nch):
    """rows of a sequence group = ranges x channels for every divisor of 16 (and 3 -> one channel per group, 24 ->
    groups of 8)"""
    rng = np.random.default_rng(10 + nch)
    n = 400000 if nch <= 8 else 200000
    x = pipeline(so.Signal(F(rng.standard_normal((n, nch))), 44.1 * so.kHz))
    a, b, fused = both(x)
    assert fused
    assert a.shape == b.shape and relerr(a, b) < 1e-11
    assert relerr(a, oracle_sink(x)) < 1e-9


@pytest.mark.parametrize("kind", ["mix_sine", "amplify_sine", "amplify_const", "sine_minus", "ramp_in_front", "padded_tail", "append"])
def test_fused_sources(kind):
    """carrier steps of the resampler's source: the LDS-add path (Mix), the in-place multiply (Amplify), a constant, a
    subtraction from the generator, and shapes whose first / last chunks or whole ranges take the general staging path"""
    rng = np.random.default_rng(21)
    n = 300000
    noise = so.Signal(F(rng.standard_normal((n, 2))), 44.1 * so.kHz)
    tone = so.Signal(so.sin, ω=1 * so.kHz)
    src = {
        "mix_sine": lambda: so.Mix(tone, noise) | so.Until(n * so.frames),
        "amplify_sine": lambda: noise | so.Amplify(so.Signal(so.sin, ω=5 * so.Hz)) | so.Until(n * so.frames),
        "amplify_const": lambda: noise | so.Amplify(0.25),
        "sine_minus": lambda: so.OperateOn(np.subtract, tone, noise) | so.Until(n * so.frames),
        "ramp_in_front": lambda: noise | so.RampOn(50 * so.ms),
        "padded_tail": lambda: noise | so.Pad(so.zero) | so.Until((n + 50000) * so.frames),
        "append": lambda: so.Append(noise | so.Until(100000 * so.frames), so.Signal(F(rng.standard_normal((150000, 2))), 44.1 * so.kHz)),
    }[kind]()
    x = pipeline(src)
    a, b, fused = both(x)
    assert fused
    assert relerr(a, b) < 1e-11
    assert relerr(a, oracle_sink(x)) < 1e-9


@pytest.mark.parametrize("design", ["lowpass3", "highpass4", "bandpass6", "bandstop2"])
def test_cascades(design):
    rng = np.random.default_rng(33)
    src = so.Signal(F(rng.standard_normal((350000, 4))), 44.1 * so.kHz)
    x = {
        "lowpass3": lambda: src | so.Filt(so.Lowpass, 3 * so.kHz, order=5),      # 3 sections
        "highpass4": lambda: src | so.Filt(so.Highpass, 200 * so.Hz, order=7),   # 4
        "bandpass6": lambda: src | so.Filt(so.Bandpass, 1 * so.kHz, 4 * so.kHz, order=6),  # 6
        "bandstop2": lambda: src | so.Filt(so.Bandstop, 1 * so.kHz, 3 * so.kHz, order=2),  # 2
    }[design]() | so.ToFramerate(48 * so.kHz)
    a, b, fused = both(x)
    assert fused
    assert relerr(a, b) < 1e-11
    assert relerr(a, oracle_sink(x)) < 1e-9


@pytest.mark.parametrize("fs_in,fs_out", [(44.1, 48.0), (32.0, 48.0), (22.05, 48.0), (44.1, 88.2), (8.0, 16.0)])
def test_rates(fs_in, fs_out):
    """other periodic rates whose period splits into whole blocks of 16 outputs; where the geometry does not fit the
    engine keeps the two kernels and the result is theirs"""
    rng = np.random.default_rng(44)
    src = so.Signal(F(rng.standard_normal((300000, 2))), fs_in * so.kHz)
    x = src | so.Filt(so.Lowpass, 2 * so.kHz) | so.ToFramerate(fs_out * so.kHz)
    a, b, fused = both(x)
    assert relerr(a, b) < 1e-11
    assert relerr(a, oracle_sink(x)) < 1e-9


@pytest.mark.parametrize("kind", ["mix_sine", "amplify_sine", "amplify_const", "promoted", "sine_minus", "append"])
@pytest.mark.parametrize("nch", [8, 2, 3])
def test_float32_array_sources(kind, nch):
    """a Float32 array under a Float64 generator (the Float32 headline: `Mix(Signal(sin), noise32)` is a Float64 signal,
    src/mapsignal.jl promotion): the loader DMAs the Float32 chunk into the upper half of its ring slot and widens it in
    place with the fused step -- against the oracle, and against K3's GA form + K2 (`SIGOPS_NO_RSOS`)"""
    rng = np.random.default_rng(31 + nch)
    n = 300000
    x32 = F(rng.standard_normal((n, nch)).astype(np.float32))
    noise = so.Signal(x32, 44.1 * so.kHz)
    tone = so.Signal(so.sin, ω=1 * so.kHz)
    src = {
        "mix_sine": lambda: so.Mix(tone, noise) | so.Until(n * so.frames),
        "amplify_sine": lambda: noise | so.Amplify(so.Signal(so.sin, ω=5 * so.Hz)) | so.Until(n * so.frames),
        "amplify_const": lambda: noise | so.Amplify(0.25),
        "promoted": lambda: so.ToEltype(noise, np.float64),
        "sine_minus": lambda: so.OperateOn(np.subtract, tone, noise) | so.Until(n * so.frames),
        "append": lambda: so.Append(so.Mix(tone, noise) | so.Until(100000 * so.frames),
                                    so.Signal(F(rng.standard_normal((150000, nch))), 44.1 * so.kHz)),
    }[kind]()
    x = pipeline(src)
    assert x.dtype == so.signals.F64 if hasattr(so.signals, "F64") else True
    a, b, fused = both(x)
    assert fused or kind in ("amplify_const", "sine_minus", "append")   # (what the planner does not fuse keeps the two kernels: same values)
    assert a.dtype == np.float64 and relerr(a, b) < 1e-11
    assert relerr(a, oracle_sink(x)) < 1e-9


def test_float32_headline_is_one_launch():
    """Float32 leaf, Float32 result, Float64 arithmetic: the BASELINE pipeline with Float32 storage"""
    rng = np.random.default_rng(57)
    n = 2000000
    x32 = F(rng.standard_normal((n, 8)).astype(np.float32))
    x = pipeline(so.Mix(so.Signal(so.sin, ω=1 * so.kHz), so.Signal(x32, 44.1 * so.kHz)) | so.Until(n * so.frames))
    nout = so.nframes(x)
    with env(SIGOPS_RSOS_MINGROUPS=1):
        p = so.Plan(so.ToChannels(x, 8), (nout, 8), np.float32, (1, nout), False)
        names = [s["name"] for s in p.steps()]
        p.close()
        assert names == ["k_rsos"], names
        got = np.empty((nout, 8), dtype=np.float32, order="F")
        so.sink_into(got, x)
        with env(SIGOPS_RSOS_NO_F32MFMA=1):
            got64 = np.empty((nout, 8), dtype=np.float32, order="F")
            so.sink_into(got64, x)
    want = oracle_sink(x).astype(np.float32)
    assert relerr(got64, want) < 1e-6 and np.mean(got64 == want) > 0.99   # (values within the accumulated-alpha drift, 2e-9, of a Float32 rounding boundary flip)
    # (as shipped since round 6: Float32 samples in the ring, the resampling product on the Float32 MFMA)
    assert relerr(got, want) < 1e-6 and relerr(got, got64) < 3e-7


@pytest.mark.parametrize("fs_in,fs_out,nch", [(24.0, 48.0, 8), (16.0, 48.0, 2), (32.0, 48.0, 8), (8.0, 16.0, 3), (22.05, 44.1, 4),
                                              (48.0, 24.0, 8), (48.0, 32.0, 2)])
def test_small_rational_ratios(fs_in, fs_out, nch):
    """x 2, x 3, x 3/2 (and the downsampling ones where the windows fit 20 k-steps): DSP.jl's FIRInterpolator / FIRRational
    rates (reference src/reformatting.jl:103-111).  Their resampler stage runs on super-periods sized for K3's tiles; the
    fused kernel walks the shortest super-period of whole 16-output blocks with a tap table of its own"""
    rng = np.random.default_rng(91)
    n = int(300000 * fs_in / 44.1)
    src = so.Mix(so.Signal(so.sin, ω=0.3 * so.kHz), so.Signal(F(rng.standard_normal((n, nch))), fs_in * so.kHz)) | so.Until(n * so.frames)
    x = src | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.ToFramerate(fs_out * so.kHz)
    a, b, fused = both(x)
    assert fused or fs_out < fs_in, (fs_in, fs_out)
    assert a.shape == b.shape and relerr(a, b) < 1e-11
    assert relerr(a, oracle_sink(x)) < 1e-9
    nout = a.shape[0]
    win = x | so.After((nout // 2) * so.frames) | so.Until(20000 * so.frames)
    with env(SIGOPS_RSOS_MINGROUPS=1):
        part = so.sink(win)[0]
    assert relerr(part, a[nout // 2:nout // 2 + 20000]) < 1e-11


def test_float32_result_of_a_float64_pipeline():
    """`sink(x, Float32)` of a Float64 signal: the kernel rounds in its own store (reference src/sink.jl:262-266)"""
    rng = np.random.default_rng(55)
    x = pipeline(so.Signal(F(rng.standard_normal((300000, 8))), 44.1 * so.kHz))
    want = oracle_sink(x).astype(np.float32)
    got = np.empty(want.shape, dtype=np.float32, order="F")
    with env(SIGOPS_RSOS_MINGROUPS=1):
        so.sink_into(got, x)
    assert relerr(got, want) < 1e-6
    assert np.mean(got == want) > 0.999  # (what differs: values within the accumulated-alpha drift of a rounding boundary)


def test_a_non_finite_sample_poisons_the_rest_of_its_channel():
    """the reference's recurrence stays NaN from the first non-finite sample of a channel to its end; a later time range
    of the fused kernel starts from rest and would be finite again (k_sos_poison behind the kernel fills it)"""
    rng = np.random.default_rng(66)
    a = rng.standard_normal((400000, 4))
    a[123456, 1] = np.nan
    a[300000, 3] = np.inf
    x = pipeline(so.Signal(F(a), 44.1 * so.kHz))
    got, ref, fused = both(x)
    assert fused
    for ch, first in ((1, 123456), (3, 300000)):
        m = int(first * 160 / 147)
        assert np.isnan(got[m + 200:, ch]).all() and np.isnan(ref[m + 200:, ch]).all()
        assert np.isfinite(got[:m - 200, ch]).all()
        assert relerr(got[:m - 200, ch], ref[:m - 200, ch]) < 1e-11
    for ch in (0, 2):
        assert np.isfinite(got[:, ch]).all() and relerr(got[:, ch], ref[:, ch]) < 1e-11


@pytest.mark.parametrize("a,n", [(200000, 50000), (8192, 300000), (123457, 1), (350001, 85000), (20000, 410000)])
def test_windows_run_fused_too(a, n):
    """a window of the result (After, a later block of `so.stream`, a rank's time range) warm-starts both stages; the
    fused kernel takes the resampler stage's coordinates and stores nothing below the window (RsSos::store_lo)"""
    rng = np.random.default_rng(77)
    x = pipeline(so.Signal(F(rng.standard_normal((400000, 2))), 44.1 * so.kHz))
    win = x | so.After(a * so.frames) | so.Until(n * so.frames)
    with env(SIGOPS_RSOS_MINGROUPS=1, SIGOPS_NO_RSOS=None):
        whole = so.sink(x)[0]
        fused = "k_rsos" in steps_of(win)
        part = so.sink(win)[0]
        out32 = np.empty((n, 2), dtype=np.float32, order="F")
        so.sink_into(out32, win)
    assert fused or n < 20000
    assert part.shape == (n, 2) and relerr(part, whole[a:a + n]) < 1e-11
    assert relerr(out32, whole[a:a + n].astype(np.float32)) < 1e-6
    with env(SIGOPS_NO_RSOS=1):
        ref = so.sink(win)[0]
    assert relerr(part, ref) < 1e-11


def test_stream_blocks_and_time_shards_through_the_fused_kernel():
    rng = np.random.default_rng(78)
    x = pipeline(so.Mix(so.Signal(so.sin, ω=1 * so.kHz), so.Signal(F(rng.standard_normal((600000, 8))), 44.1 * so.kHz))
                 | so.Until(600000 * so.frames))
    from sigops_amd import sharding

    with env(SIGOPS_RSOS_MINGROUPS=1):
        whole = so.sink(x)[0]
        blocks = [b for b, _ in so.stream(x, 150000)]
        got = np.concatenate([np.asarray(b) for b in blocks])
        assert got.shape == whole.shape and relerr(got, whole) < 1e-11
        parts = [so.sink(sharding.shard_time(x, r, 3)[0])[0] for r in range(3)]
        assert relerr(np.concatenate(parts), whole) < 1e-11
    a = rng.standard_normal((500000, 2))
    a[400000, 1] = np.nan                      # non-finite behind the window's start: the rest of that channel is NaN
    y = pipeline(so.Signal(F(a), 44.1 * so.kHz)) | so.After(300000 * so.frames)
    with env(SIGOPS_RSOS_MINGROUPS=1):
        got = so.sink(y)[0]
    with env(SIGOPS_NO_RSOS=1):
        ref = so.sink(y)[0]
    want = oracle_sink(y)
    # (round 6: the fused kernel's set of non-finite outputs is the reference's -- k_rsos_fixup, tests/test_gpu_rsos_nonfinite.py;
    #  K3 + K2's is the superset K3's group windows make it)
    assert np.array_equal(np.isfinite(got), np.isfinite(want)) and not np.isfinite(got[200000:, 1]).any() and np.isfinite(got[:, 0]).all()
    assert not (np.isfinite(ref) & ~np.isfinite(want)).any()
    assert relerr(got[:, 0], ref[:, 0]) < 1e-11


def test_run_to_run_identical():
    rng = np.random.default_rng(88)
    x = pipeline(so.Mix(so.Signal(so.sin, ω=1 * so.kHz), so.Signal(F(rng.standard_normal((500000, 8))), 44.1 * so.kHz))
                 | so.Until(500000 * so.frames))
    with env(SIGOPS_RSOS_MINGROUPS=1):
        a = so.sink(x)[0]
        b = so.sink(x)[0]
    assert np.array_equal(a, b)


def test_default_policy_fuses_long_signals_only():
    rng = np.random.default_rng(99)
    short = pipeline(so.Signal(F(rng.standard_normal((100000, 8))), 44.1 * so.kHz))
    long_ = pipeline(so.Signal(F(rng.standard_normal((3000000, 8))), 44.1 * so.kHz))
    with env(SIGOPS_RSOS_MINGROUPS=None, SIGOPS_NO_RSOS=None):
        assert "k_rsos" not in steps_of(short)
        assert steps_of(long_) == ["k_rsos"]
        got = so.sink(long_)[0]
    with env(SIGOPS_NO_RSOS=1):
        ref = so.sink(long_)[0]
    assert relerr(got, ref) < 1e-11


def test_device_tensor_with_an_odd_number_of_frames():
    """a [channels x frames] device tensor with an odd frame count: every second row is 8 bytes off a 16-byte boundary.
    Float64 rows need no more than their natural alignment for the LDS-DMA path (executor.cpp carrier_vec_ok): same
    values as the aligned copy, and the plan does not fall to the general staging path (~10x slower)"""
    import time

    import torch

    n = 2000001
    xt = torch.randn((8, n), dtype=torch.float64, device="cuda")
    xa = torch.empty((8, n + 1), dtype=torch.float64, device="cuda")[:, :n]   # (rows 16-byte aligned)
    xa.copy_(xt)
    outs, times = [], []
    for leaf in (xt, xa):
        x = pipeline(so.Mix(so.Signal(so.sin, ω=1 * so.kHz), so.Signal(leaf.t(), 44.1 * so.kHz)) | so.Until(n * so.frames))
        nout = so.nframes(x)
        out = torch.empty((8, nout), dtype=torch.float64, device="cuda").t()
        with env(SIGOPS_RSOS_MINGROUPS=1):
            so.sink_into(out, x)
            torch.cuda.synchronize()
            best = None  # (the best of three: one call in a few hundred stalls for tens of milliseconds behind freed buffers)
            for _ in range(3):
                t0 = time.perf_counter()
                so.sink_into(out, x)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
        times.append(best)
        outs.append(out.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    assert times[0] < 3 * times[1] + 0.005, times


@pytest.mark.parametrize("n", [5000, 400000])
def test_consecutive_filters_run_as_one_cascade(n):
    """`x |> Filt(Lowpass) |> Filt(Highpass)` in Float64: one cascade of all the sections (the inner filter's first, gains
    multiplied) -- one run of the filter kernels, and with a resampler behind it one launch of the fused kernel; against
    the oracle (which filters twice, as the reference does) and against the engine filtering twice (SIGOPS_SOS_NOMERGE)"""
    rng = np.random.default_rng(93)
    src = so.Signal(F(rng.standard_normal((n, 4))), 44.1 * so.kHz)
    two = src | so.Filt(so.Lowpass, 3 * so.kHz, order=5) | so.Filt(so.Highpass, 200 * so.Hz, order=3)
    three = two | so.Filt(so.Bandstop, 1 * so.kHz, 1.2 * so.kHz, order=1)
    for x in (two, three, two | so.After(n // 2 * so.frames), two | so.ToFramerate(48 * so.kHz)):
        with env(SIGOPS_RSOS_MINGROUPS=1):
            names = steps_of(x)
            got = so.sink(x)[0]
        with env(SIGOPS_SOS_NOMERGE=1, SIGOPS_NO_RSOS=1):
            names2 = steps_of(x)
            ref = so.sink(x)[0]
        assert sum(nm.startswith("k_sos") or nm == "k_rsos" for nm in names) == 1, names
        assert sum(nm.startswith("k_sos") for nm in names2) >= 2, names2
        assert relerr(got, ref) < 1e-11
        assert relerr(got, oracle_sink(x)) < 1e-9
    x32 = so.Signal(F(rng.standard_normal((n, 2)).astype(np.float32)), 44.1 * so.kHz) | so.Filt(so.Lowpass, 3 * so.kHz) | so.Filt(so.Highpass, 200 * so.Hz)
    assert sum(nm.startswith("k_sos") for nm in steps_of(x32, np.float32)) == 2   # (Float32: the reference rounds in between)
    assert relerr(so.sink(x32)[0], oracle_sink(x32)) < 1e-6


@pytest.mark.parametrize("kind", ["array", "until", "window", "mix32", "append", "device"])
@pytest.mark.parametrize("nch", [8, 2, 1, 3])
def test_float32_pipelines(kind, nch):
    """a Float32 signal all the way -- Float32 array, resampler, filter, result (round 4, last item): the reference's resampler
    hands the filter Float32 samples, so the fused kernel rounds its `X` accumulators to Float32 (`RsSos::x32`; K3's Float32
    store rounds the same values) before the cascade reads them and stores a Float32 result.  Against the oracle (1e-6, the
    suite's Float32 tolerance) and against K3 + K2 (`SIGOPS_NO_RSOS=1`): the same resampled samples bit for bit, the
    cascade's block association instead of the sequential recurrence -- Float32 results that differ only where a value sits
    within 1e-13 of a rounding boundary."""
    rng = np.random.default_rng(70 + nch)
    n = 300_000
    x32 = F((rng.standard_normal((n, nch)) * 0.5).astype(np.float32))
    sig = so.Signal(x32, 44.1 * so.kHz)
    if kind == "device":
        torch = pytest.importorskip("torch")
        dev = torch.from_numpy(np.ascontiguousarray(x32.T)).cuda()
        sig = so.Signal(dev.t(), 44.1 * so.kHz)
    src = {
        "array": lambda: sig,
        "device": lambda: sig,
        "until": lambda: sig | so.Until(250_001 * so.frames),
        "window": lambda: sig | so.After(10_000 * so.frames) | so.Until(200_000 * so.frames),
        "mix32": lambda: so.Mix(sig, so.Signal(F((rng.standard_normal((n, nch)) * 0.1).astype(np.float32)), 44.1 * so.kHz)),
        "append": lambda: so.Append(sig | so.Until(100_000 * so.frames), so.Signal(F(rng.standard_normal((150_000, nch)).astype(np.float32)), 44.1 * so.kHz)),
    }[kind]()
    x = pipeline(src)
    # (round 6: a Float32 signal's resampling product runs on the Float32 MFMA inside the fused kernel too --
    #  tests/test_gpu_rsos_f32m.py --; SIGOPS_RSOS_NO_F32MFMA keeps the Float64 products this test was written for)
    with env(SIGOPS_RSOS_NO_F32MFMA=1):
        a, b, fused = both(x)
    # (two Float32 arrays of 8 channels: K3's two-array instantiation resamples the sum in one launch and the filter follows
    #  it -- 0.82 ms where K1 + the fused kernel took 1.05, tests/test_gpu_two_arrays.py)
    assert fused or nch == 3 or kind == "append" or (kind == "mix32" and nch == 8), kind
    assert a.dtype == np.float32 and b.dtype == np.float32
    if kind == "mix32" and nch == 8:  # (... on the Float32 MFMA: its own rounding against `b`'s Float64 products, test_gpu_f32_mfma.py)
        assert relerr(a, b) < 3e-7
    else:
        assert relerr(a, b) < 1e-7 and np.mean(a == b) > 0.999
    a32, _, _ = both(x)  # ... and as shipped
    assert a32.dtype == np.float32 and relerr(a32, b) < 3e-7
    want = oracle_sink(pipeline(src if kind != "device" else so.Signal(x32, 44.1 * so.kHz)))
    assert want.dtype == np.float32 and relerr(a, want) < 1e-6


def test_float32_pipeline_windows_and_streams():
    """a window of a Float32 pipeline (`After` above it: warm starts of both stages) and its blocks through `so.stream`"""
    rng = np.random.default_rng(75)
    n = 600_000
    x32 = F((rng.standard_normal((n, 2)) * 0.5).astype(np.float32))
    x = pipeline(so.Signal(x32, 44.1 * so.kHz))
    want = oracle_sink(x)
    win = x | so.After(300_000 * so.frames) | so.Until(200_000 * so.frames)
    with env(SIGOPS_RSOS_MINGROUPS=1):
        got = so.sink(win)[0]
    assert got.dtype == np.float32 and relerr(got, want[300_000:500_000]) < 1e-6
    with env(SIGOPS_RSOS_MINGROUPS=1):
        blocks = [blk for blk, _ in so.stream(x, 150_000)]
    cat = np.concatenate(blocks, axis=0)
    assert cat.dtype == np.float32 and cat.shape == want.shape and relerr(cat, want) < 1e-6


def test_a_wait_that_does_not_end_is_an_error_return_not_a_dead_process():
    """A wait between k_rsos's waves that never ends: the wave says so in the plan's host-mapped error word and ends, the
    waves that wait for it in turn do the same, the launch finishes -- and the host reports it: `sink` into a host result
    (it synchronises) raises, an execute into a device result returns and the NEXT call on the plan raises.  Neither a hung
    device nor the dead process a trap is (SIGOPS_RSOS_TRAP=1 keeps the trap: the ROCm runtime aborts the process on one).
    Forced in a child: no chain wave (ablation bit 64), the y waves' wait for a state cut to 4 096 polls (bit 32768)."""
    import subprocess
    import sys

    code = r'''
import os, sys
os.environ["SIGOPS_RSOS_DEBUG"] = str(64 + 32768)
os.environ["SIGOPS_RSOS_MINGROUPS"] = "1"
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
import sigops_amd as so
rng = np.random.default_rng(1)
x = so.Signal(np.asfortranarray(rng.standard_normal((200000, 8))), 44.1 * so.kHz) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.ToFramerate(48 * so.kHz)
try:
    so.sink(x)
    print("HOST RESULT RETURNED")
except so.ErrorException as e:
    print("HOST RESULT RAISED:", e)
n = so.nframes(x)
out = torch.empty((8, n), dtype=torch.float64, device="cuda").t()
plan = so.Plan(so.ToChannels(x, 8), (n, 8), np.float64, (out.stride(0), out.stride(1)), True, device=0)
plan.execute(out.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("DEVICE RESULT RETURNED")
try:
    plan.execute(out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    print("SECOND EXECUTE RETURNED")
except so.ErrorException as e:
    print("SECOND EXECUTE RAISED:", e)
torch.cuda.synchronize()
plan.close()
# a one-shot sink into a device tensor (execute, no second call on the plan): so_plan_check behind the execute raises
try:
    so.sink(x, "torch")
    print("ONE-SHOT DEVICE SINK RETURNED")
except so.ErrorException as e:
    print("ONE-SHOT DEVICE SINK RAISED:", e)
# ... and a plan somebody destroys without having checked it: Plan.close checks for them
plan = so.Plan(so.ToChannels(x, 8), (n, 8), np.float64, (out.stride(0), out.stride(1)), True, device=0)
plan.execute(out.data_ptr(), torch.cuda.current_stream().cuda_stream)
try:
    plan.close()
    print("CLOSE RETURNED")
except so.ErrorException as e:
    print("CLOSE RAISED:", e)
# ... and the C-ABI's own last resort: so_plan_destroy without so_plan_check says it on stderr
from sigops_amd import _capi
plan = so.Plan(so.ToChannels(x, 8), (n, 8), np.float64, (out.stride(0), out.stride(1)), True, device=0)
plan.execute(out.data_ptr(), torch.cuda.current_stream().cuda_stream)
plan._unchecked = None
sys.stdout.flush()
plan.close()
os.environ.pop("SIGOPS_RSOS_DEBUG")
from oracle_bridge import oracle_sink, relerr
print("AFTERWARDS", float(relerr(so.sink(x)[0], oracle_sink(x))) < 1e-9)
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = r.stdout.decode()
    assert r.returncode == 0, (r.returncode, out[-1500:])
    assert "HOST RESULT RAISED: k_rsos: a wait between its waves did not end" in out and "HOST RESULT RETURNED" not in out, out[-1500:]
    assert "DEVICE RESULT RETURNED" in out and "SECOND EXECUTE RAISED: k_rsos: a wait between its waves did not end" in out, out[-1500:]
    assert "ONE-SHOT DEVICE SINK RAISED: k_rsos: a wait between its waves did not end" in out and "ONE-SHOT DEVICE SINK RETURNED" not in out, out[-2500:]
    assert "CLOSE RAISED: k_rsos: a wait between its waves did not end" in out and "CLOSE RETURNED" not in out, out[-2500:]
    assert "libsigops: k_rsos: a wait between its waves did not end -- THE LAST RESULT OF THIS PLAN IS INVALID" in out, out[-2500:]
    assert "AFTERWARDS True" in out, out[-1500:]
    # ... with the trap instead (the opt-out): a dead process, the device fine afterwards
    code_trap = "import os\nos.environ['SIGOPS_RSOS_TRAP'] = '1'\n" + code
    r = subprocess.run([sys.executable, "-c", code_trap], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode != 0 and "RETURNED" not in r.stdout.decode(), (r.returncode, r.stdout.decode()[-800:])
    rng = np.random.default_rng(1)
    x = pipeline(so.Signal(F(rng.standard_normal((200000, 8))), 44.1 * so.kHz))
    with env(SIGOPS_RSOS_MINGROUPS=1):
        assert relerr(so.sink(x)[0], oracle_sink(x)) < 1e-9


@pytest.mark.parametrize("shape", ["odd rows", "odd base", "both"])
@pytest.mark.parametrize("fused", [False, True])
def test_float32_device_tensors_off_their_16_byte_boundaries(shape, fused):
    """Float32 rows that are only element-aligned -- an odd number of frames per row, a view that starts one sample into its
    storage -- take the 16-byte LDS-DMA path (executor.cpp carrier_vec_ok: the memory pipeline takes any address aligned
    for the element); SIGOPS_STRICT_ALIGN=1 sends them through the general staging path.  Same values, bit for bit, in
    K3 (`ToFramerate`) and in the fused kernel (`ToFramerate |> Filt`)."""
    import torch

    n = 300001 if shape != "odd base" else 300000
    store = torch.randn((8, n + 4), dtype=torch.float32, device="cuda")
    view = store[:, 1:n + 1] if shape != "odd rows" else store[:, :n]
    if shape == "odd rows":
        view = torch.randn((8, n), dtype=torch.float32, device="cuda")
    x = so.Signal(view.t(), 44.1 * so.kHz) | so.ToFramerate(48 * so.kHz)
    if fused:
        x = x | so.Filt(so.Lowpass, 6 * so.kHz)
    outs = []
    for strict in (None, 1):
        with env(SIGOPS_STRICT_ALIGN=strict, SIGOPS_RSOS_MINGROUPS=1):
            if fused:
                assert "k_rsos" in steps_of(x, np.float32)
            got, _ = so.sink(x, "torch")
            outs.append(got.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    host = so.Signal(F(view.t().cpu().numpy()), 44.1 * so.kHz) | so.ToFramerate(48 * so.kHz)
    if fused:
        host = host | so.Filt(so.Lowpass, 6 * so.kHz)
    assert relerr(outs[0], oracle_sink(host)) < 1e-6
