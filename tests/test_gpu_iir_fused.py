"""A plain `Filt` as ONE pass of the block-state-space kernel (k_rsos with an identity resampler, Plan::fuse_plain_sos;
reference: the IIR's nextblock filters each block of its child once, src/filters.jl:240-255) against the CPU oracle and
against the engine's three-pass chunked scan (K2, `SIGOPS_NO_PLAIN_RSOS=1` at plan creation) on identical inputs.

`SIGOPS_RSOS_MINGROUPS=1` lets short signals take the one-pass form (by default the planner's estimate decides: from a few
million samples on).  Tolerances: Float64 1e-10 against the oracle (another association of the same sums), 1e-11 between the
two engine paths; Float32 1e-6 / results that differ in single rounding steps."""
import os

import numpy as np
import pytest

import sigops_amd as so
from oracle_bridge import oracle_sink, relerr
from test_gpu_rsos import F, env, steps_of

pytestmark = pytest.mark.gpu


def both(x, to=None):
    """(one-pass result, three-pass result, one pass taken?)"""
    with env(SIGOPS_RSOS_MINGROUPS=1, SIGOPS_NO_PLAIN_RSOS=None, SIGOPS_NO_RSOS=None):
        fused = "k_rsos" in steps_of(x, np.float32 if to is np.float32 else np.float64)
        a = so.sink(x, to)[0] if to is not None else so.sink(x)[0]
    with env(SIGOPS_NO_PLAIN_RSOS=1):
        b = so.sink(x, to)[0] if to is not None else so.sink(x)[0]
    return a, b, fused


@pytest.mark.parametrize("nch", [1, 2, 3, 4, 8, 16, 24])
def test_channel_counts(nch):
    rng = np.random.default_rng(70 + nch)
    n = 300000 if nch <= 8 else 150000
    x = so.Signal(F(rng.standard_normal((n, nch))), 44.1 * so.kHz) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz)
    a, b, fused = both(x)
    assert fused
    assert a.shape == b.shape and relerr(a, b) < 1e-11
    assert relerr(a, oracle_sink(x)) < 1e-10


@pytest.mark.parametrize("kind", ["lowpass", "highpass", "bandpass", "order1", "order12"])
def test_filters(kind):
    rng = np.random.default_rng(5)
    src = so.Signal(F(rng.standard_normal((400000, 8))), 44.1 * so.kHz)
    x = {"lowpass": src | so.Filt(so.Lowpass, 4 * so.kHz), "highpass": src | so.Filt(so.Highpass, 300 * so.Hz),
         "bandpass": src | so.Filt(so.Bandpass, 1 * so.kHz, 3 * so.kHz), "order1": src | so.Filt(so.Lowpass, 2 * so.kHz, order=1),
         "order12": src | so.Filt(so.Lowpass, 5 * so.kHz, order=12)}[kind]
    a, b, fused = both(x)
    assert fused
    assert relerr(a, b) < 1e-11 and relerr(a, oracle_sink(x)) < 1e-10


def test_config2_shape_mix_with_a_sine():
    """BASELINE config 2: Mix(sin 1 kHz, noise) |> Filt(Bandstop): the sum is formed where the chunks land in LDS"""
    rng = np.random.default_rng(1983)
    n = 600000
    noise = F(rng.standard_normal((n, 2)))
    x = so.Mix(so.Signal(so.sin, ω=1 * so.kHz) | so.Until(n * so.frames), so.Signal(noise, 44.1 * so.kHz)) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz)
    a, b, fused = both(x)
    assert fused
    assert relerr(a, b) < 1e-10 and relerr(a, oracle_sink(x)) < 1e-10


def test_amplify_with_a_sine_and_device_tensors():
    import torch

    n = 500001  # (odd: every second row of the [channels x frames] tensor is 8 bytes off a 16-byte boundary)
    xt = torch.randn((8, n), dtype=torch.float64, device="cuda")
    x = so.Amplify(so.Signal(xt.t(), 44.1 * so.kHz), so.Signal(so.sin, ω=7 * so.Hz)) | so.Until(n * so.frames) | so.Filt(so.Lowpass, 3 * so.kHz)
    a, b, fused = both(x)
    assert fused
    assert relerr(a, b) < 1e-11
    xh = so.Amplify(so.Signal(F(xt.t().cpu().numpy()), 44.1 * so.kHz), so.Signal(so.sin, ω=7 * so.Hz)) | so.Until(n * so.frames) | so.Filt(so.Lowpass, 3 * so.kHz)
    assert relerr(a, oracle_sink(xh)) < 1e-10


@pytest.mark.parametrize("nch", [8, 4, 16, 2, 3])
def test_float32_signals_in_one_pass_their_samples_stay_float32_in_the_ring(nch):
    """A plain Float32 array under a Filt: the kernel's ring keeps the Float32 samples (four bytes a frame, RsSos::ring32) and
    the y waves widen their window operands -- no widening pass by the one loader wave, which had made this form slower than
    the three passes (0.47 against 0.39 ms for 12.5 M x 8; now 0.33).  Against the oracle at Float32's 1e-6, against the
    three passes (the same Float64 recurrence in another association, rounded to Float32 once) nearly everywhere bit-equal,
    and with the widening loader (SIGOPS_RSOS_NO_RING32) bit for bit: the same values reach the same MFMAs."""
    rng = np.random.default_rng(8 + nch)
    n = 400_003
    x = so.Signal(F(rng.standard_normal((n, nch)).astype(np.float32)), 44.1 * so.kHz) | so.Filt(so.Lowpass, 4 * so.kHz)
    a, b, fused = both(x)
    assert fused and a.dtype == np.float32
    assert relerr(a, oracle_sink(x)) < 1e-6
    assert relerr(a, b) < 1e-7 and np.mean(a == b) > 0.999
    with env(SIGOPS_RSOS_MINGROUPS=1, SIGOPS_RSOS_NO_RING32=1):
        assert "k_rsos" in steps_of(x, np.float32)
        assert np.array_equal(so.sink(x)[0], a)


def test_float32_edges_windows_and_pieces_with_the_float32_ring():
    """what the loader's other paths put into a Float32 ring: zeros around the signal (warm-up of the first range, tail), the
    general staging path at the ends of an array and for the pieces of an Append, a source that starts inside its array"""
    rng = np.random.default_rng(81)
    x32 = F(rng.standard_normal((300_001, 8)).astype(np.float32))
    y32 = F(rng.standard_normal((123_457, 8)).astype(np.float32))
    X, Y = so.Signal(x32, 44.1 * so.kHz), so.Signal(y32, 44.1 * so.kHz)
    for tree in (X | so.After(12_345 * so.frames) | so.Until(200_000 * so.frames) | so.Filt(so.Lowpass, 4 * so.kHz),
                 so.Append(X | so.Until(100_000 * so.frames), Y) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz),
                 X | so.Pad(so.zero) | so.Until(350_000 * so.frames) | so.Filt(so.Highpass, 300 * so.Hz),
                 X | so.Filt(so.Lowpass, 3 * so.kHz) | so.ToFramerate(48 * so.kHz),
                 so.Append(X | so.Until(50_001 * so.frames), Y) | so.Filt(so.Lowpass, 3 * so.kHz) | so.ToFramerate(48 * so.kHz)):
        with env(SIGOPS_RSOS_MINGROUPS=1):
            dflt = so.sink(tree)[0]  # (round 6: resampled Float32 signals take the Float32 MFMA -- tests/test_gpu_rsos_f32m.py)
        with env(SIGOPS_RSOS_MINGROUPS=1, SIGOPS_RSOS_NO_F32MFMA=1):
            got = so.sink(tree)[0]
        with env(SIGOPS_RSOS_MINGROUPS=1, SIGOPS_RSOS_NO_RING32=1):
            old = so.sink(tree)[0]
        want = oracle_sink(tree)
        assert got.dtype == np.float32 and relerr(got, want) < 1e-6
        assert np.array_equal(got, old)
        assert dflt.dtype == np.float32 and relerr(dflt, want) < 1e-6 and relerr(dflt, got) < 3e-7


def test_float64_filter_into_a_float32_result():
    """sink! converts on the store (reference src/sink.jl:262-266): the kernel's own narrowing store"""
    rng = np.random.default_rng(9)
    x = so.Signal(F(rng.standard_normal((300000, 4))), 44.1 * so.kHz) | so.Filt(so.Highpass, 1 * so.kHz)
    want = oracle_sink(x).astype(np.float32)
    outs = []
    for e in (dict(SIGOPS_RSOS_MINGROUPS=1, SIGOPS_NO_PLAIN_RSOS=None), dict(SIGOPS_NO_PLAIN_RSOS=1)):
        with env(**e):
            got = np.empty(want.shape, dtype=np.float32, order="F")
            so.sink_into(got, x)
            outs.append(got)
    with env(SIGOPS_RSOS_MINGROUPS=1):
        assert "k_rsos" in steps_of(x, np.float32)
    assert relerr(outs[0], want) < 1e-6 and relerr(outs[0], outs[1]) < 1e-7


def test_non_finite_samples():
    """A filter never recovers from a non-finite sample (the reference's recurrence carries it on in its state).  The block
    form multiplies whole blocks of 16 frames -- 0 * NaN is NaN --, so k_rsos alone is non-finite from the START of the
    16-frame block that holds a channel's first non-finite sample: up to 15 frames earlier than the reference (rounds 4 - 5
    stated that superset here).  Round 6: k_rsos_fixup recomputes that block sample by sample -- the set is the reference's,
    and the frames in front of the sample carry the reference's values."""
    rng = np.random.default_rng(10)
    d = rng.standard_normal((400000, 8))
    first = {3: 123457, 5: 300000, 6: 16 * 777 + 15, 7: 16 * 999}   # (middle of a block, start of one, its last frame, its first)
    d[first[3], 3] = np.nan
    d[first[5], 5] = np.inf
    d[first[6], 6] = -np.inf
    d[first[7], 7] = np.nan
    x = so.Signal(F(d), 44.1 * so.kHz) | so.Filt(so.Lowpass, 4 * so.kHz)
    a, b, fused = both(x)
    want = oracle_sink(x)
    assert fused
    assert np.array_equal(np.isfinite(b), np.isfinite(want))  # (the three-pass form: exactly the reference's set)
    expect = np.ones(d.shape, dtype=bool)
    for c, i in first.items():
        assert not np.isfinite(want[i:, c]).any() and np.isfinite(want[:i, c]).all()
        expect[i:, c] = False
    assert np.array_equal(np.isfinite(a), expect)
    assert relerr(a[expect], want[expect]) < 1e-10
    with env(SIGOPS_RSOS_MINGROUPS=1, SIGOPS_RSOS_NO_FIXUP=1):   # (the kernel's own set, as it was)
        raw = so.sink(x)[0]
    for c, i in first.items():
        expect[i // 16 * 16:, c] = False
    assert np.array_equal(np.isfinite(raw), expect)


def test_under_append_and_with_a_ramp_behind_it():
    """window aliasing: the filtered scenes write their windows of the result themselves"""
    rng = np.random.default_rng(11)
    scenes = [so.Signal(F(rng.standard_normal((200000, 2))), 44.1 * so.kHz) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.Ramp(10 * so.ms)
              for _ in range(3)]
    x = so.Append(*scenes)
    a, b, fused = both(x)
    assert fused
    assert relerr(a, b) < 1e-11 and relerr(a, oracle_sink(x)) < 1e-10


def test_default_policy_takes_one_pass_for_long_signals_only():
    rng = np.random.default_rng(12)
    short = so.Signal(F(rng.standard_normal((100000, 8))), 44.1 * so.kHz) | so.Filt(so.Lowpass, 4 * so.kHz)
    long_ = so.Signal(F(rng.standard_normal((3000000, 8))), 44.1 * so.kHz) | so.Filt(so.Lowpass, 4 * so.kHz)
    with env(SIGOPS_RSOS_MINGROUPS=None, SIGOPS_NO_RSOS=None, SIGOPS_NO_PLAIN_RSOS=None):
        assert "k_rsos" not in steps_of(short)
        assert steps_of(long_) == ["k_rsos"]
        got = so.sink(long_)[0]
    with env(SIGOPS_NO_PLAIN_RSOS=1):
        ref = so.sink(long_)[0]
    assert relerr(got, ref) < 1e-11


def test_run_to_run_identity():
    rng = np.random.default_rng(13)
    x = so.Signal(F(rng.standard_normal((500000, 8))), 44.1 * so.kHz) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz)
    with env(SIGOPS_RSOS_MINGROUPS=1):
        a = so.sink(x)[0]
        for _ in range(4):
            assert np.array_equal(a, so.sink(x)[0])


def test_loud_then_quiet_with_a_local_tolerance():
    """Every range of the one-pass form starts from rest a warm-up early: what the cut (2^-70 of the state's level at the
    warm-up's start, SIGOPS_PLAIN_WTOL) drops must stay invisible where a range begins in near silence behind a loud passage.
    One second at level 1, then 120 dB down: BLOCKWISE (1024 frames) relative error against the sequential oracle."""
    rng = np.random.default_rng(14)
    n = 4_000_000
    lvl = np.where(np.arange(n) < 44100, 1.0, 1e-6)[:, None]
    d = rng.standard_normal((n, 8)) * lvl
    for filt in (so.Filt(so.Lowpass, 300 * so.Hz), so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz)):
        x = so.Signal(F(d), 44.1 * so.kHz) | filt
        a, b, fused = both(x)
        assert fused
        want = oracle_sink(x)
        nb = n // 1024
        w = want[: nb * 1024].reshape(nb, 1024, -1)
        for got in (a, b):
            e = np.linalg.norm(got[: nb * 1024].reshape(nb, 1024, -1) - w, axis=(1, 2)) / np.linalg.norm(w, axis=(1, 2))
            assert e.max() < 1e-9, (float(e.max()), int(e.argmax()))


def test_below_a_normpower_whose_region_starts_at_the_filters_first_frame():
    """`Filt |> Normpower` (reference src/filters.jl:296-309: vals filled from the child, one rms, every sample divided): the
    filter takes the one-pass form when every Normpower above it reads it from its first frame on (Stage::norm_df == 0) -- what
    the warm starts cut is 2^-70 of something inside the region the rms is taken over; 12.5 M x 8: 1.25 -> 0.87 ms.  A
    Normpower of a window behind an `After` -- possibly a decayed tail alone (tests/test_gpu_fences.py) -- keeps the exact scan."""
    rng = np.random.default_rng(15)
    n = 700_000
    d = rng.standard_normal((n, 8))
    d[100_000:] *= 1e-4  # (a loud start, then 80 dB down)
    x = so.Signal(F(d), 44.1 * so.kHz) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz)
    for tree, one_pass in ((x | so.Normpower, True),
                           (x | so.Until(500_000 * so.frames) | so.Normpower | so.Amplify(-20 * so.dB), True),
                           (x | so.After(300_000 * so.frames) | so.Normpower, False),
                           (so.Mix(x | so.Normpower, x | so.After(1000 * so.frames) | so.Normpower | so.Pad(so.zero) | so.Until(n * so.frames)), False)):
        with env(SIGOPS_RSOS_MINGROUPS=1, SIGOPS_NO_PLAIN_RSOS=None, SIGOPS_NO_RSOS=None):  # (the estimate aside: short signals)
            names = steps_of(tree)
            got = so.sink(tree)[0]
        assert ("k_rsos" in names) == one_pass and ("k_sos" in names) == (not one_pass), names
        want = oracle_sink(tree)
        assert relerr(got, want) < 1e-9
        with env(SIGOPS_RSOS_MINGROUPS=1, SIGOPS_NORM_EXACT_FILT=1):
            assert "k_rsos" not in steps_of(tree)
            assert relerr(so.sink(tree)[0], want) < 1e-9


def test_resampler_and_filter_in_one_launch_below_a_whole_signal_normpower():
    """`Filt |> ToFramerate |> Normpower` (the reference resamples, then filters: src/filters.jl:143-148): the fused kernel below a
    Normpower whose region starts at the filter's first frame, its warm-up cut at 2^-70 there; behind an `After` resampler and
    exact scan stay apart"""
    rng = np.random.default_rng(16)
    d = rng.standard_normal((600_000, 8))
    d[50_000:] *= 1e-3
    x = so.Signal(F(d), 44.1 * so.kHz) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.ToFramerate(48 * so.kHz)
    for tree, fused in ((x | so.Normpower, True), (x | so.After(2 * so.s) | so.Normpower, False)):
        with env(SIGOPS_RSOS_MINGROUPS=1, SIGOPS_NO_RSOS=None):
            names = steps_of(tree)
            got = so.sink(tree)[0]
        assert (names[0] == "k_rsos") == fused, names
        assert relerr(got, oracle_sink(tree)) < 1e-9
