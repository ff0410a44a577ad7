"""Pins the CPU oracle against the reference's own exact known-answer tests
(test/runtests.jl; SURVEY.md Appendix E "exact" rows).  CPU only."""
import numpy as np
import pytest

import sigops_amd as so
from sigops_amd import (Signal, Until, After, Pad, Extend, Append, Mix, Amplify, AddChannel, OperateOn,
                        RampOn, RampOff, Ramp, FadeTo, Filt, Normpower, ToFramerate, ToChannels, Lowpass,
                        Highpass, Chebyshev1, cycle, mirror, lastframe, zero, one, identity, nframes,
                        ErrorException, s, ms, Hz, kHz, frames, dB, sin, cos)
from cases import CASES, F, rng
from oracle_bridge import oracle_sink, relerr


def A(x, **kw):
    return oracle_sink(x, **kw)


def test_array_tuple_output():  # runtests.jl:66-70
    x = F(rng().random((10, 2)))
    assert np.array_equal(A(Mix(Signal(x, 10 * Hz), 1)), x + 1)


def test_function_signals():  # runtests.jl:73-87
    ref = A(Signal(sin, ω=5 * Hz, ϕ=np.pi) | Until(1 * s) | ToFramerate(20 * Hz))
    assert np.array_equal(ref, A(Signal(sin, ω=5 * Hz, ϕ=np.pi * so.rad) | Until(1 * s) | ToFramerate(20 * Hz)))
    assert np.array_equal(ref, A(Signal(sin, ω=5 * Hz, ϕ=100 * ms) | Until(1 * s) | ToFramerate(20 * Hz)))
    assert np.array_equal(ref, A(Signal(sin, ω=5 * Hz, ϕ=180 * so.deg) | Until(1 * s) | ToFramerate(20 * Hz)))
    a = A(Signal(sin, ϕ=1 * s) | Until(1 * s) | ToFramerate(20 * Hz))
    b = A(Signal(sin, ω=1 * Hz, ϕ=0) | Until(1 * s) | ToFramerate(20 * Hz))
    assert np.allclose(a, b, rtol=1e-8, atol=1e-12)
    # first sample is t = 1/fs: sinpi(2*(1/20*5 + 0.5)) = sin(pi*1.5)
    assert ref[0, 0] == pytest.approx(-1.0)
    with pytest.raises(ErrorException):
        Signal(sin, ϕ=2 * np.pi * so.rad)


def test_sink_to_arrays_bump():  # runtests.jl:90-93
    tone = A(Signal(sin, 44.1 * kHz, ω=100 * Hz) | Until(5 * s))
    assert tone[0, 0] < tone[109, 0]


def test_change_channel_count():  # runtests.jl:103-114
    tone = Signal(sin, 22 * Hz, ω=10 * Hz) | Until(5 * s)
    data = A(tone | ToChannels(2))
    assert data.shape == (110, 2)
    data2 = A(Signal(data, 22 * Hz) | ToChannels(1))
    assert np.array_equal(data2, data.sum(axis=1, keepdims=True))
    with pytest.raises(ErrorException):
        tone | ToChannels(2) | ToChannels(3)


@pytest.mark.parametrize("nch", [1, 2])
def test_cutting(nch):  # runtests.jl:117-171
    x = F(rng().random((12, nch)))
    got = A(Signal(x, 6 * Hz) | After(0.5 * s) | Until(1 * s) | Mix(0.0))
    assert np.array_equal(got, x[3:9] + 0.0)
    got = A(Signal(x, 6 * Hz) | Until(1 * s) | After(0.5 * s) | Mix(0.0))
    assert np.array_equal(got, x[3:6])
    with pytest.raises(ErrorException):
        A(Signal(np.arange(1.0, 11.0), 5 * Hz) | After(3 * s))
    x = F(rng().random((20, nch)))
    assert np.array_equal(A(so.Window(x, from_=0 * frames, to=5 * frames) | Mix(0.0)), x[0:5])
    assert np.array_equal(A(so.Window(x, from_=15 * frames, to=25 * frames) | Mix(0.0)), x[15:20])
    u = A(Until(np.arange(1.0, 11.0), 5 * frames) | Mix(0.0))
    assert np.array_equal(u[:, 0], np.arange(1.0, 6.0))
    assert A(Until(np.arange(1.0, 11.0), -5 * frames) | Mix(0.0)).shape[0] == 0


@pytest.mark.parametrize("nch", [1, 2, 3])
def test_padding(nch):  # runtests.jl:174-231
    tone = A(Signal(sin, 22 * Hz, ω=10 * Hz) | ToChannels(nch) | Until(5 * s) | Pad(zero) | Until(7 * s))
    assert np.mean(np.abs(tone[: 22 * 5])) > 0
    assert np.mean(np.abs(tone[22 * 5:])) == 0
    x = F(rng().random((10, nch)))
    z = A(x | Signal(10 * Hz) | Pad(zero) | After(15 * frames) | Until(10 * frames))
    assert np.array_equal(z, np.zeros((10, nch)))
    r = A(Pad(Signal(x, 10 * Hz), cycle) | Until(30 * frames))
    assert np.array_equal(r, np.vstack([x, x, x]))
    r = A(Pad(Signal(x, 10 * Hz), mirror) | Until(30 * frames))
    assert np.array_equal(r, np.vstack([x, x[::-1], x]))
    r = A(Pad(Signal(x, 10 * Hz), lastframe) | Until(15 * frames))
    assert np.all(r[10:] == r[9:10])
    xs = Signal(sin, 10 * Hz) | ToChannels(nch) | Until(1 * s)
    with pytest.raises(ErrorException):
        A(Pad(xs, cycle) | Until(15 * frames))
    r = A(Pad(xs, lastframe) | Until(15 * frames))
    assert np.all(r[10:] == r[9:10])
    padv = rng(5).random(nch)
    r = A(Pad(xs, padv) | Until(15 * frames))
    assert np.all(r[10:] == padv[None, :])
    x5 = F(5 * np.ones((5, nch)))
    r = A(Pad(x5, zero) | Until(10 * frames) | ToFramerate(10 * Hz))
    assert np.all(r[5:] == 0) and np.all(r[:5] == 5)


@pytest.mark.parametrize("nch", [1, 2])
def test_appending(nch):  # runtests.jl:234-257
    a = Signal(sin, 22 * Hz, ω=10 * Hz) | ToChannels(nch) | Until(5 * s)
    b = Signal(sin, 22 * Hz, ω=5 * Hz) | ToChannels(nch) | Until(5 * s)
    t = A(a | Append(b))
    assert t.shape[0] == 220
    assert np.array_equal(t, np.vstack([A(a), A(b)]))


def test_mixing():  # runtests.jl:260-277
    x = F(rng(1).random((20, 65)))
    y = F(rng(2).random((20, 65)))
    assert np.array_equal(A(Mix(x, y) | ToFramerate(20 * Hz)), x + y)
    x = F(rng(3).random((20, 2)))
    r = A(OperateOn("reverse", x, bychannel=False) | ToFramerate(20 * Hz))
    assert np.array_equal(r, x[:, ::-1])


@pytest.mark.parametrize("nch", [1, 2])
def test_padded_mix_amplify(nch):  # runtests.jl:280-311
    fs = 3 * Hz
    a = Signal(2, fs) | ToChannels(nch) | Until(2 * s) | Append(Signal(3, fs)) | Until(4 * s)
    b = Signal(3, fs) | ToChannels(nch) | Until(3 * s)
    r = A(Mix(a, b))
    want = np.concatenate([np.full(6, 2) + np.full(6, 3), np.full(3, 3) + np.full(3, 3), np.full(3, 3)])
    for ch in range(nch):
        assert np.array_equal(r[:, ch], want)
    r = A(Amplify(a, b))
    want = np.concatenate([np.full(6, 2) * np.full(6, 3), np.full(3, 3) * np.full(3, 3), np.full(3, 3)])
    for ch in range(nch):
        assert np.array_equal(r[:, ch], want)


def test_addchannel_zero_extension():  # runtests.jl:306-310
    x = F(rng(3).random((10, 2)))
    y = F(rng(4).random((5, 2)))
    z = A(Signal(x, 10 * Hz) | AddChannel(y))
    assert np.array_equal(z[:, :2], x)
    assert np.array_equal(z[:5, 2:], y)
    assert np.all(z[5:, 2:] == 0)


@pytest.mark.parametrize("nch", [1, 2])
def test_ramps(nch):  # runtests.jl:373-404
    tone = Signal(sin, 50 * Hz, ω=10 * Hz) | ToChannels(nch) | Until(5 * s)
    ramped = A(tone | Ramp(500 * ms))
    t = A(tone)
    assert np.mean(ramped[:25] ** 2) < np.mean(ramped[25:50] ** 2)
    assert np.mean(ramped[225:] ** 2) < np.mean(ramped[200:225] ** 2)
    assert np.mean(np.abs(ramped)) < np.mean(np.abs(t))
    assert np.mean(ramped) < 1e-4
    # closed form: g_on[n] = sinpi(0.5 (n-1)/R), g_off[n] = sinpi(0.5 (1-(n-M)/R))
    R, N = 25, 250
    n = np.arange(1, N + 1)
    g = np.where(n <= R, np.sin(np.pi * 0.5 * (n - 1) / R), 1.0)
    g = g * np.where(n > N - R, np.sin(np.pi * 0.5 * (1 - (n - (N - R)) / R)), 1.0)
    assert np.allclose(ramped[:, 0], t[:, 0] * g, rtol=1e-13, atol=1e-15)
    x = Signal(sin, 22 * Hz, ω=10 * Hz) | ToChannels(nch) | Until(2 * s)
    y = Signal(sin, 22 * Hz, ω=5 * Hz) | ToChannels(nch) | Until(2 * s)
    fading = FadeTo(x, y, 500 * ms)
    res = A(fading)
    assert nframes(fading) == int(np.ceil((2 + 2 - 0.5) * 22))
    assert np.array_equal(res[:33], A(x)[:33])
    assert np.array_equal(res[43:], A(y)[10:])
    r2 = A(Signal(sin, 500 * Hz, ω=20 * Hz, ϕ=np.pi / 2) | ToChannels(nch) | Until(100 * ms) | Ramp(identity))
    assert np.mean(np.abs(r2[:5])) < np.mean(np.abs(r2[5:10]))
    r2 = A(Signal(sin, 500 * Hz, ω=20 * Hz, ϕ=np.pi / 2) | ToChannels(nch) | Until(100 * ms) | RampOff(identity))
    assert np.mean(np.abs(r2[6:10])) < np.mean(np.abs(r2[:6]))


@pytest.mark.parametrize("nch", [1, 2])
def test_normpower(nch):  # runtests.jl:491-500
    tone = Signal(sin, 10 * Hz, ω=2 * Hz) | ToChannels(nch) | Until(2 * s) | Ramp | Normpower
    assert np.allclose(np.sqrt(np.mean(A(tone) ** 2, axis=0)), 1)
    res = A(tone | ToFramerate(20 * Hz))
    assert np.allclose(np.sqrt(np.mean(res ** 2, axis=0)), 1)


@pytest.mark.parametrize("nch", [1, 2])
def test_arrays_numbers_db(nch):  # runtests.jl:503-547
    tone = A(Signal(sin, 200 * Hz, ω=10 * Hz) | ToChannels(nch) | Mix(1.5) | Until(5 * s))
    assert np.all(tone >= 0.5)
    assert np.all(A(10 | ToChannels(nch) | Until(1 * s) | ToFramerate(10 * Hz)) == 10)
    assert np.all(A(Signal(1, 10 * Hz) | ToChannels(nch) | Until(1 * s) | Amplify(20 * dB)) == 10)
    assert np.all(A(Signal(1, 10 * Hz) | ToChannels(nch) | Until(1 * s) | Amplify(40 * dB)) == 100)
    tone = A(Signal(sin, 200 * Hz, ω=10 * Hz) | ToChannels(nch) | Until(10 * frames) | Mix(10.0 * np.arange(1, 11)))
    assert np.all(tone >= 10.0 * np.arange(1, 11)[:, None] - 1.0)


@pytest.mark.parametrize("nch", [1, 2])
def test_infinite_signals(nch):  # runtests.jl:550-575
    t = Signal(sin, 200 * Hz, ω=10 * Hz) | ToChannels(nch) | Until(10 * frames) | After(5 * frames) | After(2 * frames)
    assert nframes(t) == 3 and A(t).shape == (3, nch)
    t = Signal(sin, 200 * Hz, ω=10 * Hz) | ToChannels(nch) | After(5 * frames) | Until(5 * frames)
    assert A(t).shape == (5, nch) and A(t)[0, 0] > 0.9
    t = Signal(sin, 200 * Hz, ω=10 * Hz) | ToChannels(nch) | Until(10 * frames) | After(5 * frames)
    assert A(t)[0, 0] > 0.9
    with pytest.raises(ErrorException):
        A(Signal(sin, 200 * Hz) | ToChannels(nch))


def test_stress_exact():  # runtests.jl:815-821, 874-878
    a = Until(sin, 2 * s)
    b = Until(cos, 2 * s)
    x = Append(a, b) | After(3 * s)
    assert np.array_equal(A(x | ToFramerate(20 * Hz)), A(b | After(1 * s) | ToFramerate(20 * Hz)))
    x = Append(1 | Until(1 * s), 2 | Until(2 * s))
    y = Append(3 | Until(2 * s), 4 | Until(1 * s))
    r = A(Mix(x, y) | ToFramerate(10 * Hz))
    assert np.all(r[:, 0] == np.concatenate([np.full(10, 4), np.full(10, 5), np.full(10, 6)]))


def test_float32_stays_float32():  # runtests.jl:707-729
    for name in ("float32_chain", "float32_normpower", "float32_append_pad", "filt_float32",
                 "resample_441_48_f32"):
        x = CASES[name]()
        assert x.dtype == np.float32
        assert A(x).dtype == np.float32


def test_config1_known_answers():
    x = Signal(sin, ω=1 * kHz) | Until(5 * s) | Ramp | Normpower | Amplify(-20 * dB) | ToFramerate(44.1 * kHz)
    r = A(x)
    assert r.shape == (220500, 1)
    assert r[0, 0] == 0.0
    assert np.sqrt(np.mean(r ** 2)) == pytest.approx(0.1, rel=1e-12)


@pytest.mark.parametrize("name", ["filt_highpass_cheby", "filt_after", "filt_append_short_blocks",
                                  "filt_chain", "filt_long_two_channel", "resample_up2", "resample_half",
                                  "resample_3_2", "resample_441_48", "resample_then_filter",
                                  "benchmark_overall", "stress_mix_middle"])
@pytest.mark.parametrize("blocksize", [5, 64, 1000])
def test_blocksize_invariance(name, blocksize):
    """runtests.jl:353-356,430-433,797-806,861-866: results do not depend on the block size
    (bit-exact for IIR, where the reference asserts ==; ≈ for resampling)."""
    x = CASES[name]()
    base = A(x)
    if "resample" in name or name in ("benchmark_overall", "stress_mix_middle"):
        if blocksize < 64:
            pytest.skip("reference errors: blocksize too small for the resampling filter (runtests.jl:810)")
        other = A(x, blocksize=blocksize)
        assert relerr(other, base) < 1e-12
    else:
        assert np.array_equal(A(x, blocksize=blocksize), base)


@pytest.mark.parametrize("nch", [1, 2])
def test_filtering_spectral_inequalities(nch):  # runtests.jl:314-350: the reference's only value-level pins of Filt
    from spectral_checks import filtering_inequalities

    filtering_inequalities(A, nch)


def test_filter_state_after():  # runtests.jl:358-362
    full = A(CASES["filt_highpass_cheby"]())
    aft = A(CASES["filt_after"]())
    assert np.array_equal(aft, full[100:])


def test_resampler_too_small_blocksize():  # runtests.jl:808-811
    y = Signal(F(np.ones((10, 2))), 10 * Hz)
    assert A(ToFramerate(y, 40 * Hz)).shape == (40, 2)
    assert A(ToFramerate(y, 5 * Hz)).shape == (5, 2)
    with pytest.raises(ErrorException):
        A(ToFramerate(y, 40 * Hz, blocksize=5), blocksize=5)
