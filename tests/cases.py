"""Seeded tree builders shared by the oracle known-answer tests (CPU) and the GPU
parity tests.  Each case mirrors a group of reference test/runtests.jl (cited)."""
import numpy as np

import sigops_amd as so
from sigops_amd import (Signal, Until, After, Pad, Extend, Append, Prepend, Mix, Amplify, AddChannel,
                        SelectChannel, OperateOn, RampOn, RampOff, Ramp, FadeTo, Filt, Normpower,
                        ToFramerate, ToChannels, ToEltype, Lowpass, Highpass, Bandpass, Bandstop,
                        Chebyshev1, Butterworth, cycle, mirror, lastframe, zero, one, identity,
                        s, ms, Hz, kHz, frames, dB, sin, cos)

CASES = {}


def case(fn):
    CASES[fn.__name__] = fn
    return fn


def rng(seed=1983):
    return np.random.default_rng(seed)


def F(a):
    return np.asfortranarray(a)


@case
def array_plus_one():  # runtests.jl:66-70
    x = F(rng().random((10, 2)))
    return Mix(Signal(x, 10 * Hz), 1)


@case
def function_phase():  # runtests.jl:73-81
    return Signal(sin, ω=5 * Hz, ϕ=np.pi) | Until(1 * s) | ToFramerate(20 * Hz)


@case
def sin_no_omega():  # runtests.jl:80-81: sin without ω is a 1 Hz tone
    return Signal(sin, ϕ=1 * s) | Until(1 * s) | ToFramerate(20 * Hz)


@case
def cos_and_identity():
    return Mix(Signal(cos, ω=3 * Hz), Signal(identity, ω=2 * Hz), Signal(cos)) | Until(50 * frames) \
        | ToFramerate(25 * Hz)


@case
def tochannels_sum():  # runtests.jl:103-111
    tone = Signal(sin, 22 * Hz, ω=10 * Hz) | Until(5 * s)
    return tone | ToChannels(2) | ToChannels(1)


@case
def cut_after_until():  # runtests.jl:129-136
    x = F(rng().random((12, 2)))
    return Signal(x, 6 * Hz) | After(0.5 * s) | Until(1 * s) | Mix(0.0)


@case
def pad_zero_after():  # runtests.jl:186-187
    x = F(rng().random((10, 3)))
    return x | Signal(10 * Hz) | Pad(zero) | After(15 * frames) | Until(10 * frames)


@case
def pad_cycle():  # runtests.jl:194-196
    x = F(rng().random((10, 3)))
    return Pad(Signal(x, 10 * Hz), cycle) | Until(30 * frames)


@case
def pad_mirror():  # runtests.jl:197-198
    x = F(rng().random((10, 3)))
    return Pad(Signal(x, 10 * Hz), mirror) | Until(35 * frames)


@case
def pad_lastframe_array():  # runtests.jl:199-200
    x = F(rng().random((10, 2)))
    return Pad(Signal(x, 10 * Hz), lastframe) | Until(15 * frames)


@case
def pad_lastframe_fn():  # runtests.jl:202-206
    x = Signal(sin, 10 * Hz) | ToChannels(2) | Until(1 * s)
    return Pad(x, lastframe) | Until(15 * frames)


@case
def pad_vector():  # runtests.jl:207-209
    x = Signal(sin, 10 * Hz) | ToChannels(3) | Until(1 * s)
    return Pad(x, [0.25, 0.5, 0.75]) | Until(15 * frames)


@case
def append_tones():  # runtests.jl:236-241
    a = Signal(sin, 22 * Hz, ω=10 * Hz) | ToChannels(2) | Until(5 * s)
    b = Signal(sin, 22 * Hz, ω=5 * Hz) | ToChannels(2) | Until(5 * s)
    return a | Append(b)


@case
def mix_65_channels():  # runtests.jl:269-271
    x = F(rng(1).random((20, 65)))
    y = F(rng(2).random((20, 65)))
    return Mix(x, y) | ToFramerate(20 * Hz)


@case
def reverse_channels():  # runtests.jl:273-276
    x = F(rng().random((20, 2)))
    return OperateOn("reverse", x, bychannel=False) | ToFramerate(20 * Hz)


def _padded_ab(nch):
    fs = 3 * Hz
    a = Signal(2, fs) | ToChannels(nch) | Until(2 * s) | Append(Signal(3, fs)) | Until(4 * s)
    b = Signal(3, fs) | ToChannels(nch) | Until(3 * s)
    return a, b


@case
def padded_mix():  # runtests.jl:283-294
    a, b = _padded_ab(2)
    return Mix(a, b) | ToEltype(np.float64)


@case
def padded_amplify():  # runtests.jl:296-303
    a, b = _padded_ab(2)
    return Amplify(a, b) | ToEltype(np.float64)


@case
def addchannel_extend():  # runtests.jl:306-310
    x = F(rng(3).random((10, 2)))
    y = F(rng(4).random((5, 2)))
    return Signal(x, 10 * Hz) | AddChannel(y)


@case
def select_channel():
    x = F(rng(5).random((16, 3)))
    return Signal(x, 8 * Hz) | SelectChannel(2) | Amplify(2.0)


@case
def ramp_sine():  # runtests.jl:375-377
    return Signal(sin, 50 * Hz, ω=10 * Hz) | ToChannels(2) | Until(5 * s) | Ramp(500 * ms)


@case
def ramp_identity():  # runtests.jl:394-402
    return Signal(sin, 500 * Hz, ω=20 * Hz, ϕ=np.pi / 2) | ToChannels(2) | Until(100 * ms) | Ramp(identity)


@case
def rampon_frames_array():  # runtests.jl:803-804 (input side)
    x = F(np.ones((25, 2)))
    return Signal(x, 10 * Hz) | RampOn(7 * frames)


@case
def fadeto():  # runtests.jl:385-392
    x = Signal(sin, 22 * Hz, ω=10 * Hz) | ToChannels(2) | Until(2 * s)
    y = Signal(sin, 22 * Hz, ω=5 * Hz) | ToChannels(2) | Until(2 * s)
    return FadeTo(x, y, 500 * ms)


@case
def normpower_ramp():  # runtests.jl:493-495
    return Signal(sin, 10 * Hz, ω=2 * Hz) | ToChannels(2) | Until(2 * s) | Ramp | Normpower


@case
def gain_db():  # runtests.jl:521-526
    return Signal(1.0, 10 * Hz) | ToChannels(2) | Until(1 * s) | Amplify(20 * dB)


@case
def mix_number_array():  # runtests.jl:512-514, 529-531
    return Signal(sin, 200 * Hz, ω=10 * Hz) | ToChannels(2) | Until(10 * frames) | Mix(10.0 * np.arange(1, 11))


@case
def infinite_after_until():  # runtests.jl:557-561
    return Signal(sin, 200 * Hz, ω=10 * Hz) | ToChannels(2) | After(5 * frames) | Until(5 * frames)


@case
def offset_append_sum():  # runtests.jl:874-878
    x = Append(1.0 | Until(1 * s), 2.0 | Until(2 * s))
    y = Append(3.0 | Until(2 * s), 4.0 | Until(1 * s))
    return Mix(x, y) | ToFramerate(10 * Hz)


@case
def append_after_drop_first():  # runtests.jl:817-821
    a = Until(sin, 2 * s)
    b = Until(cos, 2 * s)
    return Append(a, b) | After(3 * s) | ToFramerate(20 * Hz)


@case
def many_cuts():  # runtests.jl:869-872
    return Signal(sin, ω=5 * Hz) | After(2 * s) | Until(20 * s) | After(2 * s) | Until(15 * s) | After(2 * s) \
        | After(2 * s) | Until(5 * s) | Until(2 * s) | ToFramerate(12 * Hz)


@case
def float32_chain():  # runtests.jl:707-729
    x = Signal(F(rng(6).random((100, 2)).astype(np.float32)), 10 * Hz)
    y = Signal(F(rng(7).random((50, 2)).astype(np.float32)), 10 * Hz)
    return x | Mix(y) | Ramp | Amplify(np.float32(0.5))


@case
def float32_normpower():  # runtests.jl:722
    x = Signal(F(rng(8).random((100, 2)).astype(np.float32)), 10 * Hz)
    return x | Normpower | Amplify(np.float32(-10.0) * dB)


@case
def float32_append_pad():  # runtests.jl:717-720
    x = Signal(F(rng(6).random((100, 2)).astype(np.float32)), 10 * Hz)
    y = Signal(F(rng(7).random((50, 2)).astype(np.float32)), 10 * Hz)
    return x | Append(y) | Pad(zero) | Until(17 * s)


@case
def sub_div():
    x = F(rng(9).random((40, 2)) + 1.0)
    y = F(rng(10).random((30, 2)) + 1.0)
    return OperateOn("/", OperateOn("-", Signal(x, 10 * Hz), y), y)


@case
def negate():
    import operator
    x = F(rng(11).random((40, 2)))
    return OperateOn(operator.neg, Signal(x, 10 * Hz))


@case
def strided_array():  # AxisArrays with time on dim 2 arrive as strides (SURVEY §8b layout)
    x = rng(12).random((2, 64))  # channels x time, C-order
    return Signal(x.T, 8 * Hz) | Amplify(3.0)


# ---- filters ---------------------------------------------------------------
def _cmplx(nch, fs=100):
    a = Signal(sin, fs * Hz, ω=10 * Hz) | ToChannels(nch) | Until(5 * s)
    b = Signal(sin, fs * Hz, ω=5 * Hz) | ToChannels(nch) | Until(5 * s)
    return Mix(a, b)


@case
def filt_highpass_cheby():  # runtests.jl:319-320
    return _cmplx(2) | Filt(Highpass, 8 * Hz, method=Chebyshev1(5, 1))


@case
def filt_lowpass_butter():  # runtests.jl:321-322
    return _cmplx(2) | Filt(Lowpass, 6 * Hz, method=Butterworth(5))


@case
def filt_bandpass():  # runtests.jl:325-328
    return _cmplx(1) | Filt(Bandpass, 20 * Hz, 30 * Hz, method=Chebyshev1(5, 1))


@case
def filt_bandstop():  # runtests.jl:329-332
    return _cmplx(2) | Filt(Bandstop, 2 * Hz, 12 * Hz, method=Chebyshev1(5, 1))


@case
def filt_after():  # runtests.jl:358-362: filter state under After
    return _cmplx(2) | Filt(Highpass, 8 * Hz, method=Chebyshev1(5, 1), blocksize=64) | After(1 * s)


@case
def filt_append_short_blocks():  # runtests.jl:794-798
    x = Signal(F(np.ones((25, 2))), 10 * Hz)
    y = Signal(F(np.ones((10, 2))), 10 * Hz)
    z = Signal(F(np.ones((15, 2))), 10 * Hz)
    return x | Append(y) | Append(z) | Filt(Lowpass, 3 * Hz, blocksize=5)


@case
def filt_pad_until_append():  # runtests.jl:800-801
    x = Signal(F(np.ones((25, 2))), 10 * Hz)
    y = Signal(F(np.ones((10, 2))), 10 * Hz)
    return x | Pad(zero) | Until(15 * s) | Append(y) | Filt(Lowpass, 3 * Hz, blocksize=5)


@case
def filt_ramp():  # runtests.jl:805-806
    x = Signal(F(np.ones((25, 2))), 10 * Hz)
    return x | Ramp(3 * frames) | Filt(Lowpass, 3 * Hz)


@case
def filt_long_two_channel():  # BASELINE config 2 in miniature
    noise = F(rng(13).standard_normal((60000, 2)))
    return Mix(Signal(sin, ω=1 * kHz) | Until(60000 * frames), Signal(noise, 44.1 * kHz)) \
        | Filt(Bandstop, 0.5 * kHz, 2 * kHz)


@case
def filt_chain():  # runtests.jl:847-851
    noise = F(rng(14).standard_normal((120, 1)))
    return Signal(noise, 20 * Hz) | Filt(Lowpass, 9 * Hz) | Mix(Signal(sin, ω=12 * Hz) | Until(6 * s)) \
        | Filt(Highpass, 4 * Hz, method=Chebyshev1(5, 1))


@case
def filt_float32():  # runtests.jl:721
    x = Signal(F(rng(15).random((100, 2)).astype(np.float32)), 10 * Hz)
    return x | Filt(Lowpass, 3 * Hz)


@case
def filt_slow_pole():  # poles very close to the unit circle: long state memory
    noise = F(rng(16).standard_normal((200000, 1)))
    return Signal(noise, 44.1 * kHz) | Filt(Highpass, 20 * Hz, order=3)


# ---- resampling ------------------------------------------------------------
@case
def resample_up2():  # runtests.jl:422-427 (ratio 2//1, FIRInterpolator)
    tone = Signal(sin, 20 * Hz, ω=5 * Hz) | ToChannels(2) | Until(5 * s)
    data = F(rng(17).random((100, 2)))
    return ToFramerate(Signal(data, 20 * Hz), 40 * Hz)


@case
def resample_down_075():  # runtests.jl:414-417 (0.75: arbitrary kernel since max(3,4) > 3)
    data = F(rng(18).random((100, 2)))
    return ToFramerate(Signal(data, 20 * Hz), 15 * Hz)


@case
def resample_half():  # benchmarks.jl resampling 1 kHz -> 500 Hz (1//2, FIRDecimator)
    data = F(rng(19).random((1000, 2)))
    return Signal(data, 1000 * Hz) | ToFramerate(500 * Hz)


@case
def resample_3_2():  # FIRRational
    data = F(rng(20).random((400, 2)))
    return Signal(data, 20 * Hz) | ToFramerate(30 * Hz)


@case
def resample_2_3():
    data = F(rng(21).random((400, 1)))
    return Signal(data, 30 * Hz) | ToFramerate(20 * Hz)


@case
def resample_third():
    data = F(rng(22).random((400, 1)))
    return Signal(data, 30 * Hz) | ToFramerate(10 * Hz)


@case
def resample_pi():  # benchmarks.jl resampling-irrational
    data = F(rng(23).random((2000, 2)))
    return Signal(data, 1000 * Hz) | ToFramerate(np.pi * 1000 * Hz)


@case
def resample_441_48():  # BASELINE config 3 in miniature
    noise = F(rng(24).standard_normal((44100, 4)))
    return Signal(noise, 44.1 * kHz) | Amplify(Signal(sin, ω=5 * Hz)) | Until(1 * s) | ToFramerate(48 * kHz)


@case
def resample_441_48_f32():
    noise = F(rng(25).standard_normal((22050, 2)).astype(np.float32))
    return Signal(noise, 44.1 * kHz) | ToFramerate(48 * kHz)


@case
def resample_then_filter():  # BASELINE config 5 rewrite: Filt over data -> resample first
    x = F(rng(26).random((20000, 3)))
    return Signal(x, 44.1 * kHz) | Filt(Lowpass, 4 * kHz) | ToFramerate(16 * kHz)


@case
def resample_down_8ch():  # K3r MFMA path: ct=8, pb=4 (44.1 kHz -> 16 kHz, 441:160)
    x = F(rng(41).standard_normal((12000, 8)))
    return Signal(x, 44.1 * kHz) | ToFramerate(16 * kHz)


@case
def resample_down_f32():  # K3r with fp32 tiles, 2 channels (ct=2, pb=32)
    x = F(rng(42).standard_normal((12000, 2)).astype(np.float32))
    return Signal(x, 44.1 * kHz) | ToFramerate(16 * kHz)


@case
def resample_down_then_until():  # K3r, mono, output cut short of a whole period
    x = F(rng(43).standard_normal((20000, 1)))
    return Signal(x, 48 * kHz) | ToFramerate(11.025 * kHz) | Until(4000 * frames)


@case
def resample_padded_computed():  # runtests.jl:440-443
    tone = Signal(sin, 20 * Hz, ω=5 * Hz) | ToChannels(2) | Until(5 * s)
    return tone | Pad(one) | Until(7 * s) | ToFramerate(40 * Hz)


@case
def normpower_resampled():  # runtests.jl:497-498
    tone = Signal(sin, 10 * Hz, ω=2 * Hz) | ToChannels(2) | Until(2 * s) | Ramp | Normpower
    return tone | ToFramerate(20 * Hz)


@case
def stress_multirate():  # runtests.jl:839-844
    return Signal(sin, ω=10 * Hz, fs=20 * Hz) | Until(4 * s) | ToFramerate(30 * Hz) | Filt(Lowpass, 10 * Hz) \
        | FadeTo(Signal(sin, ω=5 * Hz) | Until(4 * s), 500 * ms) | ToFramerate(22 * Hz)


@case
def stress_mix_middle():  # runtests.jl:881-888 (noise as a fixed array)
    noise = F(rng(27).standard_normal((80, 1)))
    return Signal(noise, 20 * Hz) | After(50 * ms) | Filt(Lowpass, 5 * Hz) | Mix(Signal(sin, ω=7 * Hz)) \
        | Until(3.5 * s) | Filt(Highpass, 2 * Hz) | Append(F(rng(28).random((10, 2)))) \
        | Append(F(rng(29).random((5, 2)))) | ToFramerate(20 * Hz)


@case
def readme_sound1():  # README.md:21 / runtests.jl:896-900; BASELINE config 1
    return Signal(sin, ω=1 * kHz) | Until(5 * s) | Ramp | Normpower | Amplify(-20 * dB) | ToFramerate(4 * kHz)


@case
def readme_scene():  # runtests.jl:913-918
    noise = F(rng(30).standard_normal((44100, 1)))
    x = Signal(sin, ω=1 * kHz) | Until(1 * s) | Ramp | Normpower | Amplify(-20 * dB + 5 * dB)
    y = Signal(noise, 44.1 * kHz) | Until(1 * s) | Filt(Bandstop, 0.5 * kHz, 2 * kHz) | Normpower | Amplify(-20 * dB)
    return Mix(x, y)


@case
def benchmark_overall():  # test/benchmarks.jl:88-97
    N = 10000
    x_ = F(rng(31).random((2 * N, 2)))
    return Mix(Signal(sin, ω=10 * Hz), x_) | ToFramerate(2000 * Hz) | Until(0.5 * N * frames) \
        | After(0.25 * N * frames) | Append(sin) | Until(N * frames) | Filt(Lowpass, 20 * Hz) | Normpower \
        | Amplify(-10 * dB)


@case
def resample_mix_sine_8ch():  # BASELINE metric's "Mix ... Resample": one fused add in the resampler's staging
    x = F(rng(47).standard_normal((6000, 8)))
    return Mix(Signal(x, 44.1 * kHz), Signal(sin, ω=440 * Hz)) | Until(6000 * frames) | ToFramerate(48 * kHz)


@case
def resample_sub_sine():  # the fused one-step routine's v-m form (4-channel tiles)
    x = F(rng(53).standard_normal((5100, 4)))  # (longer than the cut: no `0 - sin` tail, which is not fused)
    return OperateOn("-", Signal(x, 44.1 * kHz), Signal(sin, ω=100 * Hz)) | Until(5000 * frames) | ToFramerate(48 * kHz)


@case
def resample_sine_sub():  # ... and m-v
    x = F(rng(59).standard_normal((5000, 4)))
    return OperateOn("-", Signal(sin, ω=100 * Hz), Signal(x, 44.1 * kHz)) | Until(5000 * frames) | ToFramerate(48 * kHz)


@case
def deep_right_nested():
    """Right-nested maps seven levels deep: the fused interpreter has a 4-deep stack, the
    reference recurses without limit -- the planner materialises sub-expressions (planner.cpp
    legalise)."""
    r = rng(41)
    xs = [Signal(F(r.standard_normal((300, 2))), 100 * Hz) for _ in range(7)]
    e = xs[6]
    for k in (5, 4, 3, 2, 1, 0):
        e = Mix(xs[k], e) if k % 2 else Amplify(xs[k], e)
    return e


@case
def six_generators():
    """Six distinct generators, each scaling its own array: more per-frame slots than the fused
    kernel has (4) -- the planner materialises the surplus."""
    r = rng(43)
    parts = [Amplify(Signal(F(r.standard_normal((250, 2))), 100 * Hz), Signal(sin, 100 * Hz, ω=(3 + 2 * k) * Hz))
             | Until(250 * frames) for k in range(6)]
    return Mix(*parts) | Ramp(20 * frames)


@case
def empty_signal():  # runtests.jl:481-488
    return Signal(sin, 200 * Hz, ω=10 * Hz) | ToChannels(2) | Until(10 * frames) | Until(0 * frames)
