"""Host-side mirror of the reference API: lengths, promotion, ToFramerate rewrites, error
behaviour (reference test/runtests.jl, cited per test) and the C-ABI surface.  CPU only."""
import ctypes
import os
import re

import numpy as np
import pytest

import sigops_amd as so
from sigops_amd import (Signal, Until, After, Pad, Extend, Append, Mix, Amplify, AddChannel, SelectChannel,
                        Ramp, FadeTo, Filt, Normpower, ToFramerate, ToChannels, Lowpass, Highpass, Bandpass,
                        Bandstop, Chebyshev1, nframes, nchannels, framerate, duration, sampletype, inflen,
                        isinf, zero, one, ErrorException, s, ms, Hz, kHz, frames, dB, sin)
from sigops_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_currying():  # runtests.jl:31-44
    x = Signal(1, 10 * Hz)
    for f in (Mix(x), Amplify(x), Filt(Lowpass, 200 * Hz, 400 * Hz), Ramp(10 * ms), so.RampOn(10 * ms),
              so.RampOff(10 * ms), FadeTo(x), Amplify(20 * dB), AddChannel(x), SelectChannel(1)):
        assert callable(f)


def test_basic_signal_errors():  # runtests.jl:47-63
    with pytest.raises(ErrorException):
        Signal(np.random.rand(5), 10 * Hz) | Signal(5 * Hz)
    with pytest.raises(ErrorException):
        Signal(np.random.rand(2, 2, 2))


def test_cut_lengths():  # runtests.jl:117-148
    for nch in (1, 2):
        tone = Signal(sin, 44.1 * kHz, ω=100 * Hz) | ToChannels(nch) | Until(5 * s)
        assert not isinf(nframes(tone)) and nframes(tone) == 44100 * 5
        x = np.random.rand(12, nch)
        assert nframes(Signal(x, 6 * Hz) | After(0.5 * s) | Until(1 * s)) == 6
        assert nframes(Signal(x, 6 * Hz) | Until(1 * s) | After(0.5 * s)) == 3
        xs = x | Signal(6 * Hz)
        assert nframes(Append(Until(xs, 1 * s), After(xs, 1 * s))) == 12
        assert nframes(tone | After(2 * s)) == 44100 * 3


def test_sink_of_cut_array_is_a_view():  # runtests.jl:150-169 (DataCut, SURVEY C-11)
    x = np.random.rand(12, 2)
    xv = so.until(x, 5 * frames)
    xv[...] = 0
    assert np.all(x[:5] == 0)
    x = np.random.rand(12, 2)
    xv = so.after(x, 5 * frames)
    xv[...] = 0
    assert np.all(x[5:] == 0)


def test_pad_extend_lengths():  # runtests.jl:216-229
    x, y = np.random.rand(10, 2), np.random.rand(15, 2)
    assert nframes(Extend(x, one)) is inflen
    assert nframes(Mix(Extend(x, one), y)) == 15
    assert nframes(Mix(y, Extend(x, one))) == 15
    assert isinf(nframes(Mix(Pad(x, one), y)))
    assert nframes(Mix(1, np.random.rand(10, 2))) == 10
    assert nframes(Mix(1, Extend(np.random.rand(10, 2), zero))) == 10
    assert nframes(Mix(np.random.rand(10, 2), 1)) == 10
    assert isinf(nframes(Mix(sin, 1, np.random.rand(10, 2))))
    assert isinf(nframes(Mix(1, np.random.rand(10, 2), sin)))


def test_append_rules():  # runtests.jl:243-256
    a = Signal(2, 3) | ToChannels(2) | Until(2 * s) | Append(Signal(3, 3)) | Until(4 * s)
    assert nframes(a) == 12
    with pytest.raises(ErrorException):
        Append(sin, np.arange(1.0, 11.0))
    assert isinf(nframes(Append(np.arange(1.0, 11.0), sin)))


def test_filter_nyquist_and_lengths():  # runtests.jl:334-341
    a = Signal(sin, 100 * Hz, ω=10 * Hz) | Until(5 * s)
    for args in ((Highpass, 75 * Hz), (Lowpass, 75 * Hz), (Bandpass, 75 * Hz, 80 * Hz), (Bandstop, 75 * Hz, 80 * Hz)):
        with pytest.raises(ErrorException):
            Filt(a, *args)
    assert nframes(Filt(a, Highpass, 8 * Hz, method=Chebyshev1(5, 1))) == 500
    assert Filt(a, Highpass, 8 * Hz, blocksize=100).blocksize == 100


def test_resampling_lengths_and_rewrites():  # runtests.jl:407-458
    for nch in (1, 2):
        tone = Signal(sin, 20 * Hz, ω=5 * Hz) | ToChannels(nch) | Until(5 * s)
        up = ToFramerate(tone, 40 * Hz)
        assert framerate(up) == 40 and nframes(up) == 2 * nframes(tone)
        down = ToFramerate(tone, 15 * Hz)
        assert framerate(down) == 15 and nframes(down) == 0.75 * nframes(tone)
        assert ToFramerate(tone, 20 * Hz) is tone
        padded = tone | Pad(one) | Until(7 * s)
        assert nframes(ToFramerate(padded, 40 * Hz)) == 280
        toned = Signal(np.zeros((100, nch)), 20 * Hz)
        twice = ToFramerate(ToFramerate(toned, 15 * Hz), 50 * Hz)
        assert isinstance(twice, so.FilteredSignal) and twice.signal is toned
        a = Signal(sin, 48 * Hz, ω=10 * Hz) | ToChannels(nch) | Until(3 * s)
        high = Mix(a, a) | Filt(Highpass, 8 * Hz, method=Chebyshev1(5, 1))
        assert nframes(ToFramerate(high, 24 * Hz)) == 72
    # ratio 4 stays a Float64 (arbitrary kernel): max(num,den) > 3 (SURVEY C-9)
    assert isinstance(ToFramerate(Signal(np.zeros((10, 1)), 10 * Hz), 40 * Hz).fn.ratio, float)
    assert ToFramerate(Signal(np.zeros((10, 1)), 10 * Hz), 20 * Hz).fn.ratio == (2, 1)


def test_automatic_reformatting():  # runtests.jl:461-470
    a = Signal(sin, 200 * Hz, ω=10 * Hz) | ToChannels(2) | Until(5 * s)
    b = Signal(sin, 100 * Hz, ω=5 * Hz) | Until(3 * s)
    c = Mix(a, b)
    assert nchannels(c) == 2 and framerate(c) == 200 and nframes(c) == 1000
    assert nframes(Mix(a, b, 1)) == 1000


def test_empty_and_infinite():  # runtests.jl:481-488, 569-573
    tone = Signal(sin, 200 * Hz, ω=10 * Hz) | Until(10 * frames) | Until(0 * frames)
    assert nframes(tone) == 0
    with pytest.raises(ErrorException):
        so.sink(Signal(sin, 200 * Hz) | Normpower | Until(1 * s))
    with pytest.raises(ErrorException):
        so.sink(Signal(sin, 200 * Hz))


def test_frame_units():  # runtests.jl:602-614
    x, y = Signal(np.random.rand(100, 2), 10 * Hz), Signal(np.random.rand(50, 2), 10 * Hz)
    assert nframes(x | Until(30 * frames)) == 30
    assert nframes(x | After(30 * frames)) == 70
    assert nframes(x | Append(y) | After(20 * frames)) == 130
    assert nframes(x | Pad(zero) | Until(150 * frames)) == 150
    assert nframes(x | Ramp(10 * frames)) == 100
    assert nframes(x | FadeTo(y, 10 * frames)) == 140


def test_unknown_frame_rates():  # runtests.jl:758-790
    x, y = np.random.rand(100, 2), np.random.rand(50, 2)
    assert framerate(x) is None and duration(x) is None
    assert framerate(x | ToFramerate(10 * Hz)) == 10
    assert nframes(x | Until(3 * s) | ToFramerate(10 * Hz)) == 30
    assert nframes(x | After(3 * s) | ToFramerate(10 * Hz)) == 70
    assert nframes(x | Append(y) | Until(13 * s) | ToFramerate(10 * Hz)) == 130
    assert nframes(x | Pad(zero) | Until(15 * s) | ToFramerate(10 * Hz)) == 150
    assert nframes(x | Filt(Lowpass, 3 * Hz) | ToFramerate(10 * Hz)) == 100
    assert nframes(x | Normpower | Amplify(-10 * dB) | ToFramerate(10 * Hz)) == 100
    assert nframes(x | AddChannel(y) | ToFramerate(10 * Hz)) == 100
    with pytest.raises(ErrorException):
        x | FadeTo(y) | ToFramerate(10 * Hz)


def test_sampletype_promotion():  # runtests.jl:707-729
    x = Signal(np.random.rand(100, 2).astype(np.float32), 10 * Hz)
    y = Signal(np.random.rand(50, 2).astype(np.float32), 10 * Hz)
    for t in (x | Until(5 * s), x | Append(y), x | Pad(zero) | Until(15 * s), x | Filt(Lowpass, 3 * Hz),
              x | Normpower | Amplify(np.float32(-10.0) * dB), x | Mix(y), x | AddChannel(y), x | SelectChannel(1),
              x | Ramp, x | FadeTo(y)):
        assert sampletype(t) == np.float32
    assert sampletype(x | Mix(1)) == np.float32  # Int literals promote to the signal's type
    assert sampletype(x | Mix(1.5)) == np.float64


def test_stress_durations():  # runtests.jl:832-844, 869-872, 881-888
    x = (Signal(sin, ω=10 * Hz, fs=20 * Hz) | Until(4 * s) | ToFramerate(30 * Hz) | Filt(Lowpass, 10 * Hz)
         | FadeTo(Signal(sin, ω=5 * Hz) | Until(4 * s), 500 * ms) | ToFramerate(22 * Hz))
    assert framerate(x) == 22 and duration(x) == 7.5
    x = (Signal(sin, ω=5 * Hz) | After(2 * s) | Until(20 * s) | After(2 * s) | Until(15 * s) | After(2 * s)
         | After(2 * s) | Until(5 * s) | Until(2 * s) | ToFramerate(12 * Hz))
    assert duration(x) == 2


def test_cabi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "sigops.h")).read()
    declared = sorted(set(re.findall(r"\b(so_[a-z_]+)\s*\(", hdr)) - {"so_node", "so_out_desc", "so_stats"})
    lib = ctypes.CDLL(_capi.LIB_PATH)
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    assert set(declared) == set(_capi.EXPORTS)
    assert _capi.lib().so_abi_version() == 1


def test_cabi_exports_nothing_else():
    """-fvisibility=hidden + csrc/exports.map: the dynamic symbol table defines the header's so_* entry points and
    nothing of the engine's C++ (no kernel launchers, no STL instantiations another library could bind to)"""
    import shutil
    import subprocess

    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm, "-D", "--defined-only", _capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    syms = {ln.split()[-1].split("@")[0] for ln in out.splitlines() if ln.strip()}
    syms -= {"_init", "_fini", "_edata", "_end", "__bss_start"}
    assert syms == set(_capi.EXPORTS), sorted(syms ^ set(_capi.EXPORTS))[:10]


def test_struct_layout_matches_header():
    assert ctypes.sizeof(_capi.so_node_t) == 4 * 4 + 8 + 8 + 8 + 4 * 4 + 2 * 8 + 4 * 8 + 2 * 8 + 2 * 8
    assert ctypes.sizeof(_capi.so_out_desc_t) == 40


def test_product_fails_loudly_without_a_gpu():
    """no CPU fallback on the product path (only meaningful where no device is visible)"""
    if _capi.lib().so_device_count() > 0:
        pytest.skip("a HIP device is visible")
    with pytest.raises(ErrorException, match="no HIP device"):
        so.sink(Mix(Signal(np.ones((4, 1)), 10 * Hz), 1))


def test_stream_of_plain_data_yields_consecutive_slices():
    """so.stream: block k = sink(x |> After(k*blocksize) |> Until(blocksize)); for raw arrays that is the
    reference's aliasing time slice (src/sink.jl:65-69), no device involved"""
    import sigops_amd as so

    data = np.asfortranarray(np.arange(20.0).reshape(10, 2))
    blocks = list(so.stream(so.Signal(data, 10 * so.Hz), 4, so.Array))
    assert [b.shape[0] for b in blocks] == [4, 4, 2]
    assert np.array_equal(np.concatenate(blocks, axis=0), data)
    with pytest.raises(so.ErrorException):
        next(so.stream(so.Signal(data, 10 * so.Hz), 0))


def test_opaque_closures_run_once_per_frame_on_the_host():
    """the host evaluation of a closure the engine has no kernel for used to call it once with the whole
    array and keep any result of the right shape -- a stateful closure, or one that is not elementwise (x / max|x|),
    silently gave other values than the reference's per-frame calls (src/functions.jl:53-56, src/mapsignal.jl:249-272)"""
    from sigops_amd.lowering import _apply_host

    x = np.array([1.0, -4.0, 2.0])
    calls = []

    def counting(v):
        calls.append(v)
        return v / abs(v)  # elementwise only when called per element; over the array `abs(v).max()`-style code would not be

    assert np.array_equal(_apply_host(counting, (x,)), np.array([1.0, -1.0, 1.0])) and len(calls) == 3
    assert np.array_equal(_apply_host(lambda v: v / np.abs(v).max(), (x,)), np.ones(3) * np.sign(x))  # per frame: v / |v|
    assert np.array_equal(_apply_host(np.maximum, (x, np.zeros(3))), np.maximum(x, 0))  # a ufunc: one call
    vec = lambda v: 2 * v  # noqa: E731
    vec.vectorized = True
    assert np.array_equal(_apply_host(vec, (x,)), 2 * x)
