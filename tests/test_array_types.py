"""f2: array-type signals and sinks (reference src/SampledSignals.jl,
src/AxisArrays.jl, src/DimensionalData.jl; sink type resolution src/sink.jl:28-50), the eager
lower-case forms (e.g. src/mapsignal.jl:321 `mix(xs...) = sink(Mix(xs...))`) and the DSP.filt
overloads (src/filters.jl:68-87).  Host-logic part on CPU; the `gpu` part sinks through the engine."""
import numpy as np
import pytest
from scipy import signal as sps

import sigops_amd as so
from sigops_amd import SampleBuf, AxisArray, DimensionalArray, Signal, Mix, Ramp, Hz, s
from sigops_amd.engine import _refineroot
from cases import F, rng
from oracle_bridge import oracle_sink, relerr


def test_container_traits():
    x = rng(1).random((10, 2))
    b = SampleBuf(x, 10)
    sig = Signal(b)
    assert so.framerate(sig) == 10 and so.nframes(sig) == 10 and so.nchannels(sig) == 2
    with pytest.raises(so.ErrorException):  # inconsistent frame rate, src/SampledSignals.jl:3-9
        Signal(b, 20 * Hz)
    # runtests.jl:539-543: an AxisArray with time on the SECOND dimension
    a = AxisArray(rng(2).random((2, 10)), times=np.arange(10) * 0.1, time_axis=1)
    sa = Signal(a)
    assert so.nframes(sa) == 10 and so.nchannels(sa) == 2 and abs(so.framerate(sa) - 10.0) < 1e-9
    assert np.shares_memory(sa.data, a.data)  # strides, not a copy (AxisArrays.jl:38-39)


def test_sink_type_follows_the_root_data():
    """refineroot(root(x)) with mergeroot's priorities (src/sink.jl:30-50; runtests.jl:934,954,962)"""
    x = rng(3).random((10, 2))
    assert _refineroot(so.process_sink_params(Mix(SampleBuf(x, 10), 1))) is SampleBuf
    assert _refineroot(so.process_sink_params(Mix(1, DimensionalArray(x, step=0.1)))) is DimensionalArray
    assert _refineroot(so.process_sink_params(Mix(Signal(x, 10 * Hz), 1))) is tuple
    assert _refineroot(so.process_sink_params(Mix(x, 1))) is so.Array
    # a signal-typed container outranks a plain array; of two containers the first wins
    assert _refineroot(so.process_sink_params(Mix(x, AxisArray(x, step=0.1)))) is AxisArray
    assert _refineroot(so.process_sink_params(Mix(SampleBuf(x, 10), AxisArray(x, step=0.1)))) is SampleBuf


def test_oracle_sees_the_same_samples():
    a = AxisArray(rng(4).random((3, 50)), step=0.01, time_axis=1)
    want = oracle_sink(Signal(np.ascontiguousarray(a.data.T), 100 * Hz) | Ramp(50 * so.ms))
    assert np.array_equal(oracle_sink(Signal(a) | Ramp(50 * so.ms)), want)


@pytest.mark.gpu
def test_gpu_array_type_sinks():
    x = F(rng(5).random((40, 2)))
    buf = SampleBuf(x, 20)
    y = so.sink(Mix(buf, 1))
    assert isinstance(y, SampleBuf) and y.samplerate == 20.0 and np.array_equal(y.data, x + 1)
    d = Signal(x, 20 * Hz) | so.sink(DimensionalArray)  # runtests.jl:680,697
    assert isinstance(d, DimensionalArray) and abs(d.framerate - 20.0) < 1e-12 and np.array_equal(d.data, x)
    # runtests.jl:474-477: Signal(x) |> Ramp |> AxisArray
    ax = AxisArray(np.ones(20), times=np.linspace(0, 2, 20))
    proc = so.sink(Signal(ax) | Ramp, AxisArray)
    assert isinstance(proc, AxisArray) and proc.nframes == 20
    assert np.array_equal(proc.data, oracle_sink(Signal(ax) | Ramp))
    # time on dimension 2 arrives as a strided leaf
    a2 = AxisArray(rng(6).random((2, 30)), step=0.05, time_axis=1)
    got = so.sink(Mix(a2, 0.5))
    assert isinstance(got, AxisArray) and np.array_equal(got.data, a2.data.T + 0.5)


@pytest.mark.gpu
def test_gpu_eager_forms():
    """mix / amplify / ramp / until / toframerate ... = sink(Op(...)) (src/mapsignal.jl:321,346 etc.)"""
    x = F(rng(7).standard_normal((3000, 2)))
    y = F(rng(8).standard_normal((3000, 2)))
    sx, sy = Signal(x, 8000 * Hz), Signal(y, 8000 * Hz)
    assert np.array_equal(so.mix(sx, sy)[0], x + y)
    assert np.array_equal(so.amplify(sx, sy)[0], x * y)
    assert np.array_equal(so.until(sx, 100 * so.frames)[0], x[:100])
    assert np.array_equal(so.append(sx, sy)[0], np.concatenate([x, y]))
    r = so.ramp(sx, 10 * so.ms)[0]
    assert relerr(r, oracle_sink(Ramp(sx, 10 * so.ms))) < 1e-12
    t = so.toframerate(sx, 12000 * Hz)
    assert t[1] == 12000.0 and relerr(t[0], oracle_sink(so.ToFramerate(sx, 12000 * Hz))) < 1e-11
    assert np.array_equal(so.mix(SampleBuf(x, 8000), 1).data, x + 1)  # README.md:61-73


@pytest.mark.gpu
def test_gpu_filt_overloads():
    x = F(rng(9).standard_normal((5000, 2)))
    sig = Signal(x, 1000 * Hz) | so.Amplify(2.0)
    h = sps.firwin(21, 0.3)
    y = so.filt(h, 1.0, sig)
    assert isinstance(y, tuple) and y[1] == 1000.0
    assert relerr(y[0], sps.lfilter(h, [1.0], 2 * x, axis=0)) < 1e-12
    b, a = sps.butter(2, 0.2)
    assert relerr(so.filt(b, a, sig)[0], sps.lfilter(b, a, 2 * x, axis=0)) < 1e-11
    b4, a4 = sps.butter(4, 0.2)  # order 4 + an initial state: sink on the engine, array filt! after
    zi = sps.lfilter_zi(b4, a4)
    want, _ = sps.lfilter(b4, a4, 2 * x, axis=0, zi=np.repeat(zi[:, None], 2, axis=1))
    assert relerr(so.filt(b4, a4, sig, zi)[0], want) < 1e-11
    out = np.empty((5000, 2))
    so.filt_into(out, b, a, sig)
    assert relerr(out, sps.lfilter(b, a, 2 * x, axis=0)) < 1e-11
    assert isinstance(so.filt(h, 1.0, Signal(SampleBuf(x, 1000))), SampleBuf)
