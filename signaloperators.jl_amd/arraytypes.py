"""Array-type signals and sinks: Python stand-ins for the container types the reference adapts
through Requires.jl -- SampledSignals.SampleBuf (src/SampledSignals.jl), AxisArrays.AxisArray
(src/AxisArrays.jl) and DimensionalData.DimensionalArray (src/DimensionalData.jl).  Each carries
its own frame rate, is accepted wherever a signal is, and `sink(x, T)` / `x |> T` builds one
(`initsink(x, ::Type{T})`); `sink(x)` without a type returns the type of the tree's root data
(`refineroot(root(x))`, src/sink.jl:30-50) -- `sink(Mix(buf, 1)) isa SampleBuf` (README.md:61-73,
test/runtests.jl:934,954,962).  The samples themselves go through the HIP engine like any array
leaf; a time axis on the second dimension arrives as strides, not as a copy (AxisArrays.jl:38-39)."""
import numpy as np


class _Container:
    time_axis = 0

    def signal_view(self):
        """[nframes x nch] view of the samples (time first)"""
        d = self.data
        if d.ndim == 2 and self.time_axis == 1:
            return d.T  # PermutedDimsArray(view(x,:,indices),(2,1)): strides, not a copy
        return d

    @property
    def nframes(self):
        return self.signal_view().shape[0]

    @property
    def nchannels(self):
        v = self.signal_view()
        return 1 if v.ndim == 1 else v.shape[1]

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.signal_view(), dtype=dtype)


class SampleBuf(_Container):
    """SampledSignals.SampleBuf(data, samplerate)"""

    def __init__(self, data, samplerate):
        self.data = np.asarray(data)
        self.samplerate = float(samplerate)

    @property
    def framerate(self):
        return self.samplerate

    @classmethod
    def initsink(cls, array, fs):
        return cls(array, fs)


class AxisArray(_Container):
    """AxisArrays.AxisArray with a :time axis given as a uniform range (start, step in seconds) on
    dimension `time_axis` (0 or 1) and a :channel axis on the other"""

    def __init__(self, data, times=None, *, step=None, start=0.0, time_axis=0):
        self.data = np.asarray(data)
        self.time_axis = int(time_axis)
        if times is not None:
            times = np.asarray(times, dtype=np.float64)
            if times.size < 2:
                raise ValueError("a time axis needs at least two points")
            start, step = float(times[0]), float(times[1] - times[0])
        if step is None:
            raise ValueError("AxisArray needs `times` or `step`")
        self.start, self.step = float(start), float(step)

    @property
    def framerate(self):
        return 1.0 / self.step  # inHz(1/step(times)), src/AxisArrays.jl:12-15

    @property
    def times(self):
        return self.start + self.step * np.arange(self.nframes)

    @classmethod
    def initsink(cls, array, fs):  # src/AxisArrays.jl:41-45
        return cls(array, step=1.0 / fs, start=0.0, time_axis=0)


class DimensionalArray(AxisArray):
    """DimensionalData.DimensionalArray with a `Time` dimension (uniform range) and `SigChannel`"""

    @classmethod
    def initsink(cls, array, fs):  # src/DimensionalData.jl:36-41
        return cls(array, step=1.0 / float(fs), start=0.0, time_axis=0)
