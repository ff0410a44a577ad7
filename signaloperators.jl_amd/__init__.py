"""signaloperators.jl_amd — MI355X-native sink engine behind the SignalOperators.jl
operator API (hot path only)."""
from .units import s, ms, Hz, kHz, frames, kframes, dB, rad, deg, Quantity  # noqa: F401
from .signals import (  # noqa: F401
    ErrorException, inflen, isinf, nframes, nchannels, framerate, sampletype, duration, pipe,
    Signal, Until, After, Window, Pad, Extend, cycle, mirror, lastframe, zero, one,
    Append, Prepend, Mix, Amplify, AddChannel, SelectChannel, OperateOn, Operate,
    RampOn, RampOff, Ramp, FadeTo, sinramp, identity, randn,
    Filt, Normpower, Lowpass, Highpass, Bandpass, Bandstop, Butterworth, Chebyshev1,
    ToFramerate, ToChannels, ToEltype, Format, Uniform,
    ArraySig, NumberSig, FuncSig, CutApply, PaddedSignal, AppendSignals, RampSignal,
    MapSignal, FilteredSignal, NormedSignal, FilterFn, RawFilterFn, RawFirFn, ResamplerFn, digitalfilter, ZeroPoleGain, SecondOrderSections, Biquad,
    PolynomialRatio,
)
from numpy import sin, cos  # noqa: F401  (Signal(sin), Signal(cos))
from .engine import sink, sink_into, stream, BlockStream, Plan, Array, process_sink_params, _eager, filt, filt_into  # noqa: F401
from .arraytypes import SampleBuf, AxisArray, DimensionalArray  # noqa: F401
from .lowering import lower, design_iir, design_resample  # noqa: F401
from .wav import save_signal, load_signal  # noqa: F401

until = _eager(Until)
after = _eager(After)
window = _eager(Window)
append = _eager(Append)
prepend = _eager(Prepend)
mix = _eager(Mix)
amplify = _eager(Amplify)
addchannel = _eager(AddChannel)
selectchannel = _eager(SelectChannel)
operate = _eager(OperateOn)
rampon = _eager(RampOn)
rampoff = _eager(RampOff)
ramp = _eager(Ramp)
fadeto = _eager(FadeTo)
normpower = _eager(Normpower)
toframerate = _eager(ToFramerate)
tochannels = _eager(ToChannels)
toeltype = _eager(ToEltype)
format = _eager(Format)

# The engine library is opened when the package is imported (as a Julia package opens its library in `__init__`, before the
# first `ccall`): dlopen + the registration of its code objects take ~5 ms, which the first sink of a process does not pay.
# Without the library the package's host half still imports -- trees, units, the WAV layer -- and the first call that needs
# the engine raises EngineMissing as before.  SIGOPS_LAZY_LOAD=1 keeps the load for that first call.
import os as _os

if not _os.environ.get("SIGOPS_LAZY_LOAD"):
    try:
        from . import _capi as _capi_mod

        _capi_mod.lib()
    except Exception:  # (missing or unloadable library: reported by the call that needs it)
        pass
