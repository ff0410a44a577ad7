"""Host-side mirror of the SignalOperators.jl operator API (layers L3-L5 of
SURVEY.md §1): lazy tree construction, traits / length algebra, promotion
(`Uniform`, `Format`) and the `ToFramerate` rewrite rules.  Nothing here computes
samples: `sink` hands the tree to the HIP engine through the C-ABI
(`lowering.py` -> include/sigops.h).

The reference host language is Julia; Julia is not present in this image
(SURVEY.md probe table), so the host side above the C-ABI is written in Python with
the same names, argument meaning and error behaviour as the reference so that the
parity tests read like test/runtests.jl.  Piping `x |> Until(5s)` is spelled
`x | Until(5*s)` (or `pipe(x, Until(5*s), ...)`).

Every class/function cites the reference definition it mirrors.
"""
import math
import operator

import numpy as np

from . import units as U
from .units import Quantity


class ErrorException(Exception):
    """Julia's ErrorException: what the reference's `error(msg)` throws."""


def error(msg):
    raise ErrorException(msg)


# --------------------------------------------------------------------------
# lengths: reference src/inflen.jl, src/signal.jl:28-37, src/numbers.jl:5-9
class _InfLen:
    def __repr__(self):
        return "inflen"


inflen = _InfLen()


class Extended:  # src/signal.jl:31-36
    def __init__(self, n):
        self.len = n

    def __repr__(self):
        return f"Extended({self.len})"


class _NumberExtended:  # src/numbers.jl:5-7
    def __repr__(self):
        return "numextend"


numextend = _NumberExtended()


def isknowninf(n):  # src/inflen.jl:21-22
    return n is inflen or isinstance(n, (Extended, _NumberExtended))


def isinf(n):
    return isknowninf(n)


def cleanextend(n):  # src/signal.jl:29,35 src/numbers.jl:9
    return inflen if isknowninf(n) else n


F32 = np.dtype("float32")
F64 = np.dtype("float64")
I64 = np.dtype("int64")


def promote_type(a, b):
    """Julia promote_type restricted to Float32 / Float64 / Int64"""
    if a == F64 or b == F64:
        return F64
    if a == F32 or b == F32:
        return F32
    return I64


def float_type(t):
    return F64 if t == I64 else t


def _is_torch(x):
    return type(x).__module__.split(".")[0] == "torch"


# --------------------------------------------------------------------------
class AbstractSignal:
    """reference src/signal.jl:18-53 (IsSignal{T,Fs,L} traits as attributes)"""

    __array_ufunc__ = None
    fs = None  # framerate(x); None == missing
    nch = 1  # nchannels(x)
    dtype = F64  # sampletype(x)
    evaltrait = "computed"  # EvalTrait(x) src/signal.jl:154-175
    children = ()

    def nframes_helper(self):
        raise NotImplementedError

    def __init_subclass__(cls, **kw):
        # signals are immutable once built and their length is asked for again and again (every node of
        # every lowering, every block of so.stream): remember it per object
        super().__init_subclass__(**kw)
        f = cls.__dict__.get("nframes_helper")
        if f is not None and not getattr(f, "_memo", False):
            def helper(self, _f=f):
                d = self.__dict__
                if "_nframes_memo" not in d:
                    d["_nframes_memo"] = _f(self)
                return d["_nframes_memo"]
            helper._memo = True
            helper.__doc__ = f.__doc__
            cls.nframes_helper = helper

    def __or__(self, fn):  # x |> f
        if callable(fn):
            return fn(self)
        return NotImplemented


def nframes(x):  # src/signal.jl:27
    x = _assignal(x)
    return cleanextend(x.nframes_helper())


def nchannels(x):
    return _assignal(x).nch


def framerate(x):
    if isinstance(x, tuple) and len(x) == 2:
        return float(x[1])
    return _assignal(x).fs


def sampletype(x):
    return _assignal(x).dtype


def duration(x):
    x = _assignal(x)
    return x.duration()


class Curried:
    """`Until(5s)` etc. return functions in the reference; this wrapper also makes
    `array | Until(...)` work (numpy defers to __ror__)."""

    __array_ufunc__ = None

    def __init__(self, fn):
        self.fn = fn

    def __call__(self, x):
        return self.fn(x)

    def __ror__(self, x):
        return self.fn(x)


def pipe(x, *fns):
    for f in fns:
        x = f(x)
    return x


# --------------------------------------------------------------------------
# leaves
class ArraySig(AbstractSignal):
    """arrays and (array,fs) tuples as signals, reference src/arrays.jl:35-132"""

    evaltrait = "data"

    def __init__(self, data, fs=None):
        if _is_torch(data):
            shape = tuple(data.shape)
            dt = {"torch.float32": F32, "torch.float64": F64}.get(str(data.dtype))
            if dt is None:
                error("device arrays must be float32 or float64")
        else:
            data = np.asarray(data)
            shape = data.shape
            dt = data.dtype
            if dt.kind in "iu" or dt.kind == "b":
                dt = I64
        if len(shape) not in (1, 2):
            error("Array must have 1 or 2 dimensions to be treated as a signal.")
        self.data = data
        self.fs = None if fs is None else float(fs)
        self.nch = 1 if len(shape) == 1 else int(shape[1])
        self.n = int(shape[0])
        self.dtype = np.dtype(dt)
        self.container = None  # SampleBuf / AxisArray / DimensionalArray class the data came in (root type)
        self.virtual = None  # engine.BlockStream: (address frame 0 would have, frame stride, channel stride, first resident frame)

    def nframes_helper(self):
        return self.n

    def duration(self):
        return None if self.fs is None else self.n / self.fs


class NumberSig(AbstractSignal):
    """reference src/numbers.jl:1-64"""

    def __init__(self, val, fs=None, dB=False, dtype=None):
        if dtype is None:
            if isinstance(val, (bool, int, np.integer)):
                dtype = I64
            elif isinstance(val, np.floating) and val.dtype == F32:
                dtype = F32
            else:
                dtype = F64
        self.val = val
        self.fs = fs
        self.dB = dB
        self.dtype = np.dtype(dtype)
        self.nch = 1

    def nframes_helper(self):
        return numextend

    def duration(self):
        return inflen


SIN, COS, IDENTITY, RANDN = "sin", "cos", "identity", "randn"
OPAQUE = "opaque"  # any other callable: evaluated on the host at sink time and handed to the engine as an array leaf


def _fn_code(fn):
    import builtins  # noqa: F401

    if fn in (SIN, COS, IDENTITY, RANDN):
        return fn
    if fn is np.sin or fn is math.sin:
        return SIN
    if fn is np.cos or fn is math.cos:
        return COS
    if fn is randn or fn is np.random.randn:
        return RANDN
    if getattr(fn, "__name__", "") == "identity":
        return IDENTITY
    return None


def identity(x):
    return x


def randn(*a):
    """marker for Signal(randn) (reference src/functions.jl:98-114)"""
    return np.random.randn(*a)


class FuncSig(AbstractSignal):
    """SignalFunction, reference src/functions.jl:11-60,88-96"""

    def __init__(self, fn, fs=None, omega=None, phi=0.0, rng=None, pyfn=None):
        self.fn = fn  # code string
        self.pyfn = pyfn  # OPAQUE: the host callable itself
        self.fs = fs
        self.omega = omega
        self.phi = float(phi)
        self.rng = rng
        self.nch = 1
        self.dtype = F64

    def nframes_helper(self):
        return inflen

    def duration(self):
        return inflen


# --------------------------------------------------------------------------
class WrappedSignal(AbstractSignal):
    """reference src/wrapping.jl:3-19"""

    def __init__(self, child):
        self.signal = child
        self.children = (child,)
        self.fs = child.fs
        self.nch = child.nch
        self.dtype = child.dtype

    @property
    def evaltrait(self):
        return self.signal.evaltrait

    def nframes_helper(self):
        return self.signal.nframes_helper()

    def duration(self):
        return self.signal.duration()


def _tolen_arith(n):
    return n


class CutApply(WrappedSignal):
    """Until / After, reference src/cutting.jl:6-32,130-138"""

    def __init__(self, child, time, kind):
        super().__init__(child)
        self.time = time
        self.kind = kind  # "until" | "after"

    @property
    def evaltrait(self):
        if self.kind == "after":
            return "data"  # src/cutting.jl:138
        return self.signal.evaltrait

    def resolvelen(self):  # src/cutting.jl:32
        return U.inframes_int(U.maybeseconds(self.time), self.fs)

    def nframes_helper(self):
        n = self.signal.nframes_helper()
        L = self.resolvelen()
        if L is None or n is None:
            return None
        if self.kind == "until":  # :130
            L = max(0, L)
            return L if isknowninf(n) else min(n, L)
        if isknowninf(n):  # :134 (inflen - L == inflen)
            return n
        return min(max(n - L, 0), n)

    def duration(self):
        d = self.signal.duration()
        t = U.inseconds(U.maybeseconds(self.time), self.fs)
        if d is None or t is None:
            return None
        if self.kind == "until":
            t = max(0, t)
            return t if isknowninf(d) else min(d, t)
        if isknowninf(d):
            return d
        return min(max(d - t, 0), d)


class PaddedSignal(WrappedSignal):
    """Pad / Extend, reference src/padding.jl:3-19"""

    def __init__(self, child, pad, extending=False):
        super().__init__(child)
        self.pad = pad
        self.extending = extending

    def nframes_helper(self):
        if self.extending:
            return Extended(nframes(self.signal))
        return inflen

    def duration(self):
        return inflen


class AppendSignals(WrappedSignal):
    """reference src/appending.jl:3-17"""

    def __init__(self, signals, length, dtype):
        super().__init__(signals[0])
        self.signals = tuple(signals)
        self.children = self.signals
        self.len = length
        self.dtype = dtype

    def nframes_helper(self):
        return self.len

    def duration(self):
        ds = [c.duration() for c in self.signals]
        if any(d is None for d in ds):
            return None
        if any(isknowninf(d) for d in ds):
            return inflen
        return sum(ds)


class RampSignal(WrappedSignal):
    """reference src/ramps.jl:6-26 — a GAIN signal shaped like its child"""

    def __init__(self, direction, child, time, fn):
        super().__init__(child)
        self.direction = direction  # "on" | "off"
        self.time = time
        self.fn = fn  # "sinramp" | "identity"
        self.dtype = float_type(child.dtype)

    def resolvelen(self):  # :26
        n = U.inframes_int(U.maybeseconds(self.time), self.fs)
        return None if n is None else max(1, n)


class NormedSignal(WrappedSignal):
    """reference src/filters.jl:266-275"""

    def __init__(self, child):
        super().__init__(child)
        self.dtype = float_type(child.dtype)


class FilterFn:  # src/filters.jl:5-12
    def __init__(self, design, method, args):
        self.design = design  # "lowpass" | "highpass" | "bandpass" | "bandstop"
        self.method = method  # ("butterworth", order) | ("chebyshev1", order, ripple)
        self.args = tuple(args)  # Hz


class RawFilterFn:  # src/filters.jl:89-92
    def __init__(self, sos, gain=1.0):
        self.sos = np.ascontiguousarray(np.asarray(sos, dtype=np.float64).reshape(-1, 6))
        self.gain = float(gain)


class RawFirFn:
    """Filt(x, h) with FIR coefficients h: DF2TFilter(PolynomialRatio(h, [1])), y[n] = sum_k h[k] x[n-k]"""

    def __init__(self, h):
        self.h = np.ascontiguousarray(np.asarray(h, dtype=np.float64).ravel())


class ResamplerFn:  # src/util.jl:12-15
    def __init__(self, ratio, fs):
        self.ratio = ratio  # (num, den) tuple or float
        self.fs = fs


class FilteredSignal(WrappedSignal):
    """reference src/filters.jl:98-114,159-167"""

    evaltrait = "computed"

    def __init__(self, child, fn, blocksize, newfs):
        super().__init__(child)
        self.fn = fn
        self.blocksize = int(blocksize)
        self.fs = newfs
        self.dtype = float_type(child.dtype)

    @property
    def evaltrait(self):  # noqa: F811
        return "computed"

    def nframes_helper(self):  # :159-167
        cfs = self.signal.fs
        if cfs is None:
            return None
        n = self.signal.nframes_helper()
        if self.fs == cfs:
            return n
        if isknowninf(n) or n is None:
            return n
        return int(math.ceil(n * self.fs / cfs))

    def duration(self):
        return self.signal.duration()


# map functions (src/mapsignal.jl:308,333,360,389; src/reformatting.jl:148-184)
ADD, MUL, SUB, DIV = "add", "mul", "sub", "div"
TUPLECAT, GETCHAN, AS1CHANNEL, ASNCHANNELS, TOELTYPE, REVERSECH = (
    "tuplecat", "getchan", "as1channel", "asnchannels", "toeltype", "reversech")


def _map_code(fn):
    table = {operator.add: ADD, operator.mul: MUL, operator.sub: SUB,
             operator.truediv: DIV, operator.neg: SUB, "+": ADD, "*": MUL, "-": SUB,
             "/": DIV, reversed: REVERSECH, "reverse": REVERSECH}
    if isinstance(fn, tuple):
        return fn
    if isinstance(fn, str) and fn in (ADD, MUL, SUB, DIV, TUPLECAT, AS1CHANNEL, REVERSECH):
        return fn
    if isinstance(fn, OpaqueFn):
        return fn
    try:
        if fn in table:
            return table[fn]
    except TypeError:
        pass
    if callable(fn):
        # an arbitrary closure (reference src/mapsignal.jl:131-145): the map is evaluated on the host at
        # sink time -- its operands by the engine, the closure by NumPy -- and enters the tree as an
        # array leaf (lowering.py)
        return OpaqueFn(fn)
    error(f"OperateOn: {fn!r} is not a function")


class OpaqueFn:
    """a host callable used as a map function"""

    def __init__(self, fn):
        self.fn = fn

    def __repr__(self):
        return f"OpaqueFn({self.fn!r})"


def default_pad(fn):  # src/mapsignal.jl:274-276
    if isinstance(fn, OpaqueFn):
        return zero
    return one if fn in (MUL, DIV) else zero


class MapSignal(AbstractSignal):
    """reference src/mapsignal.jl:8-30"""

    def __init__(self, fn, signals, fs, padding, bychannel, extra=0, blocksize=4096):
        self.fn = fn
        self.extra = extra
        self.signals = tuple(signals)
        self.children = self.signals
        self.fs = fs
        self.padding = padding
        self.bychannel = bychannel
        self.blocksize = blocksize
        # test-value type inference, src/mapsignal.jl:139-143,190
        dts = [c.dtype for c in self.signals]
        t = dts[0]
        for d in dts[1:]:
            t = promote_type(t, d)
        if isinstance(fn, OpaqueFn):
            # test-value type inference (src/mapsignal.jl:139-143): the closure applied to ones
            try:
                if bychannel:
                    v = fn.fn(*[np.ones((), dtype=c.dtype)[()] for c in self.signals])
                else:
                    v = fn.fn(*[tuple(np.ones(c.nch, dtype=c.dtype)) for c in self.signals])
                    v = np.asarray(v).reshape(-1)
                    self._opaque_nch = int(v.size)
                t = np.asarray(v).dtype
                t = F32 if t == F32 else (I64 if np.issubdtype(t, np.integer) else F64)
            except Exception as e:  # noqa: BLE001
                error(f"OperateOn: could not apply {fn.fn!r} to test values: {e}")
            self.dtype = t
            self.nch = self.signals[0].nch if bychannel else self._opaque_nch
            return
        if fn == DIV and t == I64:
            t = F64
        if fn == TOELTYPE:
            t = np.dtype(extra)
        self.dtype = t
        if bychannel:
            self.nch = self.signals[0].nch
        elif fn == TUPLECAT:
            self.nch = sum(c.nch for c in self.signals)
        elif fn in (GETCHAN, AS1CHANNEL):
            self.nch = 1
        elif fn == ASNCHANNELS:
            self.nch = int(extra)
        else:
            self.nch = self.signals[0].nch

    def nframes_helper(self):  # src/mapsignal.jl:147-154
        def tolen(x):
            if isinstance(x, Extended):
                return x.len
            if x is numextend:
                return 0
            return x

        lens = [c.nframes_helper() for c in self.signals]
        acc = lens[0]
        for y in lens[1:]:
            if acc is numextend and y is numextend:
                continue
            a, b = tolen(acc), tolen(y)
            if a is None or b is None:
                acc = None
            elif a is inflen or b is inflen:
                acc = inflen
            else:
                acc = max(a, b)
        return acc

    def duration(self):
        n = nframes(self)
        if n is None or self.fs is None:
            return None
        return inflen if isknowninf(n) else n / self.fs


# pad markers (src/padding.jl:110-148)
def zero(x=None):
    return 0


def one(x=None):
    return 1


def lastframe(x):
    error("Must be passed as argument to `Pad`.")


def cycle(x, i, j):
    return x[(i - 1) % x.shape[0], j]


def mirror(x, i, j):
    n = x.shape[0]
    count, rem = divmod(i - 1, n)
    return x[rem if count % 2 == 0 else n - rem - 1, j]


def sinramp(x):  # src/ramps.jl:4
    return math.sin(math.pi * 0.5 * x)


# --------------------------------------------------------------------------
# Signal() coercion: reference src/signal.jl:96-149, src/arrays.jl:35-46,
# src/numbers.jl:47-49, src/functions.jl:88-96,109-110
def _assignal(x):
    if isinstance(x, AbstractSignal):
        return x
    return Signal(x)


def _isconsistent(fs, _fs):
    return fs is None or U.inHz(_fs) == U.inHz(fs)


def Signal(x=None, fs=None, *, ω=None, frequency=None, ϕ=0, phase=None, omega=None,
           phi=None, rng=None):
    # Signal(fs::Quantity) / Signal(;kwds...) curried forms, src/signal.jl:98-99
    if isinstance(x, Quantity) and x.unit in ("Hz", "kHz") and fs is None:
        rate = x
        return Curried(lambda y: Signal(y, rate))
    if x is None:
        kw = dict(ω=ω, frequency=frequency, ϕ=ϕ, phase=phase, omega=omega, phi=phi, rng=rng)
        return Curried(lambda y: Signal(y, fs, **kw))
    fs = U.inHz(fs)
    if omega is not None:
        ω = omega
    if phi is not None:
        ϕ = phi
    if isinstance(x, AbstractSignal):  # src/signal.jl:139-147
        if x.fs is None:
            return ToFramerate(x, fs)
        if not _isconsistent(fs, x.fs):
            error(f"Signal expected to have frame rate of {fs} Hz.")
        return x
    from .arraytypes import _Container

    if isinstance(x, _Container):  # src/SampledSignals.jl:3-9, src/AxisArrays.jl:30-36, src/DimensionalData.jl:6-14
        if not _isconsistent(fs, x.framerate):
            error(f"Signal expected to have frame rate of {fs} Hz.")
        a = ArraySig(x.signal_view(), x.framerate)
        a.container = type(x)
        return a
    if isinstance(x, tuple) and len(x) == 2 and not callable(x[0]):
        if not _isconsistent(fs, x[1]):
            error(f"Signal expected to have frame rate of {fs} Hz.")
        return ArraySig(x[0], U.inHz(x[1]))
    if isinstance(x, Quantity):
        if x.unit == "dB":  # src/numbers.jl:48-49
            return NumberSig(U.gain_ratio(x), fs, dB=True,
                             dtype=F32 if isinstance(x.value, np.floating) and x.value.dtype == F32 else F64)
        error(f"Don't know how create a signal from {x}.")
    if isinstance(x, (bool, int, float, np.integer, np.floating)):
        return NumberSig(x, fs)
    if isinstance(x, (list, range)) or isinstance(x, np.ndarray) or _is_torch(x):
        return ArraySig(x, fs)
    if callable(x) or (isinstance(x, str) and x in (SIN, COS, IDENTITY, RANDN)):
        code = _fn_code(x)
        pyfn = None
        if code is None:
            # what the engine cannot lower is materialised on the host and passed
            # as an array leaf (here: the closure is evaluated with NumPy at sink time, lowering.py)
            code, pyfn = OPAQUE, x
        if code == RANDN:  # src/functions.jl:109-110
            return FuncSig(RANDN, fs, None, 0.0, rng=rng)
        if frequency is not None and ω is None:
            pass  # keyword `frequency` is accepted but ignored (src/functions.jl:90,92)
        w = U.inHz(ω)
        ph = phase if phase is not None else ϕ
        if w is None:
            if isinstance(ph, Quantity) and ph.unit not in ("s", "ms"):
                error("phase in radians needs a frequency (reference src/functions.jl:93)")
            p = U.inseconds(ph) if isinstance(ph, Quantity) else float(ph)
        else:
            p = U.inradians(ph, w) / (2 * math.pi)
        return FuncSig(code, fs, w, p, pyfn=pyfn)
    error(f"Don't know how create a signal from {x!r}.")


# --------------------------------------------------------------------------
# cutting: src/cutting.jl:46-57,77-78,105-106
def _curry2(ctor):
    def f(*args, **kw):
        if len(args) == 1:
            a = args[0]
            return Curried(lambda x: ctor(x, a, **kw))
        return ctor(*args, **kw)

    f.__name__ = ctor.__name__
    return f


def _Until(x, time):
    return CutApply(_assignal(x), time, "until")


def _After(x, time):
    return CutApply(_assignal(x), time, "after")


Until = _curry2(_Until)
After = _curry2(_After)


def Window(x=None, *, at=None, width=None, from_=None, to=None):
    if x is None:
        return Curried(lambda y: Window(y, at=at, width=width, from_=from_, to=to))
    if (at is None) != (width is None) or (from_ is None) != (to is None) or \
            (at is None) == (from_ is None):
        error("`Window` must either use the two keywords `at` and `width` OR"
              "the two keywords `from` and `to`.")
    # the reference's `at/width` form is broken by tuple precedence (SURVEY C-10)
    if from_ is None:
        error("Window(at=,width=) is broken in the reference (src/cutting.jl:55)")
    return _Until(_After(x, from_), to - from_)


# padding: src/padding.jl:76-101
def _Pad(x, p):
    x = _assignal(x)
    return x if isknowninf(nframes(x)) else PaddedSignal(x, p)


def _Extend(x, p):
    x = _assignal(x)
    return x if isknowninf(nframes(x)) else PaddedSignal(x, p, True)


Pad = _curry2(_Pad)
Extend = _curry2(_Extend)


# --------------------------------------------------------------------------
# reformatting: src/reformatting.jl
default_blocksize = 2 ** 12


def _maybe_rationalize(r):  # src/reformatting.jl:103-111
    for num, den in ((1, 1), (2, 1), (3, 1), (1, 2), (1, 3), (3, 2), (2, 3)):
        v = num / den
        if abs(v - r) <= np.spacing(r):
            return (num, den)
    return float(r)


def _resample(x, fs, blocksize):  # __ToFramerate__ :113-122
    ratio = _maybe_rationalize(fs / x.fs)
    if ratio == (1, 1):
        return x
    return FilteredSignal(x, ResamplerFn(ratio, fs), blocksize, fs)


def _stretchtime(t, scale):  # src/cutting.jl:140-141
    if U.is_frames(t) and scale is not None:
        return Quantity(int(math.floor(t.value * scale)), "frames")
    return t


def ToFramerate(x, fs=None, blocksize=default_blocksize):
    if fs is None and not isinstance(x, (AbstractSignal, np.ndarray, tuple, list)) and not _is_torch(x):
        rate = x  # curried form ToFramerate(fs)
        return Curried(lambda y: ToFramerate(y, rate, blocksize))
    x = _assignal(x)
    fs = U.inHz(fs)
    if fs is None:  # :67,85-86
        return x
    if x.fs is not None and fs == x.fs:
        return x
    known = x.fs is not None
    bs = blocksize
    if isinstance(x, ArraySig):  # src/arrays.jl:47-50
        return _resample(x, fs, bs) if known else ArraySig(x.data, fs)
    if isinstance(x, FuncSig):  # src/functions.jl:62-63
        return FuncSig(x.fn, fs, x.omega, x.phi, rng=x.rng, pyfn=x.pyfn)
    if isinstance(x, NumberSig):  # src/numbers.jl:56-57
        return NumberSig(x.val, fs, dB=x.dB, dtype=x.dtype)
    if isinstance(x, MapSignal):  # src/mapsignal.jl:46-64
        if known and not fs < x.fs:
            return _resample(x, fs, bs)
        kids = [ToFramerate(c, fs, bs) for c in x.signals]
        return _OperateOn(x.fn, kids, padding=x.padding, bychannel=x.bychannel, extra=x.extra)
    if isinstance(x, FilteredSignal):  # src/filters.jl:143-157
        if known and x.fs != x.signal.fs:
            return _resample(x.signal, fs, bs)
        return FilteredSignal(ToFramerate(x.signal, fs, bs), x.fn, x.blocksize, fs)
    computed = x.evaltrait == "computed"
    if known and not computed:  # generic DataSignal method :88-90
        return _resample(x, fs, bs)
    if isinstance(x, CutApply):  # src/cutting.jl:142-152
        scale = fs / x.fs if known else None
        return CutApply(ToFramerate(x.signal, fs, bs), _stretchtime(x.time, scale), x.kind)
    if isinstance(x, PaddedSignal):  # src/padding.jl:16-19
        return PaddedSignal(ToFramerate(x.signal, fs, bs), x.pad)
    if isinstance(x, AppendSignals):  # src/appending.jl:77-80
        return _Append([ToFramerate(c, fs, bs) for c in x.signals])
    if isinstance(x, RampSignal):  # src/ramps.jl:28-43
        return RampSignal(x.direction, ToFramerate(x.signal, fs, bs), x.time, x.fn)
    if isinstance(x, NormedSignal):  # src/filters.jl:276-285
        return NormedSignal(ToFramerate(x.signal, fs, bs))
    error(f"Value is not a signal: {x!r}")


def ToChannels(x, ch=None):  # src/reformatting.jl:132-170
    if ch is None:
        n = x
        return Curried(lambda y: ToChannels(y, n))
    x = _assignal(x)
    if ch == x.nch:
        return x
    if ch == 1:
        return _OperateOn(AS1CHANNEL, [x], bychannel=False)
    if x.nch == 1:
        return _OperateOn(ASNCHANNELS, [x], bychannel=False, extra=ch)
    error(f"No rule to convert signal with {x.nch} channels to a signal with {ch} channels.")


def ToEltype(x, T=None):  # src/reformatting.jl:183-184
    if T is None:
        t = x
        return Curried(lambda y: ToEltype(y, t))
    return _OperateOn(TOELTYPE, [_assignal(x)], extra=np.dtype(T))


def Format(x, fs, ch=None):  # src/reformatting.jl:207-213
    x = _assignal(x)
    if ch is None:
        ch = x.nch
    if ch > 1 and x.nch == 1:
        return ToChannels(ToFramerate(x, fs), ch)
    return ToFramerate(ToChannels(x, ch), fs)


def Uniform(xs, channels=False):  # src/reformatting.jl:241-254
    xs = [_assignal(x) for x in xs]
    rates = [x.fs for x in xs if x.fs is not None]
    fs = max(rates) if rates else None
    if not channels:
        return [Format(x, fs) for x in xs]
    ch = max(x.nch for x in xs)
    return [Format(x, fs, ch) for x in xs]


# --------------------------------------------------------------------------
# mapping: src/mapsignal.jl:131-145
def _OperateOn(fn, xs, padding=None, bychannel=True, extra=0, blocksize=default_blocksize):
    if padding is None:
        padding = default_pad(fn)
    xs = Uniform(xs, channels=bychannel)
    return MapSignal(fn, xs, xs[0].fs, padding, bychannel, extra, blocksize)


def OperateOn(fn, *xs, padding=None, bychannel=True):
    code = _map_code(fn)
    if code == REVERSECH:
        bychannel = False
    if fn is operator.neg and len(xs) != 1:
        error("negation takes one signal")
    return _OperateOn(code, list(xs), padding=padding, bychannel=bychannel)


def Operate(fn, *xs, **kw):
    return Curried(lambda x: OperateOn(fn, x, *xs, **kw))


def _nary(code):
    def f(*xs):
        if len(xs) == 1:
            y = xs[0]
            return Curried(lambda x: _OperateOn(code, [x, y]))
        return _OperateOn(code, list(xs))

    return f


Mix = _nary(ADD)  # src/mapsignal.jl:307-308
Amplify = _nary(MUL)  # src/mapsignal.jl:332-333


def AddChannel(*xs):  # src/mapsignal.jl:359-360
    if len(xs) == 1:
        y = xs[0]
        return Curried(lambda x: _OperateOn(TUPLECAT, [x, y], bychannel=False))
    return _OperateOn(TUPLECAT, list(xs), bychannel=False)


def SelectChannel(x, n=None):  # src/mapsignal.jl:388-391
    if n is None:
        k = x
        return Curried(lambda y: SelectChannel(y, k))
    x = _assignal(x)
    return _OperateOn(GETCHAN, [x], bychannel=False, extra=int(n))


# --------------------------------------------------------------------------
# appending: src/appending.jl:59-76
def _Append(xs):
    xs = Uniform(xs, channels=True)
    if any(isknowninf(nframes(x)) for x in xs[:-1]):
        error("Cannot Append to the end of an infinite signal")
    El = xs[0].dtype
    for x in xs[1:]:
        El = promote_type(El, x.dtype)
    xs = [x if x.dtype == El else ToEltype(x, El) for x in xs]
    lens = [nframes(x) for x in xs]
    if any(isknowninf(n) for n in lens):
        ln = inflen
    elif any(n is None for n in lens):
        ln = None
    else:
        ln = sum(lens)
    return AppendSignals(xs, ln, El)


def Append(*xs):
    if len(xs) == 1:
        y = xs[0]
        return Curried(lambda x: _Append([x, y]))
    return _Append(list(xs))


def Prepend(*xs):  # src/appending.jl:30-31
    if len(xs) == 1:
        x0 = xs[0]
        return Curried(lambda y: _Append([x0, y]))
    return _Append(list(reversed(xs)))


# --------------------------------------------------------------------------
# ramps: src/ramps.jl:156-161,191-196,227-231,263-273
def _ramp_args(args):
    ln, fn = 10 * U.ms, "sinramp"
    for a in args:
        if callable(a) or isinstance(a, str):
            fn = a
        else:
            ln = a
    if fn is sinramp:
        fn = "sinramp"
    elif fn is identity or getattr(fn, "__name__", "") == "identity":
        fn = "identity"
    if fn not in ("sinramp", "identity"):
        error("ramp function is an opaque closure for the HIP engine (sinramp, identity supported)")
    return ln, fn


def _is_sigarg(a):
    return isinstance(a, (AbstractSignal, np.ndarray, tuple, list)) or _is_torch(a)


def _mkramp(direction):
    def R(*args):
        if not args or not _is_sigarg(args[0]):
            return Curried(lambda x: R(x, *args))
        x = _assignal(args[0])
        ln, fn = _ramp_args(args[1:])
        return _OperateOn(MUL, [x, RampSignal(direction, x, ln, fn)])

    return R


RampOn = _mkramp("on")
RampOff = _mkramp("off")


def Ramp(*args):
    if not args or not _is_sigarg(args[0]):
        return Curried(lambda x: Ramp(x, *args))
    x = _assignal(args[0])
    return RampOff(RampOn(x, *args[1:]), *args[1:])


def FadeTo(*args):
    if len(args) < 2 or not _is_sigarg(args[1]):
        y, rest = args[0], args[1:]
        return Curried(lambda x: FadeTo(x, y, *rest))
    x, y = args[0], args[1]
    ln, fn = _ramp_args(args[2:])
    x, y = Uniform((x, y))
    if x.fs is None:
        error("Unknown frame rate is not supported by `FadeTo`.")
    n = U.inframes_int(U.maybeseconds(ln), x.fs)
    silence = _Until(NumberSig(0, None, dtype=y.dtype) if y.dtype != I64 else NumberSig(0),
                     (nframes(x) - n) * U.frames)
    return Mix(RampOff(x, ln, fn), Prepend(RampOn(y, ln, fn), silence))


# --------------------------------------------------------------------------
# filters: src/filters.jl:14-20,54-66,96-97,323-326
class _Response:
    """DSP.jl response types.  The CLASS is the tag of `Filt(Lowpass, 4kHz)`; an INSTANCE,
    `Highpass(8, fs=100)`, is DSP.jl's response object for `digitalfilter` (test/runtests.jl:365-367)."""

    def __init__(self, *bounds, fs=None):
        self.bounds = tuple(U.inHz(b) for b in bounds)
        self.fs = None if fs is None else float(U.inHz(fs))


class Lowpass(_Response): pass  # noqa: E701
class Highpass(_Response): pass  # noqa: E701
class Bandpass(_Response): pass  # noqa: E701
class Bandstop(_Response): pass  # noqa: E701


class ZeroPoleGain:
    """DSP.jl ZeroPoleGain (what `digitalfilter` returns)"""

    def __init__(self, z, p, k):
        self.z = np.asarray(z, dtype=np.complex128).ravel()
        self.p = np.asarray(p, dtype=np.complex128).ravel()
        self.k = float(k)


class SecondOrderSections:
    """DSP.jl SecondOrderSections: rows [b0 b1 b2 1 a1 a2] + gain"""

    def __init__(self, sos, gain=1.0):
        self.sos = np.asarray(sos, dtype=np.float64).reshape(-1, 6)
        self.gain = float(gain)


class Biquad:
    def __init__(self, b0, b1, b2, a1, a2):
        self.row = [float(b0), float(b1), float(b2), 1.0, float(a1), float(a2)]


# largest relative l2 distance between the impulse responses of a PolynomialRatio's direct form and of its factored
# cascade the engine accepts (so_tf_to_sos's residual; the suite holds Float64 results to 1e-8)
TF_RESIDUAL_MAX = 1e-9


class PolynomialRatio:
    """DSP.jl PolynomialRatio(b, a): FIR for a == [a0], one section for orders <= 2, otherwise factored into
    second-order sections by the library (so_tf_to_sos)"""

    def __init__(self, b, a):
        self.b = np.asarray(b, dtype=np.float64).ravel()
        self.a = np.asarray(a, dtype=np.float64).ravel()


def digitalfilter(response, method):
    """digitalfilter(Highpass(8, fs=fs), Chebyshev1(5, 1)) -> ZeroPoleGain, through the library's
    design entry point (the same code the named form `Filt(Highpass, 8Hz, ...)` uses)"""
    from . import _capi as K
    import ctypes as C

    if not isinstance(response, _Response) or response.fs is None:
        error("digitalfilter needs a response object with a frame rate, e.g. Highpass(8, fs=100)")
    order = method[1]
    ripple = method[2] if len(method) > 2 else 0.0
    cap = 2 * order + 2
    z = (C.c_double * (2 * cap))()
    p = (C.c_double * (2 * cap))()
    nz, npl, k = C.c_int32(0), C.c_int32(0), C.c_double(0)
    b = response.bounds
    st = K.lib().so_design_iir_zpk(K.FILT[_DESIGNS[type(response)]], b[0], b[1] if len(b) > 1 else 0.0, response.fs,
                                   K.METHOD[method[0]], order, ripple, z, C.byref(nz), p, C.byref(npl), cap, C.byref(k))
    if st != 0:
        error(K.last_error())
    zz = np.array(z[: 2 * nz.value]).view(np.complex128)
    pp = np.array(p[: 2 * npl.value]).view(np.complex128)
    return ZeroPoleGain(zz, pp, k.value)


def tf_to_sos(b, a):
    """(sos rows, gain, residual) of so_tf_to_sos: the direct-form coefficients factored into second-order sections"""
    from . import _capi as K
    import ctypes as C

    b = np.ascontiguousarray(np.atleast_1d(b), dtype=np.float64)
    a = np.ascontiguousarray(np.atleast_1d(a), dtype=np.float64)
    cap = 6 * (max(len(a), len(b)) // 2 + 2)
    sos = (C.c_double * cap)()
    nsec, gain, resid = C.c_int32(0), C.c_double(0), C.c_double(0)
    dp = C.POINTER(C.c_double)
    st = K.lib().so_tf_to_sos(b.ctypes.data_as(dp), len(b), a.ctypes.data_as(dp), len(a), sos, cap, C.byref(nsec),
                              C.byref(gain), C.byref(resid))
    if st != 0:
        error(K.last_error())
    return np.array(sos[: 6 * nsec.value]).reshape(-1, 6), gain.value, resid.value


def tf_zero_input(b, a, si, nmax):
    """zero-input response of the direct form from state `si` (so_tf_zero_input), cut where it has decayed"""
    from . import _capi as K
    import ctypes as C

    b = np.ascontiguousarray(np.atleast_1d(b), dtype=np.float64)
    a = np.ascontiguousarray(np.atleast_1d(a), dtype=np.float64)
    si = np.ascontiguousarray(si, dtype=np.float64).ravel()
    out = np.empty(max(int(nmax), 0), dtype=np.float64)
    used = C.c_int64(0)
    dp = C.POINTER(C.c_double)
    st = K.lib().so_tf_zero_input(b.ctypes.data_as(dp), len(b), a.ctypes.data_as(dp), len(a), si.ctypes.data_as(dp),
                                  len(si), out.ctypes.data_as(dp), len(out), C.byref(used))
    if st != 0:
        error(K.last_error())
    return out[: used.value]


def _raw_filter(h):
    """RawFilterFn(h) (reference src/filters.jl:89-97): a DSP.jl filter object -> what the engine
    lowers: SOS rows + gain (IIR; resolve_filter = DF2TFilter(h)) or FIR coefficients"""
    from . import _capi as K
    import ctypes as C

    if isinstance(h, (RawFilterFn, RawFirFn)):
        return h
    if isinstance(h, SecondOrderSections):
        return RawFilterFn(h.sos, h.gain)
    if isinstance(h, Biquad):
        return RawFilterFn([h.row], 1.0)
    if isinstance(h, ZeroPoleGain):
        z = np.ascontiguousarray(h.z).view(np.float64)
        p = np.ascontiguousarray(h.p).view(np.float64)
        cap = 6 * (len(h.p) + 2)
        sos = (C.c_double * cap)()
        nsec, gain = C.c_int32(0), C.c_double(0)
        st = K.lib().so_zpk_to_sos(z.ctypes.data_as(C.POINTER(C.c_double)), len(h.z),
                                   p.ctypes.data_as(C.POINTER(C.c_double)), len(h.p), h.k, sos, cap,
                                   C.byref(nsec), C.byref(gain))
        if st != 0:
            error(K.last_error())
        return RawFilterFn(np.array(sos[: 6 * nsec.value]).reshape(-1, 6), gain.value)
    if isinstance(h, PolynomialRatio):
        a0 = h.a[0]
        b, a = h.b / a0, h.a / a0
        if len(a) == 1:
            return RawFirFn(b)
        if len(a) <= 3 and len(b) <= 3:
            b = np.concatenate([b, np.zeros(3 - len(b))])
            a = np.concatenate([a, np.zeros(3 - len(a))])
            return RawFilterFn([[b[0], b[1], b[2], 1.0, a[1], a[2]]], 1.0)
        sos, gain, resid = tf_to_sos(b, a)
        if not resid <= TF_RESIDUAL_MAX:
            error("PolynomialRatio: the polynomials of this filter are too ill-conditioned to be factored into "
                  f"second-order sections (impulse responses differ by {resid:.1e}): pass SecondOrderSections or "
                  "ZeroPoleGain")
        return RawFilterFn(sos, gain)
    arr = np.asarray(h, dtype=np.float64)
    if arr.ndim == 1 and arr.size >= 1:  # FIR coefficient vector
        return RawFirFn(arr)
    error(f"Filt: unsupported filter object {h!r}")


def Butterworth(order):
    return ("butterworth", int(order))


def Chebyshev1(order, ripple):
    return ("chebyshev1", int(order), float(ripple))


def _nyquist_check(x, hz):
    if x.fs is not None and U.inHz(hz) >= 0.5 * x.fs:
        error(f"The frequency {hz} cannot be represented at a sampling rate of {x.fs} Hz. "
              "Increase the frame rate or lower the frequency.")


_DESIGNS = {Lowpass: "lowpass", Highpass: "highpass", Bandpass: "bandpass", Bandstop: "bandstop"}


def _is_signal_like(v):
    if isinstance(v, (ZeroPoleGain, SecondOrderSections, Biquad, PolynomialRatio, RawFilterFn, RawFirFn, FilterFn)):
        return False
    if isinstance(v, AbstractSignal) or _is_torch(v):
        return True
    if isinstance(v, tuple):
        return True
    if isinstance(v, np.ndarray):
        return v.ndim == 2  # a 1-D vector on its own is a coefficient vector: Filt(h)
    return False


def Filt(*args, blocksize=default_blocksize, order=5, method=None, sos=None, gain=1.0):
    if args and isinstance(args[0], type) and args[0] in _DESIGNS:  # curried Filt(Type,bounds...)
        a = args
        return Curried(lambda x: Filt(x, *a, blocksize=blocksize, order=order, method=method))
    if not args and sos is not None:
        return Curried(lambda x: Filt(x, blocksize=blocksize, sos=sos, gain=gain))
    if len(args) == 1 and not _is_signal_like(args[0]):  # curried Filt(h)
        h = args[0]
        return Curried(lambda x: Filt(x, h, blocksize=blocksize))
    x = _assignal(args[0])
    if sos is not None:  # Filt(x,h): raw DSP.jl filter object -> SOS rows
        return FilteredSignal(x, RawFilterFn(sos, gain), blocksize, x.fs)
    if len(args) == 2 and isinstance(args[1], FilterFn):
        return FilteredSignal(x, args[1], blocksize, x.fs)
    if len(args) == 2:  # Filt(x,h): RawFilterFn(h), reference src/filters.jl:96-97
        return FilteredSignal(x, _raw_filter(args[1]), blocksize, x.fs)
    if len(args) < 3 or not (isinstance(args[1], type) and args[1] in _DESIGNS):
        error("Filt(x, Type, bounds...) expected")
    if method is None:
        method = Butterworth(order)
    bounds = args[2:]
    for b in bounds:
        _nyquist_check(x, b)
    fn = FilterFn(_DESIGNS[args[1]], method, [U.inHz(b) for b in bounds])
    return FilteredSignal(x, fn, blocksize, x.fs)


def _Normpower(x):  # src/filters.jl:323-326
    x = _assignal(x)
    return NormedSignal(x)


Normpower = Curried(_Normpower)
