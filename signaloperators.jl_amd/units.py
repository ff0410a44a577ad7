"""Units mirror of SignalOperators.Units (reference src/SignalOperators.jl:11-15,
SignalBase/Unitful semantics recorded in SURVEY.md Appendix B).

    5*s, 10*ms, 44.1*kHz, 5*frames, -20*dB, pi*rad, 180*deg

Plain numbers are seconds for times (`maybeseconds`, reference src/util.jl:23-24)
and Hz for rates (`inHz`).
"""
import math


class Quantity:
    __array_ufunc__ = None  # keep numpy from broadcasting over us
    __slots__ = ("value", "unit")

    def __init__(self, value, unit):
        self.value = value
        self.unit = unit

    def __rmul__(self, other):
        if isinstance(other, Quantity):
            return NotImplemented
        return Quantity(other * self.value, self.unit)

    def __mul__(self, other):
        if isinstance(other, Quantity):
            return NotImplemented
        return Quantity(self.value * other, self.unit)

    def __truediv__(self, other):
        return Quantity(self.value / other, self.unit)

    def __neg__(self):
        return Quantity(-self.value, self.unit)

    def __add__(self, other):
        if isinstance(other, Quantity) and other.unit == self.unit:
            return Quantity(self.value + other.value, self.unit)
        if isinstance(other, Quantity) and {self.unit, other.unit} <= {"s", "ms"}:
            return Quantity(inseconds_raw(self) + inseconds_raw(other), "s")
        return NotImplemented

    def __sub__(self, other):
        return self + (-other)

    def __repr__(self):
        return f"{self.value} {self.unit}"


s = Quantity(1, "s")
ms = Quantity(1, "ms")
Hz = Quantity(1, "Hz")
kHz = Quantity(1, "kHz")
frames = Quantity(1, "frames")
kframes = Quantity(1000, "frames")
dB = Quantity(1, "dB")
rad = Quantity(1, "rad")
deg = Quantity(1, "deg")


def inseconds_raw(q):
    if q.unit == "s":
        return float(q.value)
    if q.unit == "ms":
        return q.value / 1000  # rational scaling like Unitful (10ms == 1//100 s)
    raise ValueError(f"not a time: {q}")


def is_frames(t):
    return isinstance(t, Quantity) and t.unit == "frames"


def is_time(t):
    return isinstance(t, Quantity) and t.unit in ("s", "ms")


def maybeseconds(t):
    """reference src/util.jl:23-24: bare numbers are seconds"""
    if isinstance(t, Quantity):
        return t
    return Quantity(t, "s")


def inHz(x):
    """SignalBase.inHz: strip the unit; None (missing) passes through"""
    if x is None:
        return None
    if isinstance(x, Quantity):
        if x.unit == "Hz":
            return float(x.value)
        if x.unit == "kHz":
            return float(x.value * 1000)
        raise ValueError(f"not a frequency: {x}")
    return float(x)


def inframes_int(t, fs):
    """SignalBase.inframes(Int,t,fs) = floor(Int, seconds(t)*fs); frame quantities
    pass through (floored). Returns None when fs is missing and t is a time."""
    t = maybeseconds(t)
    if t.unit == "frames":
        return int(math.floor(t.value))
    if fs is None:
        return None
    return int(math.floor(inseconds_raw(t) * fs))


def inseconds(t, fs=None):
    t = maybeseconds(t)
    if t.unit == "frames":
        if fs is None:
            return None
        return t.value / fs
    return inseconds_raw(t)


def inradians(phi, omega=None):
    """SignalBase.inradians(Float64,ϕ,ω): time-valued ϕ => 2π·ω·t
    (reference test/runtests.jl:74-79)"""
    if isinstance(phi, Quantity):
        if phi.unit == "rad":
            return float(phi.value)
        if phi.unit == "deg":
            return float(phi.value) * math.pi / 180.0
        if phi.unit in ("s", "ms"):
            if omega is None:
                raise ValueError("time-valued phase needs a frequency")
            return 2 * math.pi * omega * inseconds_raw(phi)
        raise ValueError(f"not a phase: {phi}")
    return float(phi)


def gain_ratio(q):
    """uconvertrp(NoUnits, x dB) = 10^(x/20) (reference src/numbers.jl:47-49)"""
    return 10.0 ** (q.value / 20.0)
