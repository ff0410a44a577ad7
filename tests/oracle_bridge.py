"""Test-side loader of the CPU oracle (oracle/libsigops_oracle.so).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may use the oracle."""
import ctypes as C
import os

import numpy as np

import sigops_amd as so
from sigops_amd import _capi as K
from sigops_amd import signals as S
from sigops_amd.lowering import lower, _DT

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.environ.get("SIGOPS_ORACLE_SO") or os.path.join(ROOT, "oracle", "libsigops_oracle.so")
_lib = None


def oracle_lib():
    global _lib
    if _lib is None:
        L = C.CDLL(ORACLE_SO)
        L.so_oracle_sink.restype = C.c_int32
        L.so_oracle_sink.argtypes = [C.POINTER(K.so_node_t), C.c_int32, C.c_int32,
                                     C.POINTER(K.so_out_desc_t), C.c_void_p, C.c_int32]
        L.so_oracle_nframes.restype = C.c_int64
        L.so_oracle_nframes.argtypes = [C.POINTER(K.so_node_t), C.c_int32, C.c_int32]
        L.so_oracle_last_error.restype = C.c_char_p
        L.so_oracle_set_semantics.restype = None
        L.so_oracle_set_semantics.argtypes = [C.c_int]
        L.so_oracle_set_positions.restype = None
        L.so_oracle_set_positions.argtypes = [C.c_int]
        _lib = L
    return _lib


class oracle_positions:
    """with oracle_positions("exact"): ... -- resampler positions of the oracle's arbitrary-rate
    kernel: "accumulate" (default, DSP.jl's Float64 phase accumulator = the reference's algorithm)
    or "exact" (closed form; measurement aid, see oracle/sigops_oracle.c header)."""

    def __init__(self, mode):
        self.mode = {"accumulate": 0, "exact": 1}[mode]

    def __enter__(self):
        oracle_lib().so_oracle_set_positions(self.mode)

    def __exit__(self, *exc):
        oracle_lib().so_oracle_set_positions(-1)


class oracle_semantics:
    """with oracle_semantics("intended"): ... -- a filtered / resampled child ends after nframes(x)
    frames instead of never (reference quirk C-7, see oracle/sigops_oracle.c so_oracle_set_semantics)"""

    def __init__(self, mode):
        self.mode = {"reference": 0, "intended": 1}[mode]

    def __enter__(self):
        oracle_lib().so_oracle_set_semantics(self.mode)

    def __exit__(self, *exc):
        oracle_lib().so_oracle_set_semantics(0)


def oracle_sink_lowered(lw, nframes, nch, dtype, blocksize=0):
    res = np.empty((nframes, nch), dtype=dtype, order="F")
    desc = K.so_out_desc_t(dtype=_DT[np.dtype(dtype)], nch=nch, nframes=nframes, frame_stride=1,
                           chan_stride=max(nframes, 1), is_device=0)
    st = oracle_lib().so_oracle_sink(lw.nodes, lw.n, lw.root, C.byref(desc),
                                     res.ctypes.data_as(C.c_void_p), blocksize)
    if st != 0:
        raise S.ErrorException(oracle_lib().so_oracle_last_error().decode())
    return res


def oracle_sink(x, nframes=None, blocksize=0, rng=None, dtype=None):
    """CPU oracle analogue of sink(x, Array) (always evaluates; no DataCut view)."""
    x = so.process_sink_params(x) if nframes is None else S._assignal(x)
    n = S.nframes(x) if nframes is None else nframes
    if dtype is None:
        dtype = S.float_type(x.dtype)
    lw = lower(x, nframes_out=n, rng=rng)
    return oracle_sink_lowered(lw, n, x.nch, dtype, blocksize)


def oracle_filt(b, a, x, si=None):
    """DSP.filt(b, a, x::AbstractSignal[, si]) on the CPU oracle (reference src/filters.jl:68-79): the signal sunk by the
    oracle, then DSP.jl's direct-form recurrence (so_oracle_filt_direct)"""
    L = oracle_lib()
    dp = C.POINTER(C.c_double)
    L.so_oracle_filt_direct.restype = C.c_int
    L.so_oracle_filt_direct.argtypes = [dp, C.c_int, dp, C.c_int, dp, dp, C.c_int64, C.c_int, C.c_int64, dp, C.c_int64]
    data = np.asfortranarray(x if isinstance(x, np.ndarray) else oracle_sink(x), dtype=np.float64)
    b = np.ascontiguousarray(np.atleast_1d(b), dtype=np.float64)
    a = np.ascontiguousarray(np.atleast_1d(a), dtype=np.float64)
    n, nch = data.shape
    y = np.empty((n, nch), order="F")
    zi, zs = None, 0
    if si is not None:
        zi = np.asfortranarray(si, dtype=np.float64)
        zs = zi.shape[0] if zi.ndim == 2 else 0
    st = L.so_oracle_filt_direct(b.ctypes.data_as(dp), len(b), a.ctypes.data_as(dp), len(a), data.ctypes.data_as(dp),
                                 y.ctypes.data_as(dp), n, nch, max(n, 1), zi.ctypes.data_as(dp) if zi is not None else None, zs)
    assert st == 0
    return y


def oracle_nframes(x):
    lw = lower(S._assignal(x))
    return oracle_lib().so_oracle_nframes(lw.nodes, lw.n, lw.root)


_RELERR_LOG = {}


def _relerr_dump():
    import json

    path = os.environ.get("SIGOPS_RECORD_RELERR")
    if path and _RELERR_LOG:
        with open(path, "w") as f:
            json.dump({k: {"max": v[0], "calls": v[1]} for k, v in sorted(_RELERR_LOG.items())}, f, indent=1)


def relerr(a, b):
    """norm-wise relative error, Julia isapprox semantics (runtests.jl:356).  SIGOPS_RECORD_RELERR=<file>: the largest
    value every test saw, per result dtype, is written there at exit (profiles/r04/relerr_maxima.json: what the
    tolerances of the suite are set against)."""
    dt = getattr(a, "dtype", None)
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    nb = np.linalg.norm(b)
    r = np.linalg.norm(a - b) / (nb if nb > 0 else 1.0)
    if os.environ.get("SIGOPS_RECORD_RELERR"):
        if not _RELERR_LOG:
            import atexit

            atexit.register(_relerr_dump)
        key = "%s [%s]" % (os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], dt)
        old = _RELERR_LOG.get(key, (0.0, 0))
        _RELERR_LOG[key] = (max(old[0], float(r)) if np.isfinite(r) else old[0], old[1] + 1)
    return r
