"""Flatten a lazy operator tree (signals.py) into the `so_node_t` table of
include/sigops.h.  This is what the Julia glue (julia/SignalOperatorsHIP.jl) does on
the reference side before its `ccall`.  The same table is consumed by the HIP engine
(product) and, in tests only, by the CPU oracle."""
import ctypes as C
import math

import numpy as np

from . import _capi as K
from . import signals as S
from . import units as U

_DT = {S.F32: K.SO_F32, S.F64: K.SO_F64, S.I64: K.SO_I64}


def _len_code(n):
    if n is None:
        return K.SO_LEN_MISSING
    if S.isknowninf(n):
        return K.SO_LEN_INF
    return int(n)


_DESIGNS = {}  # (what, parameters) -> designed coefficients (designs are pure functions of their parameters)


def design_iir(fn, fs):
    """FilterFn(fs): digitalfilter(design(args...,fs=fs),method) -> SOS
    (reference src/filters.jl:10-11,94) through the library's design entry point."""
    key = ("iir", fn.design, tuple(fn.method), tuple(float(a) for a in fn.args), float(fs))
    if key not in _DESIGNS:
        _DESIGNS[key] = _design_iir(fn, fs)
    sos, gain = _DESIGNS[key]
    return sos.copy(), gain  # (callers own their copy)


def _design_iir(fn, fs):
    L = K.lib()
    method = fn.method
    order = method[1]
    ripple = method[2] if len(method) > 2 else 0.0
    cap = 6 * (2 * order + 2)
    sos = (C.c_double * cap)()
    nsec = C.c_int32(0)
    gain = C.c_double(0)
    f1 = fn.args[0]
    f2 = fn.args[1] if len(fn.args) > 1 else 0.0
    st = L.so_design_iir(K.FILT[fn.design], f1, f2, fs, K.METHOD[method[0]], order, ripple,
                         sos, cap, C.byref(nsec), C.byref(gain))
    if st != 0:
        S.error(K.last_error())
    return np.array(sos[: 6 * nsec.value], dtype=np.float64).reshape(-1, 6), gain.value


def design_resample(ratio):
    """resample_filter(ratio) (reference src/reformatting.jl:93)"""
    key = ("resample", ratio)
    if key not in _DESIGNS:
        _DESIGNS[key] = _design_resample(ratio)
    return _DESIGNS[key].copy()


def _design_resample(ratio):
    L = K.lib()
    n = C.c_int32(0)
    if isinstance(ratio, tuple):
        L.so_design_resample_rational(ratio[0], ratio[1], None, 0, C.byref(n))
        h = (C.c_double * n.value)()
        st = L.so_design_resample_rational(ratio[0], ratio[1], h, n.value, C.byref(n))
    else:
        L.so_design_resample_arbitrary(float(ratio), 32, None, 0, C.byref(n))
        h = (C.c_double * n.value)()
        st = L.so_design_resample_arbitrary(float(ratio), 32, h, n.value, C.byref(n))
    if st != 0:
        S.error(K.last_error())
    return np.array(h[:], dtype=np.float64)


_HLEN = {}


def _resample_hlen(ratio):
    if ratio not in _HLEN:
        _HLEN[ratio] = len(design_resample(ratio))
    return _HLEN[ratio]


def _pad_fields(pad, nch):
    """usepad resolution, reference src/padding.jl:150-192"""
    if pad is S.zero:
        return K.PAD["zero"], 0.0, None
    if pad is S.one:
        return K.PAD["one"], 0.0, None
    if pad is S.lastframe:
        return K.PAD["lastframe"], 0.0, None
    if pad is S.cycle:
        return K.PAD["cycle"], 0.0, None
    if pad is S.mirror:
        return K.PAD["mirror"], 0.0, None
    if isinstance(pad, (bool, int, float, np.integer, np.floating)):
        return K.PAD["value"], float(pad), None
    if isinstance(pad, (tuple, list, np.ndarray)):
        v = np.ascontiguousarray(np.asarray(pad, dtype=np.float64).reshape(-1))
        if v.size != nch:
            S.error("padding vector length must equal the channel count")
        return K.PAD["vector"], 0.0, v
    if callable(pad):
        import inspect

        try:
            nargs = len(inspect.signature(pad).parameters)
        except (TypeError, ValueError):
            nargs = -1
        if nargs not in (1, 3):
            S.error(f"Pad function ({pad}) must take 1 or 3 arguments. Refer to `Pad` documentation.")
        S.error("Pad: opaque padding closures cannot be lowered to the HIP engine")
    S.error(f"unsupported padding {pad!r}")


class Lowered:
    def __init__(self):
        self.rows = []  # python dicts, converted at the end
        self.keep = []  # keep-alive for ctypes / numpy buffers
        self.memo = {}
        self.array_nodes = []  # (index, ArraySig)
        self.nodes = None
        self.root = -1

    def add(self, **kw):
        self.rows.append(kw)
        return len(self.rows) - 1

    def finish(self, root):
        n = len(self.rows)
        arr = (K.so_node_t * n)()
        for i, r in enumerate(self.rows):
            nd = arr[i]
            kids = r.pop("children", ())
            for k, v in r.items():
                setattr(nd, k, v)
            nd.n_children = len(kids)
            if kids:
                ck = (C.c_int32 * len(kids))(*kids)
                self.keep.append(ck)
                nd.children = C.cast(ck, C.POINTER(C.c_int32))
        self.nodes = arr
        self.root = root
        self.n = n
        return self


# so.stream(): host arrays of the tree are uploaded once and every block's plan reads the device
# copy (id(ArraySig) -> torch tensor); None outside a stream
_device_cache = None


def _array_fields(x):
    d = x.data
    if _device_cache is not None and not S._is_torch(d) and d.dtype in (np.float32, np.float64) and d.size:
        t = _device_cache.get(id(x))
        if t is None:
            import torch

            dev = _device_cache["device"]
            if d.ndim == 1:
                t = torch.from_numpy(np.ascontiguousarray(d)).to(dev)
            else:  # planar on the device: channels contiguous in time
                t = torch.from_numpy(np.ascontiguousarray(d.T)).to(dev).t()
            _device_cache[id(x)] = t
            _device_cache.setdefault("keep", []).append(x)
        d = t
    if S._is_torch(d):
        ptr = d.data_ptr()
        st = d.stride()
        is_dev = 1 if d.is_cuda else 0
        fstride = st[0]
        cstride = st[1] if len(st) > 1 else 0
    else:
        ptr = d.ctypes.data
        item = d.itemsize
        fstride = d.strides[0] // item if d.shape[0] > 1 or d.strides[0] else 1
        cstride = (d.strides[1] // item) if d.ndim > 1 else 0
        is_dev = 0
    return ptr, fstride, cstride, is_dev


def _demand(x, n, out, skip=0):
    """how many frames of every randn leaf a sink of n frames reaches, and how many of the leading
    ones are SKIPPED rather than evaluated: `After` pulls the frames it drops with skip=true
    (reference src/cutting.jl:160-181) and a generator leaf draws nothing for them
    (src/functions.jl:113-114 is only reached through `frame`), while stateful nodes ignore the
    flag and evaluate their child (`Filt`, src/filters.jl:241-244)."""
    if n is None or S.isknowninf(n) or n <= 0:
        return  # (infinite demands are rejected by the planner with the reference's error)
    if isinstance(x, S.FuncSig):
        if x.fn in (S.RANDN, S.OPAQUE):
            n0, s0 = out.get(id(x), (0, skip))
            out[id(x)] = (max(n0, n), min(s0, skip))
        return
    if isinstance(x, S.ArraySig):
        n0, s0 = out.get(id(x), (0, skip))
        out[id(x)] = (max(n0, n), min(s0, skip))
        return
    if isinstance(x, (S.NumberSig, S.RampSignal)):
        return

    def capped(c, m):
        cl = S.nframes(c)
        if cl is None or S.isknowninf(cl):
            return m
        return min(m, cl)

    if isinstance(x, S.CutApply):
        L = x.resolvelen() or 0
        if x.kind == "until":
            _demand(x.signal, capped(x.signal, min(n, max(0, L))), out, skip)
        else:
            _demand(x.signal, capped(x.signal, n + max(0, L)), out, skip + max(0, L))
    elif isinstance(x, S.PaddedSignal):
        _demand(x.signal, capped(x.signal, n), out, skip)
    elif isinstance(x, S.AppendSignals):
        rem, sk = n, skip
        for c in x.signals:
            cl = S.nframes(c)
            m = rem if (cl is None or S.isknowninf(cl)) else min(rem, cl)
            _demand(c, m, out, min(sk, m))
            rem -= m
            sk = max(0, sk - m)
            if rem <= 0:
                break
    elif isinstance(x, S.MapSignal):
        if isinstance(x.fn, S.OpaqueFn):  # host-materialised as a whole: its operands are sunk on their own
            n0, s0 = out.get(id(x), (0, skip))
            out[id(x)] = (max(n0, n), min(s0, skip))
            return
        for c in x.signals:
            _demand(c, capped(c, n), out, skip)
    elif isinstance(x, S.FilteredSignal):
        if isinstance(x.fn, S.ResamplerFn):
            # newest input of the last output: position (hlen-1)/2 + (n-1)*Nphi/ratio on the fine
            # grid, i.e. (n-1)/ratio inputs plus the filter's group delay (hlen-1)/(2 Nphi)
            # (reference src/reformatting.jl:92-96 setphase!(timedelay))
            r = x.fn.ratio
            nphi = r[0] if isinstance(r, tuple) else 32
            hlen = _resample_hlen(r)
            r = r[0] / r[1] if isinstance(r, tuple) else r
            m = int(math.ceil(max(n - 1, 0) / r)) + int(math.ceil((hlen - 1) / (2 * nphi))) + 2
        else:
            m = n
        _demand(x.signal, capped(x.signal, m), out, 0)
    elif isinstance(x, S.NormedSignal):
        _demand(x.signal, S.nframes(x.signal), out, 0)
    elif isinstance(x, S.RampSignal):
        return


def _apply_host(fn, args):
    """An opaque closure over host arrays, as the reference applies it: once per frame (src/functions.jl:53-56,
    src/mapsignal.jl:249-272).  Only a NumPy ufunc -- elementwise by construction, so one call over the whole array IS
    the per-frame calls -- or a closure that says so itself (`fn.vectorized = True`) is called once with the arrays;
    anything else (a stateful or random closure, `x / abs(x).max()`, `cumsum`: same shape, different values) runs
    element by element, exactly once per element."""
    if isinstance(fn, np.ufunc) or getattr(fn, "vectorized", False):
        out = np.asarray(fn(*args))
        if out.shape == np.asarray(args[0]).shape:
            return out
        S.error("a closure marked `vectorized` must return one value per frame")
    return np.frompyfunc(fn, len(args), 1)(*args).astype(np.float64)


def _host_function_leaf(s, n):
    """`Signal(fn)` with a closure the engine has no kernel for: the reference's own arithmetic
    (src/functions.jl:53-60, every operation separately rounded, first frame t = 1/fs) in NumPy"""
    if s.fs is None:
        S.error("Unknown frame rate: function signals need a frame rate before `sink` (use ToFramerate)")
    t = np.arange(1, n + 1, dtype=np.float64) / float(s.fs)
    if s.omega is not None:
        arg = 2 * np.pi * np.fmod(t * float(s.omega) + s.phi, 1.0)
    else:
        arg = t + s.phi
    return np.asfortranarray(np.asarray(_apply_host(s.pyfn, (arg,)), dtype=np.float64).reshape(-1, 1))


def _host_map_leaf(s, n, rng):
    """`OperateOn(fn, xs...)` with an opaque closure (reference src/mapsignal.jl:131-145, frame protocol
    :249-272): the operands -- extended with the map's padding, `Extend.(signals, padding)` at :26 -- are
    evaluated by the engine, the closure by NumPy; the result is handed back as an array leaf."""
    from . import engine
    from .units import frames

    cols = []
    for c in s.signals:
        cl = S.nframes(c)
        ext = c if (cl is None or S.isknowninf(cl) or cl >= n) else S.Extend(c, s.padding)
        cols.append(engine.sink(ext | S.Until(n * frames), engine.Array, rng=rng))
    if s.bychannel:
        out = _apply_host(s.fn.fn, cols)
    else:  # the closure sees whole frames (tuples of channel values)
        rows = [s.fn.fn(*[tuple(col[i]) for col in cols]) for i in range(n)]
        out = np.asarray(rows, dtype=np.float64).reshape(n, -1)
    return np.asfortranarray(np.asarray(out, dtype=s.dtype if s.dtype != S.I64 else np.float64).reshape(n, -1))


def lower(x, nframes_out=None, rng=None):
    x = S._assignal(x)
    lw = Lowered()
    need = {}
    if nframes_out is None:
        nf = S.nframes(x)
        nframes_out = None if (nf is None or S.isknowninf(nf)) else nf
    _demand(x, nframes_out, need)

    def fs_of(s):
        return float("nan") if s.fs is None else float(s.fs)

    def common(s, kind):
        return dict(kind=kind, dtype=_DT[s.dtype], nch=s.nch, nframes=_len_code(S.nframes(s)),
                    fs=fs_of(s))

    def host_leaf(s, data, infinite, total=None):
        """a host-materialised sub-tree as an array leaf of the frames the sink reaches; the node keeps
        the sub-tree's own length (the planner cross-checks the length algebra above it)"""
        a = S.ArraySig(data, s.fs)
        lw.keep.append(a)
        r = common(a, K.NODE_ARRAY)
        r["nframes"] = K.SO_LEN_UNCHECKED
        r.update(p0=data.ctypes.data, l0=a.n, i0=0, s0=1, s1=max(a.n, 1))
        inner = lw.add(**r)
        r2 = dict(kind=K.NODE_PAD, dtype=_DT[a.dtype], nch=a.nch, nframes=K.SO_LEN_INF, fs=fs_of(s),
                  i0=K.PAD["zero"], i1=0, children=(inner,))
        idx = lw.add(**r2)
        if not infinite:
            r3 = dict(kind=K.NODE_UNTIL, dtype=_DT[a.dtype], nch=a.nch, nframes=_len_code(total), fs=fs_of(s),
                      l0=int(total), children=(idx,))
            idx = lw.add(**r3)
        return idx

    def rec(s):
        key = id(s)
        if key in lw.memo and not (isinstance(s, S.FuncSig) and s.fn == S.RANDN):
            return lw.memo[key]
        lw.keep.append(s)
        if isinstance(s, S.ArraySig):
            if s.dtype == S.I64:
                S.error("the HIP engine lowers Float32/Float64 arrays only (SURVEY.md §8(b))")
            ptr, fst, cst, dev = _array_fields(s)
            r = common(s, K.NODE_ARRAY)
            r.update(p0=ptr, l0=s.n, i0=dev, s0=fst, s1=cst)
            if getattr(s, "virtual", None) is not None:  # a bounded resident tail of an unbounded input
                vptr, vfst, vcst, first = s.virtual
                r.update(p0=vptr, i0=1, s0=vfst, s1=vcst, l1=int(first))
            idx = lw.add(**r)
            lw.array_nodes.append((idx, s))
        elif isinstance(s, S.NumberSig):
            r = common(s, K.NODE_CONST)
            r.update(d0=float(s.val), i0=_DT[s.dtype], nch=1)
            idx = lw.add(**r)
        elif isinstance(s, S.FuncSig):
            if s.fn == S.RANDN:
                # randn leaves are host-materialised (SURVEY.md §7 hard part 7): one
                # N(0,1) draw per evaluated frame, in increasing frame order
                n, skipped = need.get(id(s), (0, 0))
                g = s.rng if s.rng is not None else (rng if rng is not None else np.random.default_rng())
                data = np.zeros((max(n, 0), 1), order="F")
                data[skipped:, 0] = g.standard_normal(max(n - skipped, 0))  # skipped frames draw nothing
                a = S.ArraySig(data, s.fs)
                lw.keep.append(a)
                r = common(a, K.NODE_ARRAY)
                r["nframes"] = K.SO_LEN_UNCHECKED
                r.update(p0=data.ctypes.data, l0=a.n, i0=0, s0=1, s1=a.n)
                inner = lw.add(**r)
                # the leaf is infinite in the reference: pad is never reached because
                # demand analysis sized the array; keep lengths consistent with Pad
                r2 = dict(kind=K.NODE_PAD, dtype=K.SO_F64, nch=1, nframes=K.SO_LEN_INF, fs=fs_of(s),
                          i0=K.PAD["zero"], i1=0, children=(inner,))
                idx = lw.add(**r2)
            elif s.fn == S.OPAQUE:
                n, _ = need.get(id(s), (0, 0))
                idx = host_leaf(s, _host_function_leaf(s, max(n, 0)), infinite=True)
            else:
                if s.fs is None:
                    S.error("Unknown frame rate: function signals need a frame rate before `sink` "
                            "(use ToFramerate)")
                r = common(s, K.NODE_FUNC)
                r.update(i0=K.FN[s.fn], i1=0 if s.omega is None else 1,
                         d0=0.0 if s.omega is None else float(s.omega), d1=s.phi, nch=1)
                idx = lw.add(**r)
        elif isinstance(s, S.CutApply):
            c = rec(s.signal)
            L = s.resolvelen()
            if L is None:
                S.error("Unknown number of frames in signal.")
            r = common(s, K.NODE_UNTIL if s.kind == "until" else K.NODE_AFTER)
            r.update(l0=int(L), children=(c,))
            idx = lw.add(**r)
        elif isinstance(s, S.PaddedSignal):
            c = rec(s.signal)
            kind, val, vec = _pad_fields(s.pad, s.nch)
            r = common(s, K.NODE_PAD)
            r.update(i0=kind, i1=1 if s.extending else 0, d0=val, children=(c,))
            if vec is not None:
                lw.keep.append(vec)
                r["p0"] = vec.ctypes.data
            idx = lw.add(**r)
        elif isinstance(s, S.AppendSignals):
            kids = tuple(rec(c) for c in s.signals)
            r = common(s, K.NODE_APPEND)
            r.update(children=kids)
            idx = lw.add(**r)
        elif isinstance(s, S.RampSignal):
            c = rec(s.signal)
            R = s.resolvelen()
            if R is None:
                S.error("Unknown number of frames in signal.")
            r = common(s, K.NODE_RAMP)
            r.update(i0=0 if s.direction == "on" else 1, i1=K.RAMPFN[s.fn], l0=int(R), children=(c,))
            idx = lw.add(**r)
        elif isinstance(s, S.MapSignal) and isinstance(s.fn, S.OpaqueFn):
            n, _ = need.get(id(s), (0, 0))
            total = S.nframes(s)
            if total is None:
                S.error("Unknown number of frames in signal.")
            idx = host_leaf(s, _host_map_leaf(s, max(n, 0), rng), infinite=S.isknowninf(total), total=total)
        elif isinstance(s, S.MapSignal):
            kids = tuple(rec(c) for c in s.signals)
            pk, pv, _ = _pad_fields(s.padding, s.nch)
            r = common(s, K.NODE_MAP)
            extra = s.extra
            if s.fn == S.TOELTYPE:
                extra = _DT[np.dtype(s.extra)]
            r.update(i0=K.MAPFN[s.fn], i1=1 if s.bychannel else 0, i2=pk, d0=pv, i3=int(extra),
                     children=kids)
            idx = lw.add(**r)
        elif isinstance(s, S.FilteredSignal):
            c = rec(s.signal)
            if isinstance(s.fn, S.ResamplerFn):
                ratio = s.fn.ratio
                h = design_resample(ratio)
                lw.keep.append(h)
                r = common(s, K.NODE_RESAMPLE)
                r.update(p0=h.ctypes.data, i2=int(h.size), i3=s.blocksize, children=(c,))
                if isinstance(ratio, tuple):
                    r.update(i0=K.RS_RATIONAL, i1=int(ratio[0]), l0=int(ratio[0]), l1=int(ratio[1]))
                else:
                    r.update(i0=K.RS_ARBITRARY, i1=32, d0=float(ratio))
            elif isinstance(s.fn, S.RawFirFn):
                h = s.fn.h
                lw.keep.append(h)
                r = common(s, K.NODE_RESAMPLE)
                r.update(p0=h.ctypes.data, i0=K.RS_FIR, i1=1, i2=int(h.size), i3=s.blocksize, l0=1, l1=1, children=(c,))
            else:
                if s.fs is None:
                    S.error("Unknown frame rate: Filt needs a frame rate before `sink`")
                if isinstance(s.fn, S.RawFilterFn):
                    sos, gain = s.fn.sos, s.fn.gain
                else:
                    sos, gain = design_iir(s.fn, s.fs)  # FilterFn(fs): designed at sink time
                sos = np.ascontiguousarray(sos, dtype=np.float64)
                lw.keep.append(sos)
                r = common(s, K.NODE_FILT_SOS)
                r.update(i0=int(sos.shape[0]), p0=sos.ctypes.data, d0=float(gain), i1=s.blocksize,
                         children=(c,))
            idx = lw.add(**r)
        elif isinstance(s, S.NormedSignal):
            c = rec(s.signal)
            r = common(s, K.NODE_NORMPOWER)
            r.update(children=(c,))
            idx = lw.add(**r)
        else:
            S.error(f"Value is not a signal: {s!r}")
        lw.memo[key] = idx
        return idx

    root = rec(x)
    return lw.finish(root)
