"""`sink` / `sink!` for the HIP engine: the Python analogue of the Julia method
`sink!(result::HIPSink, x, ::IsSignal)` described in INTEGRATION.md.  Mirrors
reference src/sink.jl:28-168 for type resolution, length checks and ToChannels;
the block loop (src/sink.jl:225-260) is replaced by so_plan_create/so_plan_execute.
"""
import ctypes as C

import numpy as np

from . import _capi as K
from . import signals as S
from .lowering import lower, _DT

Array = np.ndarray


def process_sink_params(x):  # src/sink.jl:94-99
    x = S._assignal(x)
    n = S.nframes(x)
    if n is None:
        S.error("Unknown number of frames in signal.")
    if S.isknowninf(n):
        S.error("Cannot store infinite signal.")
    return x


def _timeslice(x):
    """DataCut path: sink of Until/After over raw arrays is an aliasing view
    (reference src/sink.jl:65-69, src/cutting.jl:193-214)"""
    if isinstance(x, S.ArraySig):
        return x.data
    d = _timeslice(x.signal)
    n = d.shape[0]
    L = x.resolvelen()
    if x.kind == "until":
        return d[: min(max(L, 0), n)]
    return d[min(max(L, 0), n):]


def _is_datacut(x):
    if isinstance(x, S.ArraySig):
        return True
    return isinstance(x, S.CutApply) and x.evaltrait == "data" and _is_datacut(x.signal)


def _root(x):
    """root(x) with mergeroot's priorities (reference src/sink.jl:33-50, src/mapsignal.jl:60,
    src/appending.jl:17, src/wrapping.jl:19): -> (priority, leaf) where a signal-typed container
    (SampleBuf, AxisArray, ...) has priority 2, a plain array 1, anything else 0; the first of
    equal priorities wins."""
    if isinstance(x, S.ArraySig):
        if x.container is not None:
            return 2, x
        if S._is_torch(x.data):
            return 0, x
        return (1 if x.fs is None else 0), x  # an (array, fs) tuple is not an AbstractArray
    best = None
    for c in getattr(x, "children", ()):  # mergeroot: the first of equal priorities wins
        r = _root(c)
        if best is None or r[0] > best[0]:
            best = r
    return best or (0, None)


def _refineroot(x):
    """the result type `sink(x)` picks when none is given (src/sink.jl:30-37)"""
    pr, leaf = _root(x)
    if pr == 2:
        return leaf.container
    if pr == 1:
        return Array
    return tuple


class Plan:
    """RAII wrapper of so_plan_t"""

    def __init__(self, x, result_shape, result_dtype, strides, is_device, device=0, rng=None):
        self.lowered = lower(x, nframes_out=result_shape[0], rng=rng)
        self.desc = K.so_out_desc_t(dtype=_DT[np.dtype(result_dtype)], nch=result_shape[1],
                                    nframes=result_shape[0], frame_stride=strides[0],
                                    chan_stride=strides[1], is_device=1 if is_device else 0)
        self.handle = C.c_void_p()
        L = K.lib()
        st = L.so_plan_create(self.lowered.nodes, self.lowered.n, self.lowered.root,
                              C.byref(self.desc), device, C.byref(self.handle))
        if st != 0:
            raise S.ErrorException(K.last_error())

    def execute(self, out_ptr, stream=None):
        st = K.lib().so_plan_execute(self.handle, C.c_void_p(out_ptr), C.c_void_p(stream or 0))
        if st != 0:
            raise S.ErrorException(K.last_error())
        # (an execute into a device result returns before its kernels have run: what they report is still to come)
        self._unchecked = (stream or 0) if self.desc.is_device else None

    def check(self, stream=None):
        """so_plan_check: wait for what the plan has launched and raise what only shows once the kernels have run (a
        kernel that gave up on a wait between its waves: the result is invalid).  `sink!` returns a complete result
        (reference src/sink.jl:225-241): `sink_into` calls this behind an execute into a device result."""
        self._unchecked = None
        st = K.lib().so_plan_check(self.handle, C.c_void_p(stream or 0))
        if st != 0:
            raise S.ErrorException(K.last_error())

    def set_array(self, k, data):
        """so_plan_set_array: point the k-th array leaf of the tree (depth-first order) at new data
        of the same shape, dtype and residency without re-planning."""
        idx, sig = self.lowered.array_nodes[k]
        if hasattr(data, "data_ptr"):
            ptr = data.data_ptr()
        else:
            ptr = data.ctypes.data
        self._keep_arrays = getattr(self, "_keep_arrays", {})
        self._keep_arrays[k] = data
        st = K.lib().so_plan_set_array(self.handle, idx, C.c_void_p(ptr))
        if st != 0:
            raise S.ErrorException(K.last_error())

    def set_profiling(self, on=True):
        """so_plan_set_profiling: False/0 off, True/1 per execute (synchronising), 2 deferred (events of
        every execute kept; `steps()` after a synchronize reports the mean)"""
        K.lib().so_plan_set_profiling(self.handle, int(on))

    def stats(self):
        s = K.so_stats_t()
        K.lib().so_plan_stats(self.handle, C.byref(s))
        return {f[0]: (getattr(s, f[0]).decode() if f[0] == "dominant_kernel" else getattr(s, f[0]))
                for f in K.so_stats_t._fields_}

    def counters(self):
        """so_plan_counter: how the executes so far were issued (graph replays / captures / direct)"""
        L = K.lib()
        return {"graph_replays": L.so_plan_counter(self.handle, 0), "graph_captures": L.so_plan_counter(self.handle, 1),
                "direct_executes": L.so_plan_counter(self.handle, 2), "fused_mfmas_per_block": L.so_plan_counter(self.handle, 3)}

    def steps(self):
        """per-step statistics of the last (profiled) execute: so_plan_step_info"""
        out = []
        info = K.so_step_info_t()
        n = K.lib().so_plan_step_info(self.handle, 0, C.byref(info))
        for i in range(n):
            K.lib().so_plan_step_info(self.handle, i, C.byref(info))
            out.append({"name": info.name.decode(), "algorithmic_bytes": info.algorithmic_bytes,
                        "ms": info.ms, "launches": info.launches})
        return out

    def close(self):
        if self.handle:
            err = None
            if getattr(self, "_unchecked", None) is not None:  # a device result nobody has checked: do not lose its failure
                if K.lib().so_plan_check(self.handle, C.c_void_p(self._unchecked)) != 0:
                    err = K.last_error()
                self._unchecked = None
            K.lib().so_plan_destroy(self.handle)
            self.handle = C.c_void_p()
            if err is not None:
                raise S.ErrorException(err)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _result_fields(result):
    if S._is_torch(result):
        shape = tuple(result.shape)
        st = result.stride()
        if len(shape) == 1:
            shape, st = (shape[0], 1), (st[0], 0)
        dt = {"torch.float32": S.F32, "torch.float64": S.F64}[str(result.dtype)]
        return shape, dt, st, result.is_cuda, result.data_ptr()
    shape = result.shape
    item = result.itemsize
    if result.ndim == 1:
        return (shape[0], 1), result.dtype, (result.strides[0] // item or 1, 0), False, result.ctypes.data
    return shape, result.dtype, (result.strides[0] // item or 1, result.strides[1] // item), False, \
        result.ctypes.data


def sink_into(result, x, *, device=0, rng=None, stream=None):
    """sink!(result,x): write size(result,1) frames of x (reference src/sink.jl:154-168)"""
    x = S._assignal(x)
    shape, dt, strides, is_dev, ptr = _result_fields(result)
    n = S.nframes(x)
    if n is not None and not S.isknowninf(n) and n < shape[0]:
        S.error(f"Signal is too short to fill buffer of length {shape[0]}.")
    x = S.ToChannels(x, shape[1])
    plan = Plan(x, shape, dt, strides, is_dev, device=device, rng=rng)
    try:
        plan.execute(ptr, stream)
        if is_dev:
            plan.check(stream)  # (a complete result or an exception, as the reference's sink!)
    finally:
        plan.close()
    return result


def sink(x, to=None, *, device=0, rng=None):
    """sink(x[,to]) (reference src/sink.jl:28-37,64-92).  `to` may be None (type from
    the tree's root data), `Array`/np.ndarray, `tuple`, or "torch" (device-resident
    torch tensor, column-major)."""
    if to is None and (isinstance(x, type) or x is tuple or (isinstance(x, str) and x == "torch")):
        to_ = x  # sink(to::Type) = x -> sink(x,to), reference src/sink.jl:29
        return S.Curried(lambda y: sink(y, to_, device=device, rng=rng))
    x = process_sink_params(x)
    if to is None:
        to = _refineroot(x)
    if _is_datacut(x) and to is Array:
        # aliasing view, reference src/sink.jl:65-69: only when the sink type equals the
        # type of the underlying data (a Tuple sink of a view is copied through sink!)
        return _timeslice(x)
    n = S.nframes(x)
    dt = x.dtype
    if dt == S.I64:
        S.error("the HIP engine sinks Float32/Float64 signals only (SURVEY.md §8(b)); "
                "integer signals take the stock CPU sink")
    if to == "torch":
        import torch

        tdt = torch.float32 if dt == S.F32 else torch.float64
        res = torch.empty((x.nch, n), dtype=tdt, device=f"cuda:{device}").t()
        sink_into(res, x, device=device, rng=rng)
        return res, x.fs
    res = np.empty((n, x.nch), dtype=dt, order="F")  # initsink src/sink.jl:115-119
    sink_into(res, x, device=device, rng=rng)
    if isinstance(to, type) and hasattr(to, "initsink"):  # initsink(x, ::Type{<:SampleBuf}) etc.
        if x.fs is None:
            S.error("Unknown frame rate: array-type sinks carry a frame rate")
        return to.initsink(res, x.fs)
    return res if to is Array else (res, x.fs)


def stream(x, blocksize, to=None, *, device=0):
    """Iterate over the frames of `x` in blocks of `blocksize` frames (the last one may be shorter;
    an infinite signal never stops): block k is `sink(x |> After(k*blocksize frames) |>
    Until(blocksize frames), to)`, which is what the reference's own sink loop -- `sink!(buffer, x,
    block)` with the block it returned last time, src/sink.jl:227-241 -- writes into successive
    buffers.  Results do not depend on `blocksize` (the reference's `blocksize` contract,
    src/filters.jl:3).  Stateful stages do not start over for every block: a filter starts a decay
    time before the block and a resampler a few periods before it (warm start), and the
    resampler's DSP.jl phase accumulator resumes from the previous block's end (a rate without a period -- non-integer
    frame rates, x pi -- starts exactly at the block: its stage stages taps + 2 input frames before the block's first
    output and lists the accumulator's deviations from the nearest checkpoint of an earlier replay).  Host arrays of the tree are uploaded once.  `Normpower` needs its whole child for every block and `randn` leaves
    draw new numbers for every block: neither is meant for streaming."""
    from . import lowering as LW
    from .units import frames

    x = S._assignal(x)
    blocksize = int(blocksize)
    if blocksize <= 0:
        S.error("stream: blocksize must be positive")
    n = S.nframes(x)
    if n is None:
        S.error("Unknown number of frames in signal.")
    finite = not S.isknowninf(n)
    if to is None:
        to = _refineroot(x)
    cache = {"device": f"cuda:{device}"}
    pos = 0
    while not finite or pos < n:
        m = blocksize if not finite else min(blocksize, n - pos)
        blk = S.Until(S.After(x, pos * frames) if pos else x, m * frames)
        prev, LW._device_cache = LW._device_cache, cache
        try:
            res = sink(blk, to, device=device)
        finally:
            LW._device_cache = prev
        yield res
        pos += m


class BlockStream:
    """Bounded-memory streaming of an UNBOUNDED input: `push(block)` takes the next frames of the input
    signal and returns the output frames that have become final; `finish()` returns the rest.

        bs = so.BlockStream(lambda x: x | so.Filt(so.Lowpass, 3*so.kHz) | so.ToFramerate(48*so.kHz),
                            fs=44.1*so.kHz, nch=2)
        for block in source: out = bs.push(block)      # torch tensors [frames x nch] on the device
        tail = bs.finish()

    `pipeline` builds the operator tree over the input signal (called once per block with a signal
    that covers everything received so far).  Only the last `history` input frames stay in device
    memory: the tree's array leaf is *virtual* -- its address is where frame 0 would be, its node
    says which frames are resident (include/sigops.h, ARRAY l1) -- every block is the plan of
    `tree |> After(emitted) |> Until(final - emitted)`, stateful stages start from warm starts a decay
    time before the block, and the planner refuses to read a frame that is gone
    ("raise the stream's history").  An output frame is final when every input frame it depends on
    has arrived (the lowering's demand analysis: the newest input of the last output of every
    resampler, `Filt` one to one).  Concatenated, the outputs are the sink of the pipeline over the
    whole input.  Not streamable, and refused with an error at the first push (`_streamable`): `Normpower`,
    ramps at the end of the signal (`RampOff`, `Ramp`, `FadeTo`), pads that index the end of the input --
    anything whose early outputs depend on the input's total length."""

    def __init__(self, pipeline, fs, nch=1, dtype=np.float64, history=1 << 16, device=0):
        import torch

        from .units import inHz

        self.pipeline = pipeline
        self.fs = inHz(fs)
        self.nch = int(nch)
        self.dtype = np.dtype(dtype)
        self.tdt = torch.float32 if self.dtype == np.float32 else torch.float64
        self.history = int(history)
        self.device = device
        self.dev = f"cuda:{device}"
        self.cap = 0
        self.buf = None  # [nch x cap] planar tail of the input
        self.start = 0  # absolute frame stored at buf[:, 0]
        self.received = 0
        self.emitted = 0
        self.closed = False

    # -- resident tail ---------------------------------------------------------------------------
    def _append(self, block):
        import torch

        if hasattr(block, "data_ptr"):
            b = block.to(self.dev, self.tdt)
        else:
            b = torch.from_numpy(np.ascontiguousarray(np.asarray(block, dtype=self.dtype))).to(self.dev)
        if b.dim() == 1:
            b = b[:, None]
        if b.shape[1] != self.nch:
            S.error(f"BlockStream: blocks must have {self.nch} channel(s)")
        m = int(b.shape[0])
        have = self.received - self.start
        if self.buf is None or have + m > self.cap:
            keep = min(have, self.history)
            cap = max(self.cap, 2 * (self.history + m))
            nb = torch.empty((self.nch, cap), dtype=self.tdt, device=self.dev)
            if keep:
                nb[:, :keep] = self.buf[:, have - keep:have]
            self.buf, self.cap = nb, cap
            self.start = self.received - keep
            have = keep
        self.buf[:, have:have + m] = b.t()
        self.received += m

    def _tree(self):
        esz = self.dtype.itemsize
        leaf = S.ArraySig(np.empty((0, self.nch), dtype=self.dtype), self.fs)
        leaf.n = self.received  # (no host data: the node is filled in from `virtual`)
        guard = 64 if self.start > 0 else 0  # (vector loads may touch the aligned pair below a range)
        leaf.virtual = (self.buf.data_ptr() - self.start * esz, 1, self.cap, self.start + guard)
        return leaf, self.pipeline(leaf)

    def _final_frames(self, leaf, tree, total):
        """outputs whose inputs have all arrived: the largest n with demand(n) <= received"""
        from .lowering import _demand

        def needs(n):
            out = {}
            _demand(tree, n, out)
            return out.get(id(leaf), (0, 0))[0]

        lo, hi = self.emitted, total  # needs(lo) <= received
        while lo < hi:
            mid = (lo + hi + 1) // 2
            if needs(mid) <= self.received - (0 if self.closed else 1):
                lo = mid
            else:
                hi = mid - 1
        return lo

    def _emit(self):
        import torch

        from .units import frames

        leaf, tree = self._tree()
        total = S.nframes(tree)
        if total is None or S.isknowninf(total):
            S.error("BlockStream: the pipeline must have as many frames as its input decides")
        _streamable(tree)
        upto = total if self.closed else self._final_frames(leaf, tree, int(total))
        m = upto - self.emitted
        nch_out = tree.nch
        tdt = torch.float32 if tree.dtype == S.F32 else torch.float64
        if m <= 0:
            return torch.empty((0, nch_out), dtype=tdt, device=self.dev)
        blk = S.Until(S.After(tree, self.emitted * frames) if self.emitted else tree, m * frames)
        res = torch.empty((nch_out, m), dtype=tdt, device=self.dev).t()
        sink_into(res, blk, device=self.device)
        self.emitted = upto
        return res

    def push(self, block):
        if self.closed:
            S.error("BlockStream: finished")
        self._append(block)
        return self._emit()

    def finish(self):
        self.closed = True
        if self.buf is None:
            import torch

            return torch.empty((0, self.nch), dtype=self.tdt, device=self.dev)
        return self._emit()


def _streamable(x):
    """Refuse what `BlockStream` cannot stream: nodes whose early outputs depend on the input's TOTAL
    length, which a stream does not know -- each push would silently apply them to the input received
    so far.  `Normpower` divides by the rms of everything (reference src/filters.jl:296-309);
    a ramp off / `FadeTo` is anchored at the end (src/ramps.jl:65-72); `lastframe`, `cycle` and `mirror`
    pads index from the end of the signal (src/padding.jl:132-148)."""
    if isinstance(x, S.NormedSignal):
        S.error("BlockStream: Normpower needs the whole signal (its rms); not streamable")
    if isinstance(x, S.RampSignal) and x.direction == "off":
        S.error("BlockStream: a ramp at the END of the signal (RampOff, Ramp, FadeTo) is anchored at a length the "
                "stream does not know yet; not streamable")
    if isinstance(x, S.PaddedSignal) and any(x.pad is p for p in (S.lastframe, S.cycle, S.mirror)):  # (identity: a pad may be an ndarray)
        S.error("BlockStream: lastframe / cycle / mirror padding indexes the end of the input; not streamable")
    for c in getattr(x, "children", ()) or ():
        _streamable(c)


def filt(b, a, x, si=None, *, device=0):
    """DSP.filt(b, a, x::AbstractSignal[, si]) (reference src/filters.jl:68-79: `sink(x, Array)`, then DSP.jl's
    direct-form `filt!`).  Here the filter is a `Filt` node of the signal's own plan, whatever its order: FIR for a
    scalar `a`, else second-order sections factored from the two polynomials (`so_tf_to_sos`).  An initial state `si`
    (max(|a|, |b|) - 1 entries, per channel if a matrix) enters by linearity: the direct form's zero-input response
    from `si` is computed until it has decayed (`so_tf_zero_input`) and mixed in on the device.  Only a filter whose
    polynomials are too ill-conditioned to factor (the gate of `signals.TF_RESIDUAL_MAX`: the cascade would no longer
    be the reference's filter) follows the reference literally -- `sink(x, Array)` on the engine, then the direct-form
    recurrence on the host array."""
    from .units import frames

    x = S._assignal(x)

    def ftype(v):  # (eltype of a coefficient argument: Julia literals are Float64 / Int)
        dt = getattr(v, "dtype", None)
        return dt if dt is not None and np.issubdtype(dt, np.floating) else np.dtype(np.float64)

    rt = np.result_type(ftype(b), ftype(a), S.float_type(x.dtype), *([ftype(si)] if si is not None else []))
    b = np.atleast_1d(np.asarray(b, dtype=np.float64))
    a = np.atleast_1d(np.asarray(a, dtype=np.float64))
    if len(a) > 1 and max(len(a), len(b)) > 3 and not S.tf_to_sos(b, a)[2] <= S.TF_RESIDUAL_MAX:
        return _filt_direct_host(b, a, x, si, rt, device)
    n, nch = S.nframes(x), x.nch
    # (`filt!` runs in R = promote_type(coefficients, samples) over the sunk samples: a Float32 signal under Float64
    #  coefficients is filtered in Float64, src/filters.jl:73-76)
    y = S.Filt(x if S.float_type(x.dtype) == rt else S.ToEltype(x, rt), S.PolynomialRatio(b, a))
    if n is None or S.isknowninf(n):
        S.error("filt: the signal must have a finite length")
    if si is not None:
        zi = np.asarray(si, dtype=np.float64)
        if zi.ndim == 1:
            zi = np.repeat(zi[:, None], nch, axis=1)
        if zi.ndim != 2 or zi.shape[1] != nch:
            S.error("filt: the initial state must be a vector or have one column per channel")
        cols = [S.tf_zero_input(b, a, zi[:, c], n) for c in range(nch)]
        m = max([len(c) for c in cols] + [1])
        zir = np.zeros((min(m, n), nch), order="F")
        for c, col in enumerate(cols):
            zir[: min(len(col), n), c] = col[:n]
        if zir.shape[0] and np.any(zir):
            y = S.Mix(y, S.Signal(zir, x.fs)) | S.Until(n * frames)
    out = np.empty((n, nch), dtype=rt, order="F")
    sink_into(out, y, device=device)
    return _like_root(out, x)


def _filt_direct_host(b, a, x, si, rt, device):
    """the reference's own sequence for a filter the cascade form cannot represent: the engine sinks the signal, the
    direct-form recurrence runs over the host array (reference src/filters.jl:74-76)"""
    from scipy import signal as sps

    data = np.asarray(sink(x, Array, device=device), dtype=rt)
    if si is None:
        y = sps.lfilter(b, a, data, axis=0)
    else:
        zi = np.asarray(si, dtype=np.float64)
        if zi.ndim == 1:
            zi = np.repeat(zi[:, None], data.shape[1], axis=1)
        y, _ = sps.lfilter(b, a, data, axis=0, zi=zi)
    return _like_root(np.asfortranarray(y, dtype=rt), x)


def filt_into(data, b, a, x, si=None, *, device=0):
    """DSP.filt!(data, b, a, x::AbstractSignal[, si]) (reference src/filters.jl:81-87)"""
    y = filt(b, a, x, si, device=device)
    y = np.asarray(y[0] if isinstance(y, tuple) else y)
    np.copyto(np.asarray(data), y.reshape(np.asarray(data).shape))
    return data


def _like_root(y, x):
    """initsink(ToEltype(x,R), refineroot(root(x))): the container type of the signal's root data"""
    to = _refineroot(x)
    if isinstance(to, type) and hasattr(to, "initsink") and x.fs is not None:
        return to.initsink(y, x.fs)
    return y if to is Array or x.fs is None else (y, x.fs)


# eager lower-case forms (reference: mix(xs...) = sink(Mix(xs...)) etc.)
def _eager(op):
    def f(*a, **k):
        return sink(op(*a, **k))

    return f
