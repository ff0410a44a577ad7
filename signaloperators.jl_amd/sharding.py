"""Multi-GPU sharding of the sink path: one process per GPU,
`torch.distributed` (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in CPU tests).

The path shards along two independent axes, neither of which needs a data-path
collective:
  * Append children are independent sub-trees (every stateful node starts from zero
    state per child, reference src/filters.jl:204-211, src/appending.jl:59-76)
      -> contiguous blocks of children per rank, each rank writes its own time range;
  * channels are independent for every hot-path node except Normpower / ToChannels(1)
      -> contiguous channel slabs per rank.
A third axis needs no hand-off of filter state either:
  * time.  A stateful stage asked for frames [a, b) only starts a decay time (IIR) or a few periods
    (resampler) before a (the planner's warm start), so ONE long signal is cut
    into contiguous time ranges, one per rank, each evaluated as `x |> After(a) |> Until(b-a)`
    (not for Normpower, whose rms needs the whole signal on every rank).
The only exchange step is the optional final gather of the result (uneven sizes ->
all_gather of padded slabs).
"""
import numpy as np

from . import signals as S


def block_range(n, rank, world):
    """contiguous block partition of range(n): first (n % world) ranks get one extra"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_append(x, rank, world):
    """-> (sub-signal for this rank or None, first output frame, number of frames)"""
    if not isinstance(x, S.AppendSignals):
        raise S.ErrorException("shard_append needs an Append(...) root")
    kids = x.signals
    lens = [S.nframes(k) for k in kids]
    if any(n is None or S.isknowninf(n) for n in lens):
        raise S.ErrorException("Cannot shard an Append with infinite or unknown-length children")
    lo, hi = block_range(len(kids), rank, world)
    start = int(sum(lens[:lo]))
    count = int(sum(lens[lo:hi]))
    if hi <= lo:
        return None, start, 0
    sub = kids[lo] if hi - lo == 1 else S._Append(list(kids[lo:hi]))
    return sub, start, count


def shard_time(x, rank, world, align=1):
    """-> (x restricted to this rank's contiguous time range or None, first frame, frames); range
    boundaries are multiples of `align` frames"""
    from .units import frames

    x = S._assignal(x)
    n = S.nframes(x)
    if n is None or S.isknowninf(n):
        raise S.ErrorException("shard_time needs a signal of known, finite length")
    n = int(n)
    units = -(-n // align)
    lo, hi = block_range(units, rank, world)
    a, b = min(lo * align, n), min(hi * align, n)
    if b <= a:
        return None, a, 0
    sub = x if a == 0 else S.After(x, a * frames)
    return (sub if b == n and a == 0 else S.Until(sub, (b - a) * frames)), a, b - a


def shard_channels(x, rank, world):
    """channel slab [c0,c1) of a signal whose channels are independent"""
    x = S._assignal(x)
    c0, c1 = block_range(x.nch, rank, world)
    if c1 <= c0:
        return None, c0, c1
    if c1 - c0 == x.nch:
        return x, c0, c1
    parts = [S.SelectChannel(x, c + 1) for c in range(c0, c1)]
    return (parts[0] if len(parts) == 1 else S.AddChannel(*parts)), c0, c1


def _dist_info(rank, world):
    import torch.distributed as dist

    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    return rank, world


def assemble_ranges(outs, counts, nch, total):
    """device reassembly of all-gathered, padded time slabs: outs [world][nch][width] -> [nch][total]"""
    import torch

    full = torch.empty((nch, total), dtype=outs.dtype, device=outs.device)
    pos = 0
    for r, cnt in enumerate(counts):
        full[:, pos:pos + cnt] = outs[r, :, :cnt]
        pos += cnt
    return full


def assemble_channels(outs, bounds, n):
    """device reassembly of all-gathered, padded channel slabs: outs [world][wmax][n] -> [nch][n]"""
    import torch

    nch = bounds[-1][1]
    full = torch.empty((nch, n), dtype=outs.dtype, device=outs.device)
    for r, (lo, hi) in enumerate(bounds):
        full[lo:hi] = outs[r, :hi - lo]
    return full


class NativeComm:
    """The library's own exchange (include/sigops.h so_comm_*: grouped RCCL send / recv behind the C-ABI,
    what a Julia host would call).  `NativeComm.from_torch()` bootstraps the 128-byte id over an
    initialised torch.distributed group; `NativeComm.single()` is the one-rank communicator."""

    def __init__(self, id128, world, rank, device=0):
        import ctypes as C

        from . import _capi as K

        self.world, self.rank, self.device = world, rank, device
        self.handle = C.c_void_p()
        st = K.lib().so_comm_create(id128, world, rank, device, C.byref(self.handle))
        if st != 0:
            raise S.ErrorException(K.lib().so_comm_last_error().decode())

    @staticmethod
    def new_id():
        import ctypes as C

        from . import _capi as K

        buf = C.create_string_buffer(128)
        if K.lib().so_comm_unique_id(buf) != 0:
            raise S.ErrorException(K.lib().so_comm_last_error().decode())
        return buf.raw

    @classmethod
    def single(cls, device=0):
        return cls(cls.new_id(), 1, 0, device)

    @classmethod
    def from_torch(cls, device=0):
        import torch.distributed as dist

        rank, world = dist.get_rank(), dist.get_world_size()
        box = [cls.new_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls(box[0], world, rank, device)

    def allgather(self, mine, src_row_stride, full, slabs, stream=None):
        """slabs: [(rows, row_elems, dst_offset, dst_row_stride)] per rank, in elements of `full`"""
        import ctypes as C

        from . import _capi as K

        arr = (K.so_slab_t * len(slabs))(*[K.so_slab_t(*map(int, t)) for t in slabs])
        dt = K.SO_F32 if full.element_size() == 4 else K.SO_F64
        st = K.lib().so_comm_allgather(self.handle, C.c_void_p(mine.data_ptr() if mine is not None else 0), int(src_row_stride),
                                       C.c_void_p(full.data_ptr()), arr, dt, C.c_void_p(stream or 0))
        if st != 0:
            raise S.ErrorException(K.lib().so_comm_last_error().decode())

    def reduce_sum(self, buf, root=-1, stream=None):
        """so_comm_reduce_sum: the ranks' buffers (one planar [nch][pitch] tensor each) added up in place, into every
        rank's buffer (root < 0) or into `root`'s"""
        import ctypes as C

        from . import _capi as K

        rows, row_elems = (1, buf.numel()) if buf.is_contiguous() else (buf.shape[0], buf.shape[1])
        dt = K.SO_F32 if buf.element_size() == 4 else K.SO_F64
        st = K.lib().so_comm_reduce_sum(self.handle, C.c_void_p(buf.data_ptr()), int(rows), int(row_elems),
                                        int(buf.stride(0)) if rows > 1 else int(row_elems), dt, int(root), C.c_void_p(stream or 0))
        if st != 0:
            raise S.ErrorException(K.lib().so_comm_last_error().decode())

    def close(self):
        from . import _capi as K

        if self.handle:
            K.lib().so_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def shard_mix(x, rank, world):
    """Operands of a root `Mix(xs...) = OperateOn(+, xs...)` (reference src/mapsignal.jl:307-308) in contiguous blocks per
    rank -> this rank's partial sum as a signal of the WHOLE result's length (shorter operands are zero-extended, as the
    Mix itself extends them: src/mapsignal.jl:26), or None for a rank without operands."""
    from .units import frames

    x = S._assignal(x)
    if not (isinstance(x, S.MapSignal) and x.fn == S.ADD and x.bychannel and len(x.signals) >= 2):
        raise S.ErrorException("shard_mix needs a Mix(...) root")
    n = S.nframes(x)
    if n is None or S.isknowninf(n):
        raise S.ErrorException("Cannot shard a Mix of infinite or unknown length")
    lo, hi = block_range(len(x.signals), rank, world)
    if hi <= lo:
        return None
    mine = list(x.signals[lo:hi])
    sub = mine[0] if len(mine) == 1 else S._OperateOn(S.ADD, mine)
    sub = S.ToChannels(sub, x.nch)
    m = S.nframes(sub)
    if m is None or S.isknowninf(m) or int(m) > int(n):
        sub = S.Until(sub, int(n) * frames)
    elif int(m) < int(n):
        sub = S.Until(S.Pad(sub, S.zero), int(n) * frames)
    return sub


def sink_mix_sharded(x, *, rank=None, world=None, root=-1, compute=None, device=0, comm=None):
    """Evaluate Mix(operands...) with the operands sharded over the ranks: every rank sinks the sum of its block of
    operands into a device buffer of the result's shape, and ONE reduction adds the buffers up (the only exchange
    step; BASELINE.json north_star: "RCCL ... only for the final concatenate/sum").  `comm` (a NativeComm): the
    library's own so_comm_reduce_sum; otherwise torch.distributed's all_reduce / reduce.  `compute` (tests: the CPU
    oracle under gloo) switches to the NumPy path.  Returns the full result on every rank (root < 0) or on `root`
    (None elsewhere).  The reference folds the operands left to right; the reduction associates the ranks' partial
    sums as the collective does: the same values for two ranks, <= 1 ulp per addition beyond."""
    rank, world = _dist_info(rank, world)
    x = S._assignal(x)
    sub = shard_mix(x, rank, world)
    n, nch = int(S.nframes(x)), x.nch
    dt = S.float_type(x.dtype)
    if compute is not None:
        import torch
        import torch.distributed as dist

        local = np.zeros((n, nch), dtype=dt, order="F") if sub is None else np.asfortranarray(np.asarray(compute(sub), dtype=dt))
        if world == 1:
            return local
        t = torch.from_numpy(np.ascontiguousarray(local.T))
        if dist.get_backend() == "nccl":
            t = t.cuda(device)
        if root < 0:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        else:
            dist.reduce(t, dst=root, op=dist.ReduceOp.SUM)
        return np.asfortranarray(t.cpu().numpy().T) if (root < 0 or rank == root) else None
    import torch
    import torch.distributed as dist

    from .engine import sink_into

    tdt = torch.float32 if dt == S.F32 else torch.float64
    buf = torch.zeros((nch, max(n, 1)), dtype=tdt, device=f"cuda:{device}")
    if sub is not None and n > 0:
        sink_into(buf.t()[:n], sub, device=device)
    if comm is not None:
        torch.cuda.synchronize(device)
        comm.reduce_sum(buf, root, torch.cuda.current_stream(device).cuda_stream)
        torch.cuda.synchronize(device)
    elif world > 1:
        if root < 0:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        else:
            dist.reduce(buf, dst=root, op=dist.ReduceOp.SUM)
    return buf.t()[:n] if (root < 0 or rank == root) else None


def sink_append_sharded(x, *, rank=None, world=None, gather=True, compute=None, device=0, comm=None, force_gather=False):
    """Evaluate Append(children...) with children sharded over ranks.

    Default compute = the HIP engine with a DEVICE-RESIDENT result: the rank's slab is written by
    the engine straight into its slot of the exchange buffer, the slabs are all-gathered device to
    device (RCCL over xGMI; uneven shares -> slabs padded to the widest) and the planar
    [nframes x nch] result is assembled on the device -- no host hop anywhere.  Returns a
    column-major torch tensor on every rank when gather=True, else (local_slab, start).
    `compute` (tests: the CPU oracle under gloo) switches to the NumPy path.
    `comm` (a NativeComm): the exchange goes through the library's own so_comm_allgather instead -- the
    engine writes this rank's share straight into its place of the full buffer and the other shares
    arrive there too: no padded slabs, no reassembly.  `force_gather`: run the collective even with one
    rank (exercises the device branch on a single GPU)."""
    return _sink_ranges(x, shard_append, rank, world, gather, compute, device, comm, force_gather)


def sink_time_sharded(x, *, rank=None, world=None, gather=True, compute=None, device=0, align=1, comm=None, force_gather=False):
    """Evaluate ONE signal with its time axis cut into contiguous ranges, one per rank (see the module
    docstring); same result layout and gather as `sink_append_sharded`."""
    x = S._assignal(x)
    return _sink_ranges(x, lambda y, r, w: shard_time(y, r, w, align), rank, world, gather, compute, device, comm, force_gather)


def _sink_ranges(x, shard, rank, world, gather, compute, device, comm=None, force_gather=False):
    rank, world = _dist_info(rank, world)
    sub, start, count = shard(x, rank, world)
    nch = x.nch
    dt = S.float_type(x.dtype)
    counts = [shard(x, r, world)[2] for r in range(world)]
    if compute is not None:
        return _sink_ranges_host(x, sub, start, count, counts, rank, world, gather, compute, device)
    import torch
    import torch.distributed as dist

    from .engine import sink_into

    tdt = torch.float32 if dt == S.F32 else torch.float64
    total = int(S.nframes(x))
    if comm is not None and gather:
        # the library's exchange: every share lands at its place in `full`, the own one is written there
        # by the engine itself
        full = torch.empty((nch, max(total, 1)), dtype=tdt, device=f"cuda:{device}")
        if sub is not None and count > 0:
            sink_into(full.t()[start:start + count], sub, device=device)
        starts = [sum(counts[:r]) for r in range(world)]
        slabs = [(nch, counts[r], starts[r], full.stride(0)) for r in range(world)]
        torch.cuda.synchronize(device)
        comm.allgather(full[:, start:] if count > 0 else None, full.stride(0), full, slabs,
                       torch.cuda.current_stream(device).cuda_stream)
        torch.cuda.synchronize(device)
        return full.t()[:total]
    width = max(counts) if gather and (world > 1 or force_gather) else count
    slab = torch.zeros((nch, max(width, 1)), dtype=tdt, device=f"cuda:{device}")
    if sub is not None and count > 0:
        sink_into(slab.t()[:count], sub, device=device)  # result strides (1, width): written in place
    if not gather:
        return slab.t()[:count], start
    if world == 1 and not force_gather:
        return slab.t()[:count]
    outs = torch.empty((world, nch, width), dtype=tdt, device=slab.device)
    dist.all_gather_into_tensor(outs, slab)
    return assemble_ranges(outs, counts, nch, total).t()


def sink_channels_sharded(x, *, rank=None, world=None, gather=True, compute=None, device=0, comm=None, force_gather=False):
    """Channel-striped sink of a signal whose channels are independent: each rank evaluates its
    contiguous slab of channels with a device-resident result.  Planar layout makes a slab one
    contiguous block of the full result, so with equal slabs the optional all-gather writes the
    final [nframes x nch] buffer directly (uneven: padded slabs, then assembled on the device).
    Returns the full column-major tensor when gather=True, else (local_slab, c0, c1)."""
    rank, world = _dist_info(rank, world)
    x = S._assignal(x)
    sub, c0, c1 = shard_channels(x, rank, world)
    n = int(S.nframes(x))
    dt = S.float_type(x.dtype)
    if compute is not None:
        local = np.empty((n, 0), dtype=dt) if sub is None else np.asarray(compute(sub))
        if not gather or world == 1:
            return (local, c0, c1) if not gather else local
        return _gather_channels_host(local, x.nch, n, dt, world, device)
    import torch
    import torch.distributed as dist

    from .engine import sink_into

    tdt = torch.float32 if dt == S.F32 else torch.float64
    bounds = [block_range(x.nch, r, world) for r in range(world)]
    wmax = max(hi - lo for lo, hi in bounds)
    even = all(hi - lo == wmax for lo, hi in bounds)
    if comm is not None and gather:  # the library's exchange: a channel slab is one contiguous run of the planar result
        full = torch.empty((x.nch, n), dtype=tdt, device=f"cuda:{device}")
        if sub is not None:
            sink_into(full[c0:c1].t(), sub, device=device)
        slabs = [(1, (hi - lo) * n, lo * n, 0) for lo, hi in bounds]
        torch.cuda.synchronize(device)
        comm.allgather(full[c0:] if c1 > c0 else None, 0, full, slabs, torch.cuda.current_stream(device).cuda_stream)
        torch.cuda.synchronize(device)
        return full.t()
    if gather and (world > 1 or force_gather) and even:
        full = torch.empty((x.nch, n), dtype=tdt, device=f"cuda:{device}")
        mine = full[c0:c1]  # this rank's slab of the final buffer
        sink_into(mine.t(), sub, device=device)
        dist.all_gather_into_tensor(full.view(world, wmax, n), mine.contiguous())
        return full.t()
    slab = torch.zeros((max(wmax if gather else c1 - c0, 1), n), dtype=tdt, device=f"cuda:{device}")
    if sub is not None:
        sink_into(slab[:c1 - c0].t(), sub, device=device)
    if not gather:
        return slab[:c1 - c0].t(), c0, c1
    if world == 1 and not force_gather:
        return slab[:c1 - c0].t()
    outs = torch.empty((world, wmax, n), dtype=tdt, device=slab.device)
    dist.all_gather_into_tensor(outs, slab)
    return assemble_channels(outs, bounds, n).t()


def _gather_channels_host(local, nch, n, dt, world, device):
    import torch
    import torch.distributed as dist

    bounds = [block_range(nch, r, world) for r in range(world)]
    wmax = max(hi - lo for lo, hi in bounds)
    pad = np.zeros((wmax, n), dtype=dt)
    pad[:local.shape[1]] = local.T
    t = torch.from_numpy(pad)
    if dist.get_backend() == "nccl":
        t = t.cuda(device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    full = np.empty((n, nch), dtype=dt, order="F")
    for r, (lo, hi) in enumerate(bounds):
        full[:, lo:hi] = outs[r][:hi - lo].cpu().numpy().T
    return full


def _sink_ranges_host(x, sub, start, count, counts, rank, world, gather, compute, device):
    """NumPy path (a caller-supplied `compute`, e.g. the CPU oracle in the gloo tests)"""
    import torch.distributed as dist

    nch = x.nch
    dt = S.float_type(x.dtype)
    local = np.empty((0, nch), dtype=dt) if sub is None else np.asarray(compute(sub))
    if not gather or world == 1:
        return (local, start) if not gather else local
    import torch

    total = int(S.nframes(x))
    width = max(counts)
    pad = np.zeros((width, nch), dtype=dt)
    pad[:count] = local
    t = torch.from_numpy(pad)
    backend = dist.get_backend()
    if backend == "nccl":
        t = t.cuda(device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    full = np.empty((total, nch), dtype=dt, order="F")
    pos = 0
    for r in range(world):
        full[pos:pos + counts[r]] = outs[r][:counts[r]].cpu().numpy()
        pos += counts[r]
    return full
