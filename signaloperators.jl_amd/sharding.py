"""Multi-GPU sharding of the sink path: one process per GPU,
`torch.distributed` (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in CPU tests).

The path shards along two independent axes, neither of which needs a data-path
collective:
  * Append children are independent sub-trees (every stateful node starts from zero
    state per child, reference src/filters.jl:204-211, src/appending.jl:59-76)
      -> contiguous blocks of children per rank, each rank writes its own time range;
  * channels are independent for every hot-path node except Normpower / ToChannels(1)
      -> contiguous channel slabs per rank.
The only exchange step is the optional final gather of the result (uneven sizes ->
all_gather of padded slabs).
"""
import numpy as np

from . import signals as S
from .engine import sink as _engine_sink


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


def sink_append_sharded(x, *, rank=None, world=None, gather=True, compute=None, device=0):
    """Evaluate Append(children...) with children sharded over ranks.  Returns the full
    [nframes x nch] array on every rank when gather=True, else (local_slab, start)."""
    import torch.distributed as dist

    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    compute = compute or (lambda sig: _engine_sink(sig, np.ndarray, device=device))
    sub, start, count = shard_append(x, rank, world)
    nch = x.nch
    dt = S.float_type(x.dtype)
    local = np.empty((0, nch), dtype=dt) if sub is None else np.asarray(compute(sub))
    if not gather or world == 1:
        return (local, start) if not gather else local
    import torch

    total = int(S.nframes(x))
    counts = [shard_append(x, r, world)[2] for r in range(world)]
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
