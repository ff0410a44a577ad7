"""The exchange step of the sharded sink on the device, exercised on ONE GPU
(the device `all_gather_into_tensor` branches of sharding.py had never executed):

* the device reassembly of padded, uneven slabs, fed with slabs the engine really produced (every
  rank's share evaluated on this GPU, the gathered tensor put together by hand);
* `torch.distributed` with backend nccl (= RCCL) and world size 1, `force_gather=True`: the collective
  call itself and the code after it;
* the library's own exchange behind the C-ABI (`so_comm_*`, grouped RCCL send / recv): a one-rank
  communicator, the own share routed through RCCL as a send to / receive from the same rank
  (SIGOPS_COMM_SELF_EXCHANGE), for time ranges and for channel slabs.
Every result is compared with the unsharded sink (bit-equal for Append shards and channel slabs)."""
import os

import numpy as np
import pytest

import sigops_amd as so
from sigops_amd import sharding as sh
from oracle_bridge import oracle_semantics, oracle_sink, relerr

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _scenes(rng, k=5, nch=2):
    kids = []
    for i in range(k):
        n = 9000 + 1111 * i
        x = np.asfortranarray(rng.standard_normal((n, nch)))
        kids.append(so.Signal(x, 44.1 * so.kHz) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.Ramp(10 * so.ms))
    return so.Append(*kids)


@pytest.mark.parametrize("world", [2, 3, 4])
def test_device_reassembly_of_uneven_time_slabs(world):
    tree = _scenes(np.random.default_rng(50))
    whole = so.sink(tree, so.Array)
    total, nch = whole.shape
    counts = [sh.shard_append(tree, r, world)[2] for r in range(world)]
    width = max(counts)
    outs = torch.zeros((world, nch, width), dtype=torch.float64, device="cuda")
    for r in range(world):  # what all_gather_into_tensor would deliver: every rank's padded slab
        slab, start = sh.sink_append_sharded(tree, rank=r, world=world, gather=False)
        assert slab.is_cuda and slab.shape == (counts[r], nch) and start == sum(counts[:r])
        outs[r, :, :counts[r]] = slab.t()
    full = sh.assemble_ranges(outs, counts, nch, total).t()
    assert np.array_equal(full.cpu().numpy(), whole)


@pytest.mark.parametrize("world,nch", [(2, 8), (3, 8), (4, 6)])
def test_device_reassembly_of_channel_slabs(world, nch):
    x = np.asfortranarray(np.random.default_rng(51).standard_normal((40_000, nch)))
    tree = so.Signal(x, 44.1 * so.kHz) | so.Filt(so.Lowpass, 4 * so.kHz) | so.ToFramerate(16 * so.kHz)
    whole = so.sink(tree, so.Array)
    n = whole.shape[0]
    bounds = [sh.block_range(nch, r, world) for r in range(world)]
    wmax = max(hi - lo for lo, hi in bounds)
    outs = torch.zeros((world, wmax, n), dtype=torch.float64, device="cuda")
    for r, (lo, hi) in enumerate(bounds):
        slab, c0, c1 = sh.sink_channels_sharded(tree, rank=r, world=world, gather=False)
        assert (c0, c1) == (lo, hi) and slab.shape == (n, hi - lo)
        outs[r, :hi - lo] = slab.t()
    full = sh.assemble_channels(outs, bounds, n).t()
    assert np.array_equal(full.cpu().numpy(), whole)


def test_torch_distributed_nccl_world_1_runs_the_collective():
    import torch.distributed as dist

    if dist.is_initialized():
        pytest.skip("a process group is already initialised in this process")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        tree = _scenes(np.random.default_rng(52), k=3)
        whole = so.sink(tree, so.Array)
        got = sh.sink_append_sharded(tree, force_gather=True)  # all_gather_into_tensor + assemble_ranges
        assert got.is_cuda and np.array_equal(got.cpu().numpy(), whole)
        got_t = sh.sink_time_sharded(tree, force_gather=True)
        assert np.array_equal(got_t.cpu().numpy(), whole)
        x = np.asfortranarray(np.random.default_rng(53).standard_normal((30_000, 4)))
        ct = so.Signal(x, 44.1 * so.kHz) | so.Filt(so.Lowpass, 4 * so.kHz)
        got_c = sh.sink_channels_sharded(ct, force_gather=True)
        assert np.array_equal(got_c.cpu().numpy(), so.sink(ct, so.Array))
        # ... and the library's own exchange bootstrapped over the same process group
        comm = sh.NativeComm.from_torch()
        got_n = sh.sink_append_sharded(tree, comm=comm)
        assert np.array_equal(got_n.cpu().numpy(), whole)
        comm.close()
    finally:
        dist.destroy_process_group()


def test_native_exchange_through_rccl_on_one_rank(monkeypatch):
    """so_comm_create / so_comm_allgather / so_comm_destroy (include/sigops.h) with real RCCL calls: the
    own share is sent to and received from this very rank inside ncclGroupStart / ncclGroupEnd"""
    monkeypatch.setenv("SIGOPS_COMM_SELF_EXCHANGE", "1")
    comm = sh.NativeComm.single()
    rng = np.random.default_rng(54)
    nch, n, start = 3, 5000, 700
    for dt in (torch.float64, torch.float32):
        mine = torch.from_numpy(rng.standard_normal((nch, n))).to(dt).cuda()          # a time range of all channels
        full = torch.full((nch, 9000), float("nan"), dtype=dt, device="cuda")
        comm.allgather(mine, mine.stride(0), full, [(nch, n, start, full.stride(0))], torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(full[:, start:start + n], mine)
        assert bool(torch.isnan(full[:, :start]).all()) and bool(torch.isnan(full[:, start + n:]).all())
        slab = torch.from_numpy(rng.standard_normal((2, 9000))).to(dt).cuda()          # a slab of two channels
        full2 = torch.full((nch, 9000), float("nan"), dtype=dt, device="cuda")
        comm.allgather(slab, 0, full2, [(1, 2 * 9000, 1 * 9000, 0)], torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(full2[1:3], slab) and bool(torch.isnan(full2[0]).all())
    # the sharded sinks on top of it (one rank: the engine writes the share in place, nothing travels)
    monkeypatch.delenv("SIGOPS_COMM_SELF_EXCHANGE")
    tree = _scenes(rng, k=3)
    assert np.array_equal(sh.sink_append_sharded(tree, rank=0, world=1, comm=comm).cpu().numpy(), so.sink(tree, so.Array))
    comm.close()


@pytest.mark.parametrize("world", [1, 2, 3])
def test_mix_operands_on_ranks_add_up_to_the_unsharded_sink(world):
    """sink_mix_sharded on ONE GPU: every rank's partial sum evaluated by the engine, the partial sums added by hand where
    there are several (what the reduction does), and for world = 1 through the library's own so_comm_reduce_sum on a
    one-rank RCCL communicator and through torch.distributed's all_reduce"""
    rng = np.random.default_rng(60)
    ops = []
    for k in range(3):
        n = 30000 + 4000 * k
        ops.append(so.Signal(np.asfortranarray(rng.standard_normal((n, 2))), 44.1 * so.kHz)
                   | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.ToFramerate(48 * so.kHz))
    tree = so.Mix(*ops)
    whole = so.sink(tree, so.Array)
    if world == 1:
        comm = sh.NativeComm.single()
        try:
            got = sh.sink_mix_sharded(tree, rank=0, world=1, comm=comm)
            assert got.is_cuda and np.array_equal(got.cpu().numpy(), whole)
            only = sh.sink_mix_sharded(tree, rank=0, world=1, comm=comm, root=0)
            assert np.array_equal(only.cpu().numpy(), whole)
        finally:
            comm.close()
        return
    acc = None
    for r in range(world):
        sub = sh.shard_mix(tree, r, world)
        part = np.zeros_like(whole) if sub is None else so.sink(sub, so.Array)
        assert part.shape == whole.shape
        acc = part if acc is None else acc + part
    assert relerr(acc, whole) < 1e-15
