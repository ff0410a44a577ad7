"""Multi-GPU workloads of bench.py (`--workload config4|config5|ns_time`), BASELINE.json configs[3] and [4]
and the headline pipeline cut along time.

config4  Append of 64 independent 60 s scenes (Mix(sin, noise[2 646 000 x 2]) |> Filt(Bandstop) |> Ramp),
         44.1 kHz, sharded in contiguous blocks of scenes over the ranks (reference: Append children
         are independent sub-trees, src/appending.jl:59-76).  Every rank's engine writes its time
         range straight into its slot of the exchange buffer; the slabs are all-gathered device to
         device over RCCL/xGMI.  Total work is fixed (strong scaling).  Reported: whole-job frames/s
         WITH the gather (`value`) and compute-only.
config5  x[10 000 000 x 128 per GPU] |> Filt(Lowpass 4 kHz) |> ToFramerate(16 kHz): each rank owns a
         128-channel slab of the 1024-channel signal (channels are independent, planar layout keeps a
         slab contiguous); no exchange step (weak scaling: N x 128 channels).
"""
import json
import time


def _sink_traffic(workload, world):
    """PMC bytes per execute of the whole sink on ONE GPU (the PMC passes are single-GPU runs of the full workload: quoted for the
    1-GPU line only)"""
    if world != 1:
        return None, "the PMC passes are of the 1-GPU run"
    import bench

    return bench.pmc_traffic_sink(workload)


def _kernel_traffic(name):
    import bench

    return bench.pmc_traffic(name)

import numpy as np


def _sync_time(fn, steps, warmup, torch, dist, dev):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    return el


def _agree(ok, torch, dist, dev):
    """Does EVERY rank say ok?  (all-reduce of a flag.)  A rank that failed on its own -- out of memory, a plan the
    library refused -- must not leave the others waiting for it in a barrier or an all-gather: every rank calls this at
    the same points, a failed one with ok = False, and all of them leave the workload together."""
    if dist is None:
        return bool(ok)
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def _share_gate(so, torch, scene, noises, ks, n, slab, tol=1e-8):
    """rank 0's share of config 4 against the CPU oracle (the checker, after the timed loops): the first and the last
    scene of its slab, each whole, on host copies of the very noise the engine filtered"""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
    from oracle_bridge import oracle_sink, relerr

    worst, checked = 0.0, []
    for i in sorted({0, len(ks) - 1}):
        host = np.asfortranarray(noises[i].t().cpu().numpy())
        want = oracle_sink(scene(so, host, ks[i], n))
        got = slab[:, i * n:(i + 1) * n].t().cpu().numpy()
        worst = max(worst, float(relerr(got, want)))
        checked.append(int(ks[i]))
    res = {"relerr": worst, "tolerance": tol, "scenes_checked": checked, "frames_each": int(n),
           "oracle": "CPU restatement of the reference on host copies of the same scenes, tests/oracle_bridge.py"}
    if not worst <= tol:
        raise SystemExit(f"bench: rank 0's share of config 4 differs from the CPU oracle: {res}")
    return res


def run(args, so, torch, dist, rank, local_rank, world, dev, emit=True):
    """emit=False: returns rank 0's result object instead of printing it, and leaves the process group alone"""
    from bench import HBM_PEAK_GBS, METRIC, scene, tree_config5
    from sigops_amd import sharding

    tdt = torch.float64 if args.dtype == "f64" else torch.float32
    ndt = np.float64 if args.dtype == "f64" else np.float32
    esz = 8 if args.dtype == "f64" else 4
    stream = torch.cuda.current_stream().cuda_stream
    steps, warmup = args.steps, args.warmup
    result = None
    # measurement aid on a one-GPU box: evaluate the shard rank R of W would get, without a process group
    # (SIGOPS_BENCH_AS=R/W; compute-only time of that rank, no gather)
    import os

    if world == 1 and os.environ.get("SIGOPS_BENCH_AS"):
        r_, w_ = os.environ["SIGOPS_BENCH_AS"].split("/")
        shard_rank, shard_world = int(r_), int(w_)
    else:
        shard_rank, shard_world = rank, world
    if args.workload in ("config4", "ns_time"):
        err = None
        plan = None
        gate_trees = []
        try:
            keep = []
            if args.workload == "config4":
                nscenes, nch = 64, 2
                n = int(round(args.seconds / 10.0 * 44100)) if args.seconds != 600.0 else 2_646_000  # 60 s scenes
                lo, hi = sharding.block_range(nscenes, shard_rank, shard_world)
                trees = []
                for k in range(lo, hi):
                    g = torch.Generator(device=dev)
                    g.manual_seed(1983 + k)
                    nz = torch.randn((nch, n), dtype=tdt, device=dev, generator=g)
                    keep.append(nz)
                    trees.append(scene(so, nz.t(), k, n))
                counts = [(sharding.block_range(nscenes, r, shard_world)[1] - sharding.block_range(nscenes, r, shard_world)[0]) * n
                          for r in range(shard_world)]
                total = nscenes * n
                sub = None if not trees else (trees[0] if len(trees) == 1 else so.Append(*trees))
                label = ("config4: Append of 64 scenes (Mix(sin,noise[%d x 2]) |> Filt(Bandstop 0.5-2kHz) |> Ramp(10ms)) "
                         "@44.1kHz |> sink, scenes sharded over ranks, RCCL all-gather of the device slabs" % n)
                par = f"append-shard x{world}, all_gather_into_tensor (device to device)"
                extra = {"scenes_per_rank": hi - lo}
            else:
                # ONE north-star pipeline cut along time (sharding.shard_time): every rank evaluates its range of
                # the output from a warm start (no filter state is handed over), slabs all-gathered.  Every rank
                # holds the whole synthetic input (same seed); it reads only its own range of it.
                from bench import tree_ns

                nch = 8
                n_in = int(round(args.seconds * 44100))
                g = torch.Generator(device=dev)
                g.manual_seed(1983)
                nz = torch.randn((nch, n_in), dtype=tdt, device=dev, generator=g)
                keep.append(nz)
                whole = tree_ns(so, nz.t(), n_in)
                total = so.nframes(whole)
                align = 160 * 16
                parts = [sharding.shard_time(whole, r, shard_world, align) for r in range(shard_world)]
                counts = [p[2] for p in parts]
                sub = parts[shard_rank][0]
                label = ("north-star pipeline Mix(sin 1kHz, noise[%d x 8]) |> Filt(Bandstop) |> ToFramerate(48kHz) |> sink, "
                         "ONE signal cut along time over the ranks (warm starts, no state hand-off), RCCL all-gather" % n_in)
                par = f"time-shard x{world}, all_gather_into_tensor (device to device)"
                extra = {"in_frames": n_in}
            width, count = max(counts), counts[shard_rank]
            slab = torch.zeros((nch, width), dtype=tdt, device=dev)
            plan_ms = None
            if sub is not None and count > 0:
                res = slab.t()[:count]
                t0 = time.perf_counter()
                plan = so.Plan(so.ToChannels(sub, nch), (count, nch), ndt, (res.stride(0), res.stride(1)), True, device=local_rank)
                torch.cuda.synchronize()
                plan_ms = (time.perf_counter() - t0) * 1e3
            outs = torch.empty((world, nch, width), dtype=tdt, device=dev) if world > 1 else None
            full = torch.empty((nch, total), dtype=tdt, device=dev) if world > 1 else None
            optr = slab.data_ptr()

        except Exception as exc:  # (a local failure: say so to everybody below instead of leaving them in a collective)
            err = f"rank {rank}: {type(exc).__name__}: {exc}"[:300]
        if err is None and plan is not None:
            try:  # one execute on this rank's own before anybody enters a barrier: a launch that fails, fails here
                plan.execute(slab.data_ptr(), stream)
                torch.cuda.synchronize()
            except Exception as exc:
                err = f"rank {rank}: {type(exc).__name__}: {exc}"[:300]
        if not _agree(err is None, torch, dist, dev):
            if plan is not None:
                plan.close()
            msg = err or "another rank failed during set-up; every rank skipped the workload"
            if emit:
                raise SystemExit("bench: " + msg)
            return {"error": msg} if rank == 0 else None
        def compute():
            if plan is not None:
                plan.execute(optr, stream)

        def with_gather():
            compute()
            if world > 1:
                dist.all_gather_into_tensor(outs.view(world * nch, width), slab)  # (the concatenated form: what RCCL and gloo both take)
                pos = 0
                for r in range(world):  # planar [nch x total] result on every rank
                    full[:, pos:pos + counts[r]] = outs[r, :, :counts[r]]
                    pos += counts[r]

        el_c = _sync_time(compute, steps, warmup, torch, dist, dev)
        el_g = _sync_time(with_gather, steps, warmup, torch, dist, dev)
        st = plan.stats() if plan is not None else {}
        gate = None
        if rank == 0 and args.workload == "config4" and plan is not None and not os.environ.get("SIGOPS_BENCH_NO_GATE"):
            gate = _share_gate(so, torch, scene, keep, list(range(lo, hi)), n, slab)
        if rank == 0:
            ms_g, ms_c = el_g / steps * 1e3, el_c / steps * 1e3
            algo = 2 * esz * total * nch if args.workload == "config4" else st.get("algorithmic_bytes", 0) * world
            cfg = {"workload": label, "out_frames": total, "channels": nch, "parallelism": par,
                   "compute_only_ms": ms_c, "compute_only_frames_per_s": total / (ms_c * 1e-3),
                   "gather_ms": ms_g - ms_c, "launches_per_step": st.get("n_launches"), "plan_create_ms": plan_ms}
            cfg.update(extra)
            if shard_world != world:
                cfg["measured_as"] = f"rank {shard_rank} of {shard_world} on one GPU: its shard only ({count} frames), no gather"
            result = ({
                "metric": METRIC, "value": total / (ms_g * 1e-3), "unit": "frames/s", "n_gpus": world, "steps": steps,
                "warmup": warmup, "ms_per_step": ms_g, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": args.dtype, "data": "synthetic", "config": cfg,
                "algorithmic_bytes_per_step": algo,
                "roofline": {"bound": "hbm", "achieved": algo / (ms_c * 1e-3) / 1e9 / world, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": algo / (ms_c * 1e-3) / 1e9 / world / HBM_PEAK_GBS,
                             "traffic": _sink_traffic(args.workload, world)[0], "traffic_source": _sink_traffic(args.workload, world)[1],
                             "kernel": "whole sink per GPU, compute only (timed loop)"},
                "rccl_ranks": dist.get_world_size() if dist is not None and dist.get_backend() == "nccl" else None,
                "parity_gate": gate, "cpu_baseline": None})
            if emit:
                print(json.dumps(result), flush=True)
    else:  # config5
        cpg = 128
        n = int(round(args.seconds / 600.0 * 10_000_000))
        g = torch.Generator(device=dev)
        g.manual_seed(1983 + rank)
        x = torch.rand((cpg, n), dtype=tdt, device=dev, generator=g)
        tree = tree_config5(so, x.t())
        n_out = so.nframes(tree)
        out_t = torch.empty((cpg, n_out), dtype=tdt, device=dev)
        out = out_t.t()
        t0 = time.perf_counter()
        plan = so.Plan(so.ToChannels(tree, cpg), (n_out, cpg), ndt, (out.stride(0), out.stride(1)), True, device=local_rank)
        torch.cuda.synchronize()
        plan_ms = (time.perf_counter() - t0) * 1e3
        optr = out.data_ptr()
        el = _sync_time(lambda: plan.execute(optr, stream), steps, warmup, torch, dist, dev)
        # size-independent checks at full size: finite, and the low-pass keeps the DC level of uniform(0,1)
        mean = float(out_t[:, 2000:-2000].mean().item())
        finite = bool(torch.isfinite(out_t).all().item())
        plan.set_profiling(True)
        plan.execute(optr, stream)
        stages = plan.steps()
        st = plan.stats()
        if rank == 0:
            ms = el / steps * 1e3
            algo = st["algorithmic_bytes"]
            dom = max(stages, key=lambda s: s["ms"])
            print(json.dumps({
                "metric": METRIC, "value": world * n_out / (ms * 1e-3), "unit": "frames/s", "n_gpus": world, "steps": steps,
                "warmup": warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": args.dtype, "data": "synthetic",
                "config": {"workload": "config5 slab: x[%d x %d per GPU] uniform(0,1) @44.1kHz |> Filt(Lowpass 4kHz) |> ToFramerate(16kHz) |> sink"
                                       % (n, cpg),
                           "in_frames": n, "out_frames": n_out, "channels": world * cpg,
                           "parallelism": f"channel slabs x{world}, no collective", "plan_create_ms": plan_ms,
                           "launches_per_step": st["n_launches"], "mean_of_result": mean, "finite": finite},
                "algorithmic_bytes_per_step": algo, "stages": stages,
                "roofline": {"bound": "hbm", "achieved": dom["algorithmic_bytes"] / (dom["ms"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": dom["algorithmic_bytes"] / (dom["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "traffic": _kernel_traffic("config5:" + dom["name"])[0], "traffic_source": _kernel_traffic("config5:" + dom["name"])[1],
                             "kernel": dom["name"], "kernel_ms": dom["ms"]},
                "roofline_sink": {"achieved": algo / (ms * 1e-3) / 1e9, "frac": algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "traffic": _sink_traffic("config5", 1)[0], "traffic_source": _sink_traffic("config5", 1)[1],
                                  "from": "timed loop, per GPU"},
                "cpu_baseline": None}), flush=True)
        plan.close()
    if not emit:
        if plan is not None and args.workload != "config5":
            plan.close()
        return result if rank == 0 else None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
