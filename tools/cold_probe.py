#!/usr/bin/env python
"""Per-execute time series of a freshly created plan (why is the driver's
20-step / 5-warm-up number slower than the 200 / 30 one?).

    python tools/cold_probe.py [--workload config3|ns] [--n 60] [--touch] [--idle 0.5]

Prints one JSON line: ms of every execute (events on the launch stream, recorded back to back
without host synchronisation), the same after an idle gap, and the wall time of the loop."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="config3")
    ap.add_argument("--n", type=int, default=60)
    ap.add_argument("--touch", action="store_true", help="memset the result buffer before the first execute")
    ap.add_argument("--idle", type=float, default=0.5)
    ap.add_argument("--seconds", type=float, default=600.0)
    args = ap.parse_args()
    import numpy as np
    import torch

    import bench
    import sigops_amd as so

    dev = torch.device("cuda:0")
    nch = 8
    n_in = int(round(args.seconds * 44100))
    gen = torch.Generator(device=dev)
    gen.manual_seed(1983)
    noise_t = torch.randn((nch, n_in), dtype=torch.float64, device=dev, generator=gen)
    noise = noise_t.t()
    fn = bench.tree_ns if args.workload == "ns" else bench.tree_config3
    x = fn(so, noise, n_in)
    n_out = so.nframes(x)
    out_t = torch.empty((nch, n_out), dtype=torch.float64, device=dev)
    if args.touch:
        out_t.zero_()
    out = out_t.t()
    stream = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan = so.Plan(so.ToChannels(x, nch), (n_out, nch), np.float64, (out.stride(0), out.stride(1)), True, device=0)
    torch.cuda.synchronize()
    plan_ms = (time.perf_counter() - t0) * 1e3

    def series(n):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        t0 = time.perf_counter()
        ev[0].record()
        for i in range(n):
            plan.execute(out.data_ptr(), stream)
            ev[i + 1].record()
        host_issue = (time.perf_counter() - t0) * 1e3
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        return [round(ev[i].elapsed_time(ev[i + 1]), 4) for i in range(n)], wall, host_issue

    res = {"workload": args.workload, "plan_ms": plan_ms, "touch": args.touch}
    res["first"], res["first_wall_ms"], res["first_issue_ms"] = series(args.n)
    time.sleep(args.idle)
    res["after_idle"], res["idle_wall_ms"], res["idle_issue_ms"] = series(args.n)
    # the driver's shape: 5 warm-ups, sync, 20 timed
    time.sleep(args.idle)
    for _ in range(5):
        plan.execute(out.data_ptr(), stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        plan.execute(out.data_ptr(), stream)
    torch.cuda.synchronize()
    res["driver_shape_ms_per_step"] = (time.perf_counter() - t0) * 1e3 / 20
    res["env"] = {k: v for k, v in os.environ.items() if k.startswith("SIGOPS_")}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
