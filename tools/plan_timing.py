#!/usr/bin/env python
"""Where does a one-shot sink's time go?  Plan creation (lowering / stage setup, allocation + uploads),
first execute, later executes, destroy -- for BASELINE configs 1-3 and the headline pipeline.
    python tools/plan_timing.py        (SIGOPS_DEBUG_PLAN=1 adds the library's own phase split on stderr)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
import sigops_amd as so


def timed(tree_fn, label, nch, reps=3):
    out = []
    for rep in range(reps):
        x = tree_fn()
        n_out = so.nframes(x)
        res = torch.empty((nch, n_out), dtype=torch.float64, device="cuda").t()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan = so.Plan(so.ToChannels(x, nch), (n_out, nch), np.float64, (res.stride(0), res.stride(1)), True, device=0)
        t1 = time.perf_counter()
        plan.execute(res.data_ptr(), 0)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        plan.execute(res.data_ptr(), 0)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        plan.close()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        out.append({"create_ms": round((t1 - t0) * 1e3, 3), "first_execute_ms": round((t2 - t1) * 1e3, 3),
                    "second_execute_ms": round((t3 - t2) * 1e3, 3), "destroy_ms": round((t4 - t3) * 1e3, 3)})
    print(json.dumps({"workload": label, "out_frames": int(n_out), "runs": out}), flush=True)


def main():
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev)
    gen.manual_seed(1983)
    n_in = 26_460_000
    noise8 = torch.randn((8, n_in), dtype=torch.float64, device=dev, generator=gen).t()
    noise2 = torch.randn((2, 2_646_000), dtype=torch.float64, device=dev, generator=gen).t()
    timed(lambda: (so.Signal(so.sin, ω=1 * so.kHz) | so.Until(5 * so.s) | so.Ramp | so.Normpower | so.Amplify(-20 * so.dB)
                   | so.ToFramerate(44.1 * so.kHz)), "config1", 1)
    timed(lambda: (so.Mix(so.Signal(so.sin, ω=1 * so.kHz), so.Signal(noise2, 44.1 * so.kHz)) | so.Until(2_646_000 * so.frames)
                   | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz)), "config2", 2)
    timed(lambda: bench.tree_config3(so, noise8, n_in), "config3", 8)
    timed(lambda: bench.tree_ns(so, noise8, n_in), "north-star", 8)
    small = torch.randn((2, 44_100), dtype=torch.float64, device=dev, generator=gen).t()
    timed(lambda: (so.Mix(so.Signal(so.sin, ω=1 * so.kHz), so.Signal(small, 44.1 * so.kHz)) | so.Until(44_100 * so.frames)
                   | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.ToFramerate(48 * so.kHz)), "1 s, 2 ch pipeline", 2, reps=5)


if __name__ == "__main__":
    main()
