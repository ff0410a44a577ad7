"""Timing of the irrational-rate resampler (reference test/benchmarks.jl "resampling-irrational", x pi / 3) on a quarter of
a config-3-sized signal: the persistent kernel (k_resample_arb), the tiled kernel it replaced (SIGOPS_RS_NOARB=1) and
the thread-per-output fallback (SIGOPS_RS_NOTILED=1).  Device time between two events over `reps` executes."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import sigops_amd as so

reps = int(os.environ.get("REPS", "100"))
n, nch = 26_460_000 // 4, 8
tdt, ndt = (torch.float32, np.float32) if os.environ.get("F32") else (torch.float64, np.float64)
g = torch.Generator(device="cuda")
g.manual_seed(1)
x = torch.randn((nch, n), dtype=tdt, device="cuda", generator=g).t()
tree = so.Signal(x, 44100 * so.Hz) | so.ToFramerate(44100 * np.pi / 3 * so.Hz)
n_out = so.nframes(tree)
out_t = torch.empty((nch, n_out), dtype=tdt, device="cuda")
out = out_t.t()
which = os.environ.get("ONLY", "arb,tiled,plain").split(",")
for name, env in (("arb", {}), ("tiled", {"SIGOPS_RS_NOARB": "1"}), ("plain", {"SIGOPS_RS_NOTILED": "1"})):
    if name not in which:
        continue
    os.environ.update(env)
    plan = so.Plan(so.ToChannels(tree, nch), (n_out, nch), ndt, (out.stride(0), out.stride(1)), True)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(20):
        plan.execute(out.data_ptr(), st)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps if name != "plain" else 10):
            plan.execute(out.data_ptr(), st)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / (reps if name != "plain" else 10))
    algo = (8 if ndt == np.float64 else 4) * nch * (n + n_out)
    print(json.dumps({"kernel": [s_["name"] for s_ in plan.steps()][-1], "dtype": ndt.__name__, "in_frames": n, "out_frames": n_out, "channels": nch,
                      "ms": best, "algorithmic_GBps": algo / best / 1e6, "frac_of_8TBps": algo / best / 1e6 / 8000}), flush=True)
    plan.close()
    for k in env:
        os.environ.pop(k, None)
