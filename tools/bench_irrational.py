"""Timing of the irrational-rate resampler (reference test/benchmarks.jl "resampling-irrational",
x pi) on a config-3-sized signal: k_resample_tiled vs the thread-per-output fallback."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sigops_amd as so

n, nch = 26_460_000 // 4, 8
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.randn((nch, n), dtype=torch.float64, device="cuda", generator=g).t()
tree = so.Signal(x, 44100 * so.Hz) | so.ToFramerate(44100 * np.pi / 3 * so.Hz)
n_out = so.nframes(tree)
out_t = torch.empty((nch, n_out), dtype=torch.float64, device="cuda"); out = out_t.t()
for env in ("", "1"):
    if env:
        os.environ["SIGOPS_RS_NOTILED"] = "1"
    plan = so.Plan(so.ToChannels(tree, nch), (n_out, nch), np.float64, (out.stride(0), out.stride(1)), True)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        plan.execute(out.data_ptr(), st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        plan.execute(out.data_ptr(), st)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
    algo = 8 * nch * (n + n_out)
    print(json.dumps({"kernel": "k_resample" if env else "k_resample_tiled", "in_frames": n, "out_frames": n_out, "channels": nch,
                      "ms": ms, "algorithmic_GBps": algo / ms / 1e6, "frac_of_8TBps": algo / ms / 1e6 / 8000}))
    plan.set_profiling(True)
    for _ in range(3):
        plan.execute(out.data_ptr(), st); torch.cuda.synchronize()
    print("   steps:", [(s_["name"], round(s_["ms"], 4), s_["launches"]) for s_ in plan.steps()])
    plan.close()
