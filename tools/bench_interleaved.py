"""K1 on frame-interleaved buffers (frame_stride = nch): the LDS-transposing chain path against
planar buffers and against the strided per-lane access it replaces (SIGOPS_K1_NOCHAIN=1)."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sigops_amd as so

n, nch = 26_460_000, 8
g = torch.Generator(device="cuda"); g.manual_seed(1)
xp = torch.randn((nch, n), dtype=torch.float64, device="cuda", generator=g)
xi = xp.t().contiguous()  # [n, nch] row-major: interleaved
st = torch.cuda.current_stream().cuda_stream


def run(name, leaf, out):
    tree = so.Signal(leaf, 44.1 * so.kHz) | so.Amplify(so.Signal(so.sin, ω=5 * so.Hz)) | so.Until(n * so.frames)
    plan = so.Plan(so.ToChannels(tree, nch), (n, nch), np.float64, (out.stride(0), out.stride(1)), True)
    for _ in range(3):
        plan.execute(out.data_ptr(), st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        plan.execute(out.data_ptr(), st)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
    print(json.dumps({"case": name, "ms": ms, "GBps": 16.0 * n * nch / ms / 1e6, "frac_of_8TBps": 16.0 * n * nch / ms / 1e6 / 8000,
                      "nochain": bool(os.environ.get("SIGOPS_K1_NOCHAIN"))}), flush=True)
    plan.close()


op = torch.empty((nch, n), dtype=torch.float64, device="cuda").t()
oi = torch.empty((n, nch), dtype=torch.float64, device="cuda")
run("planar leaf -> planar result", xp.t(), op)
run("interleaved leaf -> planar result", xi, op)
run("planar leaf -> interleaved result", xp.t(), oi)
run("interleaved leaf -> interleaved result", xi, oi)
