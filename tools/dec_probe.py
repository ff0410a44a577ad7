import os, sys
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
import sigops_amd as so
from oracle_bridge import oracle_sink, relerr
for nch, fi, fo in ((8, 48.0, 16.0), (8, 192.0, 48.0), (2, 96.0, 24.0), (8, 44.1, 11.025), (3, 48.0, 8.0), (8, 96.0, 11.025), (8, 32.0, 11.025)):
    n = int(fi * 1000 * 120)
    x = torch.randn((nch, n), dtype=torch.float64, device="cuda").t()
    tree = so.Signal(x, fi * so.kHz) | so.ToFramerate(fo * so.kHz)
    n_out = so.nframes(tree)
    out = torch.empty((nch, n_out), dtype=torch.float64, device="cuda").t()
    plan = so.Plan(so.ToChannels(tree, nch), (n_out, nch), np.float64, (out.stride(0), out.stride(1)), True)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        plan.execute(out.data_ptr(), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        plan.execute(out.data_ptr(), st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    names = [s_["name"] for s_ in plan.steps()]
    plan.close()
    m = 300000
    xs = np.asfortranarray(x[:m].cpu().numpy())
    t2 = so.Signal(xs, fi * so.kHz) | so.ToFramerate(fo * so.kHz)
    err = relerr(so.sink(t2)[0], oracle_sink(t2))
    print(nch, fi, fo, round(ms, 3), "ms", round(8 * nch * (n + n_out) / ms / 1e9, 2), "TB/s", names, "vs oracle", err, flush=True)
