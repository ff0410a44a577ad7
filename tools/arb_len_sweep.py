"""k_resample_arb against the tiled kernel over signal lengths (x pi / 3, 8 channels): where does the persistent form pay?"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import sigops_amd as so

nch = int(os.environ.get("NCH", "8"))
for n in (20000, 50000, 100000, 200000, 400000, 800000, 1600000, 3200000):
    x = torch.randn((nch, n), dtype=torch.float64, device="cuda").t()
    tree = so.Signal(x, 44100 * so.Hz) | so.ToFramerate(44100 * np.pi / 3 * so.Hz)
    n_out = so.nframes(tree)
    out = torch.empty((nch, n_out), dtype=torch.float64, device="cuda").t()
    res = {}
    for name, env in (("arb", {}), ("tiled", {"SIGOPS_RS_NOARB": "1"})):
        os.environ.update(env)
        plan = so.Plan(so.ToChannels(tree, nch), (n_out, nch), np.float64, (out.stride(0), out.stride(1)), True)
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(20):
            plan.execute(out.data_ptr(), st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            plan.execute(out.data_ptr(), st)
        e1.record()
        torch.cuda.synchronize()
        res[name] = (e0.elapsed_time(e1) / 100, [s_["name"] for s_ in plan.steps()][-1])
        plan.close()
        for k in env:
            os.environ.pop(k, None)
    print(n, {k: (round(v[0] * 1e3, 1), v[1]) for k, v in res.items()}, flush=True)
