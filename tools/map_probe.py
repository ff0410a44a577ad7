"""K1 on one Amplify(x, sin) map over channel counts: interpreter chain path vs the hipRTC-specialised kernel."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import sigops_amd as so

total = 100_000_000
for tdt, ndt in ((torch.float64, np.float64), (torch.float32, np.float32)):
    for nch in (1, 2, 4, 8):
        n = total // nch
        x = torch.randn((nch, n), dtype=tdt, device="cuda").t()
        tree = so.Signal(x, 44.1 * so.kHz) | so.Amplify(so.Signal(so.sin, ω=5 * so.Hz)) | so.Until(n * so.frames)
        out = torch.empty((nch, n), dtype=tdt, device="cuda").t()
        plan = so.Plan(so.ToChannels(tree, nch), (n, nch), ndt, (out.stride(0), out.stride(1)), True)
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            plan.execute(out.data_ptr(), st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            plan.execute(out.data_ptr(), st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        esz = 8 if ndt == np.float64 else 4
        print(ndt.__name__, nch, "ch", round(ms, 3), "ms", round(2 * esz * total / ms / 1e9, 2), "TB/s", [s_["name"] for s_ in plan.steps()], flush=True)
        plan.close()
