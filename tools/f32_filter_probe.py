import os, sys
import numpy as np
sys.path.insert(0, '/root/repo')
import torch
import sigops_amd as so
for tdt, ndt in ((torch.float32, np.float32), (torch.float64, np.float64)):
    nch, n = 8, 12500000
    x = torch.randn((nch, n), dtype=tdt, device="cuda").t()
    tree = so.Signal(x, 44.1 * so.kHz) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz)
    out = torch.empty((nch, n), dtype=tdt, device="cuda").t()
    plan = so.Plan(so.ToChannels(tree, nch), (n, nch), ndt, (out.stride(0), out.stride(1)), True)
    st = torch.cuda.current_stream().cuda_stream
    plan.set_profiling(True)
    for _ in range(3):
        plan.execute(out.data_ptr(), st); torch.cuda.synchronize()
    print(ndt.__name__, [(s_["name"], round(s_["ms"], 4), s_["launches"]) for s_ in plan.steps()], flush=True)
    plan.close()
