"""Which kernel does `ToFramerate` get for the common audio rate pairs, and how fast is it?  (looking for cliffs)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import sigops_amd as so

rates = [8.0, 11.025, 16.0, 22.05, 24.0, 32.0, 44.1, 48.0, 88.2, 96.0, 192.0]
secs = float(os.environ.get("SECS", "120"))
TDT, NDT, ESZ = (torch.float32, np.float32, 4) if os.environ.get("F32") else (torch.float64, np.float64, 8)
for nch in (8, 2):
    for fi in rates:
        for fo in rates:
            if fi == fo:
                continue
            n = int(fi * 1000 * secs)
            x = torch.randn((nch, n), dtype=TDT, device="cuda").t()
            tree = so.Signal(x, fi * so.kHz) | so.ToFramerate(fo * so.kHz)
            n_out = so.nframes(tree)
            out = torch.empty((nch, n_out), dtype=TDT, device="cuda").t()
            plan = so.Plan(so.ToChannels(tree, nch), (n_out, nch), NDT, (out.stride(0), out.stride(1)), True)
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
            names = [s_["name"] for s_ in plan.steps()]
            plan.close()
            tb = ESZ * nch * (n + n_out) / ms / 1e9
            print(f"{nch} ch {fi:7.3f} -> {fo:7.3f}  {ms:8.3f} ms  {tb:6.2f} TB/s  {names}", flush=True)
            del x, out
