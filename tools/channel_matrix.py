"""The headline pipeline, its two halves and a plain map over channel counts (total samples fixed): looking for cliffs."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import sigops_amd as so

total = int(float(os.environ.get("TOTAL", "1.0e8")))
dts = [(torch.float64, np.float64)] if not os.environ.get("F32") else [(torch.float32, np.float32)]
for tdt, ndt in dts:
    for nch in (1, 2, 3, 4, 5, 6, 7, 8, 12, 16, 24, 32, 64, 128):
        n = total // nch
        x = torch.randn((nch, n), dtype=tdt, device="cuda").t()
        src = so.Signal(x, 44.1 * so.kHz)
        trees = {
            "pipeline": so.Mix(so.Signal(so.sin, ω=1 * so.kHz), src) | so.Until(n * so.frames) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.ToFramerate(48 * so.kHz),
            "resample": src | so.ToFramerate(48 * so.kHz),
            "filter": src | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz),
            "map": src | so.Amplify(so.Signal(so.sin, ω=5 * so.Hz)) | so.Until(n * so.frames),
        }
        line = f"{nch:4d} ch x {n:9d}"
        for name, tree in trees.items():
            n_out = so.nframes(tree)
            out = torch.empty((nch, n_out), dtype=tdt, device="cuda").t()
            plan = so.Plan(so.ToChannels(tree, nch), (n_out, nch), ndt, (out.stride(0), out.stride(1)), True)
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
            names = "+".join(s_["name"].replace("k_resample_", "rs_").replace("k_", "") for s_ in plan.steps())
            plan.close()
            esz = 8 if ndt == np.float64 else 4
            line += f" | {name} {ms:7.3f} ms {esz * nch * (n + n_out) / ms / 1e9:5.2f} TB/s [{names}]"
            del out
        print(line, flush=True)
        del x
