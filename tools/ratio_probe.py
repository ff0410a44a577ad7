"""Fused resampler + IIR kernel against K3 + K2 for the small rational ratios (x 2, x 3, x 3/2): 8 channels, 300 s."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import sigops_amd as so

nch = 8
for fs_in, fs_out, secs in ((24.0, 48.0, 300), (16.0, 48.0, 300), (32.0, 48.0, 300), (44.1, 48.0, 300)):
    n = int(fs_in * 1000 * secs)
    x = torch.randn((nch, n), dtype=torch.float64, device="cuda").t()
    tree = (so.Mix(so.Signal(so.sin, ω=1 * so.kHz), so.Signal(x, fs_in * so.kHz)) | so.Until(n * so.frames)
            | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.ToFramerate(fs_out * so.kHz))
    n_out = so.nframes(tree)
    out = torch.empty((nch, n_out), dtype=torch.float64, device="cuda").t()
    res = {}
    ref = None
    for name, env in (("fused", {}), ("two", {"SIGOPS_NO_RSOS": "1"})):
        os.environ.update(env)
        plan = so.Plan(so.ToChannels(tree, nch), (n_out, nch), np.float64, (out.stride(0), out.stride(1)), True)
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(10):
            plan.execute(out.data_ptr(), st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            plan.execute(out.data_ptr(), st)
        e1.record()
        torch.cuda.synchronize()
        res[name] = (round(e0.elapsed_time(e1) / 40, 4), [s_["name"] for s_ in plan.steps()])
        if ref is None:
            ref = out.clone()
        else:
            res["relerr"] = float((torch.linalg.norm(out - ref) / torch.linalg.norm(ref)).item())
        plan.close()
        for k in env:
            os.environ.pop(k, None)
    algo = 8 * nch * (n + n_out)
    res["fused_TBps"] = round(algo / res["fused"][0] / 1e9, 3)
    print(fs_in, fs_out, res, flush=True)

# ... and the plain resampler (no filter): what K3 itself makes of these ratios
for fs_in, fs_out, secs in ((24.0, 48.0, 300), (16.0, 48.0, 300), (12.0, 48.0, 300), (32.0, 48.0, 300), (48.0, 24.0, 300)):
    n = int(fs_in * 1000 * secs)
    x = torch.randn((nch, n), dtype=torch.float64, device="cuda").t()
    tree = so.Signal(x, fs_in * so.kHz) | so.ToFramerate(fs_out * so.kHz)
    n_out = so.nframes(tree)
    out = torch.empty((nch, n_out), dtype=torch.float64, device="cuda").t()
    plan = so.Plan(so.ToChannels(tree, nch), (n_out, nch), np.float64, (out.stride(0), out.stride(1)), True)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        plan.execute(out.data_ptr(), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        plan.execute(out.data_ptr(), st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("plain", fs_in, fs_out, round(ms, 4), [s_["name"] for s_ in plan.steps()], "TB/s", round(8 * nch * (n + n_out) / ms / 1e9, 3), flush=True)
    plan.close()
