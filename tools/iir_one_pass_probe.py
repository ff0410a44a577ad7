"""plain Filt: one pass (k_rsos, identity resampler) against the three-pass K2 on device-resident data"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sigops_amd as so
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream().cuda_stream
def run(n, nch, dtype, env):
    for k, v in env.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    tdt = torch.float64 if dtype == "f64" else torch.float32
    x_t = torch.randn((nch, n), dtype=tdt, device=dev)
    x = so.Signal(x_t.t(), 44.1 * so.kHz) | so.Filt(so.Lowpass, 4 * so.kHz)
    out_t = torch.empty((nch, n), dtype=tdt, device=dev); out = out_t.t()
    plan = so.Plan(so.ToChannels(x, nch), (n, nch), np.float64 if dtype == "f64" else np.float32, (out.stride(0), out.stride(1)), True, device=0)
    for _ in range(30): plan.execute(out.data_ptr(), stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(60): plan.execute(out.data_ptr(), stream)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 60
    names = [s["name"] for s in plan.steps()]; st = plan.stats(); plan.close()
    return {"n": n, "nch": nch, "dtype": dtype, "ms": round(ms, 4), "steps": names, "TBps": round(st["algorithmic_bytes"] / (ms * 1e-3) / 1e12, 3)}
for n, nch in ((12_500_000, 8), (28_800_000, 8), (2_646_000, 2), (50_000_000, 2), (25_000_000, 4), (6_000_000, 16)):
    for dt in ("f64", "f32"):
        print(json.dumps(run(n, nch, dt, {"SIGOPS_NO_PLAIN_RSOS": None})), flush=True)
        print(json.dumps(run(n, nch, dt, {"SIGOPS_NO_PLAIN_RSOS": "1"})), flush=True)
