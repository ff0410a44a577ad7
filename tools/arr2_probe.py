"""round 6: Mix(x, y) |> Filt |> ToFramerate over two 12.5 M x 8 arrays: one launch (k_rsos with the two-array loader) against
the resampler's two-array form + the filter, and against the sum materialised by K1 + the fused kernel"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sigops_amd as so
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream().cuda_stream
def run(n, nch, env, op="Mix"):
    for k in ("SIGOPS_RSOS_NO_ARR2", "SIGOPS_NO_ARR2", "SIGOPS_RSOS_ARR2_DEPTH", "SIGOPS_NO_RSOS"): os.environ.pop(k, None)
    os.environ.update(env)
    X = so.Signal(torch.randn((nch, n), dtype=torch.float64, device=dev).t(), 44.1 * so.kHz)
    Y = so.Signal(torch.randn((nch, n), dtype=torch.float64, device=dev).t(), 44.1 * so.kHz)
    x = (so.Mix(X, Y) if op == "Mix" else so.Amplify(X, Y)) | so.Filt(so.Lowpass, 3 * so.kHz) | so.ToFramerate(48 * so.kHz)
    m = so.nframes(x)
    out_t = torch.empty((nch, m), dtype=torch.float64, device=dev); out = out_t.t()
    plan = so.Plan(so.ToChannels(x, nch), (m, nch), np.float64, (out.stride(0), out.stride(1)), True, device=0)
    for _ in range(30): plan.execute(out.data_ptr(), stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(60): plan.execute(out.data_ptr(), stream)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 60
    names = [s["name"] for s in plan.steps()]; plan.close()
    return {"n": n, "nch": nch, "op": op, "env": env, "ms": round(ms, 4), "steps": names}
for n, nch in ((12_500_000, 8), (6_250_000, 16), (26_460_000, 8), (25_000_000, 4), (50_000_000, 2)):
    for env in ({}, {"SIGOPS_RSOS_NO_ARR2": "1"}, {"SIGOPS_NO_ARR2": "1"}, {"SIGOPS_NO_RSOS": "1"}):
        print(json.dumps(run(n, nch, env)), flush=True)
print(json.dumps(run(12_500_000, 8, {}, "Amplify")), flush=True)
