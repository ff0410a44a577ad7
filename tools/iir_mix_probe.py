"""Mix(sine, x) |> Filt on few channels: the one-pass form (12 waves; 16 waves with the step waves) against the three-pass K2."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import sigops_amd as so
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream().cuda_stream
def run(n, nch, mix, env):
    keys = ("SIGOPS_NO_PLAIN_RSOS", "SIGOPS_PLAIN_NWAVES", "SIGOPS_RSOS_MINGROUPS", "SIGOPS_RSOS_NOGSPLIT")
    for k in keys: os.environ.pop(k, None)
    os.environ.update(env)
    x_t = torch.randn((nch, n), dtype=torch.float64, device=dev)
    src = so.Signal(x_t.t(), 44.1 * so.kHz)
    if mix: src = so.Mix(so.Signal(so.sin, ω=1 * so.kHz), src) | so.Until(n * so.frames)
    x = src | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz)
    out_t = torch.empty((nch, n), dtype=torch.float64, device=dev); out = out_t.t()
    plan = so.Plan(so.ToChannels(x, nch), (n, nch), np.float64, (out.stride(0), out.stride(1)), True, device=0)
    for _ in range(30): plan.execute(out.data_ptr(), stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(60): plan.execute(out.data_ptr(), stream)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 60
    names = [s["name"] for s in plan.steps()]; plan.close()
    return round(ms, 4), "+".join(names)
for n, nch in ((50_000_000, 2), (25_000_000, 4), (12_500_000, 8)):
    for mix in (True, False):
        r = {"n": n, "nch": nch, "mix": mix}
        r["default"] = run(n, nch, mix, {})
        r["one pass, 12 waves"] = run(n, nch, mix, {"SIGOPS_RSOS_MINGROUPS": "1", "SIGOPS_PLAIN_NWAVES": "12"})
        r["one pass, 16 waves"] = run(n, nch, mix, {"SIGOPS_RSOS_MINGROUPS": "1", "SIGOPS_PLAIN_NWAVES": "16"})
        r["three passes"] = run(n, nch, mix, {"SIGOPS_NO_PLAIN_RSOS": "1"})
        print(json.dumps(r), flush=True)
