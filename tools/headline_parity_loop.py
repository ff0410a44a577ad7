"""Repeated full-length check of the fused kernel against the two-kernel path on the headline pipeline: the reference result
is computed once, the fused plan is executed N times and every result compared (a rare bad block shows as n_bad > 0)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sigops_amd as so

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
nch, n_in = 8, int(round(600 * 44100))
gen = torch.Generator(device=dev); gen.manual_seed(1983)
noise_t = torch.randn((nch, n_in), dtype=torch.float64, device=dev, generator=gen)
noise = noise_t.t()
stream = torch.cuda.current_stream().cuda_stream
def tree():
    return (so.Mix(so.Signal(so.sin, ω=1 * so.kHz), so.Signal(noise, 44.1 * so.kHz)) | so.Until(n_in * so.frames)
            | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.ToFramerate(48 * so.kHz))
def mk(fused):
    if fused: os.environ.pop("SIGOPS_NO_RSOS", None)
    else: os.environ["SIGOPS_NO_RSOS"] = "1"
    x = tree(); n_out = so.nframes(x)
    out_t = torch.empty((nch, n_out), dtype=torch.float64, device=dev); out = out_t.t()
    plan = so.Plan(so.ToChannels(x, nch), (n_out, nch), np.float64, (out.stride(0), out.stride(1)), True, device=0)
    return plan, out_t, out
p2, y2, o2 = mk(False); p2.execute(o2.data_ptr(), stream); torch.cuda.synchronize()
pf, yf, of = mk(True)
scale = float(y2.abs().max().item()); nrm = float(torch.linalg.norm(y2).item())
bad_runs = 0; worst = 0.0
for i in range(N):
    yf.zero_()
    pf.execute(of.data_ptr(), stream); torch.cuda.synchronize()
    d = yf - y2
    rel = float(torch.linalg.norm(d).item()) / nrm
    nb = int((d.abs() > 1e-6 * scale).sum().item())
    worst = max(worst, rel)
    if nb:
        bad_runs += 1
        idx = (d.abs() > 1e-6 * scale).nonzero()
        print(json.dumps({"run": i, "n_bad": nb, "relerr": rel, "first": idx[0].tolist(), "last": idx[-1].tolist()}), flush=True)
print(json.dumps({"runs": N, "bad_runs": bad_runs, "worst_relerr": worst, "steps": [s["name"] for s in pf.steps()]}), flush=True)
