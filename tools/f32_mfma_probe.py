"""Float32 ToFramerate on the Float32 MFMA (K3 F32M) against the Float64 products rounded once (SIGOPS_RS_NO_F32MFMA=1):
time on 8 ch x 600 s at 44.1 -> 48 kHz, accuracy of both against the oracle on a prefix, and a sweep of signal kinds
(noise, a loud low tone, a sum of tones, a DC offset with small noise) whose cancellation behaviour differs."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import sigops_amd as so
from oracle_bridge import oracle_sink, relerr
dev = torch.device("cuda:0"); stream = torch.cuda.current_stream().cuda_stream
def timed(env):
    for k, v in env.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    n = int(600 * 44100)
    x_t = torch.randn((8, n), dtype=torch.float32, device=dev)
    x = so.Signal(x_t.t(), 44.1 * so.kHz) | so.ToFramerate(48 * so.kHz)
    no = so.nframes(x)
    out_t = torch.empty((8, no), dtype=torch.float32, device=dev); out = out_t.t()
    plan = so.Plan(so.ToChannels(x, 8), (no, 8), np.float32, (out.stride(0), out.stride(1)), True, device=0)
    for _ in range(40): plan.execute(out.data_ptr(), stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): plan.execute(out.data_ptr(), stream)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 100
    st = plan.stats(); names = [s["name"] for s in plan.steps()]; plan.close()
    return {"ms": round(ms, 4), "TBps": round(st["algorithmic_bytes"] / (ms * 1e-3) / 1e12, 3), "steps": names}
print(json.dumps({"f32 mfma": timed({"SIGOPS_RS_NO_F32MFMA": None}), "f64 mfma": timed({"SIGOPS_RS_NO_F32MFMA": "1"})}), flush=True)
rng = np.random.default_rng(5)
N = 300000
t = np.arange(N) / 44100.0
kinds = {"noise": rng.standard_normal((N, 8)), "low tone": np.sin(2 * np.pi * 50 * t)[:, None] * np.ones((1, 8)) * 0.9,
         "tones": sum(np.sin(2 * np.pi * f * t + p) for f, p in ((440, 0.1), (1000, 0.7), (7000, 1.3), (15000, 2.0)))[:, None] * np.ones((1, 8)),
         "dc + small noise": 1.0 + 1e-3 * rng.standard_normal((N, 8)), "sparse clicks": (rng.random((N, 8)) < 1e-3) * 1.0}
res = {}
for name, d in kinds.items():
    x32 = np.asfortranarray(d.astype(np.float32))
    for fo in (48.0, 32.0, 96.0):
        tree = so.Signal(x32, 44.1 * so.kHz) | so.ToFramerate(fo * so.kHz)
        want = oracle_sink(tree)
        out = {}
        for tag, env in (("f32mfma", None), ("f64mfma", "1")):
            if env is None: os.environ.pop("SIGOPS_RS_NO_F32MFMA", None)
            else: os.environ["SIGOPS_RS_NO_F32MFMA"] = env
            got = so.sink(tree)[0]
            out[tag] = float(relerr(got, want))
        res[f"{name} -> {fo}"] = out
print(json.dumps(res, indent=1))
