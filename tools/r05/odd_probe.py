import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch
import sigops_amd as so
os.environ["SIGOPS_RSOS_MINGROUPS"] = "1"
def pipeline(x): return x | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.ToFramerate(48 * so.kHz)
n = 2000001
xt = torch.randn((8, n), dtype=torch.float64, device="cuda")
xa = torch.empty((8, n + 1), dtype=torch.float64, device="cuda")[:, :n]
xa.copy_(xt)
for rep in range(2):
  for name, leaf in (("odd", xt), ("aligned", xa)):
    x = pipeline(so.Mix(so.Signal(so.sin, ω=1 * so.kHz), so.Signal(leaf.t(), 44.1 * so.kHz)) | so.Until(n * so.frames))
    nout = so.nframes(x)
    out = torch.empty((8, nout), dtype=torch.float64, device="cuda").t()
    for i in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        so.sink_into(out, x)
        torch.cuda.synchronize()
        print(name, rep, i, "%.3f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
