import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import sigops_amd as so
n, nch = 3_000_000, 4
x = torch.randn((nch, n), dtype=torch.float64, device="cuda").t()
y = torch.randn((nch, n), dtype=torch.float64, device="cuda").t()
tree = so.Amplify(so.Mix(so.Signal(x, 1000 * so.Hz), so.Signal(y, 1000 * so.Hz)), 0.5 + float(sys.argv[1]) * 1e-3)
out = torch.empty((nch, n), dtype=torch.float64, device="cuda").t()
so.sink_into(out, tree)     # big map: interpreter now, a background compile queued -- and the process exits right away
print("done", float(out[0, 0].item()) != 12345.0, flush=True)
