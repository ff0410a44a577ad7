import sys; sys.path.insert(0,'.')
import numpy as np, torch, sigops_amd as so
from sigops_amd import sharding
from bench import tree_ns
n_in = 26460000
g = torch.Generator(device="cuda"); g.manual_seed(1)
nz = torch.randn((8, n_in), dtype=torch.float64, device="cuda", generator=g)
whole = tree_ns(so, nz.t(), n_in)
for r, w in ((0, 1), (0, 8), (3, 8)):
    sub, start, count = sharding.shard_time(whole, r, w, 2560)
    out = torch.empty((8, count), dtype=torch.float64, device="cuda")
    p = so.Plan(so.ToChannels(sub, 8), (count, 8), np.float64, (1, count), True)
    p.set_profiling(True)
    for _ in range(3):
        p.execute(out.data_ptr(), torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    print(r, w, count, [(s["name"], round(s["ms"], 4), s["launches"], s["algorithmic_bytes"]) for s in p.steps()])
    p.close()
