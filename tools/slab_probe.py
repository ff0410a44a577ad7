import sys; sys.path.insert(0,'.')
import numpy as np, torch, sigops_amd as so
from sigops_amd import sharding
from bench import tree_ns
n_in = 26460000
g = torch.Generator(device="cuda"); g.manual_seed(1)
nz = torch.randn((8, n_in), dtype=torch.float64, device="cuda", generator=g)
whole = tree_ns(so, nz.t(), n_in)
total = so.nframes(whole)
full = torch.empty((8, total), dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for w in (1, 8, 16, 32, 64):
    plans = []
    for r in range(w):
        sub, start, count = sharding.shard_time(whole, r, w, 2560)
        res = full[:, start:start + count]
        p = so.Plan(so.ToChannels(sub, 8), (count, 8), np.float64, (1, total), True)
        plans.append((p, res.data_ptr()))
    for _ in range(10):
        for p, ptr in plans: p.execute(ptr, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        for p, ptr in plans: p.execute(ptr, st)
    e1.record(); torch.cuda.synchronize()
    print("slabs", w, "ms per whole signal", e0.elapsed_time(e1) / 30, flush=True)
    for p, _ in plans: p.close()
