"""Steps of the config-4 plan (64 scenes under one Append) on one GPU: names, launches, mean ms, bytes."""
import sys; sys.path.insert(0, '.')
import collections
import numpy as np, torch, sigops_amd as so
from bench import scene
nscenes, nch, n = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 2, 2_646_000
keep, trees = [], []
for k in range(nscenes):
    g = torch.Generator(device="cuda"); g.manual_seed(1983 + k)
    nz = torch.randn((nch, n), dtype=torch.float64, device="cuda", generator=g)
    keep.append(nz); trees.append(scene(so, nz.t(), k, n))
total = nscenes * n
out = torch.empty((nch, total), dtype=torch.float64, device="cuda")
p = so.Plan(so.ToChannels(so.Append(*trees), nch), (total, nch), np.float64, (1, total), True)
st = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    p.execute(out.data_ptr(), st)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(50):
    p.execute(out.data_ptr(), st)
ev1.record(); torch.cuda.synchronize()
print("ms per execute", ev0.elapsed_time(ev1) / 50, p.counters() if hasattr(p, "counters") else "")
p.set_profiling(True)
for _ in range(3):
    p.execute(out.data_ptr(), st); torch.cuda.synchronize()
agg = collections.OrderedDict()
for s in p.steps():
    a = agg.setdefault(s["name"], [0, 0.0, 0, 0])
    a[0] += 1; a[1] += s["ms"]; a[2] += s["launches"]; a[3] += s["algorithmic_bytes"]
for k, a in agg.items():
    print(k, "steps", a[0], "sum ms", round(a[1], 4), "launches", a[2], "bytes", a[3])
print(p.stats())
p.close()
