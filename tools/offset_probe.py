"""Does the placement of the result relative to the engine's intermediate matter to K2?  The headline plan into one
allocation at byte offsets 0 ... 1 MiB (rows keep their stride); per-stage times from the library's events."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch, sigops_amd as so
from bench import tree_ns
n_in, nch = 26_460_000, 8
g = torch.Generator(device="cuda"); g.manual_seed(1)
nz = torch.randn((nch, n_in), dtype=torch.float64, device="cuda", generator=g)
tree = tree_ns(so, nz.t(), n_in)
n_out = so.nframes(tree)
st = torch.cuda.current_stream().cuda_stream
big = torch.empty((nch * n_out + (1 << 18),), dtype=torch.float64, device="cuda")
p = so.Plan(so.ToChannels(tree, nch), (n_out, nch), np.float64, (1, n_out), True)
print("result base", hex(big.data_ptr()))
for off in (0, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1048576, 1048576 + 4096, 1048576 + 65536 + 2048):
    ptr = big.data_ptr() + off
    p.set_profiling(False)
    for _ in range(8): p.execute(ptr, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): p.execute(ptr, st)
    e1.record(); torch.cuda.synchronize()
    p.set_profiling(True)
    for _ in range(3):
        p.execute(ptr, st); torch.cuda.synchronize()
    print("offset", off, "ms", round(e0.elapsed_time(e1) / 30, 4), [(s["name"], round(s["ms"], 4)) for s in p.steps()], flush=True)
p.close()
