"""Does the alignment of the result's channel rows matter?  config 5's slab (10 000 000 x 128 -> 3 628 118 frames per
channel): result rows at a stride of n_out frames (what a Julia Array has; rows start at any multiple of 8 bytes) against
a stride rounded up to 64 frames.  Also irrational-rate geometry when asked (argv[1] == 'irr')."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch, sigops_amd as so
from bench import tree_config5
nch, n = 128, 10_000_000
x = torch.rand((nch, n), dtype=torch.float64, device="cuda")
tree = tree_config5(so, x.t())
n_out = so.nframes(tree)
st = torch.cuda.current_stream().cuda_stream
for stride in (n_out, (n_out + 63) // 64 * 64, n_out + 1, n_out + 8):
    out = torch.empty((nch, stride), dtype=torch.float64, device="cuda")
    p = so.Plan(so.ToChannels(tree, nch), (n_out, nch), np.float64, (1, stride), True)
    for _ in range(5):
        p.execute(out.data_ptr(), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        p.execute(out.data_ptr(), st)
    e1.record(); torch.cuda.synchronize()
    p.set_profiling(True)
    for _ in range(3):
        p.execute(out.data_ptr(), st); torch.cuda.synchronize()
    print("stride", stride, "mod 16 =", stride % 16, "ms", round(e0.elapsed_time(e1) / 20, 4), [(s["name"], round(s["ms"], 4)) for s in p.steps()], flush=True)
    p.close()
    del out
