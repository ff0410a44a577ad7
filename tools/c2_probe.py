import sys; sys.path.insert(0, '.')
import numpy as np, torch, sigops_amd as so
n, nch = 2_646_000, 2
nz = torch.randn((nch, n), dtype=torch.float64, device="cuda")
tree = so.Mix(so.Signal(so.sin, ω=1 * so.kHz) | so.Until(n * so.frames), so.Signal(nz.t(), 44.1 * so.kHz)) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz)
out = torch.empty((nch, n), dtype=torch.float64, device="cuda")
p = so.Plan(so.ToChannels(tree, nch), (n, nch), np.float64, (1, n), True)
st = torch.cuda.current_stream().cuda_stream
for _ in range(20): p.execute(out.data_ptr(), st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): p.execute(out.data_ptr(), st)
e1.record(); torch.cuda.synchronize()
print("ms", round(e0.elapsed_time(e1) / 200, 4))
