"""Result rows off the cache line, per kernel: the same plan into a result whose channel stride is n_out (rounded to 64)
and n_out + 1 / + 7 frames.  K3 (config 3), K1 (Mix of two arrays), K2 (headline)."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch, sigops_amd as so
from bench import tree_ns, tree_config3
n_in, nch = 26_460_000, 8
g = torch.Generator(device="cuda"); g.manual_seed(1)
nz = torch.randn((nch, n_in), dtype=torch.float64, device="cuda", generator=g)
nz2 = torch.randn((nch, n_in), dtype=torch.float64, device="cuda", generator=g)
st = torch.cuda.current_stream().cuda_stream
cases = [("config3 (K3 writes the result)", tree_config3(so, nz.t(), n_in)),
         ("Mix of two arrays (K1)", so.Mix(so.Signal(nz.t(), 44.1 * so.kHz), so.Signal(nz2.t(), 44.1 * so.kHz))),
         ("north-star (K2 writes the result)", tree_ns(so, nz.t(), n_in))]
for dt, tdt in ((np.float64, torch.float64), (np.float32, torch.float32)):
    for name, tree in cases:
        n_out = so.nframes(tree)
        for pad in (0, 1, 7):
            stride = (n_out + 63) // 64 * 64 + pad
            out = torch.empty((nch, stride), dtype=tdt, device="cuda")
            try:
                p = so.Plan(so.ToChannels(tree, nch), (n_out, nch), dt, (1, stride), True)
            except Exception as e:
                print(name, dt.__name__, "plan failed:", e); break
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
            print(name, dt.__name__, "stride pad", pad, "ms", round(e0.elapsed_time(e1) / 20, 4), [(s["name"], round(s["ms"], 4)) for s in p.steps()], flush=True)
            p.close(); del out
