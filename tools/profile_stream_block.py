"""Where one block of so.stream spends its host time (north-star pipeline, 1 s blocks)."""
import sys, time, cProfile, pstats
sys.path.insert(0, '.')
import numpy as np, torch
import sigops_amd as so
n = 44100 * 120
g = torch.Generator(device="cuda"); g.manual_seed(7)
x = torch.randn((8, n), dtype=torch.float64, device="cuda", generator=g).t()
tree = (so.Mix(so.Signal(so.sin, ω=1 * so.kHz), so.Signal(x, 44.1 * so.kHz)) | so.Until(n * so.frames)
        | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.ToFramerate(48 * so.kHz))
it = so.stream(tree, 48000, "torch")
for _ in range(5): next(it)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for _ in range(50): next(it)
torch.cuda.synchronize()
el = time.perf_counter() - t0
pr.disable()
print("ms per block", el / 50 * 1e3)
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
