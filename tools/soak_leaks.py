"""Many plans created, executed, failed and destroyed: device memory returns to where it started.
python tools/soak_leaks.py"""
import sys, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import sigops_amd as so
rng = np.random.default_rng(3)
x = so.Signal(np.asfortranarray(rng.standard_normal((60000, 2))), 44.1 * so.kHz)
good = so.Mix(so.Signal(so.sin, ω=1 * so.kHz), x) | so.Until(60000 * so.frames) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.ToFramerate(48 * so.kHz)
failing = [so.Pad(x | so.Amplify(2.0), so.mirror) | so.Until(70000 * so.frames),              # indexing pad over a non-array
           x | so.Amplify(1.0) | so.After(70000 * so.frames),                                # too short to skip
           so.Append(x | so.Until(10 * so.frames) | so.After(20 * so.frames), x | so.Filt(so.Lowpass, 3 * so.kHz) | so.Normpower)]  # ... in the first child
torch.cuda.synchronize(); so.sink(good); torch.cuda.synchronize()
free0, total = torch.cuda.mem_get_info()
nfail = 0
for it in range(1500):
    so.sink(good, so.Array)
    for t in failing:
        try:
            so.sink(t, so.Array)
        except so.ErrorException:
            nfail += 1
    if it % 500 == 499:
        torch.cuda.synchronize(); f, _ = torch.cuda.mem_get_info()
        print('iteration', it + 1, 'free memory change %.1f MB' % ((f - free0) / 1e6), flush=True)
torch.cuda.synchronize(); free1, _ = torch.cuda.mem_get_info()
print('plans', 1500 * 4, 'failed as expected', nfail, 'free memory change %.1f MB' % ((free1 - free0) / 1e6), 'OK' if abs(free1 - free0) < 64e6 and nfail == 4500 else 'BAD')
