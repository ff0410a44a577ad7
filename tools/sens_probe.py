import sys, os, itertools
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
os.environ["SIGOPS_DEBUG_PLAN"] = "1"
import numpy as np, sigops_amd as so
rng = np.random.default_rng(1)
fs = 48000.0
x = so.Signal(np.asfortranarray(rng.standard_normal((300000, 2))), fs * so.Hz)
for order, kind, f in itertools.product((2, 3, 5, 9), ("lp", "hp", "bp", "bs"), (1e-4, 1e-3, 0.01, 0.04, 0.125, 0.25, 0.375, 0.49, 0.4999)):
    if kind in ("bp", "bs") and not (f < 0.4): continue
    meth = so.Butterworth(order)
    if kind == "lp": t = so.Filt(x, so.Lowpass, f * fs * so.Hz, method=meth)
    elif kind == "hp": t = so.Filt(x, so.Highpass, f * fs * so.Hz, method=meth)
    elif kind == "bp": t = so.Filt(x, so.Bandpass, f * fs * so.Hz, min(0.4999, f * 1.5 + 0.05) * fs * so.Hz, method=meth)
    else: t = so.Filt(x, so.Bandstop, f * fs * so.Hz, min(0.4999, f * 1.5 + 0.05) * fs * so.Hz, method=meth)
    print("FILT", order, kind, f, flush=True)
    sys.stderr.flush()
    so.sink(t)
