"""How often do engine and oracle round a Float32 resampler output apart?  Where the Float64 value sits within their
Float64 difference (the drift of DSP.jl's accumulated alpha, ~1e-12) of a Float32 rounding boundary: 4-9 in 10 000."""
import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, sigops_amd as so
from oracle_bridge import oracle_sink, relerr
rng = np.random.default_rng(1)
for (fi, fo) in ((44100, 24000), (44100, 48000), (48000, 44100), (22050, 16000)):
    for n, nch in ((13917, 6), (200000, 8)):
        x32 = np.asfortranarray(rng.standard_normal((n, nch)).astype(np.float32))
        t32 = so.Signal(x32, float(fi) * so.Hz) | so.ToFramerate(float(fo) * so.Hz)
        t64 = so.Signal(np.asfortranarray(x32.astype(np.float64)), float(fi) * so.Hz) | so.ToFramerate(float(fo) * so.Hz)
        g32, w32 = so.sink(t32)[0], oracle_sink(t32)
        g64, w64 = so.sink(t64)[0], oracle_sink(t64)
        mism = np.argwhere(g32 != w32)
        # how close to a rounding boundary are the f64 values at the mismatching elements?
        info = []
        for (i, c) in mism[:5]:
            v = w64[i, c]; lo = np.float32(v); 
            nxt = np.nextafter(lo, np.float32(np.inf) if v > lo else np.float32(-np.inf))
            mid = (float(lo) + float(nxt)) / 2
            info.append((int(i), int(c), "%.3g" % abs(v - mid), "%.3g" % abs(g64[i, c] - w64[i, c])))
        print(fi, fo, n, nch, g32.dtype, "f32 mismatches", len(mism), "of", g32.size, "f64 relerr %.3g" % relerr(g64, w64), "max|g64-w64| %.3g" % np.abs(g64 - w64).max(), info, flush=True)
