"""Resampling at the edges against the oracle: rates next to 1, far below and far above it, rational rates with small
and with huge periods, signals shorter than the filter, one frame, windows at the very start and end.
python tools/soak_degenerate_rates.py"""
import sys, itertools, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import sigops_amd as so
from oracle_bridge import oracle_semantics, oracle_sink, relerr
bad = 0; n = 0; worst = 0.0
rng = np.random.default_rng(777)
pairs = [(44100.0, 44100.0 * (1 + 1e-7)), (44100.0, 44100.0 * (1 - 1e-7)), (48000.0, 480.0), (48000.0, 613.0), (480.0, 48000.0),
         (1280.0, 48000.0), (44101.0, 48000.0), (48000.0, 44101.0), (16000.0, 48000.0), (48000.0, 16000.0), (44100.0, 22050.0),
         (8000.0, 12000.0), (12000.0, 8000.0), (44100.0, 44100.5), (1000.0, 1000.0 * np.pi), (96000.0, 8000.0), (8000.0, 96000.0)]
for dt in (np.float64, np.float32):
    for (fi, fo), nfr, nch in itertools.product(pairs, (1, 2, 37, 39, 1000, 40000, 250000), (1, 3, 8)):
        if nfr * fo / fi > 3e6 or nfr * fo / fi < 1: continue
        x = so.Signal(np.asfortranarray(rng.standard_normal((nfr, nch)).astype(dt)), fi * so.Hz)
        t = x | so.ToFramerate(fo * so.Hz)
        variants = [t]
        m = so.nframes(t)
        if m > 10:
            variants.append(t | so.After((m - 7) * so.frames))
            variants.append(t | so.After(3 * so.frames) | so.Until(5 * so.frames))
        for v in variants:
            try:
                with oracle_semantics("intended"):
                    want = oracle_sink(v)
                got = so.sink(v)[0]
            except Exception as e:
                print("skip", dt.__name__, fi, fo, nfr, nch, str(e)[:100]); continue
            n += 1
            e = relerr(got, want) if got.shape == want.shape else float("inf")
            if dt == np.float64: worst = max(worst, e if np.isfinite(e) else 0)
            tol = 5e-6 if dt == np.float32 else 1e-8
            if not e <= tol:
                bad += 1
                print("BAD", dt.__name__, fi, fo, nfr, nch, "frames", so.nframes(v), "relerr %.3g" % e, flush=True)
print("checks", n, "bad", bad, "worst Float64 relerr %.3g" % worst)
