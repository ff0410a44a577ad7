"""Filters at the edges of the design space against the oracle: cut-offs at exactly fs/4 (poles at 0), next to 0 and
next to Nyquist (poles next to the unit circle: long memories, chunks lengthened until the scan fits), orders 1 ... 9,
Butterworth and Chebyshev, signals of a few and of many chunks, Float32 and Float64, an `After` window behind.
python tools/soak_degenerate_filters.py"""
import sys, itertools, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import sigops_amd as so
from oracle_bridge import oracle_semantics, oracle_sink, relerr
bad = 0; n = 0; worst = 0.0
rng = np.random.default_rng(4242)
fs = 48000.0
fracs = [1e-4, 1e-3, 0.01, 0.125, 0.25, 0.375, 0.49, 0.4999]
for dt in (np.float64, np.float32):
    for nfr in (5000, 70000, 300000):
        x = so.Signal(np.asfortranarray(rng.standard_normal((nfr, 3)).astype(dt)), fs * so.Hz)
        for order, kind, f in itertools.product((1, 2, 3, 5, 9), ("lp", "hp", "bp", "bs"), fracs):
            if kind in ("bp", "bs") and not (f < 0.4):
                continue
            meth = so.Butterworth(order) if (order + len(kind)) % 2 else so.Chebyshev1(order, 1.0)
            try:
                if kind == "lp": t = so.Filt(x, so.Lowpass, f * fs * so.Hz, method=meth)
                elif kind == "hp": t = so.Filt(x, so.Highpass, f * fs * so.Hz, method=meth)
                elif kind == "bp": t = so.Filt(x, so.Bandpass, f * fs * so.Hz, min(0.4999, f * 1.5 + 0.05) * fs * so.Hz, method=meth)
                else: t = so.Filt(x, so.Bandstop, f * fs * so.Hz, min(0.4999, f * 1.5 + 0.05) * fs * so.Hz, method=meth)
                if (order + nfr) % 3 == 0: t = t | so.After(1234 * so.frames)
                with oracle_semantics("intended"):
                    want = oracle_sink(t)
                got = so.sink(t)[0]
            except Exception as e:
                print("skip", dt.__name__, nfr, order, kind, f, str(e)[:80]); continue
            n += 1
            scale = np.abs(want).max()
            e = relerr(got, want) if np.isfinite(want).all() and scale > 0 else (0.0 if np.array_equal(got, want, equal_nan=True) else float("inf"))
            tol = 5e-6 if dt == np.float32 else 1e-6  # (north_star's bar; the long-memory designs -- cut-off 1e-4 fs, 0.4999 fs -- sit at 1e-8 ... 2e-7)
            if dt == np.float64: worst = max(worst, e)
            if not e <= tol:
                bad += 1
                print("BAD", dt.__name__, nfr, "order", order, kind, "frac", f, "relerr %.3g" % e, flush=True)
print("checks", n, "bad", bad, "worst Float64 relerr %.3g" % worst)
