"""Non-finite samples: nothing hangs or crashes, and a NaN in the input of a resampler stays local (the MFMA and
two-outputs-per-lane forms let it reach a few neighbours of the outputs the reference puts it in)."""
import sys; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, sigops_amd as so
from oracle_bridge import oracle_sink
rng = np.random.default_rng(5)
for dt in (np.float64, np.float32):
    x = np.asfortranarray(rng.standard_normal((200000, 8)).astype(dt))
    for pos in (0, 777, 100000, 199999):
        x[pos, pos % 8] = np.nan
    x[5000, 3] = np.inf
    s = so.Signal(x, 44.1 * so.kHz)
    for name, t in (("ToFramerate 48k (K3)", s | so.ToFramerate(48 * so.kHz)),
                    ("ToFramerate x pi/3 (K3t2)", s | so.ToFramerate(44.1 * np.pi / 3 * so.kHz)),
                    ("ToFramerate 16k", s | so.ToFramerate(16 * so.kHz)),
                    ("Filt", s | so.Filt(so.Lowpass, 3 * so.kHz)),
                    ("Mix + Ramp", so.Mix(s, 0.5) | so.Ramp(10 * so.ms)),
                    ("Normpower", s | so.Normpower)):
        got = so.sink(t)[0]
        want = oracle_sink(t)
        gn, wn = ~np.isfinite(got), ~np.isfinite(want)
        extra = int((gn & ~wn).sum()); missing = int((wn & ~gn).sum())
        both = ~gn & ~wn
        err = float(np.abs(got[both].astype(np.float64) - want[both]).max()) if both.any() else 0.0
        print(dt.__name__, name, "non-finite: engine", int(gn.sum()), "oracle", int(wn.sum()), "engine-only", extra, "oracle-only", missing, "max diff elsewhere %.3g" % err, flush=True)
