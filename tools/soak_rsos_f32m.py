"""Soak of the fused resampler + IIR kernel's Float32-MFMA form (k_rsos F32M): every pair of the eleven audio rates of
`rate_matrix.py` x three channel counts, a Float32 array resampled WITH A FILTER BEHIND (alternately a Float32 signal all the
way and the Float64 signal `Mix(sine, array)` into a Float32 result), against the oracle and against the engine's own Float64
products (SIGOPS_RSOS_NO_F32MFMA=1).  Records the maxima: python tools/soak_rsos_f32m.py [SEED] > profiles/r06/relerr_maxima_rsos_f32m.json"""
import collections, json, os, sys
import numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import sigops_amd as so
from oracle_bridge import oracle_sink, relerr

rates = [8.0, 11.025, 16.0, 22.05, 24.0, 32.0, 44.1, 48.0, 88.2, 96.0, 192.0]
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(62000 + seed)
os.environ["SIGOPS_RSOS_MINGROUPS"] = "1"


def into_f32(x):
    n = so.nframes(x)
    res = np.empty((n, x.nch), dtype=np.float32, order="F")
    so.sink_into(res, x)
    return res


def kernels(x, nch):
    n = so.nframes(x)
    p = so.Plan(so.ToChannels(x, nch), (n, nch), np.float32, (1, n), False)
    names = "+".join(s["name"] for s in p.steps())
    p.close()
    return names


n = bad = taken = 0
worst_oracle = worst_f64 = 0.0
worst_case = None
kern = collections.Counter()
for fi in rates:
    for fo in rates:
        if fi == fo:
            continue
        for nch in (8, 4, 2):
            N = int(rng.integers(150_000, 300_000))
            kind = n % 4
            d = rng.standard_normal((N, nch)) * 0.5
            if kind == 2:
                d = 1.0 + 1e-3 * d  # DC + small detail
            if kind == 3:
                d = (rng.random((N, nch)) < 1e-3) * 1.0  # clicks
            src = so.Signal(np.asfortranarray(d.astype(np.float32)), fi * so.kHz)
            cut = min(fi, fo) * 0.2
            if n % 2 == 0:
                tree = src | so.ToFramerate(fo * so.kHz) | so.Filt(so.Lowpass, cut * so.kHz)
            else:
                tree = so.Mix(so.Signal(so.sin, ω=0.3 * cut * so.kHz), src) | so.Until(N * so.frames) | so.Filt(so.Lowpass, cut * so.kHz) | so.ToFramerate(fo * so.kHz)
            want = oracle_sink(tree).astype(np.float32)
            os.environ.pop("SIGOPS_RSOS_NO_F32MFMA", None)
            names = kernels(tree, nch)
            got = into_f32(tree)
            os.environ["SIGOPS_RSOS_NO_F32MFMA"] = "1"
            ref = into_f32(tree)
            os.environ.pop("SIGOPS_RSOS_NO_F32MFMA", None)
            n += 1
            kern[names] += 1
            e = float(relerr(got, want)) if got.shape == want.shape else float("inf")
            e2 = float(relerr(got, ref))
            if not np.array_equal(got, ref):
                taken += 1
            if e > worst_oracle:
                worst_oracle, worst_case = e, [fi, fo, nch, N, kind, names]
            worst_f64 = max(worst_f64, e2)
            if not (e <= 1e-6 and e2 <= 3e-7):
                bad += 1
                print("BAD", fi, fo, nch, N, kind, names, "%.3g %.3g" % (e, e2), file=sys.stderr, flush=True)
print(json.dumps({"cases": n, "bad": bad, "differ_from_f64_products": taken, "worst_vs_oracle": worst_oracle, "worst_case": worst_case,
                  "worst_vs_f64_products": worst_f64, "gate_vs_oracle": 1e-6, "gate_vs_f64_products": 3e-7, "seed": seed, "kernels": dict(kern)}))
