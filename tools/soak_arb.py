"""Soak of k_resample_arb (rates without a period, persistent kernel) and of the fused kernel's Float32 sources:
random rates / channel counts / lengths / windows, the engine against itself with the kernel switched off
(SIGOPS_RS_NOARB / SIGOPS_NO_RSOS: the tiled kernel, K3 + K2) and, every few cases, against the CPU oracle.

    python3 tools/soak_arb.py [BASE_SEED] [CASES]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sigops_amd as so
from oracle_bridge import oracle_sink, relerr

os.environ.setdefault("SIGOPS_ARB_MIN", "1")  # (the persistent kernel for every length it can run, not only where it pays)
base = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200


class env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


worst = {"arb_vs_tiled": 0.0, "arb_vs_oracle": 0.0, "rsos32_vs_two": 0.0, "rsos32_vs_oracle": 0.0}
narb = 0
for i in range(cases):
    rng = np.random.default_rng(base * 100003 + i)
    nch = int(rng.choice([1, 2, 3, 4, 5, 8, 8, 16, 24]))
    n = int(rng.integers(17000, 420000))
    if nch >= 16:
        n = min(n, 120000)
    fs_in = float(rng.choice([1000.0, 44100.0, 44100.5, 48000.0, 22050.25, 8000.0]))
    ratio = float(np.exp(rng.uniform(np.log(0.3), np.log(4.0)))) * (1 + 1e-3 * rng.standard_normal())
    fs_out = fs_in * ratio
    dt = np.float32 if rng.random() < 0.35 else np.float64
    x = np.asfortranarray(rng.standard_normal((n, nch)).astype(dt))
    if rng.random() < 0.2:
        x[int(rng.integers(0, n)), int(rng.integers(0, nch))] = 0.0
    tree = so.Signal(x, fs_in * so.Hz) | so.ToFramerate(fs_out * so.Hz)
    nout = so.nframes(tree)
    if rng.random() < 0.5 and nout > 40000:
        a = int(rng.integers(1, nout - 30000))
        m = int(rng.integers(17000, nout - a))
        tree = tree | so.After(a * so.frames) | so.Until(m * so.frames)
        nout = m
    p = so.Plan(so.ToChannels(tree, nch), (nout, nch), dt, (1, nout), False)
    names = [s["name"] for s in p.steps()]
    p.close()
    got = so.sink(tree)[0]
    with env(SIGOPS_RS_NOARB=1):
        ref = so.sink(tree)[0]
    e = relerr(got, ref)
    # (Float32 results: both kernels round the same Float64 sums, which differ in their last bits: a rounding flip here and there)
    assert got.shape == ref.shape and got.dtype == dt and np.isfinite(got).all() and e < (1e-13 if dt == np.float64 else 2e-8), (i, names, e, nch, n, fs_in, fs_out)
    if "k_resample_arb" in names:
        narb += 1
        if dt == np.float64:
            worst["arb_vs_tiled"] = max(worst["arb_vs_tiled"], e)
    if i % 8 == 0:
        eo = relerr(got, oracle_sink(tree))
        assert eo < (1e-9 if dt == np.float64 else 1e-6), (i, names, eo)
        if dt == np.float64:
            worst["arb_vs_oracle"] = max(worst["arb_vs_oracle"], eo)

# Float32 sources of the fused resampler + IIR kernel
nf = 0
for i in range(max(8, cases // 8)):
    rng = np.random.default_rng(base * 7919 + i)
    nch = int(rng.choice([1, 2, 3, 4, 8, 8, 16]))
    n = int(rng.integers(150000, 600000))
    x32 = np.asfortranarray(rng.standard_normal((n, nch)).astype(np.float32))
    noise = so.Signal(x32, 44.1 * so.kHz)
    kind = int(rng.integers(0, 4))
    f = float(rng.choice([5.0, 440.0, 1000.0, 3000.5]))
    src = [lambda: so.Mix(so.Signal(so.sin, ω=f * so.Hz), noise) | so.Until(n * so.frames),
           lambda: noise | so.Amplify(so.Signal(so.sin, ω=f * so.Hz)) | so.Until(n * so.frames),
           lambda: so.ToEltype(noise, np.float64),
           lambda: so.Mix(so.Signal(so.sin, ω=f * so.Hz, ϕ=0.3), noise) | so.Until((n - 12345) * so.frames)][kind]()
    lo, hi = sorted(rng.uniform(0.2, 6.0, 2))
    tree = src | so.Filt(so.Bandstop, lo * so.kHz, (hi + 0.3) * so.kHz, order=int(rng.integers(2, 6))) | so.ToFramerate(48 * so.kHz)
    with env(SIGOPS_RSOS_MINGROUPS=1):
        nout = so.nframes(tree)
        p = so.Plan(so.ToChannels(tree, nch), (nout, nch), np.float64, (1, nout), False)
        names = [s["name"] for s in p.steps()]
        p.close()
        got = so.sink(tree)[0]
    with env(SIGOPS_NO_RSOS=1):
        ref = so.sink(tree)[0]
    e = relerr(got, ref)
    assert e < 1e-10, (i, names, e, nch, n, kind)
    if names == ["k_rsos"]:
        nf += 1
        worst["rsos32_vs_two"] = max(worst["rsos32_vs_two"], e)
    if i % 4 == 0:
        eo = relerr(got, oracle_sink(tree))
        assert eo < 1e-9, (i, names, eo)
        worst["rsos32_vs_oracle"] = max(worst["rsos32_vs_oracle"], eo)
# the fused kernel at large: rates whose period splits into blocks of 16 outputs, random cascades, channel counts,
# lengths, sources and WINDOWS (warm starts: RsSos::store_lo), every structurally possible case forced onto it
ng = 0
worst["rsos_vs_two"] = worst["rsos_vs_oracle"] = 0.0
for i in range(max(12, cases // 2)):
    rng = np.random.default_rng(base * 104729 + i)
    nch = int(rng.choice([1, 2, 3, 4, 6, 8, 8, 12, 16, 24]))
    fs_in, fs_out = [(44100.0, 48000.0), (22050.0, 48000.0), (44100.0, 96000.0), (11025.0, 48000.0), (44100.0, 48000.0),
                     (24000.0, 48000.0), (16000.0, 48000.0), (32000.0, 48000.0), (12000.0, 48000.0), (8000.0, 16000.0),
                     (48000.0, 24000.0), (48000.0, 32000.0)][int(rng.integers(0, 12))]
    n = int(rng.integers(60000, 500000) * (fs_in / 44100.0))
    if nch > 8:
        n = min(n, 150000)
    dt = np.float64
    x = np.asfortranarray(rng.standard_normal((n, nch)))
    src = so.Signal(x, fs_in * so.Hz)
    k = int(rng.integers(0, 4))
    if k == 1:
        src = so.Mix(so.Signal(so.sin, ω=float(rng.uniform(50, 3000)) * so.Hz), src) | so.Until(n * so.frames)
    elif k == 2:
        src = src | so.Amplify(so.Signal(so.sin, ω=float(rng.uniform(1, 20)) * so.Hz)) | so.Until(n * so.frames)
    elif k == 3:
        src = src | so.Amplify(float(rng.uniform(0.1, 2.0)))
    nyq = 0.5 * min(fs_in, fs_out) / 1000.0   # (the filter moves behind the resampler: reference src/filters.jl:143-148)
    kind = int(rng.integers(0, 4))
    order = int(rng.integers(1, 7 if kind < 2 else 4))
    f1 = float(rng.uniform(0.02, 0.6)) * nyq
    f2 = min(f1 + float(rng.uniform(0.05, 0.3)) * nyq, 0.95 * nyq)
    filt = [lambda s: s | so.Filt(so.Lowpass, f1 * so.kHz, order=order), lambda s: s | so.Filt(so.Highpass, f1 * so.kHz, order=order),
            lambda s: s | so.Filt(so.Bandpass, f1 * so.kHz, f2 * so.kHz, order=order),
            lambda s: s | so.Filt(so.Bandstop, f1 * so.kHz, f2 * so.kHz, order=order)][kind]
    tree = filt(src) | so.ToFramerate(fs_out * so.Hz)
    if rng.random() < 0.15:
        tree = src | so.ToFramerate(fs_out * so.Hz)   # (the resampler alone: K3's own super-periods of the small ratios)
    nout = so.nframes(tree)
    if rng.random() < 0.6 and nout > 60000:
        a = int(rng.integers(1, nout - 20000))
        m = int(rng.integers(1, nout - a))
        tree = tree | so.After(a * so.frames) | so.Until(m * so.frames)
        nout = m
    with env(SIGOPS_RSOS_MINGROUPS=1):
        p = so.Plan(so.ToChannels(tree, nch), (nout, nch), np.float64, (1, nout), False)
        names = [s["name"] for s in p.steps()]
        p.close()
        got = so.sink(tree)[0]
    with env(SIGOPS_NO_RSOS=1):
        ref = so.sink(tree)[0]
    nrm = np.linalg.norm(ref)
    e = relerr(got, ref) if nrm > 1e-200 else float(np.abs(got - ref).max())
    assert got.shape == ref.shape and e < 1e-9, (i, names, e, nch, n, fs_in, fs_out, kind, order, f1, f2)
    if "k_rsos" in names:
        ng += 1
        worst["rsos_vs_two"] = max(worst["rsos_vs_two"], e)
    if i % 6 == 0:
        eo = relerr(got, oracle_sink(tree))
        assert eo < 1e-8, (i, names, eo)
        worst["rsos_vs_oracle"] = max(worst["rsos_vs_oracle"], eo)
print({"base": base, "cases": cases, "on_k_resample_arb": narb, "fused_float32": nf, "fused_random": ng, **worst})
