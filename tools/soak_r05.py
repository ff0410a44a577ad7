"""Soak of round 5's new paths against the oracle and the paths they replace.   python tools/soak_r05.py SEED0 SEED1
 a) a plain Float64 `Filt` in one pass (k_rsos with an identity resampler, forced with SIGOPS_RSOS_MINGROUPS=1): random designs
    of 1 - 6 sections, 1 - 24 channels, host arrays and device tensors at odd offsets, plain / Mix(sin) / Amplify(sin)
    sources -- against K2 (SIGOPS_NO_PLAIN_RSOS=1) 1e-10 and the oracle 1e-9;
 b) Float32 signals through the periodic resampler on the Float32 MFMA: random rate pairs of the audio rates, 4 - 16 channels
    -- against the oracle 1e-6 and the Float64 products (SIGOPS_RS_NO_F32MFMA=1) 3e-7."""
import os, sys, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import sigops_amd as so
from oracle_bridge import oracle_sink, relerr
bad = 0; n = 0; worst = {"iir_vs_k2": 0.0, "iir_vs_oracle": 0.0, "f32m_vs_oracle": 0.0, "f32m_vs_f64": 0.0}; took = {"k_rsos": 0, "other": 0}
def setenv(**kv):
    for k, v in kv.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = str(v)
def steps(x, dt=np.float64):
    nf, nc = so.nframes(x), so.nchannels(x)
    p = so.Plan(so.ToChannels(x, nc), (nf, nc), dt, (1, nf), False); names = [s["name"] for s in p.steps()]; p.close(); return names
rates = [8.0, 11.025, 16.0, 22.05, 24.0, 32.0, 44.1, 48.0, 88.2, 96.0]
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(95000 + seed)
    # ---- a) one-pass IIR
    nch = int(rng.choice([1, 2, 3, 4, 5, 8, 8, 8, 12, 16, 24])); N = int(rng.integers(40_000, 400_000)); fs = float(rng.choice([16000, 44100, 48000, 96000]))
    d = rng.standard_normal((N, nch))
    kind = rng.choice(["lp", "hp", "bp", "bs"]); order = int(rng.integers(1, 13 if kind in ("lp", "hp") else 7))
    f1 = float(rng.uniform(0.01, 0.2)) * fs; f2 = f1 + float(rng.uniform(0.02, 0.2)) * fs
    mk = {"lp": lambda s: s | so.Filt(so.Lowpass, f1 * so.Hz, order=order), "hp": lambda s: s | so.Filt(so.Highpass, f1 * so.Hz, order=order),
          "bp": lambda s: s | so.Filt(so.Bandpass, f1 * so.Hz, f2 * so.Hz, order=order), "bs": lambda s: s | so.Filt(so.Bandstop, f1 * so.Hz, f2 * so.Hz, order=order)}[kind]
    srck = rng.choice(["plain", "mix", "amp", "dev"])
    host = so.Signal(np.asfortranarray(d), fs * so.Hz)
    if srck == "dev":
        off = int(rng.integers(0, 4)); store = torch.zeros((nch, N + 7), dtype=torch.float64, device="cuda"); store[:, off:off + N] = torch.tensor(np.ascontiguousarray(d.T), device="cuda")
        leaf = so.Signal(store[:, off:off + N].t(), fs * so.Hz)
    else:
        leaf = host
    wrap = (lambda s: s) if srck in ("plain", "dev") else (lambda s: so.Mix(so.Signal(so.sin, ω=0.013 * fs * so.Hz), s) | so.Until(N * so.frames)) if srck == "mix" else \
           (lambda s: so.Amplify(s, so.Signal(so.sin, ω=3 * so.Hz)) | so.Until(N * so.frames))
    x, xh = mk(wrap(leaf)), mk(wrap(host))
    try:
        want = oracle_sink(xh)
    except Exception as e:
        print(seed, "a oracle refused:", str(e)[:60]); want = None
    if want is not None:
        setenv(SIGOPS_RSOS_MINGROUPS=1, SIGOPS_NO_PLAIN_RSOS=None)
        names = steps(xh); a = so.sink(x)[0] if srck != "dev" else so.sink(x, "torch")[0].cpu().numpy()
        setenv(SIGOPS_NO_PLAIN_RSOS=1)
        b = so.sink(xh)[0]
        setenv(SIGOPS_NO_PLAIN_RSOS=None, SIGOPS_RSOS_MINGROUPS=None)
        took["k_rsos" if "k_rsos" in names else "other"] += 1
        e1, e2 = relerr(a, b), relerr(a, want); n += 1
        worst["iir_vs_k2"] = max(worst["iir_vs_k2"], e1); worst["iir_vs_oracle"] = max(worst["iir_vs_oracle"], e2)
        scale = max(1e-300, relerr(b, want))  # (ill-conditioned designs: K2 itself is that far from the oracle)
        if not (e1 <= max(1e-10, 10 * scale) and e2 <= max(1e-9, 10 * scale)):
            print(seed, "a BAD", kind, order, nch, N, fs, srck, names, "%.3g %.3g (k2 vs oracle %.3g)" % (e1, e2, scale), flush=True); bad += 1
    # ---- b) Float32 MFMA
    fi, fo = rng.choice(rates, 2, replace=False); nch = int(rng.choice([4, 8, 12, 16])); N = int(rng.integers(40_000, 200_000))
    x32 = np.asfortranarray((rng.standard_normal((N, nch)) * float(rng.choice([0.01, 0.5, 20.0])) + float(rng.choice([0.0, 0.0, 3.0]))).astype(np.float32))
    t = so.Signal(x32, float(fi) * so.kHz) | so.ToFramerate(float(fo) * so.kHz)
    want = oracle_sink(t)
    setenv(SIGOPS_RS_NO_F32MFMA=None); a = so.sink(t)[0]
    setenv(SIGOPS_RS_NO_F32MFMA=1); b = so.sink(t)[0]
    setenv(SIGOPS_RS_NO_F32MFMA=None)
    e1, e2 = relerr(a, want), relerr(a, b); n += 1
    worst["f32m_vs_oracle"] = max(worst["f32m_vs_oracle"], e1); worst["f32m_vs_f64"] = max(worst["f32m_vs_f64"], e2)
    if not (e1 <= 1e-6 and e2 <= 3e-7):
        print(seed, "b BAD", fi, fo, nch, N, "%.3g %.3g" % (e1, e2), flush=True); bad += 1
print("checks", n, "bad", bad, {k: float("%.3g" % v) for k, v in worst.items()}, took)
