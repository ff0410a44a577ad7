"""Soak: device-resident leaves with every stride shape torch can produce (planar, interleaved, strided frames,
channel subsets, offsets) and device results with padded pitches, through the stateful paths, against the oracle
on the same values.  python tools/soak_device_leaves.py SEED0 SEED1"""
import sys, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import sigops_amd as so
from oracle_bridge import oracle_sink, relerr
bad = 0; n = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(45000 + seed)
    nch = int(rng.choice([1, 2, 3, 8])); N = int(rng.integers(50_000, 400_000)); dt = torch.float32 if rng.random() < 0.4 else torch.float64
    base = torch.from_numpy(rng.standard_normal((nch + 2, 3 * N + 7))).to("cuda", dt)
    views = {
        'planar': base[:nch, :N].t(),
        'planar offset': base[1:nch + 1, 5:N + 5].t(),
        'interleaved': base[:nch, :N].t().contiguous(),
        'frame stride 3': base[:nch, 0:3 * N:3].t(),
        'interleaved stride 2': base[:nch, :2 * N].t().contiguous()[::2],
    }
    k = int(rng.integers(0, 4))
    for name, v in views.items():
        host = np.asfortranarray(v.cpu().numpy())
        def pipe(s):
            if k == 0: return s | so.ToFramerate(48 * so.kHz)
            if k == 1: return s | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz)
            if k == 2: return s | so.Amplify(so.Signal(so.sin, ω=5 * so.Hz)) | so.Until(N * so.frames) | so.ToFramerate(48 * so.kHz) | so.Filt(so.Lowpass, 4 * so.kHz)
            return so.Mix(s, 0.25) | so.Ramp(10 * so.ms)
        want = oracle_sink(pipe(so.Signal(host, 44.1 * so.kHz)))
        t = pipe(so.Signal(v, 44.1 * so.kHz))
        M = want.shape[0]
        out = torch.full((nch, M + 11), float("nan"), dtype=torch.float32 if want.dtype == np.float32 else torch.float64, device="cuda")
        so.sink_into(out.t()[:M], t)
        n += 1
        e = relerr(out[:, :M].t().cpu().numpy(), want)
        ok = e <= (2e-6 if want.dtype == np.float32 else 1e-9) and bool(torch.isnan(out[:, M:]).all())
        if not ok: print('BAD', seed, name, k, nch, N, str(dt), '%.3g' % e, flush=True); bad += 1
print('checks', n, 'bad', bad)
