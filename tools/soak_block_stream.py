"""Soak: so.BlockStream with random pipelines, rates, channel counts and ragged block sizes against the one-shot
sink of the same pipeline over the whole input.  python tools/soak_block_stream.py SEED0 SEED1 [SCALE]"""
import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import sigops_amd as so
from oracle_bridge import relerr
RATES = [8000.0, 16000.0, 22050.0, 44100.0, 48000.0]
bad = 0; n = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(80000 + seed)
    fs = float(rng.choice(RATES)); fo = float(rng.choice([r for r in RATES if r != fs]))
    nch = int(rng.choice([1, 2, 3, 8])); dt = np.float64 if rng.random() < 0.8 else np.float32
    N = int(rng.integers(150000, 500000)) * int(sys.argv[3] if len(sys.argv) > 3 else 1)
    x = rng.standard_normal((N, nch)).astype(dt)
    k = int(rng.integers(0, 6))
    lo, hi = 0.05 * min(fs, fo), 0.2 * min(fs, fo)
    pipes = [
        lambda s: s | so.Filt(so.Lowpass, hi * so.Hz),
        lambda s: s | so.ToFramerate(fo * so.Hz),
        lambda s: s | so.Filt(so.Bandstop, lo * so.Hz, hi * so.Hz) | so.ToFramerate(fo * so.Hz),
        lambda s: so.Mix(so.Signal(so.sin, ω=0.01 * fs * so.Hz), s) | so.Until(so.nframes(s) * so.frames) | so.Filt(so.Highpass, lo * so.Hz) | so.Amplify(0.5),
        lambda s: s | so.Amplify(so.Signal(so.sin, ω=3 * so.Hz)) | so.Until(so.nframes(s) * so.frames) | so.ToFramerate(fo * so.Hz) | so.Filt(so.Lowpass, hi * so.Hz),
        lambda s: s | so.After(1234 * so.frames) | so.Filt(so.Lowpass, hi * so.Hz) | so.ToChannels(1) if nch > 1 else s | so.After(1234 * so.frames) | so.Filt(so.Lowpass, hi * so.Hz),
    ]
    pipe = pipes[k]
    try:
        whole = so.sink(pipe(so.Signal(np.asfortranarray(x), fs * so.Hz)), so.Array)
    except so.ErrorException as e:
        continue
    bs = so.BlockStream(pipe, fs * so.Hz, nch=nch, dtype=dt, history=1 << 16)
    outs, pos = [], 0
    try:
        while pos < N:
            m = int(min(N - pos, rng.choice([1, 7, 1000, 4096, 30000, 100000]) * (int(sys.argv[3]) if len(sys.argv) > 3 else 1)))
            outs.append(bs.push(x[pos:pos + m]).cpu().numpy()); pos += m
        outs.append(bs.finish().cpu().numpy())
    except Exception as e:
        print('ERROR', seed, k, fs, fo, nch, str(e)[:160]); bad += 1; continue
    got = np.concatenate(outs, axis=0); n += 1
    e = relerr(got, whole) if got.shape == whole.shape else float('inf')
    if not e <= (2e-6 if dt == np.float32 else 1e-9):
        print('VALUE', seed, k, fs, fo, nch, dt.__name__, got.shape, whole.shape, '%.3g' % e); bad += 1
print('streams', n, 'bad', bad)
