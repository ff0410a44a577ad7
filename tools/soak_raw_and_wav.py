"""Soak: raw filter objects (random stable second-order sections, zero-pole-gain, FIR) on long signals against the
oracle and SciPy, and the WAV sink at 1-2 M frames read back by SciPy.  python tools/soak_raw_and_wav.py SEED0 SEED1"""
import sys, os, tempfile, numpy as np
from scipy import signal as sps
from scipy.io import wavfile
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import sigops_amd as so
from oracle_bridge import oracle_sink, relerr
bad = 0; n = 0
def check(tag, e, tol):
    global bad, n
    n += 1
    print(tag, '%.3g' % e, '' if e <= tol else '  <-- BAD', flush=True); bad += not e <= tol
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(25000 + seed)
    nch = int(rng.choice([1, 2, 8])); N = int(rng.integers(200_000, 1_000_000)) // (2 if nch == 8 else 1)
    xd = np.asfortranarray(rng.standard_normal((N, nch)))
    x = so.Signal(xd, 44.1 * so.kHz)
    nsec = int(rng.integers(1, 11))
    sos = np.zeros((nsec, 6))
    for s in sos:  # a pole pair inside the unit circle, zeros anywhere
        r, th = rng.uniform(0.3, 0.995), rng.uniform(0.05, 3.0)
        s[3:] = [1.0, -2 * r * np.cos(th), r * r]
        z = rng.uniform(0.2, 1.2); zt = rng.uniform(0, np.pi)
        s[:3] = np.array([1.0, -2 * z * np.cos(zt), z * z]) * rng.uniform(0.3, 1.0)
    g = float(rng.uniform(0.1, 2.0))
    t = x | so.Filt(so.SecondOrderSections(sos, g))
    want = sps.sosfilt(sos, xd, axis=0) * g
    if np.isfinite(want).all() and np.abs(want).max() < 1e8:
        got = so.sink(t, so.Array)
        check('%d sos x%d vs scipy' % (seed, nsec), relerr(got, want), 1e-8)
        check('%d sos x%d vs oracle' % (seed, nsec), relerr(got, oracle_sink(t)), 1e-8)
    h = sps.firwin(int(rng.integers(2, 400)), float(rng.uniform(0.05, 0.9)))
    t = so.Filt(x, h) | so.ToFramerate(48 * so.kHz)
    check('%d fir %d taps -> resample' % (seed, len(h)), relerr(so.sink(t, so.Array), oracle_sink(t)), 1e-9)
    # WAV sink: Float32 data, interleaved IEEE-float container, read back by SciPy
    x32 = so.Signal(np.asfortranarray(xd.astype(np.float32)), 44.1 * so.kHz) | so.Ramp(10 * so.ms) | so.Amplify(np.float32(0.5))
    path = os.path.join(tempfile.gettempdir(), 'soak_%d.wav' % seed)
    so.save_signal(path, x32)
    rate, got = wavfile.read(path)
    os.remove(path)
    want = oracle_sink(x32)
    if got.ndim == 1: got = got[:, None]
    check('%d wav %s' % (seed, got.dtype), relerr(got.astype(np.float64), want.astype(np.float64)) if got.shape == want.shape and rate == 44100 else float('inf'), 1e-6)
print('checks', n, 'bad', bad)
