"""Soak: more paths at LONG sizes against the oracle -- interleaved / strided leaves through K1, rates without a
period (K3t + accumulator replay without a closed form), many channels (K3r + IIR), Float32 everywhere.
python tools/soak_long_more.py SEED0 SEED1"""
import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import sigops_amd as so
from oracle_bridge import oracle_sink, relerr
bad = 0; n = 0
def check(tag, got, want, tol):
    global bad, n
    n += 1
    e = relerr(got.astype(np.float64), want.astype(np.float64)) if got.shape == want.shape else float('inf')
    print(tag, got.shape, got.dtype, '%.3g' % e, '' if e <= tol else '  <-- BAD', flush=True)
    bad += not e <= tol
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(30000 + seed)
    nch = int(rng.choice([2, 3, 8])); dt = np.float32 if rng.random() < 0.5 else np.float64
    N = int(rng.integers(500_000, 1_200_000))
    tolf = 2e-6 if dt == np.float32 else 1e-9
    # interleaved (C-order) and strided leaves
    xi = rng.standard_normal((N, nch)).astype(dt)                      # row-major: frames interleaved
    t = so.Signal(xi, 44.1 * so.kHz) | so.Amplify(so.Signal(so.sin, ω=7 * so.Hz)) | so.Until(N * so.frames) | so.Ramp(0.1 * so.s)
    check('il leaf -> planar', so.sink(t, so.Array), oracle_sink(t), tolf)
    res = np.empty((N, nch), dtype=dt, order="C"); so.sink_into(res, t)
    check('il leaf -> il result', res, oracle_sink(t), tolf)
    xs = np.asfortranarray(rng.standard_normal((2 * N, nch)).astype(dt))[::2]   # frame stride 2
    t = so.Mix(so.Signal(xs, 44.1 * so.kHz), 0.25) | so.Filt(so.Lowpass, 5 * so.kHz)
    check('strided leaf -> filt', so.sink(t, so.Array), oracle_sink(t), tolf)
    # a rate without a period
    rate = float(rng.choice([np.pi / 3, np.sqrt(2), 0.7234567, 1.0001]))
    x = so.Signal(np.asfortranarray(rng.standard_normal((N // 2, nch)).astype(dt)), 44.1 * so.kHz)
    t = x | so.ToFramerate(44.1 * rate * so.kHz)
    check('irrational x%.6g' % rate, so.sink(t, so.Array), oracle_sink(t), tolf)
    t = x | so.Amplify(so.Signal(so.sin, ω=2 * so.Hz)) | so.Until((N // 2) * so.frames) | so.ToFramerate(44.1 * rate * so.kHz) | so.Filt(so.Highpass, 100 * so.Hz)
    check('irrational fused', so.sink(t, so.Array), oracle_sink(t), tolf)
    if seed % 4 == 0:  # many channels: config 5's shape
        xm = so.Signal(np.asfortranarray(rng.random((200_000, 128)).astype(dt)), 44.1 * so.kHz)
        t = xm | so.Filt(so.Lowpass, 4 * so.kHz) | so.ToFramerate(16 * so.kHz)
        check('128 ch slab', so.sink(t, so.Array), oracle_sink(t), tolf)
print('checks', n, 'bad', bad)
