"""Soak: other paths at LONG sizes against the oracle -- narrowing stores, interleaved results, Append of long
filtered children (window aliasing), Float32 Normpower, deep windows.  python tools/soak_long_misc.py SEED0 SEED1"""
import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import sigops_amd as so
from oracle_bridge import oracle_sink, oracle_semantics, relerr
bad = 0; n = 0
def check(tag, got, want, tol):
    global bad, n
    n += 1
    e = relerr(got.astype(np.float64), want.astype(np.float64)) if got.shape == want.shape else float('inf')
    ok = e <= tol
    print(tag, got.shape, got.dtype, '%.3g' % e, '' if ok else '  <-- BAD', flush=True)
    bad += not ok
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(50000 + seed)
    nch = int(rng.choice([1, 2, 8])); fs = float(rng.choice([16000, 44100, 48000]))
    N = int(rng.integers(600_000, 1_500_000))
    x64 = np.asfortranarray(rng.standard_normal((N, nch)))
    # a) Float64 pipeline ending in an IIR, Float32 result (rounded in the IIR's own store)
    t = so.Mix(so.Signal(so.sin, ω=0.01 * fs * so.Hz), so.Signal(x64, fs * so.Hz)) | so.Until(N * so.frames) | so.Filt(so.Bandstop, 0.02 * fs * so.Hz, 0.05 * fs * so.Hz)
    want = oracle_sink(t)
    res = np.empty((N, nch), dtype=np.float32, order="F"); so.sink_into(res, t)
    check('a narrow', res, want.astype(np.float32), 1e-7)
    # b) interleaved host result of a pointwise tree
    t = so.Signal(x64, fs * so.Hz) | so.Amplify(so.Signal(so.sin, ω=3 * so.Hz)) | so.Until(N * so.frames) | so.Ramp(0.2 * so.s)
    res = np.empty((N, nch), dtype=np.float64, order="C"); so.sink_into(res, t)
    check('b interleaved', res, oracle_sink(t), 1e-12)
    # c) Append of long filtered children (window aliasing) with ramps
    cuts = sorted(rng.integers(100_000, N - 100_000, 2))
    while cuts[1] - cuts[0] < 20_000:  # (a middle child shorter than its 10 ms ramps is undefined in the reference: the oracle refuses it)
        cuts = sorted(rng.integers(100_000, N - 100_000, 2))
    kids = [so.Signal(np.asfortranarray(x64[a:b]), fs * so.Hz) | so.Filt(so.Lowpass, 0.1 * fs * so.Hz) | so.Ramp(10 * so.ms)
            for a, b in ((0, cuts[0]), (cuts[0], cuts[1]), (cuts[1], N))]
    t = so.Append(*kids)
    with oracle_semantics("intended"):
        want = oracle_sink(t)
    check('c append', so.sink(t, so.Array), want, 1e-9)
    # d) Float32 Normpower (bit-equal reduction order)
    x32 = np.asfortranarray(x64.astype(np.float32))
    t = so.Signal(x32, fs * so.Hz) | so.Normpower
    n += 1
    eq = np.array_equal(so.sink(t, so.Array), oracle_sink(t))
    print('d normpower f32 equal', eq, '' if eq else '  <-- BAD', flush=True); bad += not eq
    # e) a deep window of resample -> filter
    t = so.Signal(x64, fs * so.Hz) | so.ToFramerate((48000 if fs != 48000 else 44100) * so.Hz) | so.Filt(so.Highpass, 200 * so.Hz)
    M = so.nframes(t); a = int(rng.integers(M // 2, M - 50_000)); m = int(rng.integers(1000, M - a))
    want = oracle_sink(t)[a:a + m]
    check('e window', so.sink(t | so.After(a * so.frames) | so.Until(m * so.frames), so.Array), want, 1e-9)
print('checks', n, 'bad', bad)
