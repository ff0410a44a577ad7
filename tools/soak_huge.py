"""Indexing beyond 2^31 elements: 300 M frames x 8 channels (2.4 G elements, 19 GB) of a device-resident constant
signal through Amplify -> Filt(Lowpass) -> ToFramerate(48 kHz); the steady state is known (0.5 x DC gain 1), a
short prefix is checked against the oracle, the far end by value.  python tools/soak_huge.py"""
import sys, time, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import sigops_amd as so
from oracle_bridge import oracle_sink, relerr
N, nch = 300_000_000, 8
x = torch.ones((nch, N), dtype=torch.float64, device="cuda")
x[:, ::1000] = 3.0   # (something to filter: mean 1.002)
mean = 1.0 + 2.0 / 1000
for name, pipe in (("amplify", lambda s: s | so.Amplify(0.5)),
                   ("filt", lambda s: s | so.Amplify(0.5) | so.Filt(so.Lowpass, 200 * so.Hz)),
                   ("pipeline", lambda s: s | so.Amplify(0.5) | so.Filt(so.Lowpass, 200 * so.Hz) | so.ToFramerate(48 * so.kHz))):
    t = pipe(so.Signal(x.t(), 44.1 * so.kHz))
    t0 = time.perf_counter(); out, fs = so.sink(t, "torch"); torch.cuda.synchronize(); el = time.perf_counter() - t0
    M = out.shape[0]
    print(name, 'frames', M, 'elements %.3g' % (M * nch), 'seconds %.2f' % el, flush=True)
    if name == "amplify":
        ok = bool((out[-5:, :] == 0.5).all()) and float(out[M - 1000 * 7, 3]) == 1.5 and float(out[(1 << 28) * 1 + 5, 7]) in (0.5, 1.5)
        print('  exact at the far end:', ok)
    else:
        tail = out[M - 200000:, :].double()
        print('  far-end mean %.9f (expected %.9f), max deviation from it %.3g' % (float(tail.mean()), 0.5 * mean, float((tail - 0.5 * mean).abs().max())))
        mid = out[M // 2: M // 2 + 200000, :].double()
        print('  middle mean %.9f' % float(mid.mean()))
        pre = np.asfortranarray(x[:, :60000].t().cpu().numpy())
        want = oracle_sink(pipe(so.Signal(pre, 44.1 * so.kHz)))
        k = min(want.shape[0], 40000)
        print('  prefix vs oracle relerr %.3g' % relerr(out[:k].cpu().numpy(), want[:k]))
    del out
    torch.cuda.empty_cache()
