"""Soak: plans with many stages on several streams, replayed through the captured HIP graph, at long sizes:
every execute (direct, capture, replays; same and alternating result buffers) against the oracle.
python tools/soak_graph_replay.py SEED0 SEED1"""
import sys, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import sigops_amd as so
from oracle_bridge import oracle_sink, oracle_semantics, relerr
bad = 0; n = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(20000 + seed)
    nch = int(rng.choice([1, 2, 4])); K = int(rng.integers(3, 9)); fs = 44100.0
    kids = []
    for k in range(K):
        m = int(rng.integers(100_000, 400_000))
        x = so.Signal(np.asfortranarray(rng.standard_normal((m, nch))), fs * so.Hz)
        c = int(rng.integers(0, 4))
        if c == 0: kids.append(so.Mix(so.Signal(so.sin, ω=(300 + 20 * k) * so.Hz), x) | so.Until(m * so.frames) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.Ramp(10 * so.ms))
        elif c == 1: kids.append(x | so.Filt(so.Lowpass, 3 * so.kHz))
        elif c == 2: kids.append(x | so.Amplify(0.5))
        else: kids.append(x | so.ToFramerate(48 * so.kHz) | so.ToFramerate(44.1 * so.kHz) if False else x | so.Filt(so.Highpass, 200 * so.Hz) | so.Amplify(so.Signal(so.sin, ω=1 * so.Hz)) | so.Until(m * so.frames))
    tree = so.ToChannels(so.Append(*kids), nch) if nch > 1 else so.Append(*kids)
    with oracle_semantics("intended"):
        want = oracle_sink(tree)
    M = want.shape[0]
    plan = so.Plan(tree, (M, nch), np.float64, (1, M + 3), True, device=0)
    st = torch.cuda.current_stream().cuda_stream
    bufs = [torch.full((nch, M + 3), float("nan"), dtype=torch.float64, device="cuda") for _ in range(2)]
    errs = []
    for it, b in enumerate([0, 0, 0, 0, 1, 1, 1, 0, 1, 0, 0, 0]):
        bufs[b].fill_(float("nan"))
        plan.execute(bufs[b].data_ptr(), st)
        torch.cuda.synchronize()
        errs.append(relerr(bufs[b][:, :M].t().cpu().numpy(), want))
    plan.close(); n += 1
    ok = max(errs) <= 1e-9
    print(seed, nch, K, M, 'max relerr %.3g' % max(errs), '' if ok else '  <-- BAD %s' % ['%.2g' % e for e in errs], flush=True)
    bad += not ok
print('plans', n, 'bad', bad)
