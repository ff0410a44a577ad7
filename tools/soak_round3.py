"""Soak of the paths round 3 added, against the oracle: batches of independent filters (Append of scenes of one or
several orders, ragged lengths, fused sine sources, ramps) into DEVICE results whose channel rows sit at random
offsets and strides (the line-aligned output pass); rates without a period at every channel-group width (the
two-outputs-per-lane resampler), through windows; maps over several arrays big enough for the background
specialisation (first plan interpreter, later plans hipRTC: same values).
python tools/soak_round3.py SEED0 SEED1"""
import sys, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import sigops_amd as so
from sigops_amd import _capi as K
from oracle_bridge import oracle_semantics, oracle_sink, relerr
bad = 0; n = 0


def check(tag, got, want, tol):
    global bad, n
    n += 1
    e = relerr(got.astype(np.float64), want.astype(np.float64)) if got.shape == want.shape else float('inf')
    print(tag, got.shape, got.dtype, '%.3g' % e, '' if e <= tol else '  <-- BAD', flush=True)
    bad += not e <= tol


def device_sink(tree, nch, dt, rng):
    """the tree into a device result whose rows start `off` elements into an allocation, `pad` elements apart"""
    m = so.nframes(tree)
    off, pad = int(rng.integers(0, 40)), int(rng.choice([0, 1, 3, 7, 16, 29]))
    tdt = torch.float64 if dt == np.float64 else torch.float32
    flat = torch.full((off + nch * (m + pad) + 64,), float('nan'), dtype=tdt, device='cuda')
    p = so.Plan(so.ToChannels(tree, nch), (m, nch), dt, (1, m + pad), True)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):  # (the second execute: graph capture or the same direct path)
        p.execute(flat.data_ptr() + off * flat.element_size(), st)
    torch.cuda.synchronize()
    p.close()
    host = flat.cpu().numpy()
    got = np.stack([host[off + c * (m + pad): off + c * (m + pad) + m] for c in range(nch)], axis=1)
    mask = np.ones(host.shape, bool)
    for c in range(nch):
        mask[off + c * (m + pad): off + c * (m + pad) + m] = False
    assert np.all(np.isnan(host[mask])), "wrote outside the result's rows"
    return got


def one_seed(seed):
    global n, bad
    rng = np.random.default_rng(50000 + seed)
    dt = np.float32 if rng.random() < 0.4 else np.float64
    tolf = 3e-6 if dt == np.float32 else 1e-9
    nch = int(rng.choice([1, 2, 3, 5, 8]))
    fs = 44.1 * so.kHz
    mk = lambda m, c=nch: so.Signal(np.asfortranarray(rng.standard_normal((m, c)).astype(dt)), fs)
    # ---- a batch: scenes under an Append
    kids = []
    orders = rng.choice([2, 4, 5, 6, 9], size=int(rng.integers(1, 3)), replace=False)
    for k in range(int(rng.integers(2, 9))):
        m = int(rng.choice([3, 17, 64, 65, 1000, 4097, 20000, 70001, 150000]))
        x = mk(m)
        if rng.random() < 0.4:
            x = so.Mix(so.Signal(so.sin, fs, ω=(200.0 + 31 * k) * so.Hz) | so.Until(m * so.frames), x)
        elif rng.random() < 0.2:
            x = so.Amplify(x, so.Signal(so.sin, fs, ω=3 * so.Hz) | so.Until(m * so.frames))
        kind = rng.choice(["lp", "hp", "bp", "bs"])
        o = int(rng.choice(orders))
        f = (so.Filt(so.Lowpass, 3 * so.kHz, order=o) if kind == "lp" else so.Filt(so.Highpass, 300 * so.Hz, order=o) if kind == "hp"
             else so.Filt(so.Bandpass, 1 * so.kHz, 4 * so.kHz, order=o) if kind == "bp" else so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz, order=o))
        x = x | f
        if rng.random() < 0.5 and m > 2000:
            x = x | so.Ramp(5 * so.ms)
        if rng.random() < 0.2 and m > 5000:
            x = x | so.After(int(rng.integers(1, 3000)) * so.frames)
        kids.append(x)
    tree = so.Append(*kids)
    with oracle_semantics("intended"):
        want = oracle_sink(tree)
    check('batch %d scenes orders %s' % (len(kids), list(orders)), device_sink(tree, nch, dt, rng), want, tolf)
    # ---- a rate without a period, through a window
    m = int(rng.integers(30_000, 400_000))
    nc = int(rng.choice([1, 2, 3, 4, 6, 8, 16]))
    rate = float(rng.choice([np.pi / 3, np.sqrt(2), 0.7234567, 1.0001, 1 / np.e, 0.19, 2.718281828]))
    x = mk(m, nc)
    t = x | so.ToFramerate(44.1 * rate * so.kHz)
    if rng.random() < 0.5:
        t = t | so.After(int(rng.integers(1, 5000)) * so.frames)
    if rng.random() < 0.3:
        t = t | so.Until(int(rng.integers(1000, 20000)) * so.frames)
    with oracle_semantics("intended"):
        want = oracle_sink(t)
    check('rate x%.6g %d ch' % (rate, nc), device_sink(t, nc, dt, rng), want, 5e-6 if dt == np.float32 else 1e-9)
    # ---- a big map over several arrays: twice (interpreter, then the specialised kernel)
    if seed % 3 == 0:
        m = int(rng.integers(1_100_000, 1_600_000))
        a, b, c = mk(m, 4), mk(m, 4), mk(m, 4)
        t = so.Amplify(so.Mix(a, b), c) if rng.random() < 0.5 else so.Mix(a, so.Amplify(b, 0.25), c)
        want = oracle_sink(t)
        g1 = so.sink(t, so.Array)
        K.lib().so_rtc_wait_idle()
        g2 = so.sink(t, so.Array)
        check('big map first', g1, want, 1e-6 if dt == np.float32 else 1e-12)
        n += 1
        if not np.array_equal(g1, g2):
            bad += 1
            print('big map: first and second plan differ  <-- BAD', flush=True)


def run(seed0, seed1):
    """(checks, bad) over the seeds [seed0, seed1)"""
    global n, bad
    n = bad = 0
    for seed in range(seed0, seed1):
        one_seed(seed)
    return n, bad


if __name__ == "__main__":
    print('checks %d bad %d' % run(int(sys.argv[1]), int(sys.argv[2])))
