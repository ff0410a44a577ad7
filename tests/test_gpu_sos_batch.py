"""GPU parity for batched IIR launches: independent `Filt` stages of one shape -- the scenes under an `Append`
(reference src/appending.jl:59-76: every child is evaluated on its own, filter state included,
src/filters.jl:252-255) -- run their three passes inside ONE launch per pass (k_sos_tiled_batch /
k_sos_scan_batch).  The arithmetic of a member is that of its own launch: with the members' own chunk geometry
(SIGOPS_SOS_BATCH_KEEPCHUNKS) the result is bit-identical to the unbatched one (SIGOPS_SOS_NOBATCH); with the
chunks cut for the batch as a whole it agrees to rounding; all against the oracle."""
import os

import numpy as np
import pytest

import sigops_amd as so
from oracle_bridge import oracle_semantics, oracle_sink, relerr

pytestmark = pytest.mark.gpu


def _noise(rng, n, nch, dt=np.float64):
    return np.asfortranarray(rng.standard_normal((n, nch)).astype(dt))


def _both(tree, **kw):
    os.environ.pop("SIGOPS_SOS_NOBATCH", None)
    got = so.sink(tree, so.Array, **kw)
    os.environ["SIGOPS_SOS_NOBATCH"] = "1"
    try:
        ref = so.sink(tree, so.Array, **kw)
    finally:
        os.environ.pop("SIGOPS_SOS_NOBATCH", None)
    assert got.shape == ref.shape and got.dtype == ref.dtype
    # (the members of a batch are cut into chunks for the batch as a whole: same recurrence, other chunk borders)
    assert relerr(got, ref) <= (1e-12 if got.dtype == np.float64 else 1e-6), "batching the launches changed the result"
    os.environ["SIGOPS_SOS_BATCH_KEEPCHUNKS"] = "1"
    try:
        same = so.sink(tree, so.Array, **kw)
    finally:
        os.environ.pop("SIGOPS_SOS_BATCH_KEEPCHUNKS", None)
    assert np.array_equal(same, ref), "a member's arithmetic differs from that of its own launch"
    return got


def _check(tree, tol=1e-9):
    got = _both(tree)
    with oracle_semantics("intended"):
        want = oracle_sink(tree)
    assert got.shape == want.shape
    assert relerr(got, want) <= tol
    return got


def _steps(tree, nch, dt=np.float64):
    n = so.nframes(tree)
    import torch
    out = torch.empty((nch, n), dtype=torch.float64 if dt == np.float64 else torch.float32, device="cuda")
    p = so.Plan(so.ToChannels(tree, nch), (n, nch), dt, (1, n), True)
    p.set_profiling(True)
    p.execute(out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    names = [(s["name"], s["launches"]) for s in p.steps()]
    p.close()
    return names


@pytest.mark.parametrize("nch", [1, 2, 5])
@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_scenes_of_ragged_lengths(nch, dt):
    """chunked members, one-chunk members (shorter than a chunk) and a three-frame one in one batch"""
    rng = np.random.default_rng(100 + nch)
    kids = [so.Signal(_noise(rng, n, nch, dt), 44.1 * so.kHz) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz)
            for n in (50001, 17, 7777, 123456, 3, 64, 65, 30000)]
    _check(so.Append(*kids), tol=1e-6 if dt == np.float32 else 1e-9)


def test_the_batch_is_one_step_of_four_launches():
    rng = np.random.default_rng(5)
    kids = [so.Signal(_noise(rng, 40000 + k, 2), 44.1 * so.kHz) | so.Filt(so.Lowpass, 3 * so.kHz) for k in range(6)]
    names = _steps(so.Append(*kids), 2)
    assert ("k_sos_batch", 4) in names  # (state pass, scan, output pass, and the NaN fill behind a non-finite chunk)
    assert not any(n == "k_sos" for n, _ in names)


def test_different_filters_of_one_order_share_a_batch():
    rng = np.random.default_rng(6)
    kids = []
    for k, (lo, hi) in enumerate([(0.5, 2.0), (1.0, 4.0), (0.2, 0.9), (3.0, 8.0)]):
        kids.append(so.Signal(_noise(rng, 30000 + 999 * k, 2), 44.1 * so.kHz) | so.Filt(so.Bandpass, lo * so.kHz, hi * so.kHz))
    _check(so.Append(*kids))


def test_orders_are_batched_apart():
    """sections differ: two batches (and a lone stage keeps its own launches)"""
    rng = np.random.default_rng(7)
    kids = []
    for k in range(3):
        kids.append(so.Signal(_noise(rng, 20000 + k, 2), 44.1 * so.kHz) | so.Filt(so.Lowpass, 2 * so.kHz, order=4))
    for k in range(3):
        kids.append(so.Signal(_noise(rng, 25000 + k, 2), 44.1 * so.kHz) | so.Filt(so.Highpass, 1 * so.kHz, order=6))
    kids.append(so.Signal(_noise(rng, 22000, 2), 44.1 * so.kHz) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz, order=7))
    tree = so.Append(*kids)
    _check(tree)
    names = [n for n, _ in _steps(tree, 2)]
    assert names.count("k_sos_batch") == 2 and names.count("k_sos") == 1


def test_scenes_with_a_sine_formed_in_the_filter():
    """config 4's scene: Mix(sin, noise) |> Filt |> Ramp, the sine formed in the filter's loads (a batch member
    with a fused source); tones differ per scene"""
    rng = np.random.default_rng(8)
    kids = []
    for k in range(6):
        n = 30000 + 1001 * k
        nz = so.Signal(_noise(rng, n, 2), 44.1 * so.kHz)
        tone = so.Signal(so.sin, 44.1 * so.kHz, ω=(500.0 + 25 * k) * so.Hz) | so.Until(n * so.frames)
        kids.append(so.Mix(tone, nz) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.Ramp(10 * so.ms))
    tree = so.Append(*kids)
    _check(tree, tol=1e-10)
    assert any(n == "k_sos_batch" for n, _ in _steps(tree, 2))


def test_members_that_start_inside_their_array():
    """`After` moves a member's first frame into its array (in_offset) and `Until` cuts it"""
    rng = np.random.default_rng(9)
    kids = []
    for k in range(4):
        x = so.Signal(_noise(rng, 50000, 2), 44.1 * so.kHz)
        kids.append(x | so.After((100 + 37 * k) * so.frames) | so.Until((20000 + k) * so.frames) | so.Filt(so.Lowpass, 4 * so.kHz))
    _check(so.Append(*kids))


def test_batch_members_mixed_together():
    """members whose outputs are read by a later launch (no windows of the result): Mix of filtered arrays"""
    rng = np.random.default_rng(10)
    a = so.Signal(_noise(rng, 40000, 2), 44.1 * so.kHz) | so.Filt(so.Lowpass, 1 * so.kHz)
    b = so.Signal(_noise(rng, 40000, 2), 44.1 * so.kHz) | so.Filt(so.Highpass, 5 * so.kHz)
    c = so.Signal(_noise(rng, 40000, 2), 44.1 * so.kHz) | so.Filt(so.Bandpass, 2 * so.kHz, 3 * so.kHz, order=5)
    _check(so.Mix(a, b, c))
    _check(so.Amplify(a, b))


def test_batched_plan_follows_its_result_and_its_arrays():
    """the descriptor table holds result and array pointers: executes into two results alternate (graph replay
    and direct launches), then an array is replaced (so_plan_set_array)"""
    import torch
    rng = np.random.default_rng(12)
    n, nsc = 30000, 5
    arrays = [torch.from_numpy(np.ascontiguousarray(_noise(rng, n, 2).T)).cuda() for _ in range(nsc)]
    sigs = [so.Signal(a.t(), 44.1 * so.kHz) for a in arrays]
    tree = so.Append(*[s | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.Ramp(5 * so.ms) for s in sigs])
    total = so.nframes(tree)
    with oracle_semantics("intended"):
        want = oracle_sink(so.Append(*[so.Signal(np.asfortranarray(a.t().cpu().numpy()), 44.1 * so.kHz)
                                       | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.Ramp(5 * so.ms) for a in arrays]))
    p = so.Plan(so.ToChannels(tree, 2), (total, 2), np.float64, (1, total), True)
    st = torch.cuda.current_stream().cuda_stream
    A = torch.zeros((2, total), dtype=torch.float64, device="cuda")
    B = torch.zeros((2, total), dtype=torch.float64, device="cuda")
    for dst in (A, A, A, B, A, B, B, B, A):
        dst.zero_()
        p.execute(dst.data_ptr(), st)
        torch.cuda.synchronize()
        assert relerr(np.asfortranarray(dst.t().cpu().numpy()), want) <= 1e-9
    # scene 2 gets another array: the table follows it
    repl = torch.from_numpy(np.ascontiguousarray(_noise(rng, n, 2).T)).cuda()
    p.set_array(2, repl.t())
    hosts = [np.asfortranarray((repl if k == 2 else a).t().cpu().numpy()) for k, a in enumerate(arrays)]
    with oracle_semantics("intended"):
        want2 = oracle_sink(so.Append(*[so.Signal(h, 44.1 * so.kHz) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz)
                                        | so.Ramp(5 * so.ms) for h in hosts]))
    for dst in (A, A, A, B):
        dst.zero_()
        p.execute(dst.data_ptr(), st)
        torch.cuda.synchronize()
        assert relerr(np.asfortranarray(dst.t().cpu().numpy()), want2) <= 1e-9
    p.close()


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("off,pad", [(0, 0), (1, 0), (3, 5), (0, 7), (5, 15), (2, 29)])
def test_result_rows_off_the_cache_line(dt, off, pad):
    """the output pass starts its tile steps on the result's 128-byte lines (SosGeom::align_rows): a device result
    whose channel rows start `off` elements into an allocation at a stride of n + pad -- any alignment the columns
    of a Julia Array can have -- against the oracle, and bit for bit against the unaligned steps
    (SIGOPS_SOS_NOALIGN): the chunk borders, and with them every rounding, do not move"""
    import torch
    rng = np.random.default_rng(40 + off + pad)
    n, nch = 70001, 5
    x = _noise(rng, n, nch, dt)
    tree = so.Signal(x, 44.1 * so.kHz) | so.Filt(so.Bandpass, 1 * so.kHz, 3 * so.kHz) | so.After(100 * so.frames)
    with oracle_semantics("intended"):
        want = oracle_sink(tree)
    m = want.shape[0]
    tdt = torch.float64 if dt == np.float64 else torch.float32
    res = []
    for env in (None, "1"):
        if env:
            os.environ["SIGOPS_SOS_NOALIGN"] = env
        try:
            flat = torch.full((off + nch * (m + pad) + 64,), float("nan"), dtype=tdt, device="cuda")
            p = so.Plan(so.ToChannels(tree, nch), (m, nch), dt, (1, m + pad), True)
            p.execute(flat.data_ptr() + off * flat.element_size(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            p.close()
        finally:
            os.environ.pop("SIGOPS_SOS_NOALIGN", None)
        host = flat.cpu().numpy()
        got = np.stack([host[off + c * (m + pad): off + c * (m + pad) + m] for c in range(nch)], axis=1)
        # nothing outside the rows was touched
        mask = np.ones(host.shape, bool)
        for c in range(nch):
            mask[off + c * (m + pad): off + c * (m + pad) + m] = False
        assert np.all(np.isnan(host[mask]))
        assert relerr(got, want) <= (1e-9 if dt == np.float64 else 1e-6)
        res.append(got)
    assert np.array_equal(res[0], res[1])
