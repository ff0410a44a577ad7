"""GPU parity for window aliasing: stages under an `Append` (reference src/appending.jl:59-76) write
their time window of the result themselves instead of going through a root copy launch.  Every
case is run twice, with and without the optimisation (SIGOPS_NO_WINDOW_ALIAS): the two engine
results must be bit-identical, and both are checked against the oracle."""
import os

import numpy as np
import pytest

import sigops_amd as so
from oracle_bridge import oracle_semantics, oracle_sink, relerr

pytestmark = pytest.mark.gpu


def _noise(rng, n, nch, dt=np.float64):
    return np.asfortranarray(rng.standard_normal((n, nch)).astype(dt))


def _both(tree, **kw):
    os.environ.pop("SIGOPS_NO_WINDOW_ALIAS", None)
    got = so.sink(tree, so.Array, **kw)
    os.environ["SIGOPS_NO_WINDOW_ALIAS"] = "1"
    try:
        ref = so.sink(tree, so.Array, **kw)
    finally:
        os.environ.pop("SIGOPS_NO_WINDOW_ALIAS", None)
    assert got.shape == ref.shape and got.dtype == ref.dtype
    assert np.array_equal(got, ref), "window aliasing changed the result"
    return got


def _check(tree, tol=1e-9):
    got = _both(tree)
    with oracle_semantics("intended"):
        want = oracle_sink(tree)
    assert got.shape == want.shape
    assert relerr(got, want) <= tol
    return got


@pytest.mark.parametrize("nch", [1, 2, 3])
@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_append_of_filtered_children_odd_lengths(nch, dt):
    rng = np.random.default_rng(11 + nch)
    kids = [so.Signal(_noise(rng, n, nch, dt), 44.1 * so.kHz) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz)
            for n in (5001, 7777, 12345, 3)]
    _check(so.Append(*kids), tol=1e-6 if dt == np.float32 else 1e-9)


def test_scenes_filter_then_ramp():
    """BASELINE configs[3] in small: Append(Mix(sin, noise) |> Filt |> Ramp ...)"""
    rng = np.random.default_rng(3)
    kids = []
    for k in range(5):
        n = 20000 + 1001 * k
        nz = so.Signal(_noise(rng, n, 2), 44.1 * so.kHz)
        tone = so.Signal(so.sin, 44.1 * so.kHz, ω=(200.0 + 10 * k) * so.Hz) | so.Until(n * so.frames) | so.ToChannels(2)
        kids.append(so.Mix(tone, nz) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.Ramp(10 * so.ms))
    _check(so.Append(*kids))


def test_mixed_children():
    """arrays, generated pieces, resampled and filtered children in one Append"""
    rng = np.random.default_rng(4)
    a = so.Signal(_noise(rng, 9001, 2), 48 * so.kHz)
    b = so.Signal(_noise(rng, 30001, 2), 44.1 * so.kHz) | so.ToFramerate(48 * so.kHz)
    c = so.Signal(_noise(rng, 11111, 2), 48 * so.kHz) | so.Filt(so.Lowpass, 3 * so.kHz) | so.Amplify(0.5)
    d = so.Signal(so.sin, 48 * so.kHz, ω=100 * so.Hz) | so.Until(777 * so.frames) | so.ToChannels(2)
    e = so.Signal(_noise(rng, 6007, 2), 48 * so.kHz) | so.Filt(so.Highpass, 1 * so.kHz)
    _check(so.Append(a, b, c, d, e))


def test_window_in_the_middle_of_a_pad():
    rng = np.random.default_rng(5)
    x = so.Signal(_noise(rng, 10007, 2), 44.1 * so.kHz) | so.Filt(so.Lowpass, 3 * so.kHz)
    tree = so.Append(so.Signal(_noise(rng, 333, 2), 44.1 * so.kHz), x) | so.Pad(so.zero) | so.Until(20000 * so.frames)
    _check(tree)


def test_not_aliased_when_the_child_is_read_twice_or_mixed():
    rng = np.random.default_rng(6)
    f = so.Signal(_noise(rng, 8191, 2), 44.1 * so.kHz) | so.Filt(so.Lowpass, 3 * so.kHz)
    g = so.Signal(_noise(rng, 8191, 2), 44.1 * so.kHz) | so.Filt(so.Highpass, 5 * so.kHz)
    _check(so.Append(f, f))
    _check(so.Append(so.Mix(f, g), g))
    _check(so.Append(f | so.Until(4000 * so.frames), g))  # a window shorter than the stage


def test_device_result_repointed_between_executes():
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(8)
    kids = [so.Signal(_noise(rng, n, 2), 44.1 * so.kHz) | so.Filt(so.Lowpass, 3 * so.kHz) | so.Ramp(5 * so.ms)
            for n in (9999, 12001)]
    tree = so.Append(*kids)
    with oracle_semantics("intended"):
        want = oracle_sink(tree)
    n = want.shape[0]
    plan = so.Plan(so.ToChannels(tree, 2), (n, 2), np.float64, (1, n + 5), True, device=0)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        for fill in (0.0, float("nan")):
            out = torch.full((2, n + 5), fill, dtype=torch.float64, device="cuda")
            plan.execute(out.data_ptr(), st)
            torch.cuda.synchronize()
            assert relerr(out[:, :n].t().cpu().numpy(), want) <= 1e-9
    plan.close()


@pytest.mark.parametrize("single_stream", [False, True])
@pytest.mark.parametrize("orders", [(2, 4, 6), (5, 5, 5)])
def test_graph_replay_after_the_result_moved_away_and_back(orders, single_stream, monkeypatch):
    """in-place root pieces (ramp edges of window-aliased stages) read the RESULT
    through leaves that are re-pointed at every new result pointer.  With device leaves the plan is
    graph-eligible: A, A (captured for A), B (direct, leaves now point at B), A -- replaying A's graph
    while the device leaf table still pointed at B applied the ramp to B's already ramped values."""
    torch = pytest.importorskip("torch")
    if single_stream:
        # (round 5: the replayed graph of a ONE-stream plan left whole windows NaN -- the filters' "no non-finite chunk yet"
        #  words were reset by hipMemsetAsync, a memset node in the graph; they are reset by a kernel of the library's now)
        monkeypatch.setenv("SIGOPS_SINGLE_STREAM", "1")
    rng = np.random.default_rng(18)
    host = [_noise(rng, n, 2) for n in (9999, 12001, 8000)]
    dev = [torch.from_numpy(np.ascontiguousarray(h.T)).cuda() for h in host]  # [nch][n]: planar, time fastest
    def tree_of(arrs):
        return so.Append(*[so.Signal(a if isinstance(a, np.ndarray) else a.t(), 44.1 * so.kHz) | so.Filt(so.Lowpass, 3 * so.kHz, order=o) | so.Ramp(5 * so.ms)
                           for a, o in zip(arrs, orders)])
    with oracle_semantics("intended"):
        want = oracle_sink(tree_of(host))
    n = want.shape[0]
    plan = so.Plan(so.ToChannels(tree_of(dev), 2), (n, 2), np.float64, (1, n + 5), True, device=0)
    st = torch.cuda.current_stream().cuda_stream
    A = torch.full((2, n + 5), float("nan"), dtype=torch.float64, device="cuda")
    B = torch.full((2, n + 5), float("nan"), dtype=torch.float64, device="cuda")
    order = [A, A, B, A, A, B, B, B, A, B, A, A]
    for i, buf in enumerate(order):
        plan.execute(buf.data_ptr(), st)
        torch.cuda.synchronize()
        assert relerr(buf[:, :n].t().cpu().numpy(), want) <= 1e-9, (i, "A" if buf is A else "B")
    c = plan.counters()
    # filters of three different orders keep their own launches: four steps, replayed as a graph; filters of one
    # order share their launches (test_gpu_sos_batch.py): two steps, launched directly -- same sequence either way
    if len(set(orders)) == 3 and not os.environ.get("SIGOPS_NO_GRAPH"):
        assert c["graph_replays"] >= 2 and c["graph_captures"] >= 1, c  # (the graph path really ran)
    assert c["graph_replays"] + c["direct_executes"] + c["graph_captures"] == len(order), c
    plan.close()


def test_fewer_launches():
    rng = np.random.default_rng(9)
    kids = [so.Signal(_noise(rng, 50000, 2), 44.1 * so.kHz) | so.Filt(so.Lowpass, 3 * so.kHz) for _ in range(3)]
    tree = so.ToChannels(so.Append(*kids), 2)
    n = 150000

    torch = pytest.importorskip("torch")
    out = torch.empty((2, n), dtype=torch.float64, device="cuda")

    def launches():
        p = so.Plan(tree, (n, 2), np.float64, (1, n), True, device=0)
        p.execute(out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        k = p.stats()["n_launches"]
        p.close()
        return k

    with_alias = launches()
    os.environ["SIGOPS_NO_WINDOW_ALIAS"] = "1"
    try:
        without = launches()
    finally:
        os.environ.pop("SIGOPS_NO_WINDOW_ALIAS", None)
    assert with_alias < without
