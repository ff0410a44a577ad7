"""Corners where the engine used to differ from the reference's CPU sink by more than 1e-6
("documented divergences no -m gpu test pins"), each now either closed or fenced:

* ill-conditioned cascades (Chebyshev band-stops of order 7-12: `tools/soak_filters.py` seeds 101,
  300, 430, 474 of round 2, 1e-7 ... 1.1e-3 between engine and oracle): the planner measures the
  cascade's rounding sensitivity on the host and runs DSP.jl's own order of operations
  (`k_sos_exact`, reference src/filters.jl:252-255 -> DSP.jl `filt!` for second-order sections);
* `Normpower` of the tail of a filter long after its input went silent (5e-4 in round 2): a filter
  that feeds a `Normpower` is scanned without the 2^-70 cut (reference src/filters.jl:296-309);
* an empty result over a multi-block child under a root `After`.
"""
import numpy as np
import pytest

import sigops_amd as so
from sigops_amd.engine import Plan
from oracle_bridge import oracle_sink, relerr

pytestmark = pytest.mark.gpu


def _steps_of(tree, n, nch, dt=np.float64):
    """names of the stage kernels a plan of this tree runs"""
    out = np.empty((n, nch), dtype=dt, order="F")
    p = Plan(so.ToChannels(tree, nch), (n, nch), dt, (1, n), False)
    p.set_profiling(True)
    p.execute(out.ctypes.data)
    names = [s["name"] for s in p.steps()]
    p.close()
    return names, out


# (seed of round 2's tools/soak_filters.py, channels, dtype, fs, order, ripple dB, f1, f2) -- all Bandstop
ILL = [
    (101, 1, np.float64, 8000.0, 12, 2.241092965518731, 22.50567978243576, 111.38302696863563),
    (300, 1, np.float64, 8000.0, 7, 0.6047704980280488, 622.070908995388, 3920.0),
    (430, 2, np.float64, 44100.0, 11, 1.7290310069734331, 5181.802510943963, 21609.0),
    (474, 8, np.float32, 8000.0, 12, 2.685564982775896, 174.1251102200946, 1294.9524547967571),
]


@pytest.mark.parametrize("seed,nch,dt,fs,order,ripple,f1,f2", ILL)
def test_ill_conditioned_cascade_runs_in_the_reference_order(seed, nch, dt, fs, order, ripple, f1, f2):
    n = 150_000
    x = np.asfortranarray(np.random.default_rng(15000 + seed).standard_normal((n, nch)).astype(dt))
    tree = so.Signal(x, fs * so.Hz) | so.Filt(so.Bandstop, f1 * so.Hz, f2 * so.Hz, method=so.Chebyshev1(order, ripple))
    names, got = _steps_of(tree, n, nch, dt)
    assert "k_sos_exact" in names, names
    want = oracle_sink(tree)
    assert np.isfinite(want).all()
    # same operations in the same order, each rounded once on both sides: not "close", equal
    assert np.array_equal(got, want), relerr(got, want)
    # ... also through a window far from the start (no warm start for these: the whole prefix is filtered)
    a, m = 90_000, 20_000
    w = so.sink(tree | so.After(a * so.frames) | so.Until(m * so.frames), so.Array)
    assert np.array_equal(w, want[a:a + m])


def test_well_conditioned_filters_keep_the_time_parallel_kernels():
    n = 100_000
    x = np.asfortranarray(np.random.default_rng(5).standard_normal((n, 2)))
    for filt in (so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz), so.Filt(so.Lowpass, 3 * so.kHz, method=so.Chebyshev1(6, 1.0)),
                 so.Filt(so.Bandpass, 1 * so.kHz, 3 * so.kHz, method=so.Butterworth(8))):
        tree = so.Signal(x, 44.1 * so.kHz) | filt
        names, got = _steps_of(tree, n, 2)
        assert "k_sos" in names and "k_sos_exact" not in names, names
        assert relerr(got, oracle_sink(tree)) < 1e-10


def test_empty_result_under_a_root_after_over_a_filtered_child():
    """The reference's skip loop asks its child for blocks of min(maxlen, ...) = 0 frames when the
    result is empty (src/cutting.jl:167-172 with maxlen = size(result,1) = 0 from src/sink.jl:225):
    a filtered child longer than one block then never finishes skipping.  The engine returns the
    empty result (documented divergence: there is no value to get wrong)."""
    x = np.asfortranarray(np.random.default_rng(6).standard_normal((20_000, 2)))
    tree = so.Signal(x, 10 * so.kHz) | so.Filt(so.Lowpass, 1 * so.kHz) | so.After(20_000 * so.frames)
    got = so.sink(tree, so.Array)
    assert got.shape == (0, 2)
    # one frame short of empty is an ordinary sink and matches the oracle
    tree1 = so.Signal(x, 10 * so.kHz) | so.Filt(so.Lowpass, 1 * so.kHz) | so.After(19_999 * so.frames)
    got1 = so.sink(tree1, so.Array)
    want1 = oracle_sink(tree1)
    assert got1.shape == want1.shape == (1, 2) and relerr(got1, want1) < 1e-9


@pytest.mark.parametrize("nch", [1, 2])
def test_normpower_of_a_decayed_filter_tail(nch):
    """A burst, then silence: thousands of frames later the band-stop's ringing is 1e-73 ... 1e-108 of its
    peak -- far below the 2^-70 cut of the chunked scan, which used to return exact zeros there once
    the signal had more than 64 chunks (`Normpower` of that window: 5e-4 off in the round-2 soak, or a
    division by zero).  Below a Normpower the filter is scanned exactly and is not warm-started."""
    rng = np.random.default_rng(40 + nch)
    burst, total = 30_000, 1_200_000
    x = np.asfortranarray(rng.standard_normal((burst, nch)))
    filtered = (so.Signal(x, 44.1 * so.kHz) | so.Pad(so.zero) | so.Until(total * so.frames)
                | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz))
    tail = filtered | so.After((burst + 12_000) * so.frames) | so.Until(6_000 * so.frames)
    raw = oracle_sink(tail)
    assert 0 < np.abs(raw).max() < 1e-30 and np.abs(raw[-1]).max() > 1e-250  # (the window really is a decayed tail)
    tree = tail | so.Normpower
    got = so.sink(tree, so.Array)
    want = oracle_sink(tree)
    assert np.isfinite(got).all()
    assert relerr(got, want) < 1e-9
    # the unnormalised tail itself keeps its relative accuracy too when it feeds a Normpower further up
    whole = so.sink(filtered | so.Normpower | so.After((burst + 12_000) * so.frames) | so.Until(6_000 * so.frames), so.Array)
    want_whole = oracle_sink(filtered | so.Normpower | so.After((burst + 12_000) * so.frames) | so.Until(6_000 * so.frames))
    assert relerr(whole, want_whole) < 1e-9


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("kind,frac", [("lp", 0.25), ("hp", 0.25), ("lp", 0.2498), ("hp", 0.2502), ("raw", 0.0)])
def test_filters_that_forget_within_a_tile_step(dt, kind, frac):
    """A first-order Butterworth at fs/4 has its pole at 0 (pole 1e-3 just beside it): the filter's state is forgotten
    within W = 2 ... 8 frames, fewer than the 16 of a tile step.  The state pass used to walk exactly W frames and let
    the zeros that filled the rest of the step turn the state on -- to nothing -- so every chunk but the first started
    from rest (found by tools/soak_kernels.py seed 10102: 7 % off); it now walks whole steps."""
    rng = np.random.default_rng(3)
    fs = 48000.0
    x = so.Signal(np.asfortranarray(rng.standard_normal((70000, 5)).astype(dt)), fs * so.Hz)
    if kind == "lp":
        tree = so.Filt(x, so.Lowpass, frac * fs * so.Hz, method=so.Butterworth(1))
    elif kind == "hp":
        tree = so.Filt(x, so.Highpass, frac * fs * so.Hz, method=so.Butterworth(1))
    else:  # raw sections with poles at 0.001 and 0.02
        tree = so.Filt(x, sos=np.array([[1.0, 0.5, 0.0, 1.0, -0.001, 0.0], [1.0, -0.3, 0.1, 1.0, -0.02, 0.0]]), gain=0.7)
    got = so.sink(tree)[0]
    want = oracle_sink(tree)
    assert got.shape == want.shape
    assert relerr(got, want) <= (1e-12 if dt == np.float64 else 1e-6)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_band_stop_up_to_nyquist_runs_in_the_reference_order(dt):
    """ten poles next to -1 (Butterworth band-stop from 0.375 fs to 0.4999 fs): the rounding-sensitivity probe of the
    recurrence says 5e-10, but the chunked form's state hand-over -- powers of a nearly defective matrix -- was 4e-5
    away from the reference (tools/soak_degenerate_filters.py).  The planner now measures the chunked form too
    (sos_chunk_sensitivity) and runs such a cascade sequentially: the oracle's values bit for bit in Float64."""
    rng = np.random.default_rng(4242)
    fs = 48000.0
    x = so.Signal(np.asfortranarray(rng.standard_normal((70000, 3)).astype(dt)), fs * so.Hz)
    t = so.Filt(x, so.Bandstop, 0.375 * fs * so.Hz, 0.4999 * fs * so.Hz, method=so.Butterworth(5))
    got, want = so.sink(t)[0], oracle_sink(t)
    if dt == np.float64:
        assert np.array_equal(got, want)
    else:
        assert relerr(got, want) <= 1e-6
    # its neighbour with the edge at 0.49 fs stays on the time-parallel kernels, and is as close as ever
    t2 = so.Filt(x, so.Bandstop, 0.375 * fs * so.Hz, 0.49 * fs * so.Hz, method=so.Butterworth(5))
    assert relerr(so.sink(t2)[0], oracle_sink(t2)) <= (1e-11 if dt == np.float64 else 1e-6)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("shape", ["one", "window", "scenes", "cascade16"])
def test_a_filter_never_recovers_from_a_non_finite_sample(dt, shape):
    """The reference's recurrence carries a NaN or Inf on in its state: from the first non-finite sample of a channel to
    its end, everything is NaN (reference src/filters.jl:252-255 -> DSP.jl filt!).  The chunked form looks back K chunks
    only and would return finite values again K chunks later -- values the reference never computes.  The output pass
    notes the first chunk that ends in a non-finite state per channel and k_sos_poison fills what lies behind it."""
    rng = np.random.default_rng(77)
    fs = 44.1 * so.kHz

    def noisy(n, nch, bad):
        x = np.asfortranarray(rng.standard_normal((n, nch)).astype(dt))
        for (i, c, v) in bad:
            x[i, c] = v
        return so.Signal(x, fs)

    if shape == "one":
        t = noisy(200000, 4, [(777, 1, np.nan), (150000, 3, np.inf)]) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz)
    elif shape == "window":
        t = noisy(200000, 3, [(60001, 0, np.nan)]) | so.Filt(so.Lowpass, 3 * so.kHz) | so.After(50000 * so.frames) | so.Until(120000 * so.frames)
    elif shape == "scenes":  # a batch: the NaN of one scene stays in that scene
        t = so.Append(*[noisy(50000 + 1000 * k, 2, [(20000, 1, np.nan)] if k == 2 else []) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz)
                        for k in range(5)])
    else:  # two section groups filter one after the other in place
        t = noisy(150000, 2, [(90000, 0, np.nan)]) | so.Filt(so.Bandpass, 1 * so.kHz, 4 * so.kHz, order=9)
    from oracle_bridge import oracle_semantics
    with oracle_semantics("intended"):  # (filtered children under an Append: every child its own filter)
        want = oracle_sink(t)
    got = so.sink(t)[0]
    assert got.shape == want.shape
    gn, wn = ~np.isfinite(got), ~np.isfinite(want)
    assert wn.any() and np.array_equal(gn, wn)
    assert relerr(np.where(wn, 0, got), np.where(wn, 0, want)) <= (1e-9 if dt == np.float64 else 1e-6)


def test_an_indexing_pad_on_a_computed_signal_is_refused_even_where_the_outputs_never_reach_it():
    """`Pad(Signal(sin) |> Until, mirror)` is an error in the reference (src/padding.jl:169-181: a computed signal has
    no `getindex`), raised when the pad region is first pulled.  A resampler above refills its input a block at a time
    (src/filters.jl:185-199, 237-244), so the error comes even though the 265 outputs asked for depend on the first
    ~200 input frames only (tools/tree_soak_multirate.py seed 16046: the engine used to accept the tree)."""
    fs = 8 * so.kHz
    x = so.Signal(so.sin, fs, ω=100 * so.Hz) | so.Until(354 * so.frames) | so.Pad(so.mirror) | so.Until(1412 * so.frames)
    t = so.Amplify(x, 2.0) | so.ToFramerate(12 * so.kHz) | so.Until(265 * so.frames)
    with pytest.raises(Exception) as oe:
        oracle_sink(t)
    with pytest.raises(Exception) as ee:
        so.sink(t)
    assert "indexing pad function" in str(oe.value) and "indexing pad function" in str(ee.value)
    # the same tree with an array under the pad is fine on both sides
    rng = np.random.default_rng(2)
    y = so.Signal(np.asfortranarray(rng.standard_normal((354, 1))), fs) | so.Pad(so.mirror) | so.Until(1412 * so.frames)
    t2 = so.Amplify(y, 2.0) | so.ToFramerate(12 * so.kHz) | so.Until(265 * so.frames)
    assert relerr(so.sink(t2)[0], oracle_sink(t2)) <= 1e-9


def test_a_cascade_without_feedback_stays_nan_too():
    """ADVICE r3 suggested sparing sections with a1 = a2 = 0 the poison pass ("their state forgets a NaN two samples
    later").  It does not in IEEE arithmetic: DF2T forms s1 = s2 + b1 x - a1 y with y = NaN, and 0 * NaN is NaN -- the
    reference's recurrence (DSP.jl `filt!`, call site src/filters.jl:252-255) stays NaN to the end of the channel for
    such a cascade as for any other, and so do the oracle and the engine."""
    rng = np.random.default_rng(123)
    x = rng.standard_normal((300000, 2))
    x[100000, 0] = np.nan
    sos = np.array([[0.5, 0.25, 0.125, 1.0, 0.0, 0.0], [1.0, -0.5, 0.25, 1.0, 0.0, 0.0]])
    t = so.Filt(so.Signal(np.asfortranarray(x), 44.1 * so.kHz), sos=sos, gain=1.0)
    got, want = so.sink(t, so.Array), oracle_sink(t)
    assert np.isnan(want[100000:, 0]).all() and np.isfinite(want[:100000, 0]).all()
    assert np.array_equal(np.isfinite(got), np.isfinite(want))
    fin = np.isfinite(want)
    assert relerr(np.where(fin, got, 0), np.where(fin, want, 0)) <= 1e-12
