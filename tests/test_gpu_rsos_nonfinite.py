"""The set of non-finite outputs of the fused resampler + IIR kernel is the REFERENCE's (round 6, k_rsos_fixup in
csrc/k_exact.hip).  The reference filters sample by sample (src/filters.jl:252-255 -> DSP.jl filt!): a channel is non-finite from
the first output whose own taps-per-phase input window holds a non-finite sample, to its end.  k_rsos alone is non-finite from
the start of the 16-output block of the first GROUP window that holds the sample -- a superset that rounds 4 and 5 stated; the
fix-up launch recomputes that block output by output.  Asserted here: set EQUALITY with the oracle for NaN and +-Inf at a
block's start, middle and end, five rate pairs x three channel counts, Float64 and Float32 signals, a fused `Mix` in front, a
window behind -- and the oracle's values everywhere outside the set."""
import numpy as np
import pytest

import sigops_amd as so
from oracle_bridge import oracle_sink, relerr
from test_gpu_rsos import F, env, steps_of

pytestmark = pytest.mark.gpu


def spoil(d, rng, nch):
    """non-finite samples: channel 0 early, the last channel twice (the first one counts), one channel in between where there is one"""
    n = d.shape[0]
    where = {0: (n // 5 + 3, np.nan), nch - 1: (n // 2 + 1, np.inf)}
    if nch > 2:
        where[1] = (3 * n // 4, -np.inf)
    for c, (i, v) in where.items():
        d[i, c] = v
    d[min(n - 1, n // 2 + 5000), nch - 1] = np.nan
    return where


@pytest.mark.parametrize("rates", [(44.1, 48.0), (48.0, 44.1), (22.05, 24.0), (32.0, 48.0), (24.0, 48.0)])
@pytest.mark.parametrize("nch", [8, 4, 2])
def test_set_equality_with_the_reference(rates, nch):
    fi, fo = rates
    rng = np.random.default_rng(int(fi * 10) + nch)
    n = int(400_000 * fi / 44.1) + 3
    d = rng.standard_normal((n, nch))
    where = spoil(d, rng, nch)
    x = so.Signal(F(d), fi * so.kHz) | so.ToFramerate(fo * so.kHz) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz)
    with env(SIGOPS_RSOS_MINGROUPS=1):
        # (where the fused kernel does not run the geometry -- 48 -> 44.1 kHz: the row-tiled resampler, then the filter in one
        #  pass -- each of the two kernels is exact about its own part: tests/test_gpu_resampler_nonfinite.py)
        assert any(n_ in ("k_rsos",) for n_ in steps_of(x)), steps_of(x)
        got = so.sink(x)[0]
    want = oracle_sink(x)
    bad_g, bad_w = ~np.isfinite(got), ~np.isfinite(want)
    assert bad_w.any()
    for c in range(nch):
        assert np.array_equal(bad_g[:, c], bad_w[:, c]), (c, np.nonzero(bad_g[:, c])[0][:3], np.nonzero(bad_w[:, c])[0][:3])
        if c not in where:
            assert not bad_g[:, c].any()
    assert relerr(got[~bad_g], want[~bad_w]) < 1e-9


@pytest.mark.parametrize("pos", [0, 1, 7, 15, 16, 31])
@pytest.mark.parametrize("val", [np.nan, np.inf, -np.inf])
def test_every_place_in_a_block(pos, val):
    """the first non-finite OUTPUT at every kind of place of its 16-output block: the input sample is moved until it is"""
    rng = np.random.default_rng(5)
    n = 300_000
    base = rng.standard_normal((n, 8))
    x0 = so.Signal(F(base), 44.1 * so.kHz) | so.ToFramerate(48 * so.kHz) | so.Filt(so.Lowpass, 5 * so.kHz)
    # an input index whose first affected output sits at `pos` of a block (outputs ~ inputs * 160 / 147)
    i = 100_000
    for i in range(100_000, 100_200):
        d = base.copy()
        d[i, 2] = val
        first = int(np.nonzero(~np.isfinite(oracle_sink(so.Signal(F(d[: i + 200]), 44.1 * so.kHz) | so.ToFramerate(48 * so.kHz))[:, 2]))[0][0])
        if first % 32 == pos:
            break
    else:
        pytest.skip("no such input index in reach")
    d = base.copy()
    d[i, 2] = val
    x = so.Signal(F(d), 44.1 * so.kHz) | so.ToFramerate(48 * so.kHz) | so.Filt(so.Lowpass, 5 * so.kHz)
    with env(SIGOPS_RSOS_MINGROUPS=1):
        assert steps_of(x) == ["k_rsos"]
        got = so.sink(x)[0]
    want = oracle_sink(x)
    assert np.array_equal(np.isfinite(got), np.isfinite(want))
    ok = np.isfinite(want)
    assert relerr(got[ok], want[ok]) < 1e-9 and np.isfinite(got[:, [0, 1, 3, 4, 5, 6, 7]]).all()
    del x0


@pytest.mark.parametrize("kind", ["float32", "mix", "window", "f32 leaf into f32 result"])
def test_other_forms(kind):
    rng = np.random.default_rng(15)
    n = 500_000
    d = rng.standard_normal((n, 8))
    d[123_456, 1] = np.nan
    d[400_001, 6] = np.inf
    tol = 1e-9
    if kind == "float32":
        x = so.Signal(F(d.astype(np.float32)), 44.1 * so.kHz) | so.ToFramerate(48 * so.kHz) | so.Filt(so.Lowpass, 5 * so.kHz)
        tol = 1e-6
    elif kind == "mix":
        x = so.Mix(so.Signal(so.sin, ω=1 * so.kHz), so.Signal(F(d), 44.1 * so.kHz)) | so.Until(n * so.frames) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.ToFramerate(48 * so.kHz)
    elif kind == "window":
        x = so.Signal(F(d), 44.1 * so.kHz) | so.ToFramerate(48 * so.kHz) | so.Filt(so.Lowpass, 5 * so.kHz) | so.After(100_000 * so.frames) | so.Until(200_000 * so.frames)
    else:
        x = so.Mix(so.Signal(so.sin, ω=1 * so.kHz), so.Signal(F(d.astype(np.float32)), 44.1 * so.kHz)) | so.Until(n * so.frames) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.ToFramerate(48 * so.kHz)
        tol = 1e-6
    with env(SIGOPS_RSOS_MINGROUPS=1):
        if kind == "f32 leaf into f32 result":
            nout = so.nframes(x)
            got = np.empty((nout, 8), dtype=np.float32, order="F")
            so.sink_into(got, x)
            want = oracle_sink(x).astype(np.float32)
        else:
            assert "k_rsos" in steps_of(x, np.float32 if kind == "float32" else np.float64)
            got = so.sink(x)[0]
            want = oracle_sink(x)
    assert np.array_equal(np.isfinite(got), np.isfinite(want))
    ok = np.isfinite(want)
    assert relerr(got[ok], want[ok]) < tol


def test_finite_data_pays_one_empty_launch():
    """every workgroup of the fix-up launch reads its channel's word and returns"""
    rng = np.random.default_rng(25)
    x = so.Signal(F(rng.standard_normal((400_000, 8))), 44.1 * so.kHz) | so.ToFramerate(48 * so.kHz) | so.Filt(so.Lowpass, 5 * so.kHz)
    with env(SIGOPS_RSOS_MINGROUPS=1):
        a = so.sink(x)[0]
    with env(SIGOPS_RSOS_MINGROUPS=1, SIGOPS_RSOS_NO_FIXUP=1):
        b = so.sink(x)[0]
    assert np.array_equal(a, b)
