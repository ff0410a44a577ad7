"""Pins the oracle's DSP numerics (recalled DSP.jl algorithms) against
independent implementations: scipy.signal.sosfilt for the DF2T cascade, an explicit
zero-stuff + convolve + pick for the rational polyphase kernels, the vectorised closed form
of SURVEY.md Appendix A for the arbitrary-rate kernel, and an analytic sine.  CPU only."""
import os
import subprocess
import sys

import numpy as np
import pytest
from scipy import signal

import sigops_amd as so
from sigops_amd import Signal, Filt, ToFramerate, Lowpass, Bandstop, Highpass, Chebyshev1, Hz, kHz, FilterFn
from cases import F, rng
from oracle_bridge import oracle_sink, oracle_positions, relerr


@pytest.mark.parametrize("spec", [
    ("lowpass", (6.0,), ("butterworth", 5)), ("bandstop", (2.0, 12.0), ("chebyshev1", 5, 1.0)),
    ("highpass", (8.0,), ("chebyshev1", 5, 1.0)), ("bandpass", (20.0, 30.0), ("butterworth", 3))])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_iir_matches_sosfilt(spec, dtype):
    design, args, method = spec
    x = F(rng(1).standard_normal((3000, 3)).astype(dtype))
    tree = so.FilteredSignal(Signal(x, 100 * Hz), FilterFn(design, method, args), 4096, 100.0)
    got = oracle_sink(tree)
    sos, gain = so.design_iir(FilterFn(design, method, args), 100.0)
    want = signal.sosfilt(sos, x.astype(np.float64), axis=0) * gain  # DF2T, zero state
    assert got.dtype == dtype
    assert relerr(got, want.astype(dtype)) < (1e-12 if dtype == np.float64 else 2e-7)


def _polyphase_numpy(x, h, L, M, n_out):
    """y[m] = (zero-stuffed x * h)[c0 + m*M], c0 = (hLen-1)/2  """
    up = np.zeros(len(x) * L + len(h))
    up[: len(x) * L: L] = x
    full = np.convolve(up, h)
    c0 = (len(h) - 1) // 2
    idx = c0 + np.arange(n_out) * M
    return full[idx]


@pytest.mark.parametrize("ratio,fs_in,fs_out", [((2, 1), 20, 40), ((1, 2), 1000, 500), ((3, 2), 20, 30),
                                                ((2, 3), 30, 20), ((3, 1), 10, 30), ((1, 3), 30, 10)])
def test_rational_resampler_matches_convolution(ratio, fs_in, fs_out):
    x = rng(2).standard_normal(400)
    got = oracle_sink(ToFramerate(Signal(F(x[:, None]), fs_in * Hz), fs_out * Hz))[:, 0]
    h = so.design_resample(ratio)
    n_out = int(np.ceil(len(x) * fs_out / fs_in))
    want = _polyphase_numpy(x, h, ratio[0], ratio[1], n_out)
    assert got.shape[0] == n_out
    assert relerr(got, want) < 1e-12


def _arbitrary_closed_form(x, h, nphi, L, M, n_out):
    """vectorised Appendix-A formula with exact rational positions (integer rates)"""
    hlen = len(h)
    taps = -(-hlen // nphi)
    c0 = (hlen - 1) // 2
    m = np.arange(n_out, dtype=np.int64)
    N = m * (nphi * M)
    qi = c0 + N // L
    alpha = (N % L) / L
    j, p = qi // nphi, qi % nphi
    hp = np.concatenate([h, np.zeros(nphi * taps + 1)])
    dh = np.concatenate([np.diff(h), [0.0], np.zeros(nphi * taps + 1)])
    y = np.zeros(n_out)
    xp = np.concatenate([np.zeros(taps), x, np.zeros(taps + 2)])
    for k in range(taps):
        xv = xp[j - k + taps]
        y += (hp[p + nphi * k] + alpha * dh[p + nphi * k]) * xv
    return y


def test_arbitrary_resampler_exact_mode_matches_closed_form():
    """the oracle's opt-in closed-form mode against the vectorised Appendix-A formula"""
    x = rng(3).standard_normal(20000)
    with oracle_positions("exact"):
        got = oracle_sink(ToFramerate(Signal(F(x[:, None]), 44.1 * kHz), 48 * kHz))[:, 0]
    h = so.design_resample(48000 / 44100)
    want = _arbitrary_closed_form(x, h, 32, 160, 147, got.shape[0])
    assert got.shape[0] == int(np.ceil(20000 * 48000 / 44100))
    assert relerr(got, want) < 1e-12


def _fir_arbitrary_reference(x, h, nphi, rate, n_out):
    """DSP.jl 0.6 FIRArbitrary restated independently of the oracle's C:
    setphase!(timedelay), then per output  y = dot(pfb[:,ϕIdx], window) + α·dot(dpfb[:,ϕIdx], window)
    and update(): ϕAcc += Δ; wrap with div/mod; ϕIdx = floor(ϕAcc); α = ϕAcc - ϕIdx."""
    hlen = len(h)
    taps = -(-hlen // nphi)
    hp = np.concatenate([h, np.zeros(nphi * taps + 1 - hlen)])
    dh = np.concatenate([np.diff(h), [0.0], np.zeros(nphi * taps + 1 - hlen)])
    xp = np.concatenate([np.zeros(taps), x, np.zeros(taps + 2)])
    delta = nphi / rate
    tau = (hlen - 1) / 2 / nphi
    frac, whole = np.modf(tau)
    x_idx = 1 + int(round(whole))  # inputDeficit = 1 + throwaway
    acc = frac * nphi + 1.0
    y = np.empty(n_out)
    k = np.arange(taps)
    for m in range(n_out):
        phi = int(np.floor(acc))
        alpha = acc - phi
        win = xp[x_idx - 1 - k + taps]
        y[m] = np.dot(hp[phi - 1 + nphi * k], win) + alpha * np.dot(dh[phi - 1 + nphi * k], win)
        acc += delta
        if acc > nphi:
            x_idx += int(np.floor(acc - 1)) // nphi
            acc = np.mod(acc - 1, nphi) + 1
    return y


def test_arbitrary_resampler_default_is_the_phase_accumulator():
    """The oracle's DEFAULT is the reference's algorithm (DSP.jl FIRArbitrary: Float64 phase
    accumulator), checked against an independent restatement."""
    x = rng(3).standard_normal(6000)
    got = oracle_sink(ToFramerate(Signal(F(x[:, None]), 44.1 * kHz), 48 * kHz))[:, 0]
    h = so.design_resample(48000 / 44100)
    want = _fir_arbitrary_reference(x, h, 32, 48000 / 44100, got.shape[0])
    assert relerr(got, want) < 1e-13


@pytest.mark.parametrize("fs_out", [48000.0, 16000.0])
def test_resampled_sine_is_the_analytic_sine(fs_out):
    """440 Hz sine in -> 440 Hz sine out, time aligned: output m=0 sits at input n=1
    (t = 1/fs_in, reference first-frame convention); error = the 60 dB design ripple."""
    fs_in = 44100.0
    n = np.arange(1, 44101)
    x = np.sin(2 * np.pi * 440.0 * n / fs_in)
    y = oracle_sink(ToFramerate(Signal(F(x[:, None]), fs_in * Hz), fs_out * Hz))[:, 0]
    t = 1.0 / fs_in + np.arange(y.shape[0]) / fs_out
    want = np.sin(2 * np.pi * 440.0 * t)
    interior = slice(200, y.shape[0] - 200)
    assert np.max(np.abs(y[interior] - want[interior])) < 1e-3


def test_phase_accumulator_divergence_is_one_edge_tap():
    """Closed-form positions vs DSP.jl's accumulator on 44.1 -> 48 kHz.
    At a tie (closed-form alpha == 0) the accumulator sits a rounding error below: (previous
    phase, alpha ~ 1).  The interpolated taps h + alpha*dh agree there except at the two ends of
    the filter: output 80 of every 160-output period is the wrap-around tie (phase 32, xIdx not
    advanced: the tap h[0] of the next input is dropped), output 55 the last-tap tie
    (dh = [diff(h); 0] ends in 0, so h[end] survives where the closed form has moved past it).
    Everywhere else the two modes agree to rounding."""
    x = rng(4).standard_normal(8000)
    tree = ToFramerate(Signal(F(x[:, None]), 44.1 * kHz), 48 * kHz)
    acc = oracle_sink(tree)[:, 0]
    with oracle_positions("exact"):
        exact = oracle_sink(tree)[:, 0]
    assert 1e-5 < relerr(acc, exact) < 1e-4
    m = np.arange(acc.shape[0])
    wrap, last = m % 160 == 80, m % 160 == 55
    assert np.abs(acc - exact)[~(wrap | last)].max() < 1e-11
    h = so.design_resample(48000 / 44100)
    xp = np.concatenate([x, np.zeros(64)])
    # closed-form newest input j = (592 + 29.4 m) div 32 (0-based)
    j = (592 + (m * 32 * 147) // 160) // 32
    np.testing.assert_allclose((exact - acc)[wrap], h[0] * xp[j[wrap]], atol=1e-11)
    np.testing.assert_allclose((acc - exact)[last], h[-1] * xp[j[last] - 37], atol=1e-11)


def test_reference_quirk_append_never_leaves_a_long_filtered_child():
    """SURVEY quirk C-7 made visible: in the reference a FilteredSignal longer than its block
    never returns `nothing` (src/filters.jl:224-227 compares a buffer-local index with the global
    length), so `Append(filtered, y)` continues with the filter's zero-padded tail instead of y.
    The oracle follows the reference; the engine concatenates (tests/test_gpu_parity.py)."""
    rng = np.random.default_rng(31)
    fs = 8 * so.kHz
    x = np.asfortranarray(rng.standard_normal((6000, 1)))
    y = 3.0 * np.ones((9000, 1))
    rx = so.Signal(x, fs) | so.ToFramerate(12 * so.kHz)
    w = oracle_sink(so.Append(rx, so.Signal(y, 12 * so.kHz)))
    assert w.shape == (18000, 1)
    np.testing.assert_array_equal(w[:9000], oracle_sink(rx))
    assert np.abs(w[9100:]).max() < 1e-12  # the resampler's decayed tail, not y


def test_lastframe_pad_after_an_append_whose_last_child_is_empty():
    """`usepad(lastframe)` is `frame(x, block, nframes(block))` on the last block the Pad got
    (src/padding.jl:158-159); after `Append(x, <empty>)` that is still x's block -- AppendBlocks are
    immutable and advancechild returned nothing (src/appending.jl:98-110).  Found by the round-2 soak:
    the oracle's mutable child index had moved on to the empty child."""
    x = F(np.arange(10.0).reshape(5, 2))
    y = F(100 + np.arange(20.0).reshape(10, 2))
    t = so.Pad(so.Append(so.Signal(x, 50 * so.Hz), so.Signal(y, 50 * so.Hz) | so.Until(0 * so.frames)), so.lastframe) | so.Until(8 * so.frames)
    w = oracle_sink(t)
    assert np.array_equal(w[:5], x) and np.array_equal(w[5:], np.tile(x[4], (3, 1)))
