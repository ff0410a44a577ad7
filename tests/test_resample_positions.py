"""Host logic of the engine's resampler positions (no GPU): so_resample_positions exposes the
(newest input, phase, alpha) the HIP kernels use per output -- closed form + period positions
baked from DSP.jl's phase accumulator + the sparse fix-up list (csrc/accumulator.cpp
replay_phase_accumulator).  Applying them in NumPy must reproduce the oracle's DEFAULT mode,
which is the reference's FIRArbitrary algorithm (reference src/reformatting.jl:92-98,
src/filters.jl:252-255)."""
import ctypes as C

import numpy as np
import pytest

import sigops_amd as so
from sigops_amd import Signal, ToFramerate, Hz
from sigops_amd import _capi as K
from cases import F, rng
from oracle_bridge import oracle_sink, oracle_positions, relerr


def engine_positions(fs_in, fs_out, n_out, monkeypatch=None):
    rate = fs_out / fs_in
    h = so.design_resample(rate)
    j = np.empty(n_out, np.int64)
    p = np.empty(n_out, np.int32)
    a = np.empty(n_out, np.float64)
    nfix, nbaked = C.c_int64(0), C.c_int64(0)
    st = K.lib().so_resample_positions(
        float(fs_in), float(fs_out), rate, 32, h.ctypes.data_as(C.POINTER(C.c_double)), len(h), n_out,
        j.ctypes.data_as(C.POINTER(C.c_int64)), p.ctypes.data_as(C.POINTER(C.c_int32)),
        a.ctypes.data_as(C.POINTER(C.c_double)), C.byref(nfix), C.byref(nbaked))
    assert st == 0, K.last_error()
    return h, j, p, a, nfix.value, nbaked.value


def apply_positions(x, h, j, p, a, nphi=32):
    """y[m] = sum_k (h[p+Nphi k] + alpha dh[p+Nphi k]) x[j-k], dh = [diff(h); 0]"""
    hlen = len(h)
    taps = -(-hlen // nphi)
    hp = np.concatenate([h, np.zeros(nphi * taps + 1 - hlen)])
    dh = np.concatenate([np.diff(h), [0.0], np.zeros(nphi * taps + 1 - hlen)])
    xp = np.concatenate([np.zeros(taps), x, np.zeros(taps + 2)])
    y = np.zeros(len(j))
    for k in range(taps):
        y += (hp[p + nphi * k] + a * dh[p + nphi * k]) * xp[j - k + taps]
    return y


@pytest.mark.parametrize("fs_in,fs_out", [(44100, 48000), (44100, 16000), (8000, 11025), (48000, 44100),
                                          (22050, 96000), (1000, 4000), (44100.5, 48000), (100, 100 * np.pi)])
def test_engine_positions_reproduce_the_phase_accumulator(fs_in, fs_out):
    n_in = 30000
    x = rng(11).standard_normal(n_in)
    want = oracle_sink(ToFramerate(Signal(F(x[:, None]), fs_in * Hz), fs_out * Hz))[:, 0]
    h, j, p, a, nfix, nbaked = engine_positions(fs_in, fs_out, want.shape[0])
    got = apply_positions(x, h, j, p, a)
    assert relerr(got, want) < 1e-9
    # the fix-up pass stays sparse: ties the period tables cannot express
    assert nfix <= max(16, want.shape[0] // 1000), (nfix, nbaked)


def test_the_44k1_to_48k_pattern():
    """44.1 -> 48 kHz: two positions of every 160-output period follow the accumulator (the
    wrap-around tie at output 80 and the last-tap tie at output 55); nothing is left to fix up."""
    h, j, p, a, nfix, nbaked = engine_positions(44100, 48000, 160 * 500)
    assert nbaked == 2 and nfix == 0
    m = np.arange(len(j))
    q = j * 32 + p
    qe = 592 + (m * 32 * 147) // 160
    dev = np.nonzero(q != qe)[0]
    assert set(dev % 160) == {55, 80}
    assert np.all(q[dev] == qe[dev] - 1) and np.all(a[dev] == 1.0)


def test_closed_form_mode_is_the_opt_in(monkeypatch):
    monkeypatch.setenv("SIGOPS_RS_EXACT", "1")
    x = rng(12).standard_normal(8000)
    h, j, p, a, nfix, nbaked = engine_positions(44100, 48000, 8708)
    assert nfix == 0 and nbaked == 0
    with oracle_positions("exact"):
        want = oracle_sink(ToFramerate(Signal(F(x[:, None]), 44100 * Hz), 48000 * Hz))[:, 0]
    assert relerr(apply_positions(x, h, j, p, a), want) < 1e-12


@pytest.mark.parametrize("fs_in,fs_out", [(44100, 48000), (44100, 16000), (48000, 44100), (8000, 11025)])
def test_threaded_replay_is_the_sequential_replay(fs_in, fs_out, monkeypatch):
    """Long replays of an exact rational rate run on several threads from PREDICTED accumulator states
    (the accumulator is periodic up to a constant drift per period) and are verified range by range
    against the state the previous range really ended in: same positions, same fix-up list as one
    thread from output 0."""
    n_out = 5_000_003
    monkeypatch.setenv("SIGOPS_REPLAY_NOCACHE", "1")
    monkeypatch.setenv("SIGOPS_REPLAY_THREADS", "1")
    _, j1, p1, a1, nfix1, nb1 = engine_positions(fs_in, fs_out, n_out)
    monkeypatch.setenv("SIGOPS_REPLAY_THREADS", "7")
    _, j7, p7, a7, nfix7, nb7 = engine_positions(fs_in, fs_out, n_out)
    assert (nfix1, nb1) == (nfix7, nb7)
    assert np.array_equal(j1, j7) and np.array_equal(p1, p7) and np.array_equal(a1, a7)
