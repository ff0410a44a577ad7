"""The fused resampler + IIR kernel on the Float32 MFMA (k_rsos F32M, round 6): a Float32 array into a Float32 result keeps its
samples Float32 in the kernel's ring, the fast path's one step (`Mix` / `Amplify` with a sine or a constant) is done on them, and
the resampling product runs on v_mfma_f32_16x16x4_f32 -- taps rounded once, Float32 accumulators; the cascade stays Float64.
Reference: a Float32 signal stays Float32 (src/filters.jl:105, test/runtests.jl:707-729), results compared at 1e-6.  Here: against
the oracle at 1e-6, against the engine's own Float64 products (`SIGOPS_RSOS_NO_F32MFMA=1`) at 3e-7 -- the gate the soak
(tools/soak_rsos_f32m.py, profiles/r06/relerr_maxima_rsos_f32m.json) was asked to keep -- and with a local metric on a signal
of 60 dB dynamic range."""
import numpy as np
import pytest

import sigops_amd as so
from oracle_bridge import oracle_sink, relerr
from test_gpu_rsos import F, env

pytestmark = pytest.mark.gpu


def both(x, to=None):
    with env(SIGOPS_RSOS_MINGROUPS=1, SIGOPS_RSOS_NO_F32MFMA=None):
        a = so.sink(x)[0] if to is None else to(x)
    with env(SIGOPS_RSOS_MINGROUPS=1, SIGOPS_RSOS_NO_F32MFMA=1):
        b = so.sink(x)[0] if to is None else to(x)
    return a, b


def into_f32(x):
    """sink! into a Float32 buffer (reference src/sink.jl:154-168: convert on store)"""
    n = so.nframes(x)
    res = np.empty((n, x.nch), dtype=np.float32, order="F")
    so.sink_into(res, x)
    return res


def data(kind, n, nch, rng):
    t = np.arange(n) / 44100.0
    return {"noise": rng.standard_normal((n, nch)), "low tone": 0.9 * np.sin(2 * np.pi * 50 * t)[:, None] * np.ones((1, nch)),
            "dc": 1.0 + 1e-3 * rng.standard_normal((n, nch)), "clicks": (rng.random((n, nch)) < 1e-3) * 1.0}[kind]


@pytest.mark.parametrize("nch", [2, 4, 8, 16])
@pytest.mark.parametrize("kind", ["noise", "low tone", "dc", "clicks"])
def test_float32_signal_resampled_and_filtered(nch, kind):
    rng = np.random.default_rng(61 + nch)
    d = data(kind, 300000, nch, rng).astype(np.float32)
    x = so.Signal(F(d), 44.1 * so.kHz) | so.ToFramerate(48 * so.kHz) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz)
    a, b = both(x)
    want = oracle_sink(x)
    assert a.dtype == np.float32 and a.shape == want.shape
    assert relerr(a, want) < 1e-6 and relerr(b, want) < 1e-6
    assert relerr(a, b) < 3e-7
    assert not np.array_equal(a, b)  # (the Float32 instruction did run)


@pytest.mark.parametrize("nch", [2, 8])
@pytest.mark.parametrize("op", ["mix", "amplify", "mix const", "sub"])
def test_float64_signal_of_a_float32_array_into_a_float32_result(nch, op):
    """the north-star pipeline with a Float32 leaf and a Float32 result: the signal is Float64 by the reference's promotion (a sine
    generator is Float64, src/functions.jl:57-60); rounding it to Float32 on its way into the resampler instead of on its way
    into the result stays inside the contract"""
    rng = np.random.default_rng(71 + nch)
    d = rng.standard_normal((400000, nch)).astype(np.float32)
    src = so.Signal(F(d), 44.1 * so.kHz)
    g = so.Signal(so.sin, ω=1 * so.kHz)
    x = {"mix": so.Mix(g, src), "amplify": so.Amplify(src, so.Signal(so.sin, ω=5 * so.Hz)), "mix const": so.Mix(src, 0.25),
         "sub": so.OperateOn("-", src, g)}[op]
    x = x | so.Until(400000 * so.frames) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.ToFramerate(48 * so.kHz)
    a, b = both(x, into_f32)
    want = oracle_sink(x).astype(np.float32)
    assert a.dtype == np.float32 and a.shape == want.shape
    assert relerr(a, want) < 1e-6 and relerr(b, want) < 1e-6
    assert relerr(a, b) < 3e-7
    # ... and a Float64 result of the same tree does not take the path: bit-equal with and without the switch
    a64, b64 = both(x)
    assert a64.dtype == np.float64 and np.array_equal(a64, b64)


def test_local_error_on_sixty_decibels_of_dynamic_range():
    """the max-norm hides quiet passages: one second loud, one second 60 dB down, blockwise (1024 outputs) relative error of
    the Float32 MFMA form against the oracle -- a few Float32 rounding units of the LOCAL level, not of the loud one"""
    rng = np.random.default_rng(81)
    n = 88200
    env_ = np.where((np.arange(4 * n) // n) % 2 == 0, 1.0, 1e-3)[:, None]
    d = (rng.standard_normal((4 * n, 8)) * env_).astype(np.float32)
    x = so.Signal(F(d), 44.1 * so.kHz) | so.ToFramerate(48 * so.kHz) | so.Filt(so.Lowpass, 4 * so.kHz)
    a, b = both(x)
    want = oracle_sink(x).astype(np.float64)
    nb = want.shape[0] // 1024
    w = want[: nb * 1024].reshape(nb, 1024, -1)
    for got in (a, b):
        e = np.linalg.norm((got[: nb * 1024].astype(np.float64).reshape(nb, 1024, -1) - w), axis=(1, 2)) / np.linalg.norm(w, axis=(1, 2))
        # (blocks right behind a loud passage carry its decaying tail: they are judged like every other block)
        assert e.max() < 2e-6, (e.max(), int(e.argmax()))
    ea = np.linalg.norm((a[: nb * 1024].astype(np.float64).reshape(nb, 1024, -1) - w), axis=(1, 2)) / np.linalg.norm(w, axis=(1, 2))
    assert np.median(ea) < 3e-7


def test_windows_streams_and_set_array():
    rng = np.random.default_rng(91)
    d = rng.standard_normal((600000, 8)).astype(np.float32)
    x = so.Signal(F(d), 44.1 * so.kHz) | so.ToFramerate(48 * so.kHz) | so.Filt(so.Lowpass, 6 * so.kHz)
    want = oracle_sink(x)
    win = x | so.After(300_000 * so.frames) | so.Until(200_000 * so.frames)
    with env(SIGOPS_RSOS_MINGROUPS=1):
        got = so.sink(win)[0]
        assert got.dtype == np.float32 and relerr(got, want[300_000:500_000]) < 1e-6
        blocks = [blk for blk, _ in so.stream(x, 150_000)]
    cat = np.concatenate(blocks, axis=0)
    assert cat.dtype == np.float32 and cat.shape == want.shape and relerr(cat, want) < 1e-6
