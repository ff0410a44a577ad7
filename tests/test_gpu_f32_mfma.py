"""A Float32 signal through the periodic resampler on the Float32 MFMA (k_resample_periodic F32M: Float32 tile, Float32-rounded
taps, a k-ordered chain of Float32 fmas per output; reference: a Float32 signal stays Float32, src/reformatting.jl:92-98, and
Float32 results are compared at 1e-6, test/runtests.jl:707-729).  Against the oracle at the contract's 1e-6 and against the
engine's own Float64 products rounded once (`SIGOPS_RS_NO_F32MFMA=1`) at 3e-7 -- the margin the soak recorded in
profiles/r05/relerr_maxima_f32mfma.json was asked to keep (largest seen: 1.3e-7)."""
import numpy as np
import pytest

import sigops_amd as so
from oracle_bridge import oracle_sink, relerr
from test_gpu_rsos import F, env

pytestmark = pytest.mark.gpu


def both(x):
    with env(SIGOPS_RS_NO_F32MFMA=None):
        a = so.sink(x)[0]
    with env(SIGOPS_RS_NO_F32MFMA=1):
        b = so.sink(x)[0]
    return a, b


@pytest.mark.parametrize("nch", [4, 8, 16])
@pytest.mark.parametrize("kind", ["noise", "low tone", "dc", "clicks"])
def test_float32_resampling_on_the_float32_mfma(nch, kind):
    rng = np.random.default_rng(31 + nch)
    n = 250000
    t = np.arange(n) / 44100.0
    d = {"noise": rng.standard_normal((n, nch)), "low tone": 0.9 * np.sin(2 * np.pi * 50 * t)[:, None] * np.ones((1, nch)),
         "dc": 1.0 + 1e-3 * rng.standard_normal((n, nch)), "clicks": (rng.random((n, nch)) < 1e-3) * 1.0}[kind]
    x = so.Signal(F(d.astype(np.float32)), 44.1 * so.kHz) | so.ToFramerate(48 * so.kHz)
    a, b = both(x)
    want = oracle_sink(x)
    assert a.dtype == np.float32 and a.shape == want.shape
    assert relerr(a, want) < 1e-6 and relerr(b, want) < 1e-6
    assert relerr(a, b) < 3e-7
    assert not np.array_equal(a, b)  # (the Float32 instruction did run: another rounding of the same sums)


def test_windows_and_float64_signals_are_untouched():
    """only Float32 STAGES take it: a Float64 signal over the same samples gives the Float64 products' values with or without"""
    rng = np.random.default_rng(32)
    d = rng.standard_normal((200000, 8))
    x = so.Signal(F(d), 44.1 * so.kHz) | so.ToFramerate(48 * so.kHz)
    a, b = both(x)
    assert a.dtype == np.float64 and np.array_equal(a, b)
    x32 = so.Signal(F(d.astype(np.float32)), 44.1 * so.kHz) | so.ToFramerate(48 * so.kHz) | so.After(50000 * so.frames) | so.Until(60000 * so.frames)
    a, b = both(x32)
    assert relerr(a, oracle_sink(x32)) < 1e-6 and relerr(a, b) < 3e-7


def test_local_error_on_sixty_decibels_of_dynamic_range():
    """The maxima above are norm-wise: a loud passage hides what happens in a quiet one next to it.  One second loud, one second
    60 dB down, and DC plus small detail (where the taps' partial sums cancel): BLOCKWISE relative error (1024 outputs) of the
    Float32-MFMA form against the oracle -- a few Float32 rounding units of the LOCAL level -- and next to the Float64 products'
    own (the same data rounded once): the Float32 accumulation costs local accuracy a factor, not orders of magnitude."""
    rng = np.random.default_rng(33)
    n = 88200
    env_ = np.where((np.arange(4 * n) // n) % 2 == 0, 1.0, 1e-3)[:, None]
    for d in ((rng.standard_normal((4 * n, 8)) * env_), 1.0 + 1e-3 * rng.standard_normal((4 * n, 8)) * env_):
        x = so.Signal(F(d.astype(np.float32)), 44.1 * so.kHz) | so.ToFramerate(48 * so.kHz)
        a, b = both(x)
        want = oracle_sink(x).astype(np.float64)
        nb = want.shape[0] // 1024
        w = want[: nb * 1024].reshape(nb, 1024, -1)
        errs = []
        for got in (a, b):
            e = np.linalg.norm(got[: nb * 1024].astype(np.float64).reshape(nb, 1024, -1) - w, axis=(1, 2)) / np.linalg.norm(w, axis=(1, 2))
            errs.append(e)
        assert errs[0].max() < 2e-6 and np.median(errs[0]) < 3e-7, (errs[0].max(), np.median(errs[0]))
        assert errs[1].max() < 2e-6
        assert errs[0].max() < 8 * max(errs[1].max(), 6e-8)
