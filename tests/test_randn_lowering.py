"""Host logic of the `randn` leaf (reference src/functions.jl:98-114), no GPU: the leaf is
materialised on the host for the frames the sink evaluates -- one draw per EVALUATED frame in
increasing order, nothing for frames an `After` skips, and enough frames for a resampler's
look-ahead.  Checked through the CPU oracle, which consumes the same lowered node table."""
import numpy as np

import sigops_amd as so
from cases import F
from oracle_bridge import oracle_sink


def test_one_draw_per_evaluated_frame():
    d = np.random.default_rng(42).standard_normal(300)
    x = so.Signal(so.randn, 1 * so.kHz, rng=np.random.default_rng(42)) | so.Until(300 * so.frames)
    assert np.array_equal(oracle_sink(x)[:, 0], d)


def test_skipped_frames_draw_nothing():
    d = np.random.default_rng(9).standard_normal(400)
    x = so.Signal(so.randn, 1 * so.kHz, rng=np.random.default_rng(9)) | so.After(100 * so.frames) | so.Until(50 * so.frames)
    assert np.array_equal(oracle_sink(x)[:, 0], d[:50])
    # a Filt below the cut ignores the skip flag: its state needs the frames, so they are drawn
    y = (so.Signal(so.randn, 1 * so.kHz, rng=np.random.default_rng(9)) | so.Filt(so.Lowpass, 100 * so.Hz)
         | so.After(100 * so.frames) | so.Until(50 * so.frames))
    ref = (so.Signal(F(d[:, None]), 1 * so.kHz) | so.Filt(so.Lowpass, 100 * so.Hz) | so.After(100 * so.frames)
           | so.Until(50 * so.frames))
    assert np.array_equal(oracle_sink(y), oracle_sink(ref))


def test_resampler_lookahead_is_materialised():
    """the demand under a resampler includes the filter's group delay"""
    n_out = 4000
    x = (so.Signal(so.randn, 44.1 * so.kHz, rng=np.random.default_rng(7)) | so.After(100 * so.frames)
         | so.ToFramerate(16 * so.kHz) | so.Until(n_out * so.frames))
    d = np.concatenate([np.zeros(100), np.random.default_rng(7).standard_normal(20000)])
    ref = (so.Signal(F(d[:, None]), 44.1 * so.kHz) | so.After(100 * so.frames) | so.ToFramerate(16 * so.kHz)
           | so.Until(n_out * so.frames))
    assert np.array_equal(oracle_sink(x), oracle_sink(ref))
