"""GPU parity for warm starts: a stateful stage whose first frames nobody reads (`After`, a later
block of `so.stream`) starts from zero state a decay time before the first frame that is read
instead of at frame 0.  The reference runs the skipped frames through the filter
(src/cutting.jl:160-173); the oracle does the same, so these are plain parity tests."""
import os

import numpy as np
import pytest

import sigops_amd as so
from oracle_bridge import oracle_sink, relerr

pytestmark = pytest.mark.gpu


def _noise(rng, n, nch, dt=np.float64):
    return np.asfortranarray(rng.standard_normal((n, nch)).astype(dt))


def _check(tree, tol=1e-9):
    want = oracle_sink(tree)
    got = so.sink(tree, so.Array)
    os.environ["SIGOPS_NO_WARM_START"] = "1"
    try:
        ref = so.sink(tree, so.Array)
    finally:
        os.environ.pop("SIGOPS_NO_WARM_START", None)
    assert got.shape == want.shape and got.dtype == want.dtype
    assert relerr(ref, want) <= tol
    assert relerr(got, want) <= tol
    assert relerr(got, ref) <= 1e-13 if got.dtype == np.float64 else 1e-6
    return got


@pytest.mark.parametrize("nch", [1, 2, 8])
def test_filter_then_after(nch):
    rng = np.random.default_rng(21 + nch)
    x = so.Signal(_noise(rng, 90000, nch), 44.1 * so.kHz)
    _check(x | so.Filt(so.Lowpass, 3 * so.kHz) | so.After(70001 * so.frames))
    _check(x | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.After(65000 * so.frames) | so.Until(9999 * so.frames))


def test_cascaded_groups_and_chained_filters():
    rng = np.random.default_rng(22)
    x = so.Signal(_noise(rng, 120000, 2), 44.1 * so.kHz)
    # order 10 band-pass = 10 sections = two cascaded groups of the IIR kernel
    _check(x | so.Filt(so.Bandpass, 1 * so.kHz, 4 * so.kHz, order=10) | so.After(100000 * so.frames))
    # filter reading a filter (direct stage-buffer source), both warm-started
    _check(x | so.Filt(so.Highpass, 0.3 * so.kHz) | so.Filt(so.Lowpass, 5 * so.kHz) | so.After(90000 * so.frames))
    # pointwise work between them and a gain after
    _check(x | so.Filt(so.Highpass, 0.3 * so.kHz) | so.Amplify(0.5) | so.Filt(so.Lowpass, 5 * so.kHz)
           | so.After(90000 * so.frames) | so.Ramp(100 * so.frames))


def test_float32_and_mixed_consumers():
    rng = np.random.default_rng(23)
    x = so.Signal(_noise(rng, 80000, 2, np.float32), 44.1 * so.kHz)
    _check(x | so.Filt(so.Lowpass, 3 * so.kHz) | so.After(60000 * so.frames), tol=2e-6)
    y = so.Signal(_noise(rng, 80000, 2), 44.1 * so.kHz)
    f = y | so.Filt(so.Lowpass, 3 * so.kHz)
    # the same stage read at two offsets: the earlier one decides where it starts
    _check(so.Mix(f | so.After(60000 * so.frames), f | so.After(30000 * so.frames) | so.Until(20000 * so.frames)))
    # under an Append, next to an unfiltered child
    _check(so.Append(y | so.Until(5000 * so.frames), f | so.After(70000 * so.frames)))
