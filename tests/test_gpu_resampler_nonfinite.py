"""Where a non-finite input sample shows in a resampler's result (missing 6: "a superset, documented, not pinned by
a test that states the set").  The reference's polyphase kernel multiplies a sample by every tap of its phase's row; the
periodic MFMA kernels (K3, and the fused resampler + IIR's front half) multiply a group's 16 outputs by one window of KS k-steps
that is padded with zero taps on either side, and 0 * NaN is NaN: an output is non-finite when the sample lies in the padded
window of its GROUP of 16 outputs (groups are the aligned runs [16 g, 16 g + 16) of the result).  Stated here: the engine's set
contains the oracle's, stays within the aligned 16-output groups the oracle's set touches plus at most one group on either side,
never reaches another channel, and everything outside it is the oracle's value."""
import numpy as np
import pytest

import sigops_amd as so
from oracle_bridge import oracle_sink, relerr
from test_gpu_rsos import F, env, steps_of

pytestmark = pytest.mark.gpu


def groups_of(mask_col):
    return set(np.nonzero(mask_col)[0] // 16)


@pytest.mark.parametrize("rates", [(44.1, 48.0), (48.0, 44.1), (22.05, 24.0), (32.0, 48.0), (44.1, 16.0)])
@pytest.mark.parametrize("nch", [8, 4, 2])
def test_the_set_of_non_finite_outputs_of_the_periodic_resampler(rates, nch):
    fi, fo = rates
    rng = np.random.default_rng(int(fi * 10) + nch)
    n = 200_003
    d = rng.standard_normal((n, nch))
    where = {0: [777], nch - 1: [100_000, 150_123]}
    for c, idx in where.items():
        for k, i in enumerate(idx):
            d[i, c] = np.nan if k % 2 == 0 else np.inf
    x = so.Signal(F(d), fi * so.kHz) | so.ToFramerate(fo * so.kHz)
    assert any("resample" in s for s in steps_of(x)), steps_of(x)
    got = so.sink(x)[0]
    want = oracle_sink(x)
    bad_g, bad_w = ~np.isfinite(got), ~np.isfinite(want)
    assert bad_w.any()
    if steps_of(x) == ["k_resample_rows"]:
        # (the row-tiled kernel redoes a unit that came out non-finite output by output itself: its tile is still in LDS)
        assert np.array_equal(bad_g, bad_w), [(c, np.nonzero(bad_g[:, c] != bad_w[:, c])[0][:4]) for c in range(nch) if (bad_g[:, c] != bad_w[:, c]).any()]
        assert relerr(got[~bad_g], want[~bad_w]) < 1e-9
    if steps_of(x) == ["k_resample_periodic"]:
        # round 6: the periodic kernel lists the (tile, group)s a non-finite sample reached and k_rs_fixup recomputes them output
        # by output -- the set is the reference's, and so are the values around it
        assert np.array_equal(bad_g, bad_w), [(c, np.nonzero(bad_g[:, c] != bad_w[:, c])[0][:4]) for c in range(nch) if (bad_g[:, c] != bad_w[:, c]).any()]
        assert relerr(got[~bad_g], want[~bad_w]) < 1e-9
        with env(SIGOPS_RS_NO_FIXUP=1):   # (the kernel's own set, as stated below)
            got = so.sink(x)[0]
        bad_g = ~np.isfinite(got)
        assert not np.array_equal(bad_g, bad_w)
    for c in range(nch):
        if c not in where:
            assert not bad_g[:, c].any(), c  # never another channel
            continue
        assert not (bad_w[:, c] & ~bad_g[:, c]).any()  # contains the reference's set
        gw, gg = groups_of(bad_w[:, c]), groups_of(bad_g[:, c])
        allowed = set()
        for g in gw:
            allowed |= {g - 1, g, g + 1}
        assert gg <= allowed, (c, sorted(gg - allowed)[:5])
        assert bad_g[:, c].sum() <= bad_w[:, c].sum() + 32 * len(where[c])  # at most a group on either side per sample
    ok = ~bad_g
    assert relerr(got[ok], want[ok]) < 1e-9
