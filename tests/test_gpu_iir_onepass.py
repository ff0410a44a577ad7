"""K2 single pass (k_sos_onepass, opt-in with SIGOPS_SOS_ONEPASS=1): wave tiles taken in time order
by ticket, zero-state end states published per tile, look-back over the previous kt tiles, DF2T
from the propagated state.
Replaces DSP.jl's `filt!(out, DF2TFilter, x)` at reference src/filters.jl:252-255.  Checked
against the oracle's sequential DF2T, against the three-pass form (SIGOPS_SOS_3PASS=1) and for
run-to-run determinism (the kernel synchronises workgroups through global flags)."""
import numpy as np
import pytest

import sigops_amd as so
from oracle_bridge import oracle_sink, relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _onepass(monkeypatch):
    """the single-pass kernel is opt-in (the three-pass form is faster on MI355X)"""
    monkeypatch.setenv("SIGOPS_SOS_ONEPASS", "1")


def _x(seed, n, nch, dt=np.float64):
    return np.asfortranarray(np.random.default_rng(seed).standard_normal((n, nch)).astype(dt))


@pytest.mark.parametrize("nch,n,dt", [(8, 400_000, np.float64), (2, 300_000, np.float64), (1, 100_000, np.float64),
                                      (3, 70_000, np.float64), (5, 33_333, np.float32), (17, 20_001, np.float64),
                                      (8, 4096, np.float64), (4, 262_144, np.float32)])
def test_onepass_matches_sequential_df2t(nch, n, dt):
    x = _x(1, n, nch, dt)
    tree = so.Signal(x, 48 * so.kHz) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz)
    got = so.sink(tree)[0]
    want = oracle_sink(tree)
    assert got.dtype == want.dtype == dt
    assert relerr(got, want) < (1e-11 if dt == np.float64 else 2e-7)


def test_config2_full_size():
    """BASELINE config 2 at its own size: Mix(sin 1 kHz, noise[2 646 000 x 2]) |> Filt(Bandstop 0.5-2 kHz)"""
    n = 2_646_000
    noise = _x(1983, n, 2)
    tree = (so.Mix(so.Signal(so.sin, ω=1 * so.kHz) | so.Until(n * so.frames), so.Signal(noise, 44.1 * so.kHz))
            | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz))
    got, fs = so.sink(tree)
    want = oracle_sink(tree)
    assert fs == 44100.0 and got.shape == want.shape == (n, 2)
    # (1e-10, not 1e-11: the filter adds the tone in its own loads by one rotation per frame from an exact
    #  value per chunk (SosGeom::src_op); the reference rounds (n/fs)*omega per frame, which at 6e4 cycles
    #  is +-1e-11 of a cycle of its own -- the difference of the two is that noise)
    assert relerr(got, want) < 1e-10
    assert relerr(got[-100000:], want[-100000:]) < 1e-10


@pytest.mark.parametrize("spec", [
    ("lowpass 30 Hz: long memory, many look-back terms", lambda s: so.Filt(s, so.Lowpass, 30 * so.Hz)),
    ("highpass 2 Hz: memory beyond the look-back window, three-pass form", lambda s: so.Filt(s, so.Highpass, 2 * so.Hz)),
    ("order-12 bandpass: two section groups in place", lambda s: so.Filt(s, so.Bandpass, 1 * so.kHz, 3 * so.kHz, method=so.Butterworth(6))),
    ("order-9 chebyshev lowpass: 5 sections, odd order", lambda s: so.Filt(s, so.Lowpass, 5 * so.kHz, method=so.Chebyshev1(9, 1.0))),
    ("order-20 bandstop: 20 sections, three groups", lambda s: so.Filt(s, so.Bandstop, 4 * so.kHz, 9 * so.kHz, method=so.Butterworth(10))),
])
def test_filter_memory_and_section_groups(spec):
    x = _x(2, 200_000, 4)
    tree = spec[1](so.Signal(x, 44.1 * so.kHz))
    got = so.sink(tree)[0]
    want = oracle_sink(tree)
    # (2 Hz high-pass at 44.1 kHz: poles at 1 - 1e-4.  The one Float64 bound of the suite above 1e-8 (observed 1.3e-8,
    #  profiles/r04/relerr_maxima.json): this filter's memory is longer than the signal, and its own Float64 recurrence
    #  -- DSP.jl's `filt!`, the call at reference src/filters.jl:252-255 -- is that far from its 80-bit evaluation
    #  ("what remains above 1e-8"), so any other association of the same sums differs from it by as
    #  much; the sequential kernel that reproduces the reference's order bit for bit is kept for cascades whose
    #  conditioning probe fails, not for every DC blocker.)
    assert relerr(got, want) < (2e-8 if "2 Hz" in spec[0] else 1e-9), spec[0]


def test_onepass_equals_three_pass(monkeypatch):
    x = _x(3, 500_000, 8)
    tree = so.Signal(x, 48 * so.kHz) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.After(1000 * so.frames)
    one = so.sink(tree)[0]
    monkeypatch.setenv("SIGOPS_SOS_3PASS", "1")
    three = so.sink(tree)[0]
    assert relerr(one, three) < 1e-13


def test_onepass_is_deterministic():
    torch = pytest.importorskip("torch")
    g = torch.Generator(device="cuda")
    g.manual_seed(9)
    n = 48000 * 60
    noise = torch.randn((8, n), dtype=torch.float64, device="cuda", generator=g).t()
    x = so.Signal(noise, 48 * so.kHz) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz)
    out_t = torch.empty((8, n), dtype=torch.float64, device="cuda")
    out = out_t.t()
    plan = so.Plan(so.ToChannels(x, 8), (n, 8), np.float64, (out.stride(0), out.stride(1)), True)
    stream = torch.cuda.current_stream().cuda_stream
    try:
        plan.execute(out.data_ptr(), stream)
        torch.cuda.synchronize()
        ref = out_t.clone()
        assert not torch.isnan(ref).any()
        for _ in range(30):
            out_t.fill_(float("nan"))
            plan.execute(out.data_ptr(), stream)
            torch.cuda.synchronize()
            assert torch.equal(out_t, ref)
    finally:
        plan.close()


def test_linearity_at_full_length():
    """size-independent property at config 3's output length (28.8 M frames x 8): the filter is
    linear and time invariant -- a delayed impulse pair gives the delayed sum of impulse responses"""
    torch = pytest.importorskip("torch")
    n = 28_800_000
    a = torch.zeros((8, n), dtype=torch.float64, device="cuda")
    d0, d1 = 12_345, 20_000_003
    a[:, d0] = 1.0
    a[3, d1] = -2.0
    tree = so.Signal(a.t(), 48 * so.kHz) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz)
    res, _ = so.sink(tree, "torch")
    imp = np.zeros((6000, 1))
    imp[0] = 1.0
    h = oracle_sink(so.Signal(np.asfortranarray(imp), 48 * so.kHz) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz))[:, 0]
    got0 = res[d0:d0 + 6000, 0].cpu().numpy()
    got1 = res[d1:d1 + 6000, 3].cpu().numpy()
    assert np.abs(got0 - h).max() < 1e-12
    assert np.abs(got1 + 2.0 * h).max() < 1e-12
    assert float(res[:d0, :].abs().max()) == 0.0


@pytest.mark.parametrize("nch,n,fo", [(8, 300_000, 48.0), (4, 120_000, 48.0), (8, 40_000, 96.0)])
def test_state_pass_fused_into_the_resampler(nch, n, fo, monkeypatch):
    """SIGOPS_FUSE_STATE=1: a periodic resampler whose only consumer is an SOS filter computes the
    filter's per-period zero-state end states with two state waves (v = (G.Tap) x on the staged
    LDS tile); K2 then runs combine + scan + output pass only.  Same values as the three-pass form
    and as the oracle (Mix fused into the resampler's staging, as in the north-star pipeline)."""
    x = _x(11, n, nch)
    tree = (so.Mix(so.Signal(so.sin, ω=1 * so.kHz), so.Signal(x, 44.1 * so.kHz)) | so.Until(n * so.frames)
            | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.ToFramerate(fo * so.kHz))
    monkeypatch.delenv("SIGOPS_SOS_ONEPASS", raising=False)
    plain = so.sink(tree)[0]
    monkeypatch.setenv("SIGOPS_FUSE_STATE", "1")
    fused = so.sink(tree)[0]
    assert relerr(fused, plain) < 1e-12
    assert relerr(fused, oracle_sink(tree)) < 1e-9
