"""Every BASELINE.json config on the HIP path (through `so.sink`, i.e. the C-ABI) against the
oracle: miniatures of configs 1-5 live in tests/cases.py; here are config 4's miniature, the
sharded evaluation driven on one GPU, the full BASELINE sizes of configs 2 and 3 against the
oracle on identical host-generated inputs, config 5's slab at its full size through
size-independent properties, the `randn` leaf (reference src/functions.jl:98-114) and the raw
SOS filter path (src/filters.jl:89-97)."""
import numpy as np
import pytest

import sigops_amd as so
from sigops_amd import sharding
from oracle_bridge import oracle_sink, relerr

pytestmark = pytest.mark.gpu


def F(a):
    return np.asfortranarray(a)


def scenes(n=8, frames=30000, seed=100):
    out = []
    for k in range(n):
        noise = F(np.random.default_rng(seed + k).standard_normal((frames, 2)))
        tone = so.Signal(so.sin, ω=(500 + 25 * k) * so.Hz) | so.Until(frames * so.frames)
        out.append(so.Mix(tone, so.Signal(noise, 44.1 * so.kHz)) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz)
                   | so.Ramp(10 * so.ms))
    return out


def test_config4_miniature():
    """Append of 8 x (Mix(sin, noise) |> Filt(Bandstop) |> Ramp) scenes.  Each scene is longer than
    one filter block, where the reference's Append never leaves a filtered child (quirk C-7,
    tests/test_oracle_dsp.py): the engine concatenates, so the expected value is the
    concatenation of the scenes' own oracle results."""
    sc = scenes()
    got, fs = so.sink(so.Append(*sc))
    want = np.concatenate([oracle_sink(s) for s in sc])
    assert fs == 44100.0 and got.shape == want.shape == (240000, 2)
    assert relerr(got, want) < 1e-11


def test_config4_short_scenes_match_the_reference_append():
    """scenes of one filter block (<= 4096 frames): here the reference's Append does move on, and
    the oracle evaluates the Append itself"""
    sc = scenes(n=6, frames=3000)
    x = so.Append(*sc)
    got, _ = so.sink(x)
    assert relerr(got, oracle_sink(x)) < 1e-11


@pytest.mark.parametrize("world", [1, 2, 3])
def test_append_sharding_drives_the_engine(world):
    """sharding.shard_append partitions evaluated by the HIP engine (device-resident slabs,
    gather=False) and reassembled == the unsharded engine result, bit for bit"""
    sc = scenes(n=5, frames=20000)
    x = so.Append(*sc)
    whole, _ = so.sink(x)
    pieces, pos = [], 0
    for r in range(world):
        slab, start = sharding.sink_append_sharded(x, rank=r, world=world, gather=False)
        assert start == pos and slab.is_cuda
        pieces.append(slab.cpu().numpy())
        pos += slab.shape[0]
    full = np.concatenate(pieces)
    assert np.array_equal(full, whole)
    assert np.array_equal(sharding.sink_append_sharded(x, rank=0, world=1).cpu().numpy(), whole)


@pytest.mark.parametrize("world", [1, 2, 4])
def test_channel_sharding_drives_the_engine(world):
    """config 5's shape in miniature: channel slabs of Filt(Lowpass) |> ToFramerate(16 kHz)"""
    x0 = F(np.random.default_rng(7).random((30000, 6)))
    x = so.Signal(x0, 44.1 * so.kHz) | so.Filt(so.Lowpass, 4 * so.kHz) | so.ToFramerate(16 * so.kHz)
    whole, _ = so.sink(x)
    cols = []
    for r in range(world):
        slab, c0, c1 = sharding.sink_channels_sharded(x, rank=r, world=world, gather=False)
        assert slab.shape[1] == c1 - c0
        cols.append(slab.cpu().numpy())
    full = np.concatenate(cols, axis=1)
    assert np.array_equal(full, whole)
    assert relerr(full, oracle_sink(x)) < 1e-11


def test_config2_full_size():
    """BASELINE config 2 at its own size: Mix(sin 1 kHz, noise[2 646 000 x 2]) |> Filt(Bandstop 0.5-2 kHz)"""
    n = 2_646_000
    noise = F(np.random.default_rng(1983).standard_normal((n, 2)))
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


def test_config3_full_size():
    """BASELINE config 3 at its own size (26 460 000 x 8 -> 28 800 000 x 8) against the oracle on the
    same host-generated noise; the oracle's positions are DSP.jl's phase accumulator."""
    n = 26_460_000
    noise = F(np.random.default_rng(1983).standard_normal((n, 8)))
    tree = (so.Signal(noise, 44.1 * so.kHz) | so.Amplify(so.Signal(so.sin, ω=5 * so.Hz)) | so.Until(600 * so.s)
            | so.ToFramerate(48 * so.kHz))
    got, fs = so.sink(tree)
    assert fs == 48000.0 and got.shape == (28_800_000, 8)
    want = oracle_sink(tree)
    # measured 1.3e-9: the accumulator's alpha drifts from the tap tables' exact alpha by up to
    # 4e-8 of a phase step after 2.9e7 additions of the rounded delta (the bound is 1e-6)
    err = relerr(got, want)
    assert err < 1e-8, err
    assert relerr(got[-1_000_000:], want[-1_000_000:]) < 2e-8  # largest accumulated phase error
    assert np.abs(got - want).max() < 1e-7


def test_north_star_pipeline_at_config3_size():
    """Mix -> Filt(Bandstop) -> ToFramerate(48 kHz), 8 ch, 600 s: the engine's result (device
    resident) against the oracle on a 60 s prefix (the filter is causal: a prefix of the input
    gives a prefix of the output, up to the resampler's look-ahead at the prefix end) and
    finiteness / energy checks over the whole result."""
    torch = pytest.importorskip("torch")
    g = torch.Generator(device="cuda")
    g.manual_seed(1983)
    n = 26_460_000
    noise = torch.randn((8, n), dtype=torch.float64, device="cuda", generator=g).t()

    def tree(z, nn):
        return (so.Mix(so.Signal(so.sin, ω=1 * so.kHz), so.Signal(z, 44.1 * so.kHz)) | so.Until(nn * so.frames)
                | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.ToFramerate(48 * so.kHz))

    res, fs = so.sink(tree(noise, n), "torch")
    assert fs == 48000.0 and tuple(res.shape) == (28_800_000, 8)
    assert bool(torch.isfinite(res).all())
    m = 2_646_000
    pre = F(noise[:m].cpu().numpy())
    want = oracle_sink(tree(pre, m))
    k = want.shape[0] - 2000
    assert relerr(res[:k].cpu().numpy(), want[:k]) < 1e-9
    # the band-stop removes the 1 kHz tone and ~1/15 of the white noise: power stays near 1
    p = float((res[1_000_000:] ** 2).mean().item())
    assert 0.85 < p < 1.0


def test_config5_slab_full_size_properties():
    """config 5's per-GPU slab at its own size, x[10 000 000 x 128] |> Filt(Lowpass 4 kHz) |>
    ToFramerate(16 kHz) (10.24 GB in): linearity, DC gain, and a 1 kHz sine in = 1 kHz sine out
    (the resampler's 60 dB design ripple bounds the error)."""
    torch = pytest.importorskip("torch")
    n, nch = 10_000_000, 128
    g = torch.Generator(device="cuda")
    g.manual_seed(5)

    def pipe(z):
        return so.Signal(z, 44.1 * so.kHz) | so.Filt(so.Lowpass, 4 * so.kHz) | so.ToFramerate(16 * so.kHz)

    a = torch.rand((nch, n), dtype=torch.float64, device="cuda", generator=g)
    ya, fs = so.sink(pipe(a.t()), "torch")
    assert fs == 16000.0 and tuple(ya.shape) == (3_628_118, nch)
    assert bool(torch.isfinite(ya).all())
    assert abs(float(ya[5000:-5000].mean().item()) - 0.5) < 1e-3  # uniform(0,1): DC passes with gain 1
    # channel 0 <- a 1 kHz sine (first frame at t = 1/fs); linearity against the all-noise run on a prefix
    t = torch.arange(1, n + 1, dtype=torch.float64, device="cuda") / 44100.0
    b = a.clone()
    b[0] = torch.sin(2 * np.pi * 1000.0 * t)
    yb, _ = so.sink(pipe(b.t()), "torch")
    assert torch.equal(yb[:, 1:], ya[:, 1:])  # channels are independent, bit for bit
    tout = 1.0 / 44100.0 + torch.arange(yb.shape[0], dtype=torch.float64, device="cuda") / 16000.0
    # order-5 Butterworth low-pass at 4 kHz, evaluated at 1 kHz (designed at 16 kHz): gain and phase
    from scipy import signal as sps
    zb, pb, kb = sps.butter(5, 4000.0, "lowpass", fs=16000.0, output="zpk")
    _, hresp = sps.freqz_zpk(zb, pb, kb, worN=[2 * np.pi * 1000.0 / 16000.0])
    want = abs(hresp[0]) * torch.sin(2 * np.pi * 1000.0 * tout + float(np.angle(hresp[0])))
    sl = slice(20000, yb.shape[0] - 20000)
    assert float((yb[sl, 0] - want[sl]).abs().max().item()) < 2e-3


def test_randn_leaf():
    """`Signal(randn)` (reference src/functions.jl:98-114): one N(0,1) draw per evaluated frame in
    increasing frame order from the leaf's own generator; host-materialised for the engine."""
    draws = np.random.default_rng(42).standard_normal(5000)
    x = so.Signal(so.randn, 1 * so.kHz, rng=np.random.default_rng(42)) | so.Until(5000 * so.frames)
    got, fs = so.sink(x)
    assert fs == 1000.0 and np.array_equal(got[:, 0], draws)
    # mixed with an array and filtered: the same tree with the draws as an array leaf
    a = F(np.random.default_rng(1).standard_normal((5000, 2)))
    t1 = so.Mix(so.Signal(so.randn, 1 * so.kHz, rng=np.random.default_rng(42)), so.Signal(a, 1 * so.kHz)) \
        | so.Until(5000 * so.frames) | so.Filt(so.Lowpass, 100 * so.Hz)
    t2 = so.Mix(so.Signal(F(draws[:, None]), 1 * so.kHz), so.Signal(a, 1 * so.kHz)) | so.Filt(so.Lowpass, 100 * so.Hz)
    assert relerr(so.sink(t1)[0], oracle_sink(t2)) < 1e-11


def test_randn_leaf_under_a_resampler():
    """an infinite randn child resampled as a whole (`After` makes it a data signal, reference
    src/cutting.jl:138): the draws the resampler reaches include the filter's look-ahead, so the
    last outputs are noise, not the zero padding of a too-short materialisation"""
    n_out = 4000
    x = (so.Signal(so.randn, 44.1 * so.kHz, rng=np.random.default_rng(7)) | so.After(100 * so.frames)
         | so.ToFramerate(16 * so.kHz) | so.Until(n_out * so.frames))
    got, fs = so.sink(x)
    assert fs == 16000.0 and got.shape == (n_out, 1)
    # the frames `After` skips draw nothing: the first draw is frame 101
    draws = np.concatenate([np.zeros(100), np.random.default_rng(7).standard_normal(20000)])
    ref = (so.Signal(F(draws[:, None]), 44.1 * so.kHz) | so.After(100 * so.frames) | so.ToFramerate(16 * so.kHz)
           | so.Until(n_out * so.frames))
    assert relerr(got, oracle_sink(ref)) < 1e-11
    assert np.abs(got[-50:]).max() > 1e-3


def test_randn_skipped_frames_draw_nothing():
    """After over a generator: the dropped frames are pulled with skip=true and never evaluated
    (reference src/cutting.jl:160-181), so the kept frames start with the generator's FIRST draws;
    a Filt in between ignores the flag (its state needs the frames) and consumes them"""
    d = np.random.default_rng(9).standard_normal(400)
    x = so.Signal(so.randn, 1 * so.kHz, rng=np.random.default_rng(9)) | so.After(100 * so.frames) | so.Until(50 * so.frames)
    assert np.array_equal(so.sink(x)[0][:, 0], d[:50])
    y = (so.Signal(so.randn, 1 * so.kHz, rng=np.random.default_rng(9)) | so.Filt(so.Lowpass, 100 * so.Hz)
         | so.After(100 * so.frames) | so.Until(50 * so.frames))
    ref = so.Signal(F(d[:, None]), 1 * so.kHz) | so.Filt(so.Lowpass, 100 * so.Hz) | so.After(100 * so.frames) | so.Until(50 * so.frames)
    assert relerr(so.sink(y)[0], oracle_sink(ref)) < 1e-11


def test_raw_sos_filter_object():
    """Filt(x, h) with a raw filter object given as SOS rows + gain (reference src/filters.jl:89-97)
    == the same filter designed by name, bit for bit (runtests.jl:365-368 `Array(high) == Array(high4)`),
    and == scipy's sosfilt"""
    from scipy import signal as sps

    x = F(np.random.default_rng(3).standard_normal((20000, 3)))
    sig = so.Signal(x, 100 * so.Hz)
    named = sig | so.Filt(so.Highpass, 8 * so.Hz, method=so.Chebyshev1(5, 1))
    sos, gain = so.design_iir(so.FilterFn("highpass", ("chebyshev1", 5, 1.0), (8.0,)), 100.0)
    raw = sig | so.Filt(sos=sos, gain=gain)
    a, b = so.sink(named)[0], so.sink(raw)[0]
    assert np.array_equal(a, b)
    assert relerr(b, oracle_sink(raw)) < 1e-11
    assert relerr(b, sps.sosfilt(sos, x, axis=0) * gain) < 1e-11


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_time_sharding_drives_the_engine(world):
    """ONE north-star pipeline cut along time into `world` ranges (sharding.shard_time): every range is
    evaluated by the HIP engine from a warm start, the slabs reassemble to the unsharded result"""
    n = 400000
    x0 = F(np.random.default_rng(8).standard_normal((n, 8)))
    x = (so.Mix(so.Signal(so.sin, ω=1 * so.kHz), so.Signal(x0, 44.1 * so.kHz)) | so.Until(n * so.frames)
         | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.ToFramerate(48 * so.kHz))
    whole, _ = so.sink(x)
    pieces, pos = [], 0
    for r in range(world):
        slab, start = sharding.sink_time_sharded(x, rank=r, world=world, gather=False, align=160 * 16)
        assert start == pos and slab.is_cuda
        pieces.append(slab.cpu().numpy())
        pos += slab.shape[0]
    full = np.concatenate(pieces)
    assert full.shape == whole.shape
    assert np.linalg.norm(full - whole) <= 1e-12 * np.linalg.norm(whole)
