"""The HIP resampler against the reference's position rule: DSP.jl's FIRArbitrary accumulates
the phase in Float64, one addition per output (reference src/reformatting.jl:92-98 `FIRFilter(h,
ratio)` + `setphase!`, src/filters.jl:252-255 `filt!`).  The oracle's default mode restates
that; the engine's kernels use closed-form positions whose tap tables are built from a host
replay of the accumulator, plus a sparse fix-up pass (k_resample_fix) for what a periodic table
cannot express.  Bound: 1e-6 norm-wise (BASELINE.json north_star); observed ~1e-9 or better."""
import os

import numpy as np
import pytest

import sigops_amd as so
from oracle_bridge import oracle_sink, oracle_positions, relerr
from test_resample_positions import engine_positions

pytestmark = pytest.mark.gpu


def test_config3_rate_follows_the_phase_accumulator():
    """44.1 -> 48 kHz, 8 channels, 1.6 M frames (the headline kernel, K3 MFMA ring): the result is
    the accumulator's, not the closed form's (those two differ by 4e-5 norm-wise)."""
    rng = np.random.default_rng(101)
    n = 1_600_000
    x = np.asfortranarray(rng.standard_normal((n, 8)))
    tree = so.Signal(x, 44.1 * so.kHz) | so.ToFramerate(48 * so.kHz)
    got, fs = so.sink(tree)
    want = oracle_sink(tree)
    assert fs == 48000.0 and got.shape == want.shape == (1741497, 8)
    err = relerr(got, want)
    assert err < 1e-9, err
    with oracle_positions("exact"):
        closed = oracle_sink(tree)
    assert 1e-5 < relerr(got, closed) < 1e-4  # (documents the size of what is being matched)
    # the last stretch, where the accumulated phase error is largest
    assert relerr(got[-200000:], want[-200000:]) < 1e-9


def test_config3_tree_follows_the_phase_accumulator():
    """the same with config 3's fused source (Amplify by a 5 Hz sine inside the staging)"""
    rng = np.random.default_rng(102)
    n = 400_000
    x = np.asfortranarray(rng.standard_normal((n, 8)))
    tree = (so.Signal(x, 44.1 * so.kHz) | so.Amplify(so.Signal(so.sin, ω=5 * so.Hz)) | so.Until(n * so.frames)
            | so.ToFramerate(48 * so.kHz))
    assert relerr(so.sink(tree)[0], oracle_sink(tree)) < 1e-9


@pytest.mark.parametrize("fs_in,fs_out", [(44100, 12000), (32000, 11025), (999, 16000), (20, 13), (88200, 24000),
                                          (1001, 16000), (44101, 32000)])
@pytest.mark.parametrize("fused", [False, True])
def test_fixup_pass(fs_in, fs_out, fused):
    """rates whose accumulator replay leaves a non-empty fix-up list (deviations a periodic tap
    table cannot hold, or no periodic table at all): exercised for plain and fused sources"""
    rng = np.random.default_rng(103)
    n = 20000
    nch = 3
    n_out = int(np.ceil(n * fs_out / fs_in))
    nfix = engine_positions(fs_in, fs_out, n_out)[4]
    assert nfix > 0
    x = np.asfortranarray(rng.standard_normal((n, nch)))
    sig = so.Signal(x, float(fs_in) * so.Hz)
    if fused:
        sig = sig | so.Amplify(so.Signal(so.sin, ω=fs_in / 1000.0 * so.Hz)) | so.Until(n * so.frames)
    tree = sig | so.ToFramerate(float(fs_out) * so.Hz)
    got = so.sink(tree)[0]
    want = oracle_sink(tree)
    assert got.shape == want.shape
    assert relerr(got, want) < 1e-9
    # and the listed outputs themselves (a missed fix-up is ~1e-3 of ONE sample: invisible in the norm)
    assert np.abs(got - want).max() < 1e-9 * np.abs(want).max()


def test_float32_fixup_and_narrow_store():
    rng = np.random.default_rng(104)
    x = np.asfortranarray(rng.standard_normal((20000, 4)).astype(np.float32))
    tree = so.Signal(x, 44100 * so.Hz) | so.ToFramerate(12000 * so.Hz)
    got = so.sink(tree)[0]
    want = oracle_sink(tree)
    assert got.dtype == want.dtype == np.float32
    assert relerr(got, want) < 2e-7


def test_closed_form_positions_are_the_opt_in(monkeypatch):
    """SIGOPS_RS_EXACT=1: the engine's kernels with closed-form positions only == the oracle's
    opt-in exact mode (measurement aid for the divergence, not the default on either side)"""
    monkeypatch.setenv("SIGOPS_RS_EXACT", "1")
    rng = np.random.default_rng(105)
    x = np.asfortranarray(rng.standard_normal((50000, 2)))
    tree = so.Signal(x, 44.1 * so.kHz) | so.ToFramerate(48 * so.kHz)
    got = so.sink(tree)[0]
    with oracle_positions("exact"):
        want = oracle_sink(tree)
    assert relerr(got, want) < 1e-11


@pytest.mark.parametrize("fs_in,fs_out,nch,n,dt", [
    (1000.0, 1000.0 * np.pi, 2, 40000, np.float64),      # test/benchmarks.jl "resampling-irrational"
    (44100.5, 48000.0, 8, 60000, np.float64), (48000.0, 44100.5, 3, 50000, np.float64),
    (1000.0, 1000.0 / np.e, 4, 80000, np.float32), (999.0, 16000.0, 1, 9000, np.float64),
    (22050.0, 22050.0 * np.sqrt(2), 16, 30000, np.float64)])
def test_tiled_resampler_for_rates_without_a_period(fs_in, fs_out, nch, n, dt):
    """k_resample_tiled (irrational ratios, non-integer frame rates, very long periods) against
    the oracle; SIGOPS_RS_NOTILED falls back to the thread-per-output kernel: same values"""
    rng = np.random.default_rng(106)
    x = np.asfortranarray(rng.standard_normal((n, nch)).astype(dt))
    tree = so.Signal(x, fs_in * so.Hz) | so.ToFramerate(fs_out * so.Hz)
    got = so.sink(tree)[0]
    want = oracle_sink(tree)
    assert got.shape == want.shape and got.dtype == want.dtype
    assert relerr(got, want) < (1e-9 if dt == np.float64 else 2e-7)
    assert np.abs(got.astype(np.float64) - want).max() < (1e-8 if dt == np.float64 else 1e-6) * np.abs(want).max()


def test_tiled_resampler_equals_the_thread_per_output_kernel(monkeypatch):
    rng = np.random.default_rng(107)
    x = np.asfortranarray(rng.standard_normal((70000, 4)))
    tree = so.Signal(x, 1000 * so.Hz) | so.ToFramerate(1000 * np.pi * so.Hz) | so.After(123 * so.frames)
    a = so.sink(tree)[0]
    monkeypatch.setenv("SIGOPS_RS_NOTILED", "1")
    b = so.sink(tree)[0]
    assert relerr(a, b) < 1e-13


@pytest.mark.parametrize("fs_in,fs_out,nch,n,dt", [
    (1000.0, 1000.0 * np.pi, 8, 70000, np.float64),       # benchmarks.jl "resampling-irrational", eight channels
    (1000.0, 1000.0 * np.pi / 3, 16, 50000, np.float64),  # two channel groups
    (1000.0, 1000.0 * np.pi, 2, 40000, np.float64),
    (44100.5, 48000.0, 3, 60000, np.float64),              # non-integer frame rate, one channel per workgroup
    (1000.0, 1000.0 / np.e, 8, 90000, np.float32),         # down, Float32 samples
    (48000.0, 44100.5, 24, 30000, np.float32),
    (48000.0, 9000.5, 4, 90000, np.float64),               # x 0.19: the windows of a pair five or six frames apart
    (999.0, 16000.0, 8, 9000, np.float64),                 # x 16: many outputs per input
    (22050.0, 22050.0 * np.sqrt(2), 8, 2101, np.float64)])  # barely more than one tile, odd length
def test_two_outputs_per_lane_form_of_the_tiled_resampler(fs_in, fs_out, nch, n, dt, monkeypatch):
    """k_resample_tiled2 (kernels2.hip): a lane walks the union of the input windows of two neighbouring outputs --
    against the oracle (DSP.jl FIRArbitrary: yLower + alpha * yUpper per output), and bit for bit against the
    one-output form of the same tiles (SIGOPS_RS_NOPAIR): every output adds its own taps in the same order"""
    rng = np.random.default_rng(206)
    x = np.asfortranarray(rng.standard_normal((n, nch)).astype(dt))
    tree = so.Signal(x, fs_in * so.Hz) | so.ToFramerate(fs_out * so.Hz) | so.After(37 * so.frames)
    got = so.sink(tree)[0]
    want = oracle_sink(tree)
    assert got.shape == want.shape and got.dtype == want.dtype
    assert relerr(got, want) < (1e-9 if dt == np.float64 else 2e-7)
    assert np.abs(got.astype(np.float64) - want).max() < (1e-8 if dt == np.float64 else 1e-6) * np.abs(want).max()
    # (long Float64 signals take the persistent form, k_resample_arb, which forms the interpolated tap once per output:
    #  another association of the same sums, test below; the tiles themselves are compared with it switched off)
    monkeypatch.setenv("SIGOPS_RS_NOARB", "1")
    got = so.sink(tree)[0]
    assert relerr(got, want) < (1e-9 if dt == np.float64 else 2e-7)
    monkeypatch.setenv("SIGOPS_RS_NOPAIR", "1")
    one = so.sink(tree)[0]
    assert np.array_equal(got, one)


def test_persistent_form_takes_float32_signals_too(monkeypatch):
    """Float32 samples: a ring of floats (256 frames per LDS-DMA instruction), widened where the compute waves read them,
    Float32 stores -- against the oracle at the Float32 bar and against the tiled kernel (the same Float64 sums up to
    their last bits: a Float32 rounding flip here and there)"""
    monkeypatch.setenv("SIGOPS_ARB_MIN", "1")
    rng = np.random.default_rng(208)
    for nch, n, fs_in, fs_out in ((8, 400001, 44100.0, 44100.0 * np.pi / 3), (3, 250000, 44100.5, 48000.0), (2, 300002, 1000.0, 1000.0 * np.pi)):
        x = np.asfortranarray(rng.standard_normal((n, nch)).astype(np.float32))
        tree = so.Signal(x, fs_in * so.Hz) | so.ToFramerate(fs_out * so.Hz)
        nout = so.nframes(tree)
        p = so.Plan(so.ToChannels(tree, nch), (nout, nch), np.float32, (1, nout), False)
        names = [s["name"] for s in p.steps()]
        p.close()
        assert names == ["k_resample_arb"], names
        got = so.sink(tree)[0]
        assert got.dtype == np.float32 and relerr(got, oracle_sink(tree)) < 1e-6
        part = so.sink(tree | so.After(nout // 3 * so.frames) | so.Until(30000 * so.frames))[0]
        assert np.array_equal(part, got[nout // 3:nout // 3 + 30000])
        monkeypatch.setenv("SIGOPS_RS_NOARB", "1")
        ref = so.sink(tree)[0]
        monkeypatch.delenv("SIGOPS_RS_NOARB")
        assert relerr(got, ref) < 2e-8 and np.mean(got == ref) > 0.999


def test_two_outputs_per_lane_form_is_the_one_that_runs():
    import torch
    x = torch.randn((8, 100000), dtype=torch.float64, device="cuda")
    tree = so.Signal(x.t(), 1000 * so.Hz) | so.ToFramerate(1000 * np.pi * so.Hz)
    n = so.nframes(tree)
    out = torch.empty((8, n), dtype=torch.float64, device="cuda")
    for env, name in (("1", "k_resample_tiled2"), (None, "k_resample_arb")):
        if env:
            os.environ["SIGOPS_RS_NOARB"] = env
        else:
            os.environ.pop("SIGOPS_RS_NOARB", None)
        p = so.Plan(so.ToChannels(tree, 8), (n, 8), np.float64, (1, n), True)
        p.set_profiling(True)
        p.execute(out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert [s["name"] for s in p.steps()] == [name]
        p.close()


@pytest.mark.parametrize("fs_in,fs_out,nch,n,arb", [
    (1000.0, 1000.0 * np.pi, 8, 300000, True),          # benchmarks.jl "resampling-irrational"
    (44100.0, 44100.0 * np.pi / 3, 8, 700001, True),    # tools/bench_irrational.py's rate; several ranges, odd length
    (1000.0, 1000.0 * np.pi / 3, 16, 150000, True),     # two channel groups
    (44100.5, 48000.0, 3, 200000, True),                # non-integer frame rate, one channel per workgroup
    (44100.5, 48000.0, 4, 123457, True),
    (48000.0, 44100.5, 2, 260000, True),                # down
    (1000.0, 1000.0 / np.e, 8, 400000, None),           # x 0.37: a long filter (whichever kernel its tables leave room for)
    (48000.0, 9000.5, 4, 500000, None),                 # x 0.19: windows of a pair five or six frames apart
    (999.0, 16000.0, 8, 40000, True)])                  # x 16: many outputs per input
def test_persistent_form_of_the_arbitrary_rate_resampler(fs_in, fs_out, nch, n, arb, monkeypatch):
    """k_resample_arb (k_resample_arb.hip): loader wave + LDS ring + compute waves that form DSP.jl's interpolated tap
    h[p + 32 k] + alpha dh[p + 32 k] once per output -- against the oracle (yLower + alpha yUpper per output: 1e-9 as for
    every resampler, the accumulated-alpha drift) and against the tiled kernel (the two associations of the same sum:
    1e-14), windows included (warm start: g.m0 / g.j0)"""
    monkeypatch.setenv("SIGOPS_ARB_MIN", "1")   # (by default from 1.5 M output samples on: below, the tiled kernel's shorter start wins)
    rng = np.random.default_rng(207)
    x = np.asfortranarray(rng.standard_normal((n, nch)))
    x[n // 3, 0] = 0.0
    tree = so.Signal(x, fs_in * so.Hz) | so.ToFramerate(fs_out * so.Hz)
    nout = so.nframes(tree)
    p = so.Plan(so.ToChannels(tree, nch), (nout, nch), np.float64, (1, nout), False)
    names = [s["name"] for s in p.steps()]
    p.close()
    assert names == ["k_resample_arb"] or (arb is None and names == ["k_resample_tiled2"]), names
    got = so.sink(tree)[0]
    assert relerr(got, oracle_sink(tree)) < 1e-9
    part = so.sink(tree | so.After(nout // 2 * so.frames) | so.Until(20000 * so.frames))[0]
    assert np.array_equal(part, got[nout // 2:nout // 2 + 20000])
    monkeypatch.setenv("SIGOPS_RS_NOARB", "1")
    ref = so.sink(tree)[0]
    assert relerr(got, ref) < 1e-14
    assert np.abs(got - ref).max() < 1e-13 * np.abs(ref).max()
    monkeypatch.delenv("SIGOPS_RS_NOARB")
    monkeypatch.setenv("SIGOPS_ARB_NO", "4")   # four outputs per lane (opt-in; eight-channel groups at rates near 1)
    four = so.sink(tree)[0]
    assert relerr(four, ref) < 1e-14


@pytest.mark.parametrize("nch", [8, 4])
@pytest.mark.parametrize("kind", ["44.1 -> 16 kHz", "fir 101 taps", "fir + gain"])
def test_float32_long_windows_run_the_periodic_kernel(kind, nch):
    """16-row tiles of K3 (windows of 28 / 36 k-steps) for Float32 signals too: a 101-tap `Filt(x, h)` or 44.1 -> 16 kHz on
    Float32 data used to go to the row-tiled kernel (1.04 ms where Float64 took 0.66: `profiles/r04/operator_matrix_f32.txt`).
    Same arithmetic as the Float64 instantiation (Float64 MFMAs over the widened tile), a Float32 store."""
    rng = np.random.default_rng(900 + nch)
    n = 400_000
    x32 = np.asfortranarray((rng.standard_normal((n, nch)) * 0.5).astype(np.float32))
    sig = so.Signal(x32, 44.1 * so.kHz)
    tree = {"44.1 -> 16 kHz": lambda: sig | so.ToFramerate(16 * so.kHz),
            "fir 101 taps": lambda: so.Filt(sig, np.hanning(101) / 50.0),
            "fir + gain": lambda: so.Filt(sig | so.Amplify(so.Signal(so.sin, ω=3 * so.Hz)) | so.Until(n * so.frames) | so.ToEltype(np.float32), np.hanning(120) / 60.0)}[kind]()
    nout = so.nframes(tree)
    p = so.Plan(so.ToChannels(tree, nch), (nout, nch), np.float32, (1, nout), False)
    names = [s["name"] for s in p.steps()]
    p.close()
    assert "k_resample_periodic" in names and "k_resample_rows" not in names, names
    got = so.sink(tree)[0]
    want = oracle_sink(tree)
    assert got.dtype == want.dtype == np.float32 and got.shape == want.shape
    assert relerr(got, want) < 1e-6
    os.environ["SIGOPS_RS_NOQ1"] = "1"  # the row-tiled kernel on the same data
    try:
        rows = so.sink(tree)[0]
    finally:
        del os.environ["SIGOPS_RS_NOQ1"]
    assert relerr(got, rows) < 1e-6
