"""hipRTC-specialised pointwise kernels on the GPU (rtc.cpp): forced for every
pointwise step (SIGOPS_RTC=1) the shared case trees must give the interpreter kernel's values BIT FOR BIT
(same leaf evaluators, -ffp-contract=off) and match the oracle; in the default mode a map nest too deep
for the interpreter's 4-deep stack / 4 per-frame values runs as ONE specialised launch instead of a chain
of materialising launches (reference shape: one loop per map nest, src/mapsignal.jl:249-272)."""
import os

import numpy as np
import pytest

import sigops_amd as so
from sigops_amd.engine import Plan
from cases import CASES
from oracle_bridge import oracle_sink, relerr

pytestmark = pytest.mark.gpu


def _names(tree, res):
    p = Plan(so.ToChannels(tree, res.shape[1]), res.shape, res.dtype, (1, res.shape[0]), False)
    p.set_profiling(True)
    p.execute(res.ctypes.data)
    names = [s["name"] for s in p.steps()]
    n_launches = p.stats()["n_launches"]
    p.close()
    return names, n_launches


@pytest.mark.parametrize("name", sorted(CASES))
def test_cases_with_every_pointwise_step_specialised(name, monkeypatch):
    x = CASES[name]()
    try:
        want = oracle_sink(x)
    except so.ErrorException:
        pytest.skip("an error case")
    monkeypatch.setenv("SIGOPS_RTC", "0")
    ref = so.sink(x, so.Array)
    monkeypatch.setenv("SIGOPS_RTC", "1")
    got = so.sink(x, so.Array)
    assert got.shape == ref.shape == want.shape and got.dtype == ref.dtype
    assert np.array_equal(got, ref, equal_nan=True), f"{name}: specialised kernel differs from the interpreter ({relerr(got, ref):.3e})"


def _deep_tree(n, nch, rng):
    """six generators and three arrays in one nest: beyond 4 per-frame values and a 4-deep stack"""
    a, b, c = (so.Signal(np.asfortranarray(rng.standard_normal((n, nch))), 44.1 * so.kHz) for _ in range(3))
    g = [so.Signal(so.sin, ω=(100.0 + 37 * k) * so.Hz, ϕ=0.01 * k) for k in range(6)]
    t = so.Mix(so.Amplify(a, g[0]), so.Amplify(b, so.Mix(g[1], g[2])), so.Amplify(c, so.Amplify(g[3], so.Mix(g[4], g[5]))))
    return t | so.Until(n * so.frames) | so.Ramp(10 * so.ms)


def test_a_nest_too_deep_for_the_interpreter_is_one_specialised_launch(monkeypatch):
    rng = np.random.default_rng(31)
    n, nch = 600_000, 2
    tree = _deep_tree(n, nch, rng)
    want = oracle_sink(tree)
    res = np.empty((n, nch), order="F")
    monkeypatch.setenv("SIGOPS_RTC", "0")
    names0, launches0 = _names(tree, res)
    ref = res.copy()
    monkeypatch.delenv("SIGOPS_RTC")
    names, launches = _names(tree, res)  # default mode: big enough and over the interpreter's limits
    assert "k_pointwise_rtc" in names and "k_pointwise_rtc" not in names0, (names, names0)
    assert launches < launches0, (launches, launches0)  # no materialising launches
    assert relerr(res, want) < 1e-12 and relerr(ref, want) < 1e-12
    # a second plan of the same shape (other data, other constants) reuses the compiled kernel
    tree2 = _deep_tree(n, nch, np.random.default_rng(32))
    assert relerr(so.sink(tree2, so.Array), oracle_sink(tree2)) < 1e-12


def test_small_or_simple_steps_keep_the_interpreter():
    x = np.asfortranarray(np.random.default_rng(33).standard_normal((50_000, 2)))
    tree = so.Signal(x, 44.1 * so.kHz) | so.Amplify(so.Signal(so.sin, ω=5 * so.Hz)) | so.Until(50_000 * so.frames) | so.Ramp(10 * so.ms)
    res = np.empty((50_000, 2), order="F")
    names, _ = _names(tree, res)
    assert names == ["k_pointwise"]


_ASYNC_SNIPPET = r"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
import sigops_amd as so
from sigops_amd import _capi as K
torch.manual_seed(5)
a = torch.randn((2, 3_000_000), dtype=torch.float64, device="cuda")
b = torch.randn((2, 3_000_000), dtype=torch.float64, device="cuda")
tree = so.Amplify(so.Mix(so.Signal(a.t(), 44.1 * so.kHz), so.Signal(b.t(), 44.1 * so.kHz)), so.Signal(b.t(), 44.1 * so.kHz))
n = so.nframes(tree)
def run():
    out = torch.empty((2, n), dtype=torch.float64, device="cuda")
    p = so.Plan(so.ToChannels(tree, 2), (n, 2), np.float64, (1, n), True)
    p.set_profiling(True)
    p.execute(out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    names = [s["name"] for s in p.steps()]
    p.close()
    return names, out
first, o1 = run()
K.lib().so_rtc_wait_idle()
second, o2 = run()
want = (a + b) * b
print("RESULT", first, second, bool(torch.equal(o1, o2)), bool(torch.equal(o1, want)))
"""


def test_big_steps_are_specialised_in_the_background_and_kept_on_disk(tmp_path):
    """a pointwise step the interpreter can run is never compiled at the caller's expense: the first plan of a new
    shape runs the interpreter and queues the compile, plans after it use the specialised kernel (same values), and
    the next PROCESS finds the code object in the cache directory"""
    import subprocess, sys
    env = dict(os.environ, SIGOPS_CACHE_DIR=str(tmp_path / "cache"))
    env.pop("SIGOPS_RTC", None)
    out1 = subprocess.run([sys.executable, "-c", _ASYNC_SNIPPET], env=env, capture_output=True, text=True, timeout=300,
                          cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    line1 = [l for l in out1.stdout.splitlines() if l.startswith("RESULT")]
    assert line1, out1.stdout + out1.stderr
    assert line1[0] == "RESULT ['k_pointwise'] ['k_pointwise_rtc'] True True", line1[0]
    files = list((tmp_path / "cache").glob("*.gfx950.co"))
    assert len(files) == 1 and files[0].stat().st_size > 1000
    out2 = subprocess.run([sys.executable, "-c", _ASYNC_SNIPPET], env=env, capture_output=True, text=True, timeout=300,
                          cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    line2 = [l for l in out2.stdout.splitlines() if l.startswith("RESULT")]
    assert line2 and line2[0] == "RESULT ['k_pointwise_rtc'] ['k_pointwise_rtc'] True True", out2.stdout + out2.stderr
    # switched off: the interpreter both times, nothing written
    env3 = dict(env, SIGOPS_RTC_NOASYNC="1", SIGOPS_CACHE_DIR=str(tmp_path / "cache3"))
    out3 = subprocess.run([sys.executable, "-c", _ASYNC_SNIPPET], env=env3, capture_output=True, text=True, timeout=300,
                          cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    line3 = [l for l in out3.stdout.splitlines() if l.startswith("RESULT")]
    assert line3 and line3[0] == "RESULT ['k_pointwise'] ['k_pointwise'] True True", out3.stdout + out3.stderr
    assert not (tmp_path / "cache3").exists()
