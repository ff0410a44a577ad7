"""What the engine keeps between processes under SIGOPS_CACHE_DIR (accumulator.cpp): the replay of DSP.jl's phase
accumulator per (rate pair, length) and the cascade's two rounding-sensitivity probes -- host-side analyses that depend on
nothing but their key and are most of a first sink's plan creation (bench.py `one_shot`).  A cached value must give the
result the computation gives, bit for bit; a file that does not check out (truncated, another key under the same name) is
ignored and rewritten.  Each sink runs in a fresh child process: the in-process caches would hide the files."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import sigops_amd as so
rng = np.random.default_rng(7)
x = so.Signal(rng.standard_normal((30000, 2)), 44.1 * so.kHz)
y = so.Mix(so.Signal(so.sin, ω=1 * so.kHz), x) | so.Until(30000 * so.frames) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.ToFramerate(48 * so.kHz)
np.save(sys.argv[1], so.sink(y)[0])
"""


def _run(tmp_path, tag, env_extra):
    out = str(tmp_path / (tag + ".npy"))
    env = dict(os.environ)
    env.pop("SIGOPS_CACHE_DIR", None)
    env.update(env_extra)
    subprocess.run([sys.executable, "-c", CHILD % ROOT, out], env=env, check=True, timeout=300)
    return np.load(out)


def test_cached_analyses_give_the_computed_result_and_bad_files_are_ignored(tmp_path):
    cache = tmp_path / "cache"
    cache.mkdir()
    plain = _run(tmp_path, "plain", {})
    first = _run(tmp_path, "first", {"SIGOPS_CACHE_DIR": str(cache)})
    files = sorted(os.path.basename(p) for p in glob.glob(str(cache / "sigops_*.bin")))
    kinds = {f.split("_")[1] for f in files}
    assert {"acc", "sens", "chunk"} <= kinds, files
    assert not glob.glob(str(cache / "*.tmp")), "temporary names left behind"
    second = _run(tmp_path, "second", {"SIGOPS_CACHE_DIR": str(cache)})
    assert np.array_equal(plain, first) and np.array_equal(first, second)
    # damaged files: one truncated, one with a flipped byte in its key, one empty
    sizes = {}
    for i, f in enumerate(files):
        p = cache / f
        b = p.read_bytes()
        sizes[f] = len(b)
        if i % 3 == 0:
            p.write_bytes(b[: len(b) // 2])
        elif i % 3 == 1:
            p.write_bytes(b[:20] + bytes([b[20] ^ 0x5A]) + b[21:])
        else:
            p.write_bytes(b"")
    third = _run(tmp_path, "third", {"SIGOPS_CACHE_DIR": str(cache)})
    assert np.array_equal(plain, third)
    for f in files:  # ... and rewritten
        assert (cache / f).stat().st_size == sizes[f], f
    # SIGOPS_REPLAY_NOCACHE: nothing read, nothing written
    empty = tmp_path / "empty"
    empty.mkdir()
    fourth = _run(tmp_path, "fourth", {"SIGOPS_CACHE_DIR": str(empty), "SIGOPS_REPLAY_NOCACHE": "1"})
    assert np.array_equal(plain, fourth) and not os.listdir(empty)
