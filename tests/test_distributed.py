"""N>1 path on CPU: world_size-2 `gloo` processes exercise the sharding logic
(signaloperators.jl_amd/sharding.py) end to end; the oracle stands in for the HIP engine
as the per-shard compute so the test needs no GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

import sigops_amd as so
from sigops_amd import sharding
from cases import F, rng

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def scenes(n=5, frames=3000):
    out = []
    for k in range(n):
        noise = F(rng(100 + k).standard_normal((frames, 2)))
        tone = so.Signal(so.sin, ω=(500 + 25 * k) * so.Hz) | so.Until(frames * so.frames)
        out.append(so.Mix(tone, so.Signal(noise, 44.1 * so.kHz)) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz)
                   | so.Ramp(10 * so.ms))
    return out


def test_block_partition():
    for n in (1, 5, 8, 64):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                lo, hi = sharding.block_range(n, r, world)
                cover += list(range(lo, hi))
            assert cover == list(range(n))


def test_shard_append_offsets():
    x = so.Append(*scenes())
    total = 0
    for r in range(2):
        sub, start, count = sharding.shard_append(x, r, 2)
        assert start == total and so.nframes(sub) == count
        total += count
    assert total == so.nframes(x)


def test_shard_channels():
    x = so.Signal(F(rng(1).random((50, 6))), 10 * so.Hz) | so.Amplify(2.0)
    got = []
    for r in range(4):
        sub, c0, c1 = sharding.shard_channels(x, r, 4)
        got += list(range(c0, c1))
        if sub is not None:
            assert sub.nch == c1 - c0
    assert got == list(range(6))


WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch.distributed as dist
import sigops_amd as so
from sigops_amd import sharding
from oracle_bridge import oracle_sink
from test_distributed import scenes
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=int(sys.argv[3]), world_size=2)
x = so.Append(*scenes())
full = sharding.sink_append_sharded(x, compute=oracle_sink)
want = oracle_sink(x)
ok = np.array_equal(full, want)
local, start = sharding.sink_append_sharded(x, compute=oracle_sink, gather=False)
ok = ok and np.array_equal(local, want[start:start + local.shape[0]])
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if ok else 3)
'''


WORKER_CH = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch.distributed as dist
import sigops_amd as so
from sigops_amd import sharding
from oracle_bridge import oracle_sink
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=int(sys.argv[3]), world_size=2)
x0 = np.asfortranarray(np.random.default_rng(3).random((6000, 5)))
x = so.Signal(x0, 44.1 * so.kHz) | so.Filt(so.Lowpass, 4 * so.kHz) | so.ToFramerate(16 * so.kHz)
full = sharding.sink_channels_sharded(x, compute=oracle_sink)   # uneven slabs: 3 + 2 channels
want = oracle_sink(x)
ok = np.array_equal(full, want)
local, c0, c1 = sharding.sink_channels_sharded(x, compute=oracle_sink, gather=False)
ok = ok and np.array_equal(local, want[:, c0:c1])
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if ok else 3)
'''


def test_channel_sharding_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker_ch.py"
    script.write_text(WORKER_CH)
    port = str(31500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r)]) for r in range(2)]
    codes = [p.wait(timeout=300) for p in procs]
    assert codes == [0, 0]


def test_append_sharding_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = str(29500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r)]) for r in range(2)]
    codes = [p.wait(timeout=300) for p in procs]
    assert codes == [0, 0]


def test_shard_time_ranges():
    x = so.Signal(np.arange(1003.0), 1 * so.kHz) | so.Filt(so.Lowpass, 100 * so.Hz)
    got = []
    for r in range(3):
        sub, start, count = sharding.shard_time(x, r, 3, align=100)
        assert start % 100 == 0 and (sub is None) == (count == 0)
        if sub is not None:
            assert so.nframes(sub) == count
        got.append((start, count))
    assert got == [(0, 400), (400, 400), (800, 203)]
    assert sharding.shard_time(x, 5, 8, align=500)[2] == 0  # more ranks than ranges


WORKER_T = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch.distributed as dist
import sigops_amd as so
from sigops_amd import sharding
from oracle_bridge import oracle_sink
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=int(sys.argv[3]), world_size=2)
x0 = np.asfortranarray(np.random.default_rng(4).standard_normal((30001, 2)))
x = so.Mix(so.Signal(so.sin, ω=1 * so.kHz), so.Signal(x0, 44.1 * so.kHz)) | so.Until(30001 * so.frames) \
    | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.ToFramerate(48 * so.kHz)
# the oracle evaluates each rank's `After(a) |> Until(n)` by running the skipped frames through the filters:
# one signal cut along time, identical to the unsharded sink
full = sharding.sink_time_sharded(x, compute=oracle_sink)
want = oracle_sink(x)
ok = full.shape == want.shape and float(np.abs(full - want).max()) <= 1e-12
local, start = sharding.sink_time_sharded(x, compute=oracle_sink, gather=False, align=160)
ok = ok and start % 160 == 0 and float(np.abs(local - want[start:start + local.shape[0]]).max()) <= 1e-12
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if ok else 3)
'''


def test_time_sharding_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker_t.py"
    script.write_text(WORKER_T)
    port = str(33500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r)]) for r in range(2)]
    codes = [p.wait(timeout=300) for p in procs]
    assert codes == [0, 0]


def test_shard_mix_operands():
    rng = np.random.default_rng(6)
    a = so.Signal(np.asfortranarray(rng.standard_normal((5000, 2))), 44.1 * so.kHz)
    b = so.Signal(np.asfortranarray(rng.standard_normal((3000, 2))), 44.1 * so.kHz) | so.Filt(so.Lowpass, 3 * so.kHz)
    c = so.Signal(so.sin, ω=1 * so.kHz) | so.Until(4000 * so.frames)
    x = so.Mix(a, b, c)
    subs = [sharding.shard_mix(x, r, 4) for r in range(4)]
    assert subs[3] is None and all(s is not None and so.nframes(s) == 5000 and so.nchannels(s) == 2 for s in subs[:3])
    with pytest.raises(so.ErrorException, match="Mix"):
        sharding.shard_mix(so.Amplify(a, b), 0, 2)


WORKER_MIX = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch.distributed as dist
import sigops_amd as so
from sigops_amd import sharding
from oracle_bridge import oracle_sink, oracle_semantics
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=int(sys.argv[3]), world_size=2)
rng = np.random.default_rng(5)
ops = []
for k in range(3):  # three heavy operands of different lengths: filtered, resampled noise
    n = 20000 + 3000 * k
    ops.append(so.Signal(np.asfortranarray(rng.standard_normal((n, 2))), 44.1 * so.kHz) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz)
               | so.ToFramerate(48 * so.kHz))
x = so.Mix(*ops)
# (the documented meaning of a Mix of filtered operands of different lengths -- an operand ends after nframes(x) frames,
#  what the engine implements; the reference's quirk C-7 keeps pulling a long filtered child's tail inside the Mix,
#  which no partition of the operands can reproduce: tests/test_oracle_dsp.py)
with oracle_semantics("intended"):
    want = oracle_sink(x)
    full = sharding.sink_mix_sharded(x, compute=oracle_sink)  # operands [0, 1] on rank 0, [2] on rank 1; all_reduce over gloo
    only0 = sharding.sink_mix_sharded(x, compute=oracle_sink, root=0)
ok = full.shape == want.shape and float(np.abs(full - want).max()) <= 1e-12 * float(np.abs(want).max())
ok = ok and ((only0 is None) if dist.get_rank() == 1 else np.array_equal(only0, full))
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if ok else 3)
'''


def test_mix_sharding_world_size_2_gloo(tmp_path):
    """the operands of a root Mix on different ranks, one reduction (reference src/mapsignal.jl:307-308: Mix = OperateOn(+))"""
    script = tmp_path / "worker_mix.py"
    script.write_text(WORKER_MIX)
    port = str(35500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r)]) for r in range(2)]
    codes = [p.wait(timeout=300) for p in procs]
    assert codes == [0, 0]
