"""`python bench.py --gpus N` launches its N ranks itself (the flag used to be parsed and ignored, ranks came
only from RANK / WORLD_SIZE, i.e. from a launcher the driver might not use)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=timeout)
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    assert len(lines) == 1, "exactly one JSON line (rank 0's)"
    return json.loads(lines[0])


def test_gpus_2_without_a_device_runs_the_launch_path():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a HIP device is present: covered by the gpu test below")
    res = run_bench("--gpus", "2", "--steps", "3", "--warmup", "1")
    assert res["n_gpus"] == 2 and res["config"]["ranks_seen"] == 2
    assert res["valid"] is False and res["value"] == 0.0 and "NO HIP DEVICE" in res["config"]["workload"]


@pytest.mark.gpu
def test_gpus_2_on_a_one_gpu_box_is_labelled_oversubscribed():
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("more than one GPU: a real two-rank run, not the oversubscribed path")
    res = run_bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--seconds", "20", "--cpu-seconds", "0", "--no-secondary")
    assert res["n_gpus"] == 2 and res["value"] > 0
    assert "OVERSUBSCRIBED" in res["config"]["parallelism"]


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["config4", "config5"])
def test_gpus_2_on_a_one_gpu_box_runs_the_sharded_workloads_too(workload):
    """the N > 1 branches of bench_multi.py (shares, agreement over the ranks, the gather or the channel slabs) on one GPU,
    two ranks sharing it over gloo: not a measurement -- the first run of these branches must not be the driver's 8-GPU run"""
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("more than one GPU: a real two-rank run")
    res = run_bench("--gpus", "2", "--workload", workload, "--steps", "2", "--warmup", "1", "--seconds", "20")
    assert res["n_gpus"] == 2 and res["value"] > 0
    if workload == "config4":
        assert res["scaling"] == "strong" and res["config"]["scenes_per_rank"] == 32 and res["parity_gate"]["relerr"] < 1e-8
    else:
        assert res["scaling"] == "weak" and res["config"]["channels"] == 256 and res["config"]["finite"]
