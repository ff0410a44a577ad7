#!/usr/bin/env python
"""A map nest too deep for the interpreter kernel (three arrays, six generators): hipRTC-specialised single
launch against the interpreter with materialised sub-expressions (SIGOPS_RTC=0).
    python tools/bench_rtc.py [frames] [channels]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import numpy as np
    import torch

    import sigops_amd as so

    n, nch = int(sys.argv[2]), int(sys.argv[3])
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    arrs = [torch.randn((nch, n), dtype=torch.float64, device=dev, generator=gen).t() for _ in range(3)]
    a, b, c = (so.Signal(x, 44.1 * so.kHz) for x in arrs)
    g = [so.Signal(so.sin, ω=(100.0 + 37 * k) * so.Hz, ϕ=0.01 * k) for k in range(6)]
    tree = (so.Mix(so.Amplify(a, g[0]), so.Amplify(b, so.Mix(g[1], g[2])), so.Amplify(c, so.Amplify(g[3], so.Mix(g[4], g[5]))))
            | so.Until(n * so.frames) | so.Ramp(10 * so.ms))
    out = torch.empty((nch, n), dtype=torch.float64, device=dev).t()
    t0 = time.perf_counter()
    plan = so.Plan(so.ToChannels(tree, nch), (n, nch), np.float64, (out.stride(0), out.stride(1)), True, device=0)
    create_ms = (time.perf_counter() - t0) * 1e3
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(10):
        plan.execute(out.data_ptr(), st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        plan.execute(out.data_ptr(), st)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / 50
    s = plan.stats()
    names = [x["name"] for x in plan.steps()]
    algo = 4 * n * nch * 8
    print(json.dumps({"rtc": os.environ.get("SIGOPS_RTC", "auto"), "frames": n, "channels": nch, "ms_per_execute": round(ms, 4),
                      "plan_create_ms": round(create_ms, 1), "launches": s["n_launches"], "steps": names,
                      "algorithmic_GBps": round(algo / ms / 1e6, 1), "checksum": float(out[:: max(1, n // 1000)].abs().sum().item())}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "run":
        run()
    else:
        n = sys.argv[1] if len(sys.argv) > 1 else "13000000"
        nch = sys.argv[2] if len(sys.argv) > 2 else "2"
        for mode in ("0", None):
            env = dict(os.environ)
            if mode is None:
                env.pop("SIGOPS_RTC", None)
            else:
                env["SIGOPS_RTC"] = mode
            subprocess.run([sys.executable, os.path.abspath(__file__), "run", n, nch], env=env, check=False)
