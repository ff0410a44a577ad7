#!/usr/bin/env python
"""Secondary measurements for the other BASELINE.json configs (not the driver's bench contract):
frames/s of one `sink` per config on one GPU with device-resident leaves and result.
    python bench_configs.py [--scale 1.0]
Prints one JSON line per config."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def timed(plan, optr, stream, torch, steps):
    for _ in range(2):
        plan.execute(optr, stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        plan.execute(optr, stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def run(name, tree, so, torch, steps=10, dtype="f64"):
    import numpy as np
    n, nch = so.nframes(tree), tree.nch
    tdt = torch.float64 if dtype == "f64" else torch.float32
    out_t = torch.empty((nch, n), dtype=tdt, device="cuda")
    out = out_t.t()
    t0 = time.perf_counter()
    plan = so.Plan(tree, (n, nch), np.float64 if dtype == "f64" else np.float32, (out.stride(0), out.stride(1)), True)
    torch.cuda.synchronize()
    plan_ms = (time.perf_counter() - t0) * 1e3
    dt = timed(plan, out.data_ptr(), torch.cuda.current_stream().cuda_stream, torch, steps)
    plan.set_profiling(True)
    plan.execute(out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    st = plan.stats()
    res = {"config": name, "frames": n, "channels": nch, "dtype": dtype, "ms": dt * 1e3, "frames_per_s": n / dt,
           "algorithmic_GBps": st["algorithmic_bytes"] / dt / 1e9, "launches": st["n_launches"],
           "dominant_kernel": st["dominant_kernel"], "dominant_kernel_ms": st["dominant_kernel_ms"],
           "plan_create_ms": plan_ms}
    print(json.dumps(res), flush=True)
    plan.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    import torch
    import __graft_entry__ as ge
    ge.build()
    import sigops_amd as so
    from sigops_amd import Signal, Until, Ramp, Normpower, Amplify, ToFramerate, Mix, Filt, Bandstop, Lowpass, Append
    from sigops_amd import s, kHz, Hz, dB, ms, frames, sin
    g = torch.Generator(device="cuda")
    g.manual_seed(1983)

    def dev(nf, nch, dt=torch.float64, uniform=False):
        t = (torch.rand if uniform else torch.randn)((nch, nf), dtype=dt, device="cuda", generator=g)
        return t.t()

    sel = set(a.only.split(",")) if a.only else None
    if not sel or "1" in sel:  # config 1: README sound1
        run("config1 sound1 (sin|Until 5s|Ramp|Normpower|Amplify -20dB) @44.1kHz",
            Signal(sin, ω=1 * kHz) | Until(5 * s) | Ramp | Normpower | Amplify(-20 * dB) | ToFramerate(44.1 * kHz), so, torch)
    if not sel or "2" in sel:  # config 2: Mix(sin, noise) |> Filt(Bandstop), 2ch, 60 s
        n2 = int(2646000 * a.scale)
        noise = dev(n2, 2)
        run("config2 Mix(sin 1kHz, noise[2ch,60s]) |> Filt(Bandstop 0.5-2kHz)",
            Mix(Signal(sin, ω=1 * kHz) | Until(n2 * frames), Signal(noise, 44.1 * kHz)) | Filt(Bandstop, 0.5 * kHz, 2 * kHz), so, torch)
    if not sel or "ns" in sel:  # north_star's pipeline at config 3's size: Mix -> Filt -> ToFramerate, 8 ch, 600 s
        nn = int(26460000 * a.scale)
        noise = dev(nn, 8)
        run("north-star pipeline: Mix(sin 1kHz, noise[8ch,600s]) |> Filt(Bandstop 0.5-2kHz) |> ToFramerate(48kHz)",
            Mix(Signal(sin, ω=1 * kHz), Signal(noise, 44.1 * kHz)) | Until(nn * frames)
            | Filt(Bandstop, 0.5 * kHz, 2 * kHz) | ToFramerate(48 * kHz), so, torch, steps=10)
        del noise
        torch.cuda.empty_cache()
    if not sel or "4" in sel:  # config 4 (one GPU's share: 8 scenes of 60 s)
        n4 = int(2646000 * a.scale)
        scenes = []
        for k in range(8):
            noise = dev(n4, 2)
            scenes.append(Mix(Signal(sin, ω=(500 + 25 * k) * Hz) | Until(n4 * frames), Signal(noise, 44.1 * kHz))
                          | Filt(Bandstop, 0.5 * kHz, 2 * kHz) | Ramp(10 * ms))
        run("config4 share: Append(8 x [Mix+Filt(Bandstop)+Ramp], 60s, 2ch)", Append(*scenes), so, torch, steps=5)
    if sel and "k1" in sel:  # K1 alone: pointwise map over a large array (not a BASELINE config)
        nk = int(26460000 * a.scale)
        xk = dev(nk, 8)
        run("K1 probe: Amplify(x[%d x 8], sin 5Hz) |> Ramp(1s)" % nk,
            Amplify(Signal(xk, 44.1 * kHz), Signal(sin, ω=5 * Hz)) | Until(nk * frames) | Ramp(1 * s), so, torch, steps=5)
        run("K1 probe: Amplify(x[%d x 8], 0.5)" % nk, Amplify(Signal(xk, 44.1 * kHz), 0.5), so, torch, steps=5)
    if not sel or "5" in sel:  # config 5 (one GPU's slab, time scaled down): 128 ch
        n5 = int(1000000 * a.scale)
        x = dev(n5, 128, uniform=True)
        run("config5 slab: x[%d x 128] |> Filt(Lowpass 4kHz) |> ToFramerate(16kHz)" % n5,
            Signal(x, 44.1 * kHz) | Filt(Lowpass, 4 * kHz) | ToFramerate(16 * kHz), so, torch, steps=3)


if __name__ == "__main__":
    main()
