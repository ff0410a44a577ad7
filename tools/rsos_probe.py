"""Fused resampler -> IIR kernel (k_rsos) against the two-kernel path (K3 + K2) on the headline pipeline.

    python3 tools/rsos_probe.py [--seconds 60] [--channels 8] [--reps 20] [--oracle 150000]

Both plans sink the same device-resident noise; the result of the fused plan is compared with the two-kernel
plan's (the same engine, SIGOPS_NO_RSOS=1 at plan creation) over the whole length, and with the CPU oracle on a
prefix.  Times are means of `reps` executes between two events on the launch stream.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--channels", type=int, default=8)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--oracle", type=int, default=150000)
    ap.add_argument("--plain", action="store_true", help="no Mix: the noise alone through the resampler and the filter")
    ap.add_argument("--only-fused", action="store_true")
    ap.add_argument("--f32", action="store_true", help="a Float32 leaf and a Float32 result")
    ap.add_argument("--pad-out", type=int, default=0, help="extra elements between the result's channels (pitch = frames + pad)")
    ap.add_argument("--pad-in", type=int, default=0, help="extra elements between the input's channels")
    ap.add_argument("--warm", type=int, default=3, help="untimed executes before the timed ones (the chip's power management settles after ~30)")
    args = ap.parse_args()
    import numpy as np
    import torch

    import sigops_amd as so

    dev = torch.device("cuda:0")
    nch = args.channels
    n_in = int(round(args.seconds * 44100))
    gen = torch.Generator(device=dev)
    gen.manual_seed(1983)
    tdt = torch.float32 if args.f32 else torch.float64
    ndt = np.float32 if args.f32 else np.float64
    noise_t = torch.randn((nch, n_in + args.pad_in), dtype=tdt, device=dev, generator=gen)[:, :n_in]
    noise = noise_t.t()
    stream = torch.cuda.current_stream().cuda_stream

    def tree(arr, n):
        src = so.Signal(arr, 44.1 * so.kHz)
        if not args.plain:
            src = so.Mix(so.Signal(so.sin, ω=1 * so.kHz), src)
        return src | so.Until(n * so.frames) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.ToFramerate(48 * so.kHz)

    def run(fused):
        if fused:
            os.environ.pop("SIGOPS_NO_RSOS", None)
        else:
            os.environ["SIGOPS_NO_RSOS"] = "1"
        x = tree(noise, n_in)
        n_out = so.nframes(x)
        out_t = torch.empty((nch, n_out + args.pad_out), dtype=tdt, device=dev)[:, :n_out]
        out = out_t.t()
        plan = so.Plan(so.ToChannels(x, nch), (n_out, nch), ndt, (out.stride(0), out.stride(1)), True, device=0)
        for _ in range(args.warm):
            plan.execute(out.data_ptr(), stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            plan.execute(out.data_ptr(), stream)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        names = [s["name"] for s in plan.steps()]
        st = plan.stats()
        plan.close()
        return out_t, ms, names, st, n_out

    res = {"seconds": args.seconds, "channels": nch, "in_frames": n_in}
    yf, msf, namesf, stf, n_out = run(True)
    res.update(fused_ms=msf, fused_steps=namesf, out_frames=n_out,
               fused_TBps=stf["algorithmic_bytes"] / (msf * 1e-3) / 1e12, fused_finite=bool(torch.isfinite(yf).all().item()))
    if not args.only_fused:
        y2, ms2, names2, _, _ = run(False)
        d = (yf - y2)
        nrm = float(torch.linalg.norm(y2).item())
        res.update(two_kernel_ms=ms2, two_kernel_steps=names2, relerr_vs_two_kernel=float(torch.linalg.norm(d).item()) / nrm,
                   maxabs_vs_two_kernel=float(d.abs().max().item()))
        bad = (d.abs() > 1e-6 * float(y2.abs().max().item())).nonzero()
        if bad.numel():
            res["first_bad"] = [int(v) for v in bad[0].tolist()]
            res["n_bad"] = int(bad.shape[0])
            fr = bad[:, 1]
            res["bad_frames_minmax"] = [int(fr.min().item()), int(fr.max().item())]
    if args.oracle > 0:
        from oracle_bridge import oracle_sink, relerr

        m = min(n_in, args.oracle)
        pre = np.asfortranarray(noise_t[:, :m].t().cpu().numpy())
        want = oracle_sink(tree(pre, m))
        k = max(0, want.shape[0] - 4096)
        got = yf[:, :k].t().cpu().numpy()
        res["relerr_vs_oracle_prefix"] = float(relerr(got, want[:k]))
        res["oracle_frames"] = int(k)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
