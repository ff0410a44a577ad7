"""One operator at a time over 1.25e7 frames x 8 channels (Float64): looking for cliffs outside the benches' shapes.
Per row: WARM untimed executes (default 20: the chip raises its clock over the first ~25 ms of a kernel after an idle gap, and
with 3 the matrix-heavy rows read 10 - 25 % above their steady state), then the mean of REPS (default 30) between two events."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import sigops_amd as so

nch, n = int(os.environ.get("NCH", "8")), int(float(os.environ.get("FRAMES", "12.5e6")))
fs = 44.1 * so.kHz
TDT, NDT = (torch.float32, np.float32) if os.environ.get("F32") else (torch.float64, np.float64)
x = torch.randn((nch, n), dtype=TDT, device="cuda").t()
y = torch.randn((nch, n), dtype=TDT, device="cuda").t()
z = torch.randn((nch, n // 2), dtype=TDT, device="cuda").t()
X, Y, Z = so.Signal(x, fs), so.Signal(y, fs), so.Signal(z, fs)
tone = so.Signal(so.sin, ω=1 * so.kHz)
cases = {
    "copy (Until)": lambda: X | so.Until(n * so.frames),
    "Amplify(const)": lambda: X | so.Amplify(0.5),
    "Amplify(sin)": lambda: X | so.Amplify(tone) | so.Until(n * so.frames),
    "Mix(x, y)": lambda: so.Mix(X, Y),
    "Mix(x, y, sin) * 0.5": lambda: so.Mix(X, Y, tone) | so.Until(n * so.frames) | so.Amplify(0.5),
    "Amplify(x, y)": lambda: so.Amplify(X, Y),
    "Ramp 10 ms": lambda: X | so.Ramp(10 * so.ms),
    "RampOn 1 s": lambda: X | so.RampOn(1 * so.s),
    "FadeTo": lambda: so.FadeTo(X, Y, 1 * so.s),
    "After 1 s": lambda: X | so.After(1 * so.s),
    "Pad(zero) + 50 %": lambda: Z | so.Pad(so.zero) | so.Until(n * so.frames),
    "Pad(mirror) + 50 %": lambda: Z | so.Pad(so.mirror) | so.Until((n // 2 + n // 4) * so.frames),
    "Append(z, z)": lambda: so.Append(Z, Z),
    "Append of 16 pieces": lambda: so.Append(*[so.Signal(x[k * (n // 16):(k + 1) * (n // 16)], fs) for k in range(16)]),
    "Normpower": lambda: X | so.Normpower,
    "Normpower | Amplify(-20dB)": lambda: X | so.Normpower | so.Amplify(-20 * so.dB),
    "ToChannels(1) (sum)": lambda: X | so.ToChannels(1),
    "mono ToChannels(8)": lambda: so.Signal(x[:, :1], fs) | so.ToChannels(8),
    "Filt Lowpass o5": lambda: X | so.Filt(so.Lowpass, 3 * so.kHz),
    "Filt Lowpass o1": lambda: X | so.Filt(so.Lowpass, 3 * so.kHz, order=1),
    "Filt Bandpass o8": lambda: X | so.Filt(so.Bandpass, 1 * so.kHz, 4 * so.kHz, order=8),
    "Filt | Filt": lambda: X | so.Filt(so.Lowpass, 3 * so.kHz) | so.Filt(so.Highpass, 100 * so.Hz),
    "Filt | Normpower": lambda: X | so.Filt(so.Lowpass, 3 * so.kHz) | so.Normpower,
    "Filt FIR 101 taps": lambda: so.Filt(X, np.hanning(101) / 50.0),
    "ToFramerate 48k": lambda: X | so.ToFramerate(48 * so.kHz),
    "ToFramerate 48k | Filt": lambda: X | so.ToFramerate(48 * so.kHz) | so.Filt(so.Lowpass, 3 * so.kHz),
    "Mix(x, y) | ToFramerate": lambda: so.Mix(X, Y) | so.ToFramerate(48 * so.kHz),
    "Amplify(x, y) | ToFramerate": lambda: so.Amplify(X, Y) | so.ToFramerate(48 * so.kHz),
    "Mix(x, y) | Filt | ToFramerate": lambda: so.Mix(X, Y) | so.Filt(so.Lowpass, 3 * so.kHz) | so.ToFramerate(48 * so.kHz),
    "ToEltype(Float32)": lambda: so.ToEltype(X, np.float32),
}
only = os.environ.get("ONLY")
WARM, REPS = int(os.environ.get("WARM", "20")), int(os.environ.get("REPS", "30"))
for name, mk in cases.items():
    if only and only not in name:
        continue
    try:
        tree = mk()
        nout, co = so.nframes(tree), so.nchannels(tree)
        odt = np.float32 if ("Float32" in name or NDT == np.float32) else np.float64
        tdt = torch.float32 if odt == np.float32 else torch.float64
        out = torch.empty((co, nout), dtype=tdt, device="cuda").t()
        plan = so.Plan(so.ToChannels(tree, co), (nout, co), odt, (out.stride(0), out.stride(1)), True)
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(WARM):
            plan.execute(out.data_ptr(), st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS):
            plan.execute(out.data_ptr(), st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / REPS
        stt = plan.stats()
        names = "+".join(s_["name"].replace("k_resample_", "rs_").replace("k_", "") for s_ in plan.steps())
        plan.close()
        print(f"{name:28s} {ms:8.3f} ms  {stt['algorithmic_bytes'] / ms / 1e9:6.2f} TB/s  [{names}]", flush=True)
        if os.environ.get("STEPS"):  # per-step device time of one profiled execute
            plan = so.Plan(so.ToChannels(tree, co), (nout, co), odt, (out.stride(0), out.stride(1)), True)
            plan.execute(out.data_ptr(), st)
            plan.set_profiling(1)
            plan.execute(out.data_ptr(), st)
            torch.cuda.synchronize()
            print("    " + "  ".join(f"{s_['name']} {s_['ms']:.3f}" for s_ in plan.steps()), flush=True)
            plan.close()
        del out
    except Exception as exc:  # (a case the mirror does not spell this way)
        print(f"{name:28s} skipped: {type(exc).__name__}: {str(exc)[:80]}", flush=True)
