"""Where Normpower's time goes: the sum of squares, the dividing pass, a multiply of the same shape (12.5 M x 8, Float64 / F32=1)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sigops_amd as so

nch, n = 8, int(12.5e6)
fs = 44.1 * so.kHz
TDT, NDT = (torch.float32, np.float32) if os.environ.get("F32") else (torch.float64, np.float64)
x = torch.randn((nch, n), dtype=TDT, device="cuda").t()
X = so.Signal(x, fs)
cases = {
    "copy": lambda: X | so.Until(n * so.frames),
    "x * 0.5": lambda: X | so.Amplify(0.5),
    "x / 3": lambda: so.OperateOn("/", X, 3.0),
    "Normpower": lambda: X | so.Normpower,
    "Normpower | Amplify(-20dB)": lambda: X | so.Normpower | so.Amplify(-20 * so.dB),
    "Filt | Normpower": lambda: X | so.Filt(so.Lowpass, 3 * so.kHz) | so.Normpower,
    "Filt": lambda: X | so.Filt(so.Lowpass, 3 * so.kHz),
    "ToFramerate | Normpower": lambda: X | so.ToFramerate(48 * so.kHz) | so.Normpower,
}
for name, mk in cases.items():
    tree = mk()
    n_out = so.nframes(tree)
    out = torch.empty((nch, n_out), dtype=TDT, device="cuda").t()
    plan = so.Plan(so.ToChannels(tree, nch), (n_out, nch), NDT, (out.stride(0), out.stride(1)), True)
    plan.set_profiling(2)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(20):
        plan.execute(out.data_ptr(), st)
    torch.cuda.synchronize()
    plan.set_profiling(2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        plan.execute(out.data_ptr(), st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    steps = " + ".join("%s %.3f" % (s["name"].replace("k_", ""), s["ms"]) for s in plan.steps())
    plan.close()
    print(f"{name:28s} {ms:7.3f} ms   [{steps}]", flush=True)
