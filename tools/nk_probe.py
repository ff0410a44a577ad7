"""round 6: the state's k-steps follow the cascade's order (rsos_chain<NY, NK>: 2 MFMAs per block on the chain wave for 3 - 4
sections, 1 for 1 - 2, instead of always 3; the y waves' S^T C^T as many).  A plain Filt in one pass, a resampler + Filt fused,
filters of 2 / 3 / 4 / 5 sections, with the reduction and without (SIGOPS_RSOS_DEBUG=131072: three k-steps whatever the order)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sigops_amd as so
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream().cuda_stream
def run(what, n, nch, order, kind, dbg):
    if dbg: os.environ["SIGOPS_RSOS_DEBUG"] = "131072"
    else: os.environ.pop("SIGOPS_RSOS_DEBUG", None)
    x_t = torch.randn((nch, n), dtype=torch.float64, device=dev)
    f = so.Filt(so.Lowpass, 4 * so.kHz, order=order) if kind == "low" else so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz, order=order)
    x = so.Signal(x_t.t(), 44.1 * so.kHz)
    x = (x | f) if what == "filt" else (x | so.ToFramerate(48 * so.kHz) | f)
    m = so.nframes(x)
    out_t = torch.empty((nch, m), dtype=torch.float64, device=dev); out = out_t.t()
    plan = so.Plan(so.ToChannels(x, nch), (m, nch), np.float64, (out.stride(0), out.stride(1)), True, device=0)
    for _ in range(30): plan.execute(out.data_ptr(), stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(60): plan.execute(out.data_ptr(), stream)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 60
    names = [s["name"] for s in plan.steps()]; plan.close()
    return {"what": what, "n": n, "nch": nch, "filter": kind, "order": order, "three_ksteps": bool(dbg), "ms": round(ms, 4), "steps": names}
for what, n, nch in (("filt", 12_500_000, 8), ("filt", 3_628_118, 128), ("filt", 25_000_000, 4), ("resample+filt", 12_500_000, 8)):
    for kind, order in (("low", 3), ("low", 5), ("low", 8), ("stop", 5)):
        for dbg in (0, 1):
            print(json.dumps(run(what, n, nch, order, kind, dbg)), flush=True)
