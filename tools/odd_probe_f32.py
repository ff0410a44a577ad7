import os, sys
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
import sigops_amd as so
from oracle_bridge import oracle_sink, relerr
for nch, n in ((8, 12500001), (8, 12500002), (8, 12500003), (3, 400001), (8, 300002)):
    x = torch.randn((nch, n), dtype=torch.float32, device="cuda").t()
    for name, mk in (("pipeline", lambda leaf: so.Mix(so.Signal(so.sin, ω=1 * so.kHz), so.Signal(leaf, 44.1 * so.kHz)) | so.Until(n * so.frames) | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.ToFramerate(48 * so.kHz)),
                     ("resample", lambda leaf: so.Signal(leaf, 44.1 * so.kHz) | so.ToFramerate(48 * so.kHz))):
        tree = mk(x)
        n_out = so.nframes(tree)
        odt = torch.float64 if name == "pipeline" else torch.float32
        out = torch.empty((nch, n_out), dtype=odt, device="cuda").t()
        os.environ["SIGOPS_RSOS_MINGROUPS"] = "1"
        plan = so.Plan(so.ToChannels(tree, nch), (n_out, nch), np.float64 if name == "pipeline" else np.float32, (out.stride(0), out.stride(1)), True)
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(2):
            plan.execute(out.data_ptr(), st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            plan.execute(out.data_ptr(), st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        names = [s_["name"] for s_ in plan.steps()]
        plan.close()
        err = None
        if n <= 500000:
            xs = np.asfortranarray(x.cpu().numpy())
            err = relerr(out.cpu().numpy(), oracle_sink(mk(xs)))
        print(nch, n, name, round(ms, 3), "ms", names, "finite", bool(torch.isfinite(out).all().item()), "err", err, flush=True)
