import sys; sys.path.insert(0, '.')
import numpy as np, torch, sigops_amd as so
n_in, nch = 26_460_000, 8
st = torch.cuda.current_stream().cuda_stream
for dt, tdt in ((np.float64, torch.float64), (np.float32, torch.float32)):
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    a = torch.randn((nch, n_in), dtype=tdt, device="cuda", generator=g)
    b = torch.randn((nch, n_in), dtype=tdt, device="cuda", generator=g)
    c = torch.randn((nch, n_in), dtype=tdt, device="cuda", generator=g)
    A, B, C = (so.Signal(t.t(), 44.1 * so.kHz) for t in (a, b, c))
    cases = [("Mix(a,b)", so.Mix(A, B), 3), ("Mix(a,b,c)", so.Mix(A, B, C), 4), ("Amplify(Mix(a,b),c)", so.Amplify(so.Mix(A, B), C), 4),
             ("Amplify(a, 0.5)", so.Amplify(A, 0.5), 2), ("a |> Ramp", A | so.Ramp(100 * so.ms), 2)]
    for name, tree, narr in cases:
        n_out = so.nframes(tree)
        out = torch.empty((nch, n_out), dtype=tdt, device="cuda")
        p = so.Plan(so.ToChannels(tree, nch), (n_out, nch), dt, (1, n_out), True)
        for _ in range(5): p.execute(out.data_ptr(), st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): p.execute(out.data_ptr(), st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        p.set_profiling(True); p.execute(out.data_ptr(), st); torch.cuda.synchronize()
        by = narr * nch * n_out * (8 if dt == np.float64 else 4)
        print(name, dt.__name__, "ms", round(ms, 4), "TB/s", round(by / ms / 1e9, 2), [s["name"] for s in p.steps()], flush=True)
        p.close(); del out
    del a, b, c
