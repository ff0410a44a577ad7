import sys; sys.path.insert(0, '.')
import numpy as np, torch, sigops_amd as so
n_in, nch = 26_460_000, 8
st = torch.cuda.current_stream().cuda_stream
for dt, tdt, esz in ((np.float64, torch.float64, 8), (np.float32, torch.float32, 4)):
    for pad_b, pad_o in ((0, 0), (520, 0), (520, 1040), (65536 // esz + 24, 2 * 65536 // esz + 48), (1 << 18, 1 << 19)):
        # b and the result start `pad` elements into their allocations and their channel rows are `pad` longer
        a = torch.randn((nch, n_in), dtype=tdt, device="cuda")
        bb = torch.randn((nch, n_in + pad_b), dtype=tdt, device="cuda")
        b = bb[:, pad_b // 2: pad_b // 2 + n_in]
        A = so.Signal(a.t(), 44.1 * so.kHz); B = so.Signal(b.t(), 44.1 * so.kHz)
        tree = so.Mix(A, B)
        oo = torch.empty((nch, n_in + pad_o), dtype=tdt, device="cuda")
        p = so.Plan(so.ToChannels(tree, nch), (n_in, nch), dt, (1, n_in + pad_o), True)
        ptr = oo.data_ptr() + (pad_o // 2) * esz
        for _ in range(5): p.execute(ptr, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): p.execute(ptr, st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(dt.__name__, "pad_b", pad_b, "pad_o", pad_o, "ms", round(ms, 4), "TB/s", round(3 * nch * n_in * esz / ms / 1e9, 2), flush=True)
        p.close(); del a, bb, b, oo
