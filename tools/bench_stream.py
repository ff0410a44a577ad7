"""Streaming throughput: the north-star pipeline evaluated block by block (`so.stream`), device-resident
blocks, against the one-shot sink of the same tree.  One JSON line per block size.
python tools/bench_stream.py [seconds_of_signal]"""
import json, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import sigops_amd as so

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
n = int(secs * 44100)
g = torch.Generator(device="cuda"); g.manual_seed(7)
x = torch.randn((8, n), dtype=torch.float64, device="cuda", generator=g).t()
tree = (so.Mix(so.Signal(so.sin, ω=1 * so.kHz), so.Signal(x, 44.1 * so.kHz)) | so.Until(n * so.frames)
        | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.ToFramerate(48 * so.kHz))
m = so.nframes(tree)
for _ in range(2):
    whole, _fs = so.sink(tree, "torch")
torch.cuda.synchronize()
t0 = time.perf_counter(); whole, _fs = so.sink(tree, "torch"); torch.cuda.synchronize(); t_whole = time.perf_counter() - t0
for bs in (48000 * 600 // 4, 48000 * 60, 48000 * 10, 48000):
    if bs > m: continue
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        err = 0.0; pos = 0; nb = 0
        for blk, _ in so.stream(tree, bs, "torch"):
            if rep == 1 and nb % max(1, (m // bs) // 8) == 0:
                err = max(err, float((blk - whole[pos:pos + blk.shape[0]]).abs().max()))
            pos += blk.shape[0]; nb += 1
        torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(json.dumps({"workload": "north-star pipeline, %g s, 8 ch, streamed" % secs, "block_frames": bs, "blocks": nb,
                      "ms_total": el * 1e3, "ms_per_block": el * 1e3 / nb, "frames_per_s": m / el,
                      "one_shot_sink_ms_incl_plan": t_whole * 1e3, "max_abs_diff_vs_one_shot(sampled)": err}), flush=True)
