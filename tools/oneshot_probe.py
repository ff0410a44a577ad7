"""One-shot cost of the headline sink in a FRESH process: plan create + first execute + destroy -- what a user calling
sink(x, HIPSink) once pays (reference: the user's call is `sink`, src/sink.jl:154-168) --, then the same twice more in the
process (its caches warm).  Leaf and result are device tensors; SIGOPS_CACHE_DIR (if set) keeps the replay of DSP.jl's phase
accumulator between processes.  bench.py runs this as a child process for its `one_shot` object."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
_t_imp = time.perf_counter()
import sigops_amd as so  # (opens the engine library: dlopen + code-object registration, reported as import_ms, not part of a sink)
import_ms = round((time.perf_counter() - _t_imp) * 1e3, 3)
dev = torch.device("cuda:0")
nch, n_in = 8, int(round(600 * 44100))
gen = torch.Generator(device=dev); gen.manual_seed(1983)
noise_t = torch.randn((nch, n_in), dtype=torch.float64, device=dev, generator=gen)
noise = noise_t.t()
stream = torch.cuda.current_stream().cuda_stream
x = (so.Mix(so.Signal(so.sin, ω=1 * so.kHz), so.Signal(noise, 44.1 * so.kHz)) | so.Until(n_in * so.frames)
     | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.ToFramerate(48 * so.kHz))
n_out = so.nframes(x)
out_t = torch.empty((nch, n_out), dtype=torch.float64, device=dev); out = out_t.t()
if os.environ.get('ONESHOT_PRECOPY'):
    _h = torch.zeros(1 << int(os.environ['ONESHOT_PRECOPY']), dtype=torch.uint8)
    _d = _h.to(dev); torch.cuda.synchronize()
torch.cuda.synchronize()
res = []
for i in range(3):
    t0 = time.perf_counter()
    plan = so.Plan(so.ToChannels(x, nch), (n_out, nch), np.float64, (out.stride(0), out.stride(1)), True, device=0)
    t1 = time.perf_counter()
    plan.execute(out.data_ptr(), stream); torch.cuda.synchronize()
    t2 = time.perf_counter()
    plan.close()
    t3 = time.perf_counter()
    res.append({"create_ms": round((t1 - t0) * 1e3, 3), "execute_ms": round((t2 - t1) * 1e3, 3), "destroy_ms": round((t3 - t2) * 1e3, 3),
                "one_shot_ms": round((t3 - t0) * 1e3, 3)})
print(json.dumps({"cache_dir": os.environ.get("SIGOPS_CACHE_DIR"), "import_ms": import_ms, "calls": res}), flush=True)
