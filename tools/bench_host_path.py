"""PCIe-inclusive host-to-host rate of config 3 (NumPy leaf and result): one-shot sink (plan creation
included) and a reused plan; with and without page-locking the host buffers (SIGOPS_NO_PIN=1)."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sigops_amd as so

n, nch = 26_460_000, 8
noise = np.asfortranarray(np.random.default_rng(1).standard_normal((n, nch)))
tree = so.Signal(noise, 44.1 * so.kHz) | so.Amplify(so.Signal(so.sin, ω=5 * so.Hz)) | so.Until(600 * so.s) | so.ToFramerate(48 * so.kHz)
n_out = so.nframes(tree)
res = np.empty((n_out, nch), order="F")
t = []
for _ in range(3):
    t0 = time.perf_counter(); so.sink_into(res, tree); t.append(time.perf_counter() - t0)
plan = so.Plan(so.ToChannels(tree, nch), (n_out, nch), np.float64, (1, n_out), False)
r = []
for _ in range(4):
    t0 = time.perf_counter(); plan.execute(res.ctypes.data); r.append(time.perf_counter() - t0)
plan.close()
print(json.dumps({"pinned": not os.environ.get("SIGOPS_NO_PIN"), "one_shot_ms": [round(x * 1e3, 1) for x in t],
                  "reused_plan_ms": [round(x * 1e3, 1) for x in r], "bytes_h2d": noise.nbytes, "bytes_d2h": res.nbytes}))
