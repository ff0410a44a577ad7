"""diagnostic: load libsigops first, then torch; both must see the GPU"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t = time.time()
import sigops_amd as so
from sigops_amd import _capi
print("devices via libsigops:", _capi.lib().so_device_count(), flush=True)
import numpy as np
x = so.Signal(np.ones((100, 2)), 10 * so.Hz) | so.Amplify(2.0)
print("sink before torch:", so.sink(x)[0].sum(), flush=True)
import torch
print("torch cuda:", torch.cuda.is_available(), flush=True)
a = torch.ones(4, device="cuda")
print("torch sum:", float(a.sum()), time.time() - t, flush=True)
