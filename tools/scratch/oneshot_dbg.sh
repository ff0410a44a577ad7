#!/bin/bash
cd $GRAFT_REPO_ROOT
export SIGOPS_CACHE_DIR=/tmp/sigops_cache_dbg
mkdir -p $SIGOPS_CACHE_DIR
python3 tools/oneshot_probe.py 2>/dev/null
echo "== second process, warm cache dir, debug prints"
SIGOPS_DEBUG_PLAN=1 python3 tools/oneshot_probe.py 2>&1 | grep "plan_create\|calls\|accumul\|cache"
echo "== python profile of the first Plan()"
python3 - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import sigops_amd as so
import cProfile, pstats
dev = torch.device("cuda:0")
nch, n_in = 8, int(round(600 * 44100))
noise = torch.randn((nch, n_in), dtype=torch.float64, device=dev).t()
x = (so.Mix(so.Signal(so.sin, ω=1 * so.kHz), so.Signal(noise, 44.1 * so.kHz)) | so.Until(n_in * so.frames)
     | so.Filt(so.Bandstop, 0.5 * so.kHz, 2 * so.kHz) | so.ToFramerate(48 * so.kHz))
n_out = so.nframes(x)
out = torch.empty((nch, n_out), dtype=torch.float64, device=dev).t()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
plan = so.Plan(so.ToChannels(x, nch), (n_out, nch), np.float64, (out.stride(0), out.stride(1)), True, device=0)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
PY
