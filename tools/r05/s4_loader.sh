#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_s4; mkdir -p $O
C=$PWD/signaloperators.jl_amd/csrc
probe() { timeout 300 python3 tools/rsos_probe.py --seconds 600 --only-fused --oracle 0 --warm ${WARM:-40} --reps ${REPS:-100} 2>$O/err_$1.txt | grep -o '"fused_ms": [0-9.]*'; }
echo "== new"; probe new
for d in 16 32 48 8 4 12 28 2 2048; do echo "== debug=$d"; SIGOPS_RSOS_DEBUG=$d probe d$d; done
echo "== chunk 64"; SIGOPS_RSOS_CHUNK=64 probe ch64
for dp in 2 3; do echo "== depth $dp"; SIGOPS_RSOS_DEPTH=$dp probe dp$dp; done
echo "== ring 512"; SIGOPS_RSOS_RING=512 probe r512
echo "== plain (no Mix)"; timeout 300 python3 tools/rsos_probe.py --plain --seconds 600 --only-fused --oracle 0 --warm 40 --reps 100 2>/dev/null | grep -o '"fused_ms": [0-9.]*'
