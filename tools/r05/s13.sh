#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_s13; mkdir -p $O
C=$PWD/signaloperators.jl_amd/csrc
probe() { timeout 300 python3 tools/rsos_probe.py --seconds 600 --only-fused --oracle 0 --warm ${WARM:-40} --reps ${REPS:-100} $EXTRA 2>$O/err_$1.txt | grep -o '"fused_ms": [0-9.]*'; }
echo "== main"; probe new
echo "== chunk 64"; SIGOPS_RSOS_CHUNK=64 probe ch64
echo "== counts (launches 100..102)"
SIGOPS_LIB=$C/libsigops_count.so SIGOPS_RSOS_TRACE=2 SIGOPS_RSOS_TRACE_SKIP=100 WARM=60 REPS=43 probe count
grep rsos-count $O/err_count.txt | tail -16
