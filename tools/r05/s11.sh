#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_s11; mkdir -p $O
C=$PWD/signaloperators.jl_amd/csrc
probe() { timeout 300 python3 tools/rsos_probe.py --seconds 600 --only-fused --oracle 0 --warm ${WARM:-40} --reps ${REPS:-100} $EXTRA 2>$O/err_$1.txt | grep -o '"fused_ms": [0-9.]*'; }
for i in 1 2; do
echo "== main"; probe new
echo "== ns14 (ring 768)"; SIGOPS_LIB=$C/libsigops_ns14.so probe ns14
echo "== ns14 ring 640"; SIGOPS_RSOS_RING=640 SIGOPS_LIB=$C/libsigops_ns14.so probe ns14r640
done
echo "== main depth 2 / 3"; SIGOPS_RSOS_DEPTH=2 probe d2; SIGOPS_RSOS_DEPTH=3 probe d3
