#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_s16; mkdir -p $O
C=$PWD/signaloperators.jl_amd/csrc
probe() { timeout 300 python3 tools/rsos_probe.py --seconds 600 --only-fused --oracle 0 --warm ${WARM:-40} --reps ${REPS:-100} $EXTRA 2>$O/err_$1.txt | grep -o '"fused_ms": [0-9.]*'; }
for i in 1 2; do
echo "== main (early)"; probe new$i
echo "== noearly"; SIGOPS_LIB=$C/libsigops_noearly.so probe ne$i
done
echo "== y alone early / noearly"; SIGOPS_RSOS_DEBUG=344 probe ye; SIGOPS_RSOS_DEBUG=344 SIGOPS_LIB=$C/libsigops_noearly.so probe yne
echo "== parity main"; timeout 900 python3 tools/r05/parity_loop.py 30 2>/dev/null | tail -3
