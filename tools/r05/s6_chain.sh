#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_s6; mkdir -p $O
C=$PWD/signaloperators.jl_amd/csrc
probe() { timeout 300 python3 tools/rsos_probe.py --seconds 600 --only-fused --oracle 0 --warm ${WARM:-40} --reps ${REPS:-100} $EXTRA 2>$O/err_$1.txt | grep -o '"fused_ms": [0-9.]*'; }
echo "== new"; probe new
for d in 388 900 1412 1924 344 72 60 8 4 16 32; do echo "== debug=$d"; SIGOPS_RSOS_DEBUG=$d probe d$d; done
for w in 60 56 52; do echo "== wtol $w"; SIGOPS_RSOS_WTOL=$w probe w$w; done
echo "== counts"
SIGOPS_LIB=$C/libsigops_count.so SIGOPS_RSOS_TRACE=2 WARM=5 REPS=3 probe count
grep rsos-count $O/err_count.txt | tail -16
