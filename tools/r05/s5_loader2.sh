#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_s5; mkdir -p $O
C=$PWD/signaloperators.jl_amd/csrc
probe() { timeout 300 python3 tools/rsos_probe.py --seconds 600 --only-fused --oracle 0 --warm ${WARM:-40} --reps ${REPS:-100} $EXTRA 2>$O/err_$1.txt | grep -o '"fused_ms": [0-9.]*'; }
SIGOPS_DEBUG_PLAN=1 timeout 120 python3 tools/rsos_probe.py --seconds 600 --only-fused --oracle 0 --warm 1 --reps 1 2>&1 | grep "k_rsos" | head -5
echo "== r4"; SIGOPS_LIB=$C/libsigops_r4.so probe r4
echo "== new"; probe new
echo "== new old policy"; SIGOPS_RSOS_DEBUG=16384 probe oldpol
echo "== new exact bases"; SIGOPS_RSOS_DEBUG=65536 probe exact
for dp in 2 3; do echo "== depth $dp"; SIGOPS_RSOS_DEPTH=$dp probe dp$dp; done
echo "== ring 640"; SIGOPS_RSOS_RING=640 probe r640
echo "== no gain"; SIGOPS_RSOS_DEBUG=2 probe nogain
echo "== plain"; EXTRA=--plain probe plain
echo "== wtol 52"; SIGOPS_RSOS_WTOL=52 probe w52
echo "== parity"
timeout 600 python3 tools/rsos_probe.py --seconds 600 --warm 3 --reps 5 2>$O/err_parity.txt
SIGOPS_LIB=$C/libsigops_r4.so timeout 600 python3 tools/rsos_probe.py --seconds 600 --warm 3 --reps 5 2>>$O/err_parity.txt
