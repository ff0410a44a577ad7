#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_s7; mkdir -p $O
C=$PWD/signaloperators.jl_amd/csrc
probe() { timeout 300 python3 tools/rsos_probe.py --seconds 600 --only-fused --oracle 0 --warm ${WARM:-40} --reps ${REPS:-100} $EXTRA 2>$O/err_$1.txt | grep -o '"fused_ms": [0-9.]*'; }
echo "== main (chain order 1, saddr stores)"; probe new; probe new2
echo "== co0 (chain order 0)"; SIGOPS_LIB=$C/libsigops_co0.so probe co0
echo "== chain alone main / co0"; SIGOPS_RSOS_DEBUG=388 probe c1; SIGOPS_RSOS_DEBUG=388 SIGOPS_LIB=$C/libsigops_co0.so probe c0
echo "== y alone"; SIGOPS_RSOS_DEBUG=344 probe y
echo "== wtol 56"; SIGOPS_RSOS_WTOL=56 probe w56
echo "== plain"; EXTRA=--plain probe plain
echo "== parity"
timeout 600 python3 tools/rsos_probe.py --seconds 600 --warm 3 --reps 5 2>$O/err_parity.txt
