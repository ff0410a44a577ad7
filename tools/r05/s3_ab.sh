#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_s3; mkdir -p $O
C=$PWD/signaloperators.jl_amd/csrc
probe() { timeout 300 python3 tools/rsos_probe.py --seconds 600 --only-fused --oracle 0 --warm ${WARM:-40} --reps ${REPS:-100} 2>$O/err_$1.txt | grep -o '"fused_ms": [0-9.]*'; }
echo "== r4 lib"; SIGOPS_LIB=$C/libsigops_r4.so probe r4
echo "== new, no trim"; SIGOPS_RSOS_NOTRIM=1 probe notrim
echo "== new"; probe new
echo "== new again"; probe new2
echo "== new wtol 52"; SIGOPS_RSOS_WTOL=52 probe w52
echo "== new wtol 45"; SIGOPS_RSOS_WTOL=45 probe w45
echo "== chain alone r4 / new"; SIGOPS_RSOS_DEBUG=388 SIGOPS_LIB=$C/libsigops_r4.so probe c_r4; SIGOPS_RSOS_DEBUG=388 probe c_new
echo "== y alone r4 / new"; SIGOPS_RSOS_DEBUG=344 SIGOPS_LIB=$C/libsigops_r4.so probe y_r4; SIGOPS_RSOS_DEBUG=344 probe y_new
echo "== counts"
SIGOPS_LIB=$C/libsigops_count.so SIGOPS_RSOS_TRACE=2 WARM=5 REPS=3 probe count
grep rsos-count $O/err_count.txt | tail -16
echo "== parity (full length vs two kernels, oracle prefix)"
timeout 600 python3 tools/rsos_probe.py --seconds 600 --warm 3 --reps 5 2>$O/err_parity.txt
SIGOPS_RSOS_WTOL=52 timeout 600 python3 tools/rsos_probe.py --seconds 600 --warm 3 --reps 5 2>>$O/err_parity.txt
