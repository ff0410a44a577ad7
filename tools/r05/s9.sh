#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_s9; mkdir -p $O
C=$PWD/signaloperators.jl_amd/csrc
probe() { timeout 300 python3 tools/rsos_probe.py --seconds 600 --only-fused --oracle 0 --warm ${WARM:-40} --reps ${REPS:-100} $EXTRA 2>$O/err_$1.txt | grep -o '"fused_ms": [0-9.]*'; }
echo "== main"; probe new; probe new2
echo "== co0"; SIGOPS_LIB=$C/libsigops_co0.so probe co0
echo "== wtol 56"; SIGOPS_RSOS_WTOL=56 probe w56
for v in main co0; do
  echo "== parity $v"
  if [ $v = main ]; then timeout 900 python3 tools/r05/parity_loop.py 100 2>/dev/null | tail -4
  else SIGOPS_LIB=$C/libsigops_$v.so timeout 900 python3 tools/r05/parity_loop.py 40 2>/dev/null | tail -4; fi
done
