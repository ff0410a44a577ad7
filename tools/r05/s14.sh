#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_s14; mkdir -p $O
C=$PWD/signaloperators.jl_amd/csrc
probe() { timeout 300 python3 tools/rsos_probe.py --seconds 600 --only-fused --oracle 0 --warm ${WARM:-40} --reps ${REPS:-100} $EXTRA 2>$O/err_$1.txt | grep -o '"fused_ms": [0-9.]*'; }
for i in 1 2; do
echo "== main (13,16)"; probe new$i
for v in noprio p1417 p1215; do echo "== $v"; SIGOPS_LIB=$C/libsigops_$v.so probe $v$i; done
done
echo "== parity main"; timeout 900 python3 tools/r05/parity_loop.py 30 2>/dev/null | tail -3
