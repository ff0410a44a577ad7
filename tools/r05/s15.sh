#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_s15; mkdir -p $O
probe() { timeout 300 python3 tools/rsos_probe.py --seconds 600 --only-fused --oracle 0 --warm ${WARM:-40} --reps ${REPS:-100} $EXTRA 2>$O/err_$1.txt | grep -o '"fused_ms": [0-9.]*'; }
echo "== main"; probe new
# y alone = 344; +1 no stores; +131072 no window reads; +262144 no dx writes
for d in 344 345 131416 262488 393560 393561; do echo "== debug=$d"; SIGOPS_RSOS_DEBUG=$d probe d$d; done
