#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_s12; mkdir -p $O
C=$PWD/signaloperators.jl_amd/csrc
probe() { timeout 300 python3 tools/rsos_probe.py --seconds 600 --only-fused --oracle 0 --warm ${WARM:-40} --reps ${REPS:-100} $EXTRA 2>$O/err_$1.txt | grep -o '"fused_ms": [0-9.]*'; }
echo "== main"; probe new
echo "== nosaddr"; SIGOPS_LIB=$C/libsigops_nosaddr.so probe nosaddr
echo "== main"; probe new2
echo "== nosaddr"; SIGOPS_LIB=$C/libsigops_nosaddr.so probe nosaddr2
for d in 344 72 388 164 224 60 1; do echo "== debug=$d"; SIGOPS_RSOS_DEBUG=$d probe d$d; done
echo "== y alone nosaddr"; SIGOPS_RSOS_DEBUG=344 SIGOPS_LIB=$C/libsigops_nosaddr.so probe ynos
echo "== plain"; EXTRA=--plain probe plain
echo "== bench 20/5"; python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('20/5', d['ms_per_step'], d.get('steady_state_ms'), d['roofline']['frac'])"
echo "== bench 100/30"; python3 bench.py --steps 100 --warmup 30 --cpu-seconds 0 --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('100/30', d['ms_per_step'], d.get('steady_state_ms'))"
