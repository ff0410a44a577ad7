#!/bin/bash
# round 5, GPU session 1: where k_rsos stands (baseline, role ablations, cycle trace of workgroup 0)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_s1; mkdir -p $O
C=signaloperators.jl_amd/csrc
probe() { timeout 300 python3 tools/rsos_probe.py --seconds 600 --only-fused --oracle 0 --reps ${REPS:-20} 2>$O/err_$1.txt | grep -o '"fused_ms": [0-9.]*'; }
echo "== baseline"; probe base; probe base2
for d in 344 72 388 164 224 60 1; do echo "== debug=$d"; SIGOPS_RSOS_DEBUG=$d probe d$d; done
echo "== trace build"
SIGOPS_LIB=$PWD/$C/libsigops_trace.so probe trace_off
SIGOPS_LIB=$PWD/$C/libsigops_trace.so SIGOPS_RSOS_TRACE=1 REPS=3 probe trace_on
grep rsos-trace $O/err_trace_on.txt > $O/trace.txt
python3 tools/rsos_trace_summary.py $O/trace.txt | tee $O/trace_summary.txt
echo "== bench 20/5 and 100/30"
python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('20/5', d['ms_per_step'], d.get('steady_state_ms'))"
python3 bench.py --steps 100 --warmup 30 --cpu-seconds 0 --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('100/30', d['ms_per_step'], d.get('steady_state_ms'))"
