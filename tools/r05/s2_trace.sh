#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_s2; mkdir -p $O
C=signaloperators.jl_amd/csrc
probe() { timeout 300 python3 tools/rsos_probe.py --seconds 600 --only-fused --oracle 0 --warm ${WARM:-40} --reps ${REPS:-100} 2>$O/err_$1.txt | grep -o '"fused_ms": [0-9.]*'; }
echo "== baseline"; probe base
for d in 344 72 388 164 224 60 1 128; do echo "== debug=$d"; SIGOPS_RSOS_DEBUG=$d probe d$d; done
echo "== trace"
SIGOPS_LIB=$PWD/$C/libsigops_trace.so SIGOPS_RSOS_TRACE=1 WARM=3 REPS=3 probe trace_on
grep rsos-trace $O/err_trace_on.txt > $O/trace.txt
python3 tools/rsos_trace_summary.py $O/trace.txt | tee $O/trace_summary.txt
echo "== trace y alone (344)"
SIGOPS_RSOS_DEBUG=344 SIGOPS_LIB=$PWD/$C/libsigops_trace.so SIGOPS_RSOS_TRACE=1 WARM=3 REPS=3 probe trace_y
grep rsos-trace $O/err_trace_y.txt > $O/trace_y.txt
python3 tools/rsos_trace_summary.py $O/trace_y.txt | tee $O/trace_y_summary.txt
