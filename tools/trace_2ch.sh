#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r06
cd $R
export SIGOPS_LIB=$R/signaloperators.jl_amd/csrc/libsigops_trace.so SIGOPS_RSOS_TRACE=1 SIGOPS_RSOS_TRACE_SKIP=30
for tag in "b:0" "c:131072"; do
  t=${tag%%:*}; d=${tag##*:}
  SIGOPS_RSOS_DEBUG=$d python3 tools/rsos_probe.py --only-fused --oracle 0 --warm 40 --reps 3 --seconds 2400 --channels 2 2> gpurun_out/r06/trace_2ch_mix_$t.txt | tail -1 | cut -c1-120
  python3 tools/rsos_trace_summary.py gpurun_out/r06/trace_2ch_mix_$t.txt | grep -i "loader\|w01\|chain"
done
