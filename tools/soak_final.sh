#!/bin/bash
# A time-boxed pass of the soak tools over the seeds BASE .. (run on the GPU box via gpurun):
#   gpurun --timeout 3900 -- 'bash tools/soak_final.sh 9000'
# Tails under gpurun_out/soak_final_BASE.log.  profiles/r03/soak_final*.txt: bases 7000, 8000, 9000 on the round-3 code.
B=${1:-7000}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
L=$R/gpurun_out/soak_final_$B.log
mkdir -p $R/gpurun_out; : > $L
run() { echo "== $*" >> $L; timeout 600 python3 "$@" 2>&1 | tail -3 >> $L; }
run tools/tree_soak_long.py $B $((B+100))
run tools/tree_soak_long.py $((B+200)) $((B+260)) 1000 multirate
run tools/tree_soak_multirate.py $B $((B+100))
run tools/soak_long_misc.py $B $((B+100))
run tools/soak_long_more.py $B $((B+100))
run tools/soak_long_resample.py $B $((B+80))
run tools/soak_kernels.py $B $((B+300))
run tools/soak_device_leaves.py $B $((B+100))
run tools/soak_time_shards.py $B $((B+20))
run tools/soak_block_stream.py $B $((B+30))
run tools/soak_stream_long.py $B $((B+16))
run tools/soak_raw_and_wav.py $B $((B+80))
run tools/soak_round3.py $B $((B+400))
cat $L
