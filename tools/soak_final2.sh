#!/bin/bash
# A longer time-boxed pass over fresh seeds (run on the GPU box via gpurun); tails under gpurun_out/soak_final2.log
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
L=$R/gpurun_out/soak_final2.log
mkdir -p $R/gpurun_out; : > $L
run() { echo "== $*" >> $L; timeout 420 python3 "$@" 2>&1 | tail -3 >> $L; }
run tools/tree_soak_long.py 8000 8080
run tools/tree_soak_long.py 8100 8140 1000 multirate
run tools/tree_soak_multirate.py 8000 8080
run tools/soak_long_misc.py 8000 8080
run tools/soak_long_more.py 8000 8080
run tools/soak_long_resample.py 8000 8060
run tools/soak_kernels.py 8000 8200
run tools/soak_device_leaves.py 8000 8080
run tools/soak_time_shards.py 8000 8016
run tools/soak_block_stream.py 8000 8020
run tools/soak_stream_long.py 8000 8012
run tools/soak_raw_and_wav.py 8000 8060
cat $L
