#!/bin/bash
# A longer time-boxed pass over fresh seeds (run on the GPU box via gpurun); tails under gpurun_out/soak_final3.log
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
L=$R/gpurun_out/soak_final3.log
mkdir -p $R/gpurun_out; : > $L
run() { echo "== $*" >> $L; timeout 600 python3 "$@" 2>&1 | tail -3 >> $L; }
run tools/tree_soak_long.py 9000 9100
run tools/tree_soak_long.py 9200 9260 1000 multirate
run tools/tree_soak_multirate.py 9000 9100
run tools/soak_long_misc.py 9000 9100
run tools/soak_long_more.py 9000 9100
run tools/soak_long_resample.py 9000 9080
run tools/soak_kernels.py 9000 9300
run tools/soak_device_leaves.py 9000 9100
run tools/soak_time_shards.py 9000 9020
run tools/soak_block_stream.py 9000 9030
run tools/soak_stream_long.py 9000 9016
run tools/soak_raw_and_wav.py 9000 9080
cat $L
