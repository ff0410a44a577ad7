#!/bin/bash
# A time-boxed pass of the soak tools over fresh seeds (run on the GPU box via gpurun); tails under gpurun_out/soak_final.log
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
L=$R/gpurun_out/soak_final.log
mkdir -p $R/gpurun_out; : > $L
run() { echo "== $*" >> $L; timeout 240 python3 "$@" 2>&1 | tail -4 >> $L; }
run tools/tree_soak_long.py 7000 7040
run tools/tree_soak_long.py 7100 7120 1000 multirate
run tools/tree_soak_multirate.py 7000 7040
run tools/soak_long_misc.py 7000 7040
run tools/soak_long_more.py 7000 7040
run tools/soak_long_resample.py 7000 7030
run tools/soak_kernels.py 7000 7100
run tools/soak_device_leaves.py 7000 7040
run tools/soak_time_shards.py 7000 7008
run tools/soak_block_stream.py 7000 7010
run tools/soak_stream_long.py 7000 7006
run tools/soak_raw_and_wav.py 7000 7030
cat $L
