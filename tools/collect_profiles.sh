#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + two separate PMC passes of bench.py.
# Writes raw output under gpurun_out/prof/ and the summaries to gpurun_out/prof_summary/.
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof
SUM=$R/gpurun_out/prof_summary
rm -rf $OUT $SUM; mkdir -p $OUT $SUM
ARGS="--steps 200 --warmup 30 --cpu-seconds 0"
cd $R
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench --output-format csv -- python3 bench.py $ARGS > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o bench --output-format csv -- python3 bench.py $ARGS > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o bench --output-format csv -- python3 bench.py $ARGS > $OUT/write.log 2>&1
python3 tools/summarize_profiles.py $OUT $SUM
ls -la $SUM
