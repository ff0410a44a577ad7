#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace of the default bench.py run, then two
# separate PMC passes (FETCH_SIZE, WRITE_SIZE) per workload.  Raw output under gpurun_out/prof/,
# summaries under gpurun_out/prof_summary/ (copied to profiles/rNN/ by hand).
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof
SUM=$R/gpurun_out/prof_summary
rm -rf $OUT $SUM; mkdir -p $OUT $SUM
cd $R
# (the hash of the kernel sources the counters belong to: taken HERE, before the passes, not when they are summarised)
python3 -c "import bench; print(bench.kernel_sources_sha16())" > $OUT/kernel_sources_sha16.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench --output-format csv -- python3 bench.py --cpu-seconds 0 > $OUT/trace.log 2>&1
for W in ns config3; do
  ARGS="--workload $W --no-secondary --steps 100 --warmup 10 --cpu-seconds 0"
  timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch_$W -o bench --output-format csv -- python3 bench.py $ARGS > $OUT/fetch_$W.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/write_$W -o bench --output-format csv -- python3 bench.py $ARGS > $OUT/write_$W.log 2>&1
done
python3 tools/summarize_profiles.py $OUT $SUM
grep "^{" $OUT/trace.log > $SUM/bench_default.json
ls -la $SUM
