#!/bin/bash
# Round-4 evidence, all on one GPU box (via gpurun): rocprofv3 kernel trace + PMC passes of the default bench.py
# command (tools/collect_profiles.sh), then the other workloads.  Summaries under gpurun_out/prof_summary/
# (copied to profiles/r04/).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/tools/collect_profiles.sh > $R/gpurun_out/collect.log 2>&1
S=$R/gpurun_out/prof_summary
cd $R
python3 bench.py --steps 20 --warmup 5 > $S/bench_20_5.json 2>/dev/null
python3 bench.py --steps 200 --warmup 30 > $S/bench_200_30.json 2>/dev/null
SIGOPS_NO_RSOS=1 python3 bench.py --steps 200 --warmup 30 --cpu-seconds 0 --no-secondary > $S/bench_200_30_two_kernels.json 2>/dev/null
python3 bench.py --dtype f32 --steps 200 --warmup 30 --cpu-seconds 0 > $S/bench_ns_f32.json 2>/dev/null
python3 bench.py --workload config3 --dtype f32 --steps 200 --warmup 30 --cpu-seconds 0 > $S/bench_config3_f32.json 2>/dev/null
python3 bench.py --workload config4 --steps 100 --warmup 20 > $S/bench_config4_1gpu.json 2>/dev/null
python3 bench.py --workload config5 --steps 50 --warmup 10 > $S/bench_config5_slab.json 2>/dev/null
python3 bench_configs.py > $S/bench_configs.jsonl 2>/dev/null
python3 tools/bench_irrational.py > $S/bench_irrational.jsonl 2>/dev/null
ls -la $S
