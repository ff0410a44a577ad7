#!/bin/bash
# Round-4 side measurements quoted in DESIGN.md (gpurun_out/prof_summary/, copied to profiles/r04/)
R=${GRAFT_REPO_ROOT:-/root/repo}
S=$R/gpurun_out/prof_summary
mkdir -p $S
cd $R
bash tools/rsos_len_sweep.sh > $S/rsos_len_sweep.txt 2>/dev/null
python3 tools/arb_len_sweep.py > $S/arb_len_sweep.txt 2>/dev/null
NCH=2 python3 tools/arb_len_sweep.py >> $S/arb_len_sweep.txt 2>/dev/null
bash tools/arb_abl.sh > $S/arb_abl.txt 2>/dev/null
SIGOPS_BENCH_AS=3/8 python3 bench.py --workload ns_time --steps 100 --warmup 20 --cpu-seconds 0 > $S/bench_ns_time_shard3of8.json 2>/dev/null
SIGOPS_RSOS_NOWINDOWS=1 SIGOPS_BENCH_AS=3/8 python3 bench.py --workload ns_time --steps 100 --warmup 20 --cpu-seconds 0 > $S/bench_ns_time_shard3of8_two_kernels.json 2>/dev/null
bash tools/ab_headline.sh > $S/headline_repeat.txt 2>/dev/null
ls -la $S
