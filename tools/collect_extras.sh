#!/bin/bash
# Secondary measurements kept under profiles/rNN/ (run on the GPU box after tools/collect_profiles.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}; S=$R/gpurun_out/prof_summary; mkdir -p $S; cd $R
python bench.py --workload config3 --dtype f32 --no-secondary --cpu-seconds 0 2>/dev/null | grep "^{" > $S/bench_config3_f32.json
python bench.py --workload config3 --no-secondary --cpu-seconds 0 2>/dev/null | grep "^{" > $S/bench_config3_f64.json
python bench.py --workload config4 --steps 20 --warmup 5 2>/dev/null | grep "^{" > $S/bench_config4_1gpu.json
python bench.py --workload config5 --steps 10 --warmup 3 2>/dev/null | grep "^{" > $S/bench_config5_slab.json
python bench_configs.py --only 1,2,ns,4,5,k1 2>/dev/null > $S/bench_configs.jsonl
SIGOPS_SOS_ONEPASS=1 python bench_configs.py --only ns,2 2>/dev/null > $S/bench_configs_onepass.jsonl
python tools/bench_irrational.py 2>/dev/null > $S/bench_irrational.jsonl
python tools/bench_interleaved.py 2>/dev/null > $S/bench_interleaved.jsonl
python tools/bench_stream.py 600 2>/dev/null | grep "^{" > $S/bench_stream.jsonl
python bench.py --workload ns_time --steps 50 --warmup 10 2>/dev/null | grep "^{" > $S/bench_ns_time_1gpu.json
SIGOPS_BENCH_AS=3/8 python bench.py --workload ns_time --steps 50 --warmup 10 2>/dev/null | grep "^{" > $S/bench_ns_time_rank3of8.json
python bench.py --dtype f32 --no-secondary --cpu-seconds 0 2>/dev/null | grep "^{" > $S/bench_ns_f32.json
ls -la $S
