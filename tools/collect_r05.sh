#!/bin/bash
# Round-5 evidence, all on one GPU box (via gpurun): rocprofv3 kernel trace + PMC passes of the default bench.py
# command (tools/collect_profiles.sh), then the other workloads and probes.  Summaries under gpurun_out/prof_summary/
# (copied to profiles/r05/).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/tools/collect_profiles.sh > $R/gpurun_out/collect.log 2>&1
S=$R/gpurun_out/prof_summary
cd $R
python3 bench.py --steps 20 --warmup 5 > $S/bench_20_5.json 2>/dev/null
python3 bench.py --steps 200 --warmup 30 --no-one-shot > $S/bench_200_30.json 2>/dev/null
SIGOPS_NO_RSOS=1 python3 bench.py --steps 200 --warmup 30 --cpu-seconds 0 --no-secondary > $S/bench_200_30_two_kernels.json 2>/dev/null
python3 bench.py --dtype f32 --steps 200 --warmup 30 --cpu-seconds 0 --no-one-shot > $S/bench_ns_f32.json 2>/dev/null
python3 bench.py --workload config3 --dtype f32 --steps 200 --warmup 30 --cpu-seconds 0 > $S/bench_config3_f32.json 2>/dev/null
python3 bench.py --workload config4 --steps 100 --warmup 20 > $S/bench_config4_1gpu.json 2>/dev/null
python3 bench.py --workload config5 --steps 50 --warmup 10 > $S/bench_config5_slab.json 2>/dev/null
python3 bench_configs.py > $S/bench_configs.jsonl 2>/dev/null
python3 tools/bench_irrational.py > $S/bench_irrational.jsonl 2>/dev/null
python3 tools/operator_matrix.py > $S/operator_matrix.txt 2>/dev/null
F32=1 python3 tools/operator_matrix.py > $S/operator_matrix_f32.txt 2>/dev/null
python3 tools/iir_one_pass_probe.py 2>/dev/null > $S/iir_one_pass.jsonl
python3 tools/f32_mfma_probe.py 2>/dev/null > $S/f32_mfma_probe.txt
python3 tools/headline_parity_loop.py 50 2>/dev/null > $S/headline_parity_loop.txt
for d in 344 72 388 164 224 60 1; do echo "debug=$d $(SIGOPS_RSOS_DEBUG=$d python3 tools/rsos_probe.py --seconds 600 --only-fused --oracle 0 --warm 40 --reps 100 2>/dev/null | grep -o '"fused_ms": [0-9.]*')"; done > $S/rsos_ablation.txt
ls -la $S
