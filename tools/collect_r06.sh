#!/bin/bash
# Round-6 evidence, all on one GPU box (via gpurun): rocprofv3 kernel trace of the default bench.py command, then two
# separate PMC passes (FETCH_SIZE, WRITE_SIZE) for EVERY line DESIGN.md section 5 quotes -- headline, config 3, config 4,
# config 5, the Float32 headline, the two-channel headline -- then the bench lines themselves and the probes.
# Summaries under gpurun_out/prof_summary/ (copied to profiles/r06/).
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof
S=$R/gpurun_out/prof_summary
rm -rf $OUT $S; mkdir -p $OUT $S
cd $R
python3 -c "import bench; print(bench.kernel_sources_sha16())" > $OUT/kernel_sources_sha16.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench --output-format csv -- python3 bench.py --cpu-seconds 0 > $OUT/trace.log 2>&1
pmc() {  # tag, bench.py arguments
  local tag=$1; shift
  timeout 900 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch_$tag -o bench --output-format csv -- python3 bench.py "$@" > $OUT/fetch_$tag.log 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE -d $OUT/write_$tag -o bench --output-format csv -- python3 bench.py "$@" > $OUT/write_$tag.log 2>&1
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/stats_$tag -o bench --output-format csv -- python3 bench.py "$@" > $OUT/stats_$tag.log 2>&1
  f=$(find $OUT/stats_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-400 "$f" | head -12 > $S/kernel_stats_$tag.csv
}
pmc ns --workload ns --no-secondary --steps 100 --warmup 10 --cpu-seconds 0 --no-one-shot
pmc config3 --workload config3 --no-secondary --steps 100 --warmup 10 --cpu-seconds 0 --no-one-shot
pmc config4 --workload config4 --steps 50 --warmup 10 --cpu-seconds 0 --no-one-shot
pmc config5 --workload config5 --steps 20 --warmup 5 --cpu-seconds 0 --no-one-shot
pmc ns_f32 --workload ns --dtype f32 --no-secondary --steps 100 --warmup 10 --cpu-seconds 0 --no-one-shot
pmc ns_2ch --workload ns --channels 2 --seconds 2400 --no-secondary --steps 100 --warmup 10 --cpu-seconds 0 --no-one-shot
python3 tools/summarize_profiles.py $OUT $S
grep "^{" $OUT/trace.log > $S/bench_default.json
python3 bench.py --steps 20 --warmup 5 > $S/bench_20_5.json 2>/dev/null
python3 bench.py --steps 200 --warmup 30 --no-one-shot > $S/bench_200_30.json 2>/dev/null
SIGOPS_NO_RSOS=1 python3 bench.py --steps 200 --warmup 30 --cpu-seconds 0 --no-secondary --no-one-shot > $S/bench_200_30_two_kernels.json 2>/dev/null
python3 bench.py --dtype f32 --steps 200 --warmup 30 --cpu-seconds 0 --no-one-shot > $S/bench_ns_f32.json 2>/dev/null
SIGOPS_RSOS_NO_F32MFMA=1 python3 bench.py --dtype f32 --steps 200 --warmup 30 --cpu-seconds 0 --no-one-shot --no-secondary > $S/bench_ns_f32_f64products.json 2>/dev/null
python3 bench.py --channels 2 --seconds 2400 --steps 200 --warmup 30 --cpu-seconds 0 --no-one-shot --no-secondary > $S/bench_ns_2ch.json 2>/dev/null
python3 bench.py --workload config3 --dtype f32 --steps 200 --warmup 30 --cpu-seconds 0 --no-one-shot > $S/bench_config3_f32.json 2>/dev/null
python3 bench.py --workload config4 --steps 100 --warmup 20 --no-one-shot > $S/bench_config4_1gpu.json 2>/dev/null
python3 bench.py --workload config5 --steps 50 --warmup 10 --no-one-shot > $S/bench_config5_slab.json 2>/dev/null
python3 bench_configs.py > $S/bench_configs.jsonl 2>/dev/null
python3 tools/operator_matrix.py > $S/operator_matrix.txt 2>/dev/null
F32=1 python3 tools/operator_matrix.py > $S/operator_matrix_f32.txt 2>/dev/null
python3 tools/norm_probe.py > $S/norm_probe.txt 2>/dev/null
bash tools/probe_r06d.sh > /dev/null 2>&1; cp gpurun_out/r06/few_channel_mix.txt $S/few_channel_mix.txt
python3 tools/iir_one_pass_probe.py 2>/dev/null > $S/iir_one_pass.jsonl
python3 tools/nk_probe.py 2>/dev/null | grep '^{' > $S/nk_probe.txt
bash tools/probe_r06g.sh > /dev/null 2>&1; cp gpurun_out/probe_r06g.txt $S/config4_batch_shards.txt
python3 tools/iir_mix_probe.py 2>/dev/null > $S/iir_mix_probe.txt
python3 tools/arr2_probe.py 2>/dev/null | grep '^{' > $S/arr2_probe.txt
python3 tools/headline_parity_loop.py 100 2>/dev/null > $S/headline_parity_loop.txt
python3 tools/soak_rsos_f32m.py 0 > $S/relerr_maxima_rsos_f32m.json 2> $S/relerr_maxima_rsos_f32m.err
for d in 344 72 388 164 224 60 1; do echo "debug=$d $(SIGOPS_RSOS_DEBUG=$d python3 tools/rsos_probe.py --seconds 600 --only-fused --oracle 0 --warm 40 --reps 100 2>/dev/null | grep -o '"fused_ms": [0-9.]*')"; done > $S/rsos_ablation.txt
ls -la $S
