#!/bin/bash
# round 6: config 4 through the batched one-pass launch (k_rsos_batch) against the three-pass batch, whole and as the shards
# ranks of 2 / 4 / 8 GPUs get (SIGOPS_BENCH_AS: rank 0's share on this GPU, compute only)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for as in "" 0/2 0/4 0/8; do
  for b in -1 1 0; do
    echo "== shard '${as}' SIGOPS_RSOS_BATCH='${b}'"
    SIGOPS_BENCH_AS=$as SIGOPS_RSOS_BATCH=$b SIGOPS_DEBUG_PLAN=1 python3 bench.py --workload config4 --steps 100 --warmup 20 --no-one-shot 2>&1 | \
      python3 -c "
import sys, json
for l in sys.stdin:
    if 'batched single-pass IIR estimate' in l: print('  ', l.strip())
    if l.startswith('{'):
        r = json.loads(l); print('   ms/step %.4f  compute-only %.4f  launches %s  gate %s' % (r['ms_per_step'], r['config']['compute_only_ms'], r['config']['launches_per_step'], (r.get('parity_gate') or {}).get('relerr')))
"
  done
done
} > gpurun_out/probe_r06g.txt 2>&1
