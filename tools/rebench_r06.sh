#!/bin/bash
# the bench lines once more with profiles/r06/bench_pmc_hbm.json (collected on these kernel sources) in place: `roofline.traffic` quoted
R=${GRAFT_REPO_ROOT:-/root/repo}
S=$R/gpurun_out/prof_summary
mkdir -p $S
cd $R
python3 bench.py --steps 20 --warmup 5 > $S/bench_20_5.json 2>/dev/null
python3 bench.py --steps 200 --warmup 30 --no-one-shot > $S/bench_200_30.json 2>/dev/null
python3 bench.py --dtype f32 --steps 200 --warmup 30 --cpu-seconds 0 --no-one-shot > $S/bench_ns_f32.json 2>/dev/null
python3 bench.py --channels 2 --seconds 2400 --steps 200 --warmup 30 --cpu-seconds 0 --no-one-shot --no-secondary > $S/bench_ns_2ch.json 2>/dev/null
python3 bench.py --workload config4 --steps 100 --warmup 20 --no-one-shot > $S/bench_config4_1gpu.json 2>/dev/null
python3 bench.py --workload config5 --steps 50 --warmup 10 --no-one-shot > $S/bench_config5_slab.json 2>/dev/null
for f in bench_20_5 bench_200_30 bench_ns_f32 bench_ns_2ch bench_config4_1gpu bench_config5_slab; do python3 -c "
import json
d=json.loads(open('$S/$f.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$f', round(d['ms_per_step'],4), r.get('frac'), r.get('traffic'), (d.get('roofline_sink') or {}).get('traffic'))"; done
