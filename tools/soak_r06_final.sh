#!/bin/bash
# round 6, last library: the specialised soaks once more (seeds 36000..), every tool time-boxed
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
L=$R/gpurun_out/soak_r06_final.log
: > $L
run() { echo "== $*" >> $L; timeout 300 "$@" 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-400 >> $L; }
run python3 tools/soak_norm.py 36000 36150
run python3 tools/soak_r05.py 36000 36200
run python3 tools/soak_two_arrays.py 300 36000
run env SIGOPS_RSOS_NO_F32MFMA=1 python3 tools/soak_f32_ring.py 300 36000
run python3 tools/soak_f32_ring.py 300 36001
run python3 tools/soak_rates_f32.py 3
run python3 tools/soak_degenerate_filters.py 36000 36060
run python3 tools/soak_degenerate_rates.py 36000 36060
cat $L
