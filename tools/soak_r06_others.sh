#!/bin/bash
# round 6: the specialised soaks on the final library (seeds 27000..): Normpower (the chain path, RmsPatch), the round-5 paths, two
# arrays, the Float32 ring (with the Float64 products: its bit-equality half), the Float32-MFMA form of k_rsos on a second seed
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
L=$R/gpurun_out/soak_r06_others.log
: > $L
run() { echo "== $*" >> $L; timeout 900 "$@" 2>&1 | tail -3 >> $L; }
run python3 tools/soak_norm.py 27000 27150
run python3 tools/soak_r05.py 27000 27200
run python3 tools/soak_two_arrays.py 400 27000
run env SIGOPS_RSOS_NO_F32MFMA=1 python3 tools/soak_f32_ring.py 300 27000
run python3 tools/soak_f32_ring.py 300 27001
run python3 tools/soak_rsos_f32m.py 1
run python3 tools/soak_rates_f32.py 3
run python3 tools/soak_degenerate_filters.py 27000 27060
run python3 tools/soak_degenerate_rates.py 27000 27060
cat $L
