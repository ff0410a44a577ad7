#!/bin/bash
# HBM bytes of the fused kernel's two-array form (`Mix(x, y) |> Filt |> ToFramerate`, two 12.5 M x 8 Float64 arrays, one launch:
# k_rsos with the second array through the step waves' registers) from two separate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE;
# corrected as MI355X_MICROARCH.md prescribes: 2 x FETCH on gfx950, KB units) against its algorithmic bytes.
# Runs on the GPU box (gpurun); result: gpurun_out/pmc_two_arrays_fused.txt.
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_a2f
rm -rf $OUT; mkdir -p $OUT
cd $R
export ONLY="Mix(x, y) | Filt | ToFramerate"
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o m --output-format csv -- python3 tools/operator_matrix.py > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o m --output-format csv -- python3 tools/operator_matrix.py > $OUT/write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o m --output-format csv -- python3 tools/operator_matrix.py > $OUT/stats.log 2>&1
python3 - <<'PY' > $R/gpurun_out/pmc_two_arrays_fused.txt
import csv, glob, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
def per_launch(kind, counter):
    tot, n = 0.0, 0
    for f in glob.glob(R + "/gpurun_out/pmc_a2f/%s/**/*counter_collection.csv" % kind, recursive=True):
        for row in csv.DictReader(open(f)):
            if "so::k_rsos<" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                tot += float(row["Counter_Value"]); n += 1
    return tot / max(n, 1), n
f, nf = per_launch("fetch", "FETCH_SIZE")
w, nw = per_launch("write", "WRITE_SIZE")
n_in, n_out, nch = 12_500_000, 13_605_443, 8
alg = (2 * n_in + n_out) * nch * 8
got = 2 * f * 1024 + w * 1024
print({"kernel": "k_rsos (two arrays)", "launches": [nf, nw], "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
       "corrected_bytes": got, "algorithmic_bytes": alg, "ratio": got / alg})
for p in glob.glob(R + "/gpurun_out/pmc_a2f/stats/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(p)):
        if "so::" in row["Name"]:
            print({"kernel": row["Name"][:70], "calls": row["Calls"], "average_ns": row["AverageNs"]})
PY
cat $R/gpurun_out/pmc_two_arrays_fused.txt
