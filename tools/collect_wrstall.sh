#!/bin/bash
# Write-side stall counters of K3 on config 3 and of the fused kernel on the headline (separate --pmc passes of bench.py,
# like tools/collect_profiles.sh).  Output: gpurun_out/prof_summary/wrstall_<workload>.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_wr
S=$R/gpurun_out/prof_summary
rm -rf $OUT; mkdir -p $OUT $S
cd $R
for W in config3 ns; do
  ARGS="--workload $W --no-secondary --steps 50 --warmup 10 --cpu-seconds 0"
  i=0
  for C in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum" "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" "TCC_EA0_WRREQ_LEVEL_sum GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $C -d $OUT/${W}_$i -o bench --output-format csv -- python3 bench.py $ARGS > $OUT/${W}_$i.log 2>&1
  done
  python3 - $OUT $W > $S/wrstall_$W.txt <<'PY'
import csv, glob, sys, collections
out, w = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/{w}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "k_rsos" in k or "k_resample_periodic" in k:
            acc[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:44s} mean per dispatch {sum(v)/len(v):.4g}   ({len(v)} dispatches)")
PY
done
cat $S/wrstall_*.txt
