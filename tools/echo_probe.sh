#!/bin/bash
# Measurement aid (round 3): what does reading the resampler's freshly stored tiles back through L2 / the
# Infinity Cache and storing them again cost inside k_resample_periodic?  SIGOPS_RS_DEBUG bits 8..15 = lag.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/echo; mkdir -p $O
cd $R
for LAG in 0 1 2 4 8 16 32; do
  SIGOPS_RS_DEBUG=$((LAG*256)) python3 bench.py --workload config3 --steps 150 --warmup 40 --cpu-seconds 0 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lag $LAG ms_per_step %.4f kernel_ms %.4f'%(d['ms_per_step'], d['roofline']['kernel_ms']))" >> $O/times.txt
done
cd /tmp && export TMPDIR=/tmp
for LAG in 0 2 16; do
  for C in FETCH_SIZE WRITE_SIZE; do
    SIGOPS_RS_DEBUG=$((LAG*256)) timeout 300 rocprofv3 --pmc $C -d $O/pmc_${LAG}_$C -o b --output-format csv -- python3 $R/bench.py --workload config3 --steps 30 --warmup 5 --cpu-seconds 0 > $O/pmc_${LAG}_$C.log 2>&1
    python3 - <<P >> $O/pmc.txt
import csv,glob
fs=glob.glob('$O/pmc_${LAG}_$C/**/*counter_collection.csv',recursive=True)
tot=0;n=0
for f in fs:
    for r in csv.DictReader(open(f)):
        if r.get('Counter_Name')=='$C' and 'k_resample_periodic' in r.get('Kernel_Name',''):
            tot+=float(r['Counter_Value']); n+=1
print('lag $LAG $C mean_KB', tot/max(n,1), 'n', n)
P
    rm -rf $O/pmc_${LAG}_$C
  done
done
cat $O/times.txt $O/pmc.txt
