#!/bin/bash
# Round 6, VERDICT item 2(i): where do two-channel signals lose their time in k_rsos?  The same number of samples as 8 channels
# (4 x longer), plain and with the fused Mix, with and without result stores (ablation bit 1), with other pitches and range
# counts -- then the write-side counters of both shapes side by side.  Output: gpurun_out/stereo/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/stereo
mkdir -p $O
cd $R
P="python3 tools/rsos_probe.py --only-fused --oracle 0 --warm 40 --reps 100"
ms() { grep -o '"fused_ms": [0-9.]*' | cut -d' ' -f2; }
{
for ch in 8 2 4; do
  sec=$((4800 / ch))
  for plain in "" "--plain"; do
    for dbg in 0 1; do
      echo "ch=$ch sec=$sec ${plain:-mix} debug=$dbg: $(SIGOPS_RSOS_DEBUG=$dbg $P --seconds $sec --channels $ch $plain 2>/dev/null | ms) ms"
    done
  done
done
# pitches: the result's / the input's channel rows 64, 32+, 16+ elements further apart
for pad in "--pad-out 32" "--pad-out 16" "--pad-out 528" "--pad-in 32" "--pad-in 528 --pad-out 528"; do
  echo "ch=2 plain $pad: $($P --seconds 2400 --channels 2 --plain $pad 2>/dev/null | ms) ms"
  echo "ch=8 plain $pad: $($P --seconds 600 --channels 8 --plain $pad 2>/dev/null | ms) ms"
done
# other range counts (ranges per channel; default: 256 groups x 16 rows / channels)
for rg in 512 1024 4096; do
  echo "ch=2 plain ranges=$rg: $(SIGOPS_RSOS_RANGES=$rg $P --seconds 2400 --channels 2 --plain 2>/dev/null | ms) ms"
done
for rg in 256 1024; do
  echo "ch=8 plain ranges=$rg: $(SIGOPS_RSOS_RANGES=$rg $P --seconds 600 --channels 8 --plain 2>/dev/null | ms) ms"
done
} > $O/times.txt 2>&1
cat $O/times.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_avail.txt 2>&1
for ch in 8 2; do
  sec=$((4800 / ch))
  for set in "WRITE_SIZE TCC_EA0_WRREQ_STALL_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WR_UNCACHED_32B_sum" "TCC_REQ_sum TCC_WRITE_sum TCC_WRITEBACK_sum TCC_HIT_sum" "TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"; do
    tag=$(echo $set | tr ' ' '+' | cut -c1-60)
    rm -rf $O/pmc_${ch}_$tag
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_${ch}_$tag -- python3 $R/tools/rsos_probe.py --only-fused --oracle 0 --warm 5 --reps 10 --plain --seconds $sec --channels $ch > $O/pmc_${ch}_$tag.log 2>&1
  done
done
python3 - <<'PY' > $O/pmc_summary.txt 2>&1
import csv, glob, os, collections
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/stereo")
for d in sorted(glob.glob(O + "/pmc_*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if "k_rsos" not in r.get("Kernel_Name", ""): continue
            a = acc[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
        for k, (v, n) in sorted(acc.items()):
            print(os.path.basename(d)[:8], k, "per launch", v / max(n, 1), "launches", n)
PY
cat $O/pmc_summary.txt
find $O -name "*.csv" -size +2M -delete
ls $O | head -50
