#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
P="python3 tools/rsos_probe.py --only-fused --warm 40 --reps 100"
ms() { grep -o '"fused_ms": [0-9.]*\|"relerr_vs_oracle_prefix": [0-9.e-]*' | cut -d' ' -f2 | tr '\n' ' '; }
{
for ch in 2 4 8; do
  sec=$((4800 / ch))
  echo "ch=$ch mix: $($P --seconds $sec --channels $ch --oracle 100000 2>/dev/null | ms)   plain: $($P --seconds $sec --channels $ch --plain --oracle 100000 2>/dev/null | ms)"
done
echo "ch=2 mix, step in the loaders (SIGOPS_RSOS_NOGSPLIT): $(SIGOPS_RSOS_NOGSPLIT=1 $P --seconds 2400 --channels 2 --oracle 0 2>/dev/null | ms)"
echo "ch=2 mix nwaves=12: $(SIGOPS_RSOS_NWAVES=12 $P --seconds 2400 --channels 2 --oracle 0 2>/dev/null | ms)"
echo "ch=2 mix chunk 64: $(SIGOPS_RSOS_CHUNK=64 $P --seconds 2400 --channels 2 --oracle 0 2>/dev/null | ms)"
echo "ch=2 mix depth 2: $(SIGOPS_RSOS_DEPTH=2 $P --seconds 2400 --channels 2 --oracle 0 2>/dev/null | ms)"
echo "ch=2 mix nwaves=8: $(SIGOPS_RSOS_NWAVES=8 $P --seconds 2400 --channels 2 --oracle 0 2>/dev/null | ms)"
echo "ch=4 mix nwaves=16 (step waves): $(SIGOPS_RSOS_NWAVES=16 $P --seconds 1200 --channels 4 --oracle 100000 2>/dev/null | ms)"
echo "ch=4 mix nwaves=16, step in the loaders: $(SIGOPS_RSOS_NOGSPLIT=1 SIGOPS_RSOS_NWAVES=16 $P --seconds 1200 --channels 4 --oracle 0 2>/dev/null | ms)"
} > $O/few_channel_mix.txt 2>&1
cat $O/few_channel_mix.txt
