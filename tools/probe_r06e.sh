#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
P="python3 tools/rsos_probe.py --only-fused --warm 40 --reps 100 --oracle 0"
ms() { grep -o '"fused_ms": [0-9.]*' | cut -d' ' -f2 | tr '\n' ' '; }
for ch in 2 4; do
  sec=$((4800 / ch))
  for d in 0 131072; do
    echo "ch=$ch debug=$d mix: $(SIGOPS_RSOS_DEBUG=$d $P --seconds $sec --channels $ch 2>/dev/null | ms)  plain: $(SIGOPS_RSOS_DEBUG=$d $P --seconds $sec --channels $ch --plain 2>/dev/null | ms)  noalignpr mix: $(SIGOPS_RSOS_NOALIGNPR=1 SIGOPS_RSOS_DEBUG=$d $P --seconds $sec --channels $ch 2>/dev/null | ms)"
  done
done
