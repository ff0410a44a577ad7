#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
P="python3 tools/rsos_probe.py --only-fused --oracle 0 --warm 40 --reps 100 --f32 --seconds 600 --channels 8"
ms() { grep -o '"fused_ms": [0-9.]*' | cut -d' ' -f2; }
{
for dbg in 0 1 2 128 256 64 384; do
  echo "f32 mix debug=$dbg: $(SIGOPS_RSOS_DEBUG=$dbg $P 2>/dev/null | ms) ms   f64 products: $(SIGOPS_RSOS_NO_F32MFMA=1 SIGOPS_RSOS_DEBUG=$dbg $P 2>/dev/null | ms) ms"
done
for dbg in 0 1 128 256; do
  echo "f32 plain debug=$dbg: $(SIGOPS_RSOS_DEBUG=$dbg $P --plain 2>/dev/null | ms) ms   f64 products: $(SIGOPS_RSOS_NO_F32MFMA=1 SIGOPS_RSOS_DEBUG=$dbg $P --plain 2>/dev/null | ms) ms"
done
} > $O/f32m_ablation.txt 2>&1
cat $O/f32m_ablation.txt
