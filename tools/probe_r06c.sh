#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
P="python3 tools/rsos_probe.py --only-fused --oracle 0 --warm 40 --reps 100 --seconds 600 --channels 8"
ms() { grep -o '"fused_ms": [0-9.]*' | cut -d' ' -f2; }
{
for dbg in 0 2; do
  echo "f32 mix debug=$dbg: $(SIGOPS_RSOS_DEBUG=$dbg $P --f32 2>/dev/null | ms) ms"
  echo "f64 mix debug=$dbg: $(SIGOPS_RSOS_DEBUG=$dbg $P 2>/dev/null | ms) ms"
done
python -m pytest tests/test_gpu_rsos_f32m.py -x -q -m gpu 2>&1 | tail -3
} > $O/f32m_ablation2.txt 2>&1
cat $O/f32m_ablation2.txt
