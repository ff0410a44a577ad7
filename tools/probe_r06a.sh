#!/bin/bash
# round 6, GPU session 2: stereo after the spill fix; the Float32-MFMA form of k_rsos
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_rsos.py -x -q -m gpu -k "wait_that_does_not_end" 2>&1 | tail -40 > $O/test_wait.txt
python -m pytest tests/test_gpu_rsos_f32m.py -x -q -m gpu 2>&1 | tail -30 > $O/test_f32m.txt
P="python3 tools/rsos_probe.py --only-fused --oracle 0 --warm 40 --reps 100"
ms() { grep -o '"fused_ms": [0-9.]*' | cut -d' ' -f2; }
{
for ch in 2 8; do
  sec=$((4800 / ch))
  for plain in "" "--plain"; do
    for dbg in 0 1; do
      echo "ch=$ch sec=$sec ${plain:-mix} debug=$dbg: $(SIGOPS_RSOS_DEBUG=$dbg $P --seconds $sec --channels $ch $plain 2>/dev/null | ms) ms"
    done
  done
done
echo "ch=2 plain nwaves=12: $(SIGOPS_RSOS_NWAVES=12 $P --seconds 2400 --channels 2 --plain 2>/dev/null | ms) ms"
echo "ch=2 mix nwaves=12: $(SIGOPS_RSOS_NWAVES=12 $P --seconds 2400 --channels 2 2>/dev/null | ms) ms"
} > $O/stereo_after_fix.txt 2>&1
python3 bench.py --dtype f32 --steps 200 --warmup 30 --cpu-seconds 0 --no-one-shot > $O/bench_ns_f32.json 2>$O/bench_ns_f32.err
SIGOPS_RSOS_NO_F32MFMA=1 python3 bench.py --dtype f32 --steps 200 --warmup 30 --cpu-seconds 0 --no-one-shot > $O/bench_ns_f32_f64products.json 2>/dev/null
python3 bench.py --steps 20 --warmup 5 > $O/bench_20_5.json 2>$O/bench_20_5.err
