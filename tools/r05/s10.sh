#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_s10; mkdir -p $O
C=$PWD/signaloperators.jl_amd/csrc
probe() { timeout 300 python3 tools/rsos_probe.py --seconds 600 --only-fused --oracle 0 --warm ${WARM:-40} --reps ${REPS:-100} $EXTRA 2>$O/err_$1.txt | grep -o '"fused_ms": [0-9.]*'; }
echo "== main"; probe new; probe new2
echo "== wtol 70"; SIGOPS_RSOS_WTOL=70 probe w70
echo "== parity main"
timeout 900 python3 tools/r05/parity_loop.py 100 2>/dev/null | tail -4
python -m pytest tests/test_gpu_rsos.py tests/test_gpu_windows.py tests/test_gpu_configs.py tests/test_gpu_accumulator.py tests/test_gpu_soak.py tests/test_gpu_fuzz.py -q -m gpu -x 2>&1 | tail -5
