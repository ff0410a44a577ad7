#!/bin/bash
# AddressSanitizer + UBSan over the CPU builds: (1) the oracle (gcc), the whole
# `-m "not gpu"` suite; (2) the host side of libsigops -- planner / filter design / C-ABI -- built by
# hipcc's host pass with -fsanitize=address,undefined -fno-gpu-sanitize and linked against the
# normal kernel objects; the host-logic tests run against it (no device is touched: plan creation stops
# at "no HIP device", the design / position entry points run fully).  GPU ASan is not available
# on this pool.  Usage: bash tools/sanitize_cpu.sh | tee profiles/rNN/sanitizers.txt
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
T=/tmp/sigops_asan; rm -rf $T; mkdir -p $T
cd $R
echo "== oracle: gcc -fsanitize=address,undefined, pytest -m 'not gpu' =="
gcc -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -fPIC -std=gnu11 \
    -ffp-contract=off -shared -o $T/libsigops_oracle.so oracle/sigops_oracle.c -lm
ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
SIGOPS_ORACLE_SO=$T/libsigops_oracle.so python -m pytest tests -q -m "not gpu" -x -p no:cacheprovider 2>&1 | tail -3
echo "== libsigops host side: hipcc -fsanitize=address,undefined -fno-gpu-sanitize (planner.cpp stages.cpp accumulator.cpp executor.cpp design.cpp capi.cpp comm.cpp rtc.cpp) =="
C=signaloperators.jl_amd/csrc
for f in planner stages accumulator executor design capi comm rtc; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-gpu-sanitize \
      -fno-omit-frame-pointer -x hip -c $C/$f.cpp -o $T/$f.o 2>&1 | grep -v "warning\|^ \|^$" | head -5
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -o $T/libsigops.so $T/planner.o $T/stages.o $T/accumulator.o $T/executor.o $T/design.o $T/capi.o $T/comm.o $T/rtc.o $C/k_pointwise.o $C/k_sos.o $C/k_small.o $C/k_exact.o $C/k_resample_u*.o $C/k_rsos_ks*.o $C/k_resample_arb.o $C/kernels2.o -ldl
RT=$(/opt/rocm/lib/llvm/bin/clang --print-file-name=libclang_rt.asan-x86_64.so)
ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 LD_PRELOAD=$RT SIGOPS_LIB=$T/libsigops.so \
python -m pytest tests/test_design.py tests/test_host_api.py tests/test_resample_positions.py tests/test_oracle_dsp.py tests/test_randn_lowering.py \
    tests/test_oracle_golden.py tests/test_rtc.py tests/test_tf_filters.py tests/test_julia_glue_layout.py -q -m "not gpu" -k "not exports_nothing_else" -x -p no:cacheprovider 2>&1 | tail -3
