#!/bin/bash
# A/B builds of one kernel translation unit: tools/build_variant.sh NAME UNIT.hip [extra hipcc flags...]
# -> signaloperators.jl_amd/csrc/libsigops_NAME.so (git-ignored; SIGOPS_LIB=<that path> selects it at run time).
# The other objects are the ones build.py left next to the sources.
set -eu
name=$1; unit=$2; shift 2
here=$(cd "$(dirname "$0")/../signaloperators.jl_amd/csrc" && pwd)
obj=$here/${unit%.hip}_$name.o
if [ -n "${RELINK_ONLY:-}" ] && [ -f "$obj" ]; then :; else
/opt/rocm/bin/hipcc -mllvm -disable-machine-licm --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall \
    -Wno-unused-function "$@" -x hip -c "$here/$unit" -o "$obj"
fi
objs=""
# (k_rsos.hip and k_resample.hip are built as several units, k_rsos_ks*.o / k_resample_u*.o; as a variant each is ONE unit with everything in it)
for s in k_pointwise k_sos k_small k_exact k_resample k_rsos k_resample_arb kernels2 planner stages accumulator executor design capi comm rtc; do
    if [ "$s.hip" = "$unit" ]; then objs="$objs $obj"
    elif [ "$s" = k_rsos ]; then objs="$objs $(ls $here/k_rsos_ks*.o | tr '\n' ' ')"
    elif [ "$s" = k_resample ]; then objs="$objs $(ls $here/k_resample_u*.o | tr '\n' ' ')"
    else objs="$objs $here/$s.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$here/exports.map -o "$here/libsigops_$name.so" $objs -ldl
echo "$here/libsigops_$name.so"
