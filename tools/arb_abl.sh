#!/bin/bash
# k_resample_arb: geometry sweep and ablations on the irrational-rate bench (SIGOPS_ARB_DEBUG bits: 1 nobody waits,
# 2 one tap pair per batch, 4 no loader; results are wrong with any bit set).
cd "$(dirname "$0")/.."
run() { echo -n "$* : "; env "$@" ONLY=arb timeout 120 python3 tools/bench_irrational.py 2>/dev/null | sed 's/.*"ms": \([0-9.]*\).*frac_of_8TBps": \([0-9.]*\).*/\1 ms  \2/'; }
for no in 2 4; do for depth in 2 3 4 6; do for nc in 4 6 7 8 11; do
  if [ $no = 4 ] && [ $nc -gt 7 ]; then continue; fi
  run SIGOPS_ARB_NO=$no SIGOPS_ARB_DEPTH=$depth SIGOPS_ARB_NC=$nc
done; done; done
run SIGOPS_ARB_DEBUG=5
run SIGOPS_ARB_DEBUG=3
run SIGOPS_ARB_DEBUG=7
run SIGOPS_ARB_NO=2 SIGOPS_ARB_DEBUG=5
run SIGOPS_ARB_NO=2 SIGOPS_ARB_DEBUG=3
