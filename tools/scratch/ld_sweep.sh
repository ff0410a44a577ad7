#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python3 tools/rsos_probe.py --seconds 600 --only-fused --reps 20 --warm 30 --oracle 0 $EXTRA 2>/dev/null | grep -o '"fused_ms": [0-9.]*'; }
run A=0
run SIGOPS_RSOS_DEBUG=2
run SIGOPS_RSOS_DEBUG=2048
run SIGOPS_RSOS_DEBUG=3
run SIGOPS_RSOS_DEBUG=74
run SIGOPS_RSOS_DEBUG=62
EXTRA=--plain
run A=plain
run SIGOPS_RSOS_DEBUG=224
run SIGOPS_RSOS_DEBUG=1
run SIGOPS_RSOS_DEBUG=72
