#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
C=$PWD/signaloperators.jl_amd/csrc
for v in main co0 nosaddr count r4; do
  echo "== $v"
  if [ $v = main ]; then timeout 600 python3 tools/r05/parity_loop.py 40 2>/dev/null | tail -4
  else SIGOPS_LIB=$C/libsigops_$v.so timeout 600 python3 tools/r05/parity_loop.py 40 2>/dev/null | tail -4; fi
done
