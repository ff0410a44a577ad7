cd $GRAFT_REPO_ROOT
for ch in 1 2 4; do
  secs=$((2267 / ch))
  for nw in 12 16; do
    SIGOPS_RSOS_NWAVES=$nw SIGOPS_RSOS_MINGROUPS=1 python3 tools/rsos_probe.py --seconds $secs --channels $ch --reps 5 --oracle 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ch', $ch, 'nwaves', $nw, 'fused', round(d['fused_ms'],3), 'two', round(d['two_kernel_ms'],3), 'rel', d['relerr_vs_two_kernel'])"
  done
done
