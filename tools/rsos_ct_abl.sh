cd $GRAFT_REPO_ROOT
for ch in 1 2 4 8; do
  secs=$((2267 / ch))
  for dbg in 0 224 344 452; do
    SIGOPS_RSOS_MINGROUPS=1 SIGOPS_RSOS_DEBUG=$dbg python3 tools/rsos_probe.py --seconds $secs --channels $ch --reps 5 --oracle 0 --only-fused 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ch', $ch, 'debug', $dbg, round(d['fused_ms'],3), d['fused_steps'])"
  done
done
