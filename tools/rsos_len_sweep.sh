cd $GRAFT_REPO_ROOT
for s in 20 30 46 60 75 120 200; do
  SIGOPS_RSOS_MINGROUPS=1 python3 tools/rsos_probe.py --seconds $s --reps 50 --oracle 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print($s, 'fused', round(d['fused_ms'],4), d['fused_steps'], 'two', round(d['two_kernel_ms'],4), 'rel', d['relerr_vs_two_kernel'])"
done
python3 tools/rsos_probe.py --seconds 40 --reps 20 --oracle 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('default policy 40 s:', d['fused_steps'])"
python3 tools/rsos_probe.py --seconds 60 --reps 20 --oracle 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('default policy 60 s:', d['fused_steps'])"
