cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
python3 bench.py --steps 100 --warmup 30 --cpu-seconds 0 --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('f64', d['ms_per_step'], d.get('steady_state_ms'))"
done
python3 bench.py --dtype f32 --steps 100 --warmup 30 --cpu-seconds 0 --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('f32', d['ms_per_step'], d.get('steady_state_ms'))"
