for sub in 1 2 4 8 16; do echo "sub=$sub"; SIGOPS_K1_SUB=$sub python bench_configs.py --only k1 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('  ', r['config'][:60], round(r['ms'],4), round(r['algorithmic_GBps']))"; done
