import sys, os
sys.path.insert(0,'tests'); sys.path.insert(0,'.'); sys.path.insert(0,'tools')
import numpy as np, sigops_amd as so
from oracle_bridge import oracle_sink, relerr
_a = sys.argv[:]; sys.argv = [sys.argv[0]]
src = open("tools/soak_repro.py").read().split("args = [a for a in")[0]
ns = {}; exec(compile(src, "h", "exec"), ns)
tree, info = ns["gen"]('a', 1396, 7)
def find(s):
    if type(s).__name__ == 'CutApply' and s.kind == 'after' and type(s.signal).__name__ == 'FilteredSignal': return s
    for c in (getattr(s, "signals", None) or ([s.signal] if hasattr(s, "signal") else [])):
        r = find(c)
        if r is not None: return r
a = find(tree); f = a.signal
def rng_(d, tol=1e-12):
    b = np.argwhere(d.max(axis=1) > tol).ravel(); return (int(b[0]), int(b[-1]), len(b)) if len(b) else None
order = _a[1] if len(_a) > 1 else "of,ef,oa,ea"
R = {}
for step in order.split(','):
    if step == 'of': R['of'] = oracle_sink(f)
    if step == 'ef': R['ef'] = so.sink(f, so.Array)
    if step == 'oa': R['oa'] = oracle_sink(a)
    if step == 'ea': R['ea'] = so.sink(a, so.Array)
    if step == 'ou': R['ou'] = oracle_sink(f.signal)
    if step == 'eu': R['eu'] = so.sink(f.signal, so.Array)
k = 543
print(order, '| engine a vs f[k:]', rng_(np.abs(R['ea'] - R['ef'][k:])), '| oracle a vs f[k:]', rng_(np.abs(R['oa'] - R['of'][k:])), '| f e/o', rng_(np.abs(R['ef'] - R['of'])), '| a e/o', rng_(np.abs(R['ea'] - R['oa'])))
