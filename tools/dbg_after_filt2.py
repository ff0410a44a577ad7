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
ef, ea, of, oa = so.sink(f, so.Array), so.sink(a, so.Array), oracle_sink(f), oracle_sink(a)
k = ef.shape[0] - ea.shape[0]
def rng_(d, tol=1e-12):
    b = np.argwhere(d.max(axis=1) > tol).ravel(); return (int(b[0]), int(b[-1]), len(b)) if len(b) else None
print('k', k, 'engine a vs f[k:]', rng_(np.abs(ea - ef[k:])), '| oracle a vs f[k:]', rng_(np.abs(oa - of[k:])), '| engine f vs oracle f', rng_(np.abs(ef - of)), '| engine a vs oracle a', rng_(np.abs(ea - oa)))
print('time of after:', a.time, 'fs', f.fs, 'child len', so.nframes(f.signal), type(f.signal).__name__, f.signal.kind if hasattr(f.signal,'kind') else '')
