"""Evaluate sub-trees of one soak tree on both sides to find where a difference starts.
python tools/soak_subtrees.py a SEED I"""
import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import sigops_amd as so
from oracle_bridge import oracle_sink, relerr
_argv = sys.argv[:]; sys.argv = [sys.argv[0]]
import importlib.util
spec = importlib.util.spec_from_file_location("soak_repro_mod", "tools/soak_repro.py")
src = open("tools/soak_repro.py").read().split("args = [a for a in")[0]
ns = {}
exec(compile(src, "soak_repro_head", "exec"), ns)
tree, info = ns["gen"](_argv[1], int(_argv[2]), int(_argv[3]))
seen = set()
def walk(s, depth=0):
    if id(s) in seen: return
    seen.add(id(s))
    for c in (getattr(s, "signals", None) or ([s.signal] if hasattr(s, "signal") else [])):
        walk(c, depth + 1)
    n = so.nframes(s)
    if n is None or so.signals.isknowninf(n) or n == 0: return
    try:
        w = oracle_sink(s); g = so.sink(s, so.Array)
        d = np.abs(g.astype(float) - w.astype(float)); bad = np.argwhere(d > 1e-7 * max(1.0, float(np.abs(w).max())))
        print(type(s).__name__, getattr(s, "kind", ""), "n=%d nch=%d" % (n, s.nch), "relerr %.3g maxabs %.3g |w|max %.3g" % (relerr(g, w), d.max(), np.abs(w).max()), "bad frames", (bad[:2].tolist(), bad[-1:].tolist(), len(bad)) if len(bad) else "-")
    except Exception as e:
        print(type(s).__name__, "n=%s" % n, "ERR", str(e)[:80])
walk(tree)
