"""Debug helper: run every shared case on the GPU in a fresh subprocess each (a GPU fault
kills the process) and report pass/fail + error."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import numpy as np
    import sigops_amd as so
    from cases import CASES
    from oracle_bridge import oracle_sink, relerr
    name = sys.argv[2]
    x = CASES[name]()
    want = oracle_sink(x)
    got = so.sink(x, so.Array)
    exact = np.array_equal(got, want)
    print(f"RESULT {name} shape={got.shape} exact={exact} relerr={relerr(got, want):.3e}")
    sys.exit(0)

from cases import CASES
names = sys.argv[1:] or sorted(CASES)
for n in names:
    p = subprocess.run([sys.executable, __file__, "--one", n], capture_output=True, text=True, timeout=300)
    out = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
    if p.returncode == 0 and out:
        print(out[0], flush=True)
    else:
        tail = (p.stderr.strip().splitlines() or ["?"])
        msg = [l for l in tail if "Error" in l or "fault" in l.lower() or "Abort" in l][-3:]
        print(f"FAIL {n} rc={p.returncode} :: {' | '.join(msg) or tail[-1]}", flush=True)
