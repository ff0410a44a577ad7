import sys, os
sys.path.insert(0,'tests'); sys.path.insert(0,'.'); sys.path.insert(0,'tools')
import numpy as np, sigops_amd as so
from oracle_bridge import oracle_sink, relerr
rng = np.random.default_rng(5)
x0 = np.asfortranarray(rng.standard_normal((294, 3)))
for name, src in (("pad-one", so.Pad(so.Signal(x0, 50 * so.Hz), so.one) | so.Until(1038 * so.frames)),
                  ("plain", so.Signal(np.asfortranarray(rng.standard_normal((1038, 3))), 50 * so.Hz))):
    f = src | so.Filt(so.Lowpass, 5.6156 * so.Hz)
    wf = so.sink(f, so.Array)
    print(name, 'filt vs oracle', relerr(wf, oracle_sink(f)))
    for k in (100, 500, 543, 600):
        for env in (None, "SIGOPS_NO_WINDOW_ALIAS"):
            if env: os.environ[env] = "1"
            w = so.sink(f | so.After(k * so.frames), so.Array)
            n = so.sink(f | so.After(k * so.frames) | so.Normpower, so.Array)
            if env: os.environ.pop(env)
            d = np.abs(w - wf[k:]).max(axis=1); b = np.argwhere(d > 1e-12).ravel()
            ref = wf[k:] / np.sqrt(np.mean(wf[k:] ** 2))
            d2 = np.abs(n - ref).max(axis=1); b2 = np.argwhere(d2 > 1e-9).ravel()
            print('  After', k, env, 'max diff %.3g' % d.max(), 'bad', (b[0] + k, b[-1] + k) if len(b) else None,
                  '| normpower: %.3g' % d2.max(), (b2[0] + k, b2[-1] + k) if len(b2) else None)
