import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import mc_identify as mi
base = np.array([mi.PARAMS[k][0] for k in mi.NAMES])
star = np.array([0.214, 0.091, -0.073, 0.0175, 0.011, 1.0, 0.0094, 0.0196, -0.023, 0.0])
lo = np.array([mi.PARAMS[k][1] for k in mi.NAMES]); hi = np.array([mi.PARAMS[k][2] for k in mi.NAMES])
for seed in (1, 2):
    p = mi.Probe(1024, seed=seed)
    for f in (0.8, 0.9, 1.0, 1.1, 1.2, 1.3, 1.5):
        th = np.clip(base + f * (star - base), lo, hi)
        r = p.run(th, terms=True)
        print("seed=%d frac=%.1f F=%.3f len=%.1f R=%.3f theta=%s" % (seed, f, r["F"], r["len"], r["R"], np.round(th, 4).tolist()), flush=True)
    p.env.close()
