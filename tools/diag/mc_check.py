import sys, os, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import mc_identify as mi
cands = {
 "identified_rounded": [0.214, 0.091, -0.073, 0.0178, 0.011, 1.08, 0.0094, 0.0196, -0.023, 0.0],
 "identified_exact": [0.2137154891421651, 0.0906280729290419, -0.07305437756561722, 0.017793319311795293, 0.011122618393849742, 1.0776726753521668, 0.009386248353590046, 0.019560327788474392, -0.022988144869011698, 0.0],
 "rounded_mu1_toer_table": [0.214, 0.091, -0.073, 0.0175, 0.011, 1.0, 0.0094, 0.0196, -0.023, 0.0],
 "table_r02": [mi.PARAMS[k][0] for k in mi.NAMES],
}
for n, seed in ((1024, 1), (1024, 2), (256, 1)):
    p = mi.Probe(n, seed=seed)
    for name, th in cands.items():
        r = p.run(np.array(th), terms=True)
        print("n=%d seed=%d %-24s F=%.3f len=%.1f R=%.3f %s" % (n, seed, name, r["F"], r["len"], r["R"], r["reasons"]), flush=True)
    p.env.close()
