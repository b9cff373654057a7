#!/usr/bin/env python3
"""Where the accepted candidates of each tools/identify_r6.py run live (CPU; reads the --dump-all files): per run the top decile by min-J of the
accepted candidates - median [5 % .. 95 %] of every varied entry, the share with each switched feature on - and which medians sit on an
edge of the box.  usage: python tools/cv_accept_stats.py gpurun_out/r06cv/split*_candidates.jsonl.gz [...]"""
import gzip
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import identify_r6 as ir      # noqa: E402

for path in sys.argv[1:]:
    c = [json.loads(l) for l in gzip.open(path, "rt")]
    fit = list(c[0]["fit"])
    robot = "mini_cheetah" if fit[0].startswith("minicheetah") else "laikago"
    spec = ir.spec_of(robot, "hip_x" not in c[0]["theta"], "com_x" not in c[0]["theta"])
    acc = [x for x in c if min(x["fit"][p]["F"] for p in fit) >= spec["accept"]]
    if not acc:
        print("%s: %d candidates, none accepted" % (os.path.basename(path), len(c)))
        continue
    J = np.array([min(x["fit"][p]["J"] for p in fit) for x in acc])
    top = [x for x, j in zip(acc, J) if j >= np.percentile(J, 90)]
    print("%s  fit %s: %d candidates, %d accepted; top decile of the accepted by min-J (%d candidates, min-J >= %.3f, best %.3f):" % (
        os.path.basename(path), " + ".join(fit), len(c), len(acc), len(top), np.percentile(J, 90), J.max()))
    edge = []
    for k, (v0, lo, hi) in spec["params"].items():
        v = np.array([x["theta"][k] for x in top])
        med = float(np.median(v))
        at = "  <- lower edge" if med < lo + 0.05 * (hi - lo) else ("  <- upper edge" if med > hi - 0.05 * (hi - lo) else "")
        if at:
            edge.append(k)
        print("   %-14s %9.4g  [%9.4g .. %9.4g]   box [%g, %g], reference point %g%s" % (k, med, np.percentile(v, 5), np.percentile(v, 95), lo, hi, v0, at))
    sw = {k: float(np.mean([x["theta"][k] for x in top])) for k in spec["switches"]}
    soft = [x for x in top if x["theta"]["soft"]]
    print("   switched on: " + ", ".join("%s %.2f" % kv for kv in sw.items()) +
          ("; soft toes k %.3g [%.3g .. %.3g] N/m, d %.3g [%.3g .. %.3g] N s/m" % (
              np.median([x["theta"]["soft_k"] for x in soft]), np.percentile([x["theta"]["soft_k"] for x in soft], 5), np.percentile([x["theta"]["soft_k"] for x in soft], 95),
              np.median([x["theta"]["soft_d"] for x in soft]), np.percentile([x["theta"]["soft_d"] for x in soft], 5), np.percentile([x["theta"]["soft_d"] for x in soft], 95)) if soft else ""))
    print("   medians on an edge of the box: %s" % (", ".join(edge) or "none"))
