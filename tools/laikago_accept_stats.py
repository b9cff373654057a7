#!/usr/bin/env python3
"""What does acceptance hang on?  Statistics over ALL candidates of a fit-set-only survey run of tools/laikago_identify.py
(`--no-holdout --dump-all file.jsonl`; the hold-out policies are never involved).

For every varied entry: the acceptance rate (F >= 0.8 on both fit policies) of the candidates in the lower / middle / upper third of the
entry's interval, and for every switch the rate with it on / off - separately for the RANDOM stage (uniform in the box + clouds around
round 4's table: an unbiased look at the box) and for the LOCAL stage (children of the best candidates: a look at the basin the search
settled in).  A continuous entry "matters" where the three rates differ by much more than their binomial noise.

usage: python tools/laikago_accept_stats.py survey.jsonl [> profiles/r05_laikago_accept_stats.txt]
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import laikago_identify as li      # noqa: E402


def main():
    rows = [json.loads(ln) for ln in open(sys.argv[1])]
    acc = np.array([all(r["F"][p] >= 0.8 for p in r["F"]) for r in rows])
    score = np.array([min(r["F"].values()) for r in rows])
    stage = np.array([r["stage"] for r in rows])
    print("%d candidates: %d random (%d accepted, best fit score %.3f), %d local (%d accepted)" % (
        len(rows), (stage == "random").sum(), acc[stage == "random"].sum(), score[stage == "random"].max(), (stage == "local").sum(), acc[stage == "local"].sum()))
    for st in ("random", "local"):
        m = stage == st
        if not m.any():
            continue
        # random stage: hardly anything is accepted, so the statistic there is the mean fit score (min over the two policies of F)
        stat = score if st == "random" else acc.astype(float)
        what = "mean fit score" if st == "random" else "acceptance rate"
        print("\n== %s stage (%d candidates): %s by thirds of each entry's interval (low | mid | high), spread = max - min" % (st, m.sum(), what))
        out = []
        for k in li.NAMES:
            _, lo, hi, _ = li.PARAMS[k]
            v = np.array([r["theta"][k] for r in rows])[m]
            t = np.clip(((v - lo) / (hi - lo) * 3).astype(int), 0, 2)
            r3 = [float(stat[m][t == i].mean()) if (t == i).any() else float("nan") for i in range(3)]
            n3 = [int((t == i).sum()) for i in range(3)]
            out.append((np.nanmax(r3) - np.nanmin(r3), k, r3, n3))
        for spread, k, r3, n3 in sorted(out, reverse=True):
            print("  %-18s %.3f | %.3f | %.3f   (n %4d %4d %4d)   spread %.3f" % (k, r3[0], r3[1], r3[2], n3[0], n3[1], n3[2], spread))
        for k in li.SWITCHES:
            v = np.array([int(r["theta"][k]) for r in rows])[m]
            on, off = stat[m][v == 1], stat[m][v == 0]
            print("  switch %-11s on %.3f (n %d) | off %.3f (n %d)" % (k, on.mean() if len(on) else float("nan"), len(on), off.mean() if len(off) else float("nan"), len(off)))
    a = [r for r, ok in zip(rows, acc) if ok]
    if a:
        print("\n== the accepted candidates (%d): median [5 %% .. 95 %%] of each entry, round 4's value, interval" % len(a))
        for k in li.NAMES:
            v0, lo, hi, _ = li.PARAMS[k]
            v = np.array([r["theta"][k] for r in a])
            print("  %-18s %9.4g [%9.4g .. %9.4g]   round 4: %9.4g   box %g .. %g" % (k, np.median(v), np.percentile(v, 5), np.percentile(v, 95), v0, lo, hi))
        for k in li.SWITCHES:
            print("  switch %-11s on in %.3f of the accepted" % (k, np.mean([int(r["theta"][k]) for r in a])))
        sk = np.array([r["theta"]["soft_k"] for r in a if r["theta"]["soft"]])
        sd = np.array([r["theta"]["soft_d"] for r in a if r["theta"]["soft"]])
        if len(sk):
            print("  soft toes: stiffness median %.0f [%.0f .. %.0f] N/m, damping median %.0f [%.0f .. %.0f] N s/m" % (
                np.median(sk), np.percentile(sk, 5), np.percentile(sk, 95), np.median(sd), np.percentile(sd, 5), np.percentile(sd, 95)))


if __name__ == "__main__":
    main()
