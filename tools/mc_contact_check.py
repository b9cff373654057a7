#!/usr/bin/env python3
"""Mini-cheetah (IN SAMPLE: its table was identified against this very policy in round 3) under the contact settings the Laikago
identification of round 5 settled on - soft toes, toe friction below 1 - and under a millimetre-scale contact margin (DESIGN.md section
7.3).  A consistency check, not a fit: nothing is chosen from it.  usage: python tools/mc_contact_check.py [--robots 1024]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--robots", type=int, default=1024)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "mc_contact_check.json"))
    args = ap.parse_args()
    import policy_probe
    rows = []
    soft = {"contact_stiffness": 25335.0, "contact_damping": 2110.9}
    for label, mo, co in (("shipped (rigid toes, mu 1, margin 0.02)", {}, {}),
                          ("mu 0.75", {"foot_friction": 0.75}, {}), ("mu 0.5", {"foot_friction": 0.5}, {}),
                          ("soft toes (25335, 2110.9), mu 1", dict(soft), {}), ("soft toes, mu 0.75", dict(soft, foot_friction=0.75), {}),
                          ("soft toes, mu 0.5", dict(soft, foot_friction=0.5), {}),
                          ("contact_margin 0.004", {}, {"contact_margin": 0.004}),
                          ("soft toes, mu 0.75, contact_margin 0.004", dict(soft, foot_friction=0.75), {"contact_margin": 0.004})):
        o = policy_probe.run("minicheetah_trot", "minicheetah_trot", "mini_cheetah", args.robots, 1, model_over=mo or None, config_over=co or None)
        rows.append({"setting": label, "finished": o["finished"], "len": o["len"], "reward_per_step": o["reward_per_step"], "reasons": o["reasons"]})
        print("%-45s finished %.3f  len %5.1f  r/step %.3f  %s" % (label, o["finished"], o["len"], o["reward_per_step"], o["reasons"]), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(rows, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
