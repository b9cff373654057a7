"""All five shipped policies on the shipped tables under config.PYBULLET_REMEMBERED (the solver constants PyBullet is remembered to set:
erp 0.08, warm start 0.1, contact margin 0.004) next to the shipped library defaults.  A record, nothing is chosen from it.
usage: python tools/diag/remembered_constants_check.py [robots]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import policy_probe
from openroborl_amd import config
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for pol, clip, robot, named in policy_probe.policy_table():
    if not named:
        continue
    for label, co in (("library defaults (shipped)", None), ("PYBULLET_REMEMBERED", dict(config.PYBULLET_REMEMBERED))):
        o = policy_probe.run(pol, clip, robot, n, 1, config_over=co)
        print("%-17s %-28s finished %.3f  len %5.1f  r/step %.3f  %s" % (pol, label, o["finished"], o["len"], o["reward_per_step"], o["reasons"]), flush=True)
