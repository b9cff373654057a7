"""EXPLORATION after round 6's protocol had run (in sample on all four policies; a RECORD, nothing ships from it).  The turning clip pins
the three geometric entries of the identification at round 4's values (hip_x 0.21, hip_y 0.0828, com_x 0: tools/diag/clip_hip_x_slip.py)
while the round's first table (P6 + P7) has hip_x 0.192, com_x +0.058, and putting either back ALONE costs the policies their walk (P7:
profiles/r06_laikago_minimal.txt).  Put back TOGETHER?  (front hips: shipped 0.134, com_x alone back 0.192, both back 0.21 m ahead of the
base COM.)  Also recorded, because it costs nothing: the trunk's inertia with its axes as an unpermuted y-up URDF would give them (pitch
0.0733 instead of 0.2507) - a hypothesis for the missing pitch behaviour that the policies reject.
usage (GPU box): python tools/diag/geometry_pinned_probe.py [robots]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import policy_probe
from openroborl_amd import robots

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ship = dict(robots.LAIKAGO_R04, **robots.laikago_theta_kwargs(robots.LAIKAGO_R06_P6_MOVED))     # round 6's FIRST table (P6 + P7), on which this probe was made
geo = dict(hip_xy=[0.21, 0.1157 - 0.032875], com_x=0.0)
I = (0.073348887, 0.250684593, 0.254469458)
VARIANTS = [
    ("round 6's first table (hip_x 0.192, com_x +0.058)", {}),
    ("com_x alone back to 0", dict(com_x=0.0)),
    ("hip_x alone back to 0.21", dict(hip_xy=[0.21, 0.1157 - 0.032875])),
    ("BOTH back: the geometry the turning clip pins (hip_x 0.21, com_x 0)", geo),
    ("  + trunk inertia (roll, pitch, yaw) = (0.2545, 0.0733, 0.2507): axes of an unpermuted y-up URDF", dict(geo, base_inertia=[I[2], I[0], I[1]])),
    ("  + pitch inertia alone 0.0733", dict(geo, base_inertia=[1.4195 * I[0], I[0], 1.4195 * I[2]])),
    ("round 4's table + soft toes (10 kN/m, 745 N s/m) + toe friction 0.53 only", dict(robots.LAIKAGO_R04, contact_stiffness=10000.0, contact_damping=744.99, foot_friction=0.53185)),
]
pols = [("laikago_pace", "laikago_pace"), ("laikago_spin", "laikago_spin"), ("laikago_trot", "laikago_trot"), ("laikago_trot0", "laikago_trot")]
for name, over in VARIANTS:
    cells = []
    for pol, clip in pols:
        o = policy_probe.run(pol, clip, "laikago", n, 1, model_over={"_build": dict(ship, **over)})
        cells.append("%s %.2f / %.2f" % (pol.replace("laikago_", ""), o["finished"], o["return_per_nominal_step"]))
    print("%-100s %s" % (name, "   ".join(cells)), flush=True)
