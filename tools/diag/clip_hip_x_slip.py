"""Kinematic check of the hips' lengthwise position against the clips (no physics, no policy; POST-HOC in round 6, nothing was chosen from it):
the clips were made by IK on the real URDF, so on a table with the right geometry the STANCE toes of a clip move least over the ground.
For straight gaits the measure is flat in hip_x (VERDICT r5 tried the mean over all clips); for the TURNING clips it is not: the toe's
path over the ground is the yaw rate times its distance from the turning centre.  Round 4's table with hip_x alone varied; stance = toe
sphere within `mm` of the ground; value = mean planar speed of the stance toes [m/s].  usage: python tools/diag/clip_hip_x_slip.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from openroborl_amd import config, motion, robots      # noqa: E402
from tests import oracle_lib as ol                      # noqa: E402


def slip(clip_name, model, stance_mm, samples=600):
    clip = motion.MotionClip(clip_name)
    cfg = config.make_config(1, mode="test", enable_randomizer=False, auto_reset=False)
    orc = ol.OracleEnv(cfg, [model, None, None, None], [clip], 1, robot_type=0, clip_id=0)
    lay = orc.lay
    st = orc.state[0].copy()
    dur = clip.frame_duration * (clip.num_frames - 1)
    ts = np.linspace(0.0, dur, samples + 1)
    toes = np.zeros((samples + 1, 4, 3))
    for i, t in enumerate(ts):
        f = np.zeros(19)
        orc.L.orc_clip_calc_frame(orc.h, 0, C.c_double(t), ol.P(f))
        s = st.copy()
        s[lay.sl("POS")] = f[:3]
        s[lay.sl("QUAT")] = f[3:7]
        s[lay.sl("Q")] = f[7:]
        out = np.zeros(34 * 3)
        masses = np.zeros(13)
        orc.L.orc_fk_probe(orc.h, ol.P(s), ol.P(out), ol.P(masses))
        toes[i] = out[26 * 3:].reshape(8, 3)[1::2]
    orc.close()
    dt = ts[1] - ts[0]
    stance = (toes[:-1, :, 2] - model["toe_radius"]) < stance_mm * 1e-3
    v = np.diff(toes[:, :, :2], axis=0) / dt
    return float((np.hypot(v[..., 0], v[..., 1]) * stance).sum() / max(stance.sum(), 1))


GRID = [0.15, 0.16, 0.17, 0.18, 0.19, 0.20, 0.21, 0.22, 0.23, 0.24, 0.25, 0.26, 0.27, 0.28]
print("hip_x [m]:                          " + "  ".join("%.2f " % x for x in GRID) + "  minimum at")
for clipn in ("laikago_turn", "laikago_spin", "laikago_pace", "laikago_trot", "laikago_inplace_steps"):
    for mm in (4.0, 6.0, 10.0):
        row = [slip(clipn, robots.laikago(**dict(robots.LAIKAGO_R04, hip_xy=[hx, 0.1157 - 0.032875])), mm) for hx in GRID]
        print("%-22s stance %2.0f mm:  " % (clipn, mm) + "  ".join("%.3f" % v for v in row) + "  %.2f" % GRID[int(np.argmin(row))])
print()
print("the same for the other two geometric entries of the identification, laikago_turn, stance 6 mm (hip_x 0.21):")
CX = [-0.03, -0.02, -0.01, 0.0, 0.01, 0.02, 0.03, 0.04, 0.05, 0.06]
row = [slip("laikago_turn", robots.laikago(**dict(robots.LAIKAGO_R04, com_x=cx)), 6.0) for cx in CX]
print("com_x [m] (root in front of the hips' centre)  " + "  ".join("%+.2f" % x for x in CX) + "   minimum at")
print("                                               " + "  ".join("%.3f" % v for v in row) + "   %+.2f" % CX[int(np.argmin(row))])
HY = [0.06, 0.07, 0.08, 0.082825, 0.09, 0.10, 0.11, 0.12]
row = [slip("laikago_turn", robots.laikago(**dict(robots.LAIKAGO_R04, hip_xy=[0.21, hy])), 6.0) for hy in HY]
print("hip_y [m]                                      " + "  ".join("%.3f" % x for x in HY) + "   minimum at")
print("                                               " + "  ".join("%.3f" % v for v in row) + "   %.3f" % HY[int(np.argmin(row))])
p6 = dict(robots.LAIKAGO_R04, **robots.laikago_theta_kwargs(robots.LAIKAGO_R06_P6_MOVED))
print("shipped table (P9: hip_x 0.21, com_x +0.06): %.3f;  round 6's first table (0.192, +0.058): %.3f;  round 4's (0.21, 0): %.3f;  round 5's (0.227, +0.021, hip_y 0.098): %.3f" % (
    slip("laikago_turn", robots.laikago(), 6.0), slip("laikago_turn", robots.laikago(**p6), 6.0), slip("laikago_turn", robots.laikago(**robots.LAIKAGO_R04), 6.0),
    slip("laikago_turn", robots.laikago(**robots.LAIKAGO_R05), 6.0)))
