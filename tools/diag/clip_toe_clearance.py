"""Kinematic check of a Laikago table against the clips (no physics, no policy): the clips were made by IK on the real URDF, so in every frame
the lowest toe of a correct table touches the ground.  usage: python tools/diag/clip_toe_clearance.py"""
import os, sys, numpy as np, ctypes as C
sys.path.insert(0, '.')
from openroborl_amd import _abi, config, motion, robots
from tests import oracle_lib as ol
P = ol.P
for name, table in (("round-4 table", robots.LAIKAGO_R04), ("search winner (hip plane at -0.068)", {"hip_z": -0.068136}), ("shipped table", {})):
    print(name)
    for clipn in ("laikago_pace", "laikago_trot", "laikago_spin", "laikago_inplace_steps", "laikago_turn"):
        clip = motion.MotionClip(clipn)
        cfg = config.make_config(1, mode="test", enable_randomizer=False, auto_reset=False)
        m = robots.laikago(**table)
        orc = ol.OracleEnv(cfg, [m, None, None, None], [clip], 1, robot_type=0, clip_id=0)
        lay = orc.lay
        st = orc.state[0].copy()
        lows = []
        for f in clip.frames:
            s = st.copy(); s[lay.sl("POS")] = f[:3]; s[lay.sl("QUAT")] = f[3:7]; s[lay.sl("Q")] = f[7:]
            out = np.zeros(34*3); masses = np.zeros(13)
            orc.L.orc_fk_probe(orc.h, P(s), P(out), P(masses))
            toes = out[26*3:].reshape(8,3)[1::2, 2] - m["toe_radius"]
            lows.append(np.sort(toes)[:2])
        lows = np.array(lows)
        print("   %-24s lowest toe clearance per frame: median %+.1f mm [min %+.1f, max %+.1f]; second lowest: median %+.1f mm" % (
            clipn, 1000*np.median(lows[:,0]), 1000*lows[:,0].min(), 1000*lows[:,0].max(), 1000*np.median(lows[:,1])))
        orc.close()
