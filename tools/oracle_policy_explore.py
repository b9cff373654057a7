#!/usr/bin/env python3
"""The shipped Laikago policies on the float64 CPU ORACLE under variations of the contact model / table (no GPU).

Exploration aid for SURVEY 8a row C (round 4: HISTORY.md section 7c): which of Bullet's toe-contact features - friction anchor, spinning friction,
contact stiffness / damping, the URDF's lateral friction, PyBullet's solver constants - change the fate of the PyBullet-trained policies
on this engine.  HOLD-OUT RULE (fixed in round 5 before anything was run): only `laikago_trot` and `laikago_spin` may be looked at while
features or table entries are chosen; `laikago_trot0` and `laikago_pace` are evaluated once, at the end, on the chosen candidate
(--holdout).  Test-mode protocol of run.py:151-183 (no randomiser, 2 ms latency, deterministic actions), like tools/policy_probe.py.

usage: python tools/oracle_policy_explore.py [--robots 48] [--steps 600] [--set name=value ...] [--holdout]
  names: any keyword of robots._build (toe_m, foot_friction, spinning_friction, friction_anchor, contact_stiffness ...) or any float /
  int field of orr_config (contact_erp, friction_erp, warmstart_factor, contact_margin, solver_iters ...)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from openroborl_amd import _abi, config, motion, robots      # noqa: E402
from tests import oracle_lib as ol                             # noqa: E402

FIT = [("laikago_trot", "laikago_trot"), ("laikago_spin", "laikago_spin")]
HOLDOUT = [("laikago_trot0", "laikago_trot"), ("laikago_pace", "laikago_pace")]
CFG_FIELDS = {n for n, _ in _abi.OrrConfig._fields_}


def split_overrides(over):
    build, cfg = {}, {}
    for k, v in over.items():
        (cfg if k in CFG_FIELDS else build)[k] = v
    return build, cfg


def run(policy, clip_name, n, steps, seed, over, threads=8, table="shipped"):
    W = np.load(os.path.join(ol.GOLDEN, "policy_%s.npz" % policy))
    w = {k: W[k].astype(np.float64) for k in W.files}
    clip = motion.MotionClip(clip_name)
    build, cfg_over = split_overrides(over)
    spin = build.pop("spinning_friction", 0.0)          # oracle-only experiments (orc_set_experimental)
    nofw = build.pop("no_friction_warmstart", 0)
    cfg = config.make_config(n, sim_params=config.load_sim_params(None), mode="test", enable_randomizer=False, seed=seed, num_procs=1,
                             auto_reset=False, legacy_grid=False)
    for k, v in cfg_over.items():
        setattr(cfg, k, type(getattr(cfg, k))(v))
    model = robots.laikago(**dict(robots.LAIKAGO_R04 if table == "r04" else {}, **build))
    orc = ol.OracleEnv(cfg, [model, None, None, None], [clip], n, robot_type=np.zeros(n, dtype=np.int32), clip_id=np.zeros(n, dtype=np.int32),
                       threads=threads)
    if spin or nofw:
        import ctypes as C
        orc.L.orc_set_experimental.argtypes = [C.c_void_p, C.c_int, C.c_double]
        orc.L.orc_set_experimental(orc.h, 0, float(spin))
        orc.L.orc_set_experimental(orc.h, 1, float(nofw))
    orc.field("FOOT_MU")[:] = model["foot_friction"]
    obs = orc.reset()
    orc.field("FOOT_MU")[:] = model["foot_friction"]
    alive = np.ones(n, dtype=bool)
    length = np.zeros(n)
    ret = np.zeros(n)
    reasons = np.zeros(n, dtype=int)
    lay = orc.lay
    for _ in range(steps):
        h = np.maximum(obs @ w["model__pi_fc0__w_0"] + w["model__pi_fc0__b_0"], 0.0)
        h = np.maximum(h @ w["model__pi_fc1__w_0"] + w["model__pi_fc1__b_0"], 0.0)
        a = np.clip(h @ w["model__pi__w_0"] + w["model__pi__b_0"], -2 * np.pi, 2 * np.pi)
        obs, rew, done = orc.step(a)
        length += alive
        ret += rew * alive
        reason = orc.state[:, lay.sl("DONE_REASON")][:, 0].astype(int)
        failed = done & ((reason & ~_abi.DONE_TIME_LIMIT) != 0)
        reasons = np.where(alive & failed, reason, reasons)
        alive &= ~failed
        if not alive.any():
            break
    orc.close()
    return {"finished": float(alive.mean()), "len": float(length.mean()), "r": float((ret / np.maximum(length, 1)).mean()),
            "fall": int(((reasons & 1) != 0).sum()), "pos": int(((reasons & 2) != 0).sum()), "rot": int(((reasons & 4) != 0).sum())}


def parse_value(v):
    try:
        return json.loads(v)
    except ValueError:
        return v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--robots", type=int, default=48)
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--set", nargs="*", default=[])
    ap.add_argument("--holdout", action="store_true", help="ALSO run the two held-out policies (only once a candidate is chosen)")
    ap.add_argument("--table", default="shipped", choices=["shipped", "r04"], help="r04: start from round 4's hand-authored table (robots.LAIKAGO_R04)")
    args = ap.parse_args()
    over = {}
    for kv in args.set:
        k, v = kv.split("=", 1)
        over[k] = parse_value(v)
    rows = FIT + (HOLDOUT if args.holdout else [])
    t0 = time.time()
    out = []
    for pol, clip in rows:
        o = run(pol, clip, args.robots, args.steps, args.seed, over, table=args.table)
        out.append(o)
        print("%-14s finished %.2f  len %5.1f  r/step %.3f  fall %d pos %d rot %d" % (pol, o["finished"], o["len"], o["r"], o["fall"], o["pos"], o["rot"]), flush=True)
    print("# table %s + %s  (%.0f s)" % (args.table, json.dumps(over), time.time() - t0))


if __name__ == "__main__":
    main()
