#!/usr/bin/env python3
"""All five PyBullet-trained policies the reference ships, run on this engine (runs on the GPU box).

The reference's task/policies/*.zip were trained in PyBullet on the real URDFs: they are the only PyBullet-derived artefacts in the
tree, i.e. the only behavioural evidence about SURVEY 8a row C (the physics engine) that exists here.  Status of the five anchors since
round 5: `laikago_trot` and `laikago_spin` are IN SAMPLE (the Laikago table was identified against them,
tools/laikago_identify.py), `laikago_trot0` and `laikago_pace` were HELD OUT by that protocol, `minicheetah_trot` is in sample (round 3).
(Round 4, whose --sensitivity sweep and LAIKAGO_SHIPPED values below refer to robots.LAIKAGO_R04: pace looked at while the table was
written, the other three Laikago policies out of sample - and falling.)  Each zip is matched to its clip
by the 76 target-observation bounds pickled inside it (tests/golden/policy_clips.json, written by tests/golden/make_golden.py); a clip
and its time reversal have equal bounds, so both are run.

Protocol = the reference's test mode (run.py:151-183: no randomiser, 2 ms latency, 600-step limit, deterministic actions):
  finished   = fraction of robots whose first termination is the 600-step time limit
  len        = mean number of steps until the first failure (600 if none)
  reward     = mean reward per step while alive, and its five terms (imitation_task.py:341-356).  Pose, velocity, root-pose and
               root-velocity terms are recomputed here in torch from the state record (reference pose of the PREVIOUS update:
               quadruped_gym_env.py:230-232); the end-effector term is what remains of the kernel's reward:
               (r - sum_k w_k term_k) / w_ee
  reasons    = first failure: fall contact / root position / root rotation / non-finite
  splits     = by the episode's WARMUP flag (imitation_task.py:183-188: one reset in ten starts from the default pose, not from the
               clip), by the phase of the clip at reset, by the step of the fall
  control    = the same with ref_state_init_prob = 1.0 (no warm-up episodes) and = 0.0 (warm-up episodes only)

usage: python tools/policy_probe.py [--robots 1024] [--seeds 1 2] [--out gpurun_out/policy_probe.json]
"""
import argparse
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

DONE_FALL, DONE_POS, DONE_ROT, DONE_TIME, DONE_NAN = 1, 2, 4, 8, 16


def policy_table():
    with open(os.path.join(GOLDEN, "policy_clips.json")) as f:
        match = json.load(f)
    rows = []
    for pol in sorted(match):
        for clip in match[pol]["clips_with_equal_bounds"]:
            rows.append((pol, clip, "mini_cheetah" if clip.startswith("minicheetah") else "laikago", clip == match[pol]["clip"]))
    return rows


def _qmul(t, a, b):
    ax, ay, az, aw = a.unbind(-1)
    bx, by, bz, bw = b.unbind(-1)
    return t.stack([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                    aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz], dim=-1)


def reward_terms(t, env, ref_pose, ref_vel, reward):
    """The five terms of imitation_task.py:341-516 for the step that just ran.  ref_pose / ref_vel: snapshot BEFORE the step."""
    c = env.cfg
    w, sc = list(c.reward_w), list(c.reward_scale)
    q, qd = env.field("Q"), env.field("QD")
    pose = t.exp(-sc[0] * ((ref_pose[:, 7:] - q) ** 2).sum(1))
    vel = t.exp(-sc[1] * ((ref_vel[:, 6:] - qd) ** 2).sum(1))
    quat = env.field("QUAT")
    conj = quat * t.tensor([-1.0, -1.0, -1.0, 1.0], device=quat.device)
    dq = _qmul(t, ref_pose[:, 3:7], conj)
    ang = 2.0 * t.atan2(dq[:, :3].norm(dim=1), dq[:, 3])          # pose3d.py:139-187
    ang = t.remainder(ang + math.pi, 2.0 * math.pi) - math.pi      # pose3d.py:304-322
    rpose = t.exp(-sc[4] * (((ref_pose[:, :3] - env.field("POS")) ** 2).sum(1) + 0.5 * ang * ang))
    rvel = t.exp(-sc[5] * (((ref_vel[:, :3] - env.field("LINVEL")) ** 2).sum(1)
                           + 0.1 * ((ref_vel[:, 3:6] - env.field("ANGVEL")) ** 2).sum(1)))
    ee = (reward - w[0] * pose - w[1] * vel - w[3] * rpose - w[4] * rvel) / w[2]
    return t.stack([pose, vel, ee, rpose, rvel], dim=1)


def run(pol, clip, robot, n, seed, steps=600, ref_state_init_prob=None, model_over=None, config_over=None, raw=False):
    import torch
    from openroborl_amd import policy as polmod
    from openroborl_amd.env import VecQuadrupedEnv
    over = dict(config_over or {})
    if ref_state_init_prob is not None:
        over["ref_state_init_prob"] = float(ref_state_init_prob)
    env = VecQuadrupedEnv(num_robot=n, seed=seed, robot=robot, motion_file=clip, mode="test", enable_randomizer=False,
                          auto_reset=False, config_overrides=over, model_overrides={robot: model_over} if model_over else None)
    dev = env.device
    model = polmod.MLPPolicy.from_file(os.path.join(GOLDEN, "policy_%s.npz" % pol), dev)
    obs = env.reset()
    warm = env.field_int("WARMUP")[:, 0].clone().bool()
    dur = env.clips[0].frame_duration * (env.clips[0].num_frames - 1)
    phase0 = (env.field("TIME_OFFSET")[:, 0].clone() / dur) % 1.0
    # what the teleport into the reference state does to the robot in its very first sub-step (zero motor torques): normal impulses per
    # leg and the change of the base's vertical velocity - the reference pose may put toes into the ground (the contact rows then push them
    # out with erp x depth / dt) or leave the robot in the air
    start = env.state.clone()
    vz0 = env.field("LINVEL")[:, 2].clone()
    env.debug_physics(torch.zeros(n, 12, device=dev), 1)
    kick = (env.field("LINVEL")[:, 2] - vz0).clone()
    lam_n = env.field("LAMBDA")[:, 0::3].clone()                 # normal impulse per leg [N s]
    env.state.copy_(start)
    alive = torch.ones(n, dtype=torch.bool, device=dev)
    length = torch.zeros(n, device=dev)
    ret = torch.zeros(n, device=dev)
    terms = torch.zeros(n, 5, device=dev)
    first_reason = torch.zeros(n, dtype=torch.int32, device=dev)
    reason_f = env.field_int("DONE_REASON")[:, 0]
    trace = []      # every 10th step, means over the robots still up: how the simulated robot sits relative to the reference it tracks
    for k in range(steps):
        act, _, _ = model.act(obs, deterministic=True)
        rp, rv = env.field("REF_POSE").clone(), env.field("REF_VEL").clone()
        obs, rew, done, _ = env.step(act.contiguous())
        if k % 10 == 9 and k < 200 and bool(alive.any()):
            m = alive
            ref_now = env.field("REF_POSE")                       # the pose the robot should be in NOW (updated inside the step)
            dq = (env.field("Q") - ref_now[:, 7:])[m]              # URDF joint order: (hip-x, upper, lower) x 4 legs
            trace.append({"step": k + 1, "alive": float(m.float().mean()),
                          "dz": float((env.field("POS")[:, 2] - ref_now[:, 2])[m].mean()),
                          "dvx_world": float((env.field("LINVEL")[:, 0] - env.field("REF_VEL")[:, 0])[m].mean()),
                          "dq_hip_x": float(dq[:, 0::3].mean()), "dq_upper": float(dq[:, 1::3].mean()), "dq_lower": float(dq[:, 2::3].mean()),
                          "dq_front_minus_rear_upper": float((dq[:, [1, 4]].mean() - dq[:, [7, 10]].mean())),
                          "rms_dq": float((dq ** 2).mean().sqrt()),
                          "normal_impulse_front": float(env.field("LAMBDA")[:, 0:6:3][m].sum(1).mean()),
                          "normal_impulse_rear": float(env.field("LAMBDA")[:, 6:12:3][m].sum(1).mean())})
        a = alive.float()
        ret += rew * a
        terms += reward_terms(torch, env, rp, rv, rew) * a[:, None]
        length += a
        failed = done.bool() & ((reason_f & ~DONE_TIME) != 0)
        first_reason = torch.where(alive & failed, reason_f, first_reason)
        alive &= ~failed
    L = length.clamp(min=1)
    r = first_reason.cpu().numpy()
    al = alive.cpu().numpy()
    wm = warm.cpu().numpy()
    ln = length.cpu().numpy()
    ph = phase0.cpu().numpy()
    tm = (terms / L[:, None]).cpu().numpy()

    def frac(mask):
        return None if mask.sum() == 0 else float(al[mask].mean())

    out = {"policy": pol, "clip": clip, "robot": robot, "robots": n, "seed": seed,
           "ref_state_init_prob": float(env.cfg.ref_state_init_prob),
           "finished": float(al.mean()), "len": float(ln.mean()), "reward_per_step": float((ret / L).mean()),
           # J of tools/identify_r6.py: the episode return (rewards until the first failure) per NOMINAL step - what PPO maximised
           "return_per_nominal_step": float((ret / float(steps)).mean()),
           "terms": {k: float(tm[:, i].mean()) for i, k in enumerate(("pose", "velocity", "end_effector", "root_pose", "root_velocity"))},
           "terms_finishers": {k: (float(tm[al, i].mean()) if al.any() else None)
                               for i, k in enumerate(("pose", "velocity", "end_effector", "root_pose", "root_velocity"))},
           "reasons": {"finished": int(al.sum()), "fall": int((r & DONE_FALL != 0).sum()), "root_pos": int((r & DONE_POS != 0).sum()),
                       "root_rot": int((r & DONE_ROT != 0).sum()), "non_finite": int((r & DONE_NAN != 0).sum())},
           "warmup": {"episodes": int(wm.sum()), "finished": frac(wm), "mean_len": (float(ln[wm].mean()) if wm.any() else None)},
           "ref_state_init": {"episodes": int((~wm).sum()), "finished": frac(~wm), "mean_len": (float(ln[~wm].mean()) if (~wm).any() else None)},
           # fallers among the reference-state-init episodes, by the clip phase they were dropped into (8 bins)
           "finished_by_reset_phase": [frac((~wm) & (ph >= k / 8.0) & (ph < (k + 1) / 8.0)) for k in range(8)],
           "fall_step_histogram": {"edges": [0, 5, 10, 20, 40, 80, 160, 320, 600],
                                   "counts": np.histogram(ln[~al], bins=[0, 5, 10, 20, 40, 80, 160, 320, 600])[0].tolist()},
           "fall_step_histogram_warmup": np.histogram(ln[(~al) & wm], bins=[0, 5, 10, 20, 40, 80, 160, 320, 600])[0].tolist()}
    # reference-state-init episodes by the clip phase they start in, 32 bins: finished fraction, legs with a ground contact in the first
    # sub-step, largest normal impulse [N s] and vertical velocity change [m/s] of that sub-step
    kc, ln_ = kick.cpu().numpy(), lam_n.cpu().numpy()
    detail = []
    for k in range(32):
        m = (~wm) & (ph >= k / 32.0) & (ph < (k + 1) / 32.0)
        if m.sum() == 0:
            detail.append(None)
            continue
        detail.append({"phase": (k + 0.5) / 32.0, "episodes": int(m.sum()), "finished": float(al[m].mean()),
                       "legs_in_contact": float((ln_[m] > 0).sum(axis=1).mean()), "max_normal_impulse": float(ln_[m].max(axis=1).mean()),
                       "dvz_first_substep": float(kc[m].mean())})
    out["by_reset_phase_32"] = detail
    out["trace"] = trace
    if raw:      # per-robot arrays (tests): finished, warm-up episode, clip phase at reset, steps survived
        out["_raw"] = {"finished": al, "warmup": wm, "phase": ph, "len": ln}
    env.close()
    return out


def fmt_phase(o):
    lines = ["  %s on %s: reference-state-init episodes by the clip phase at reset (finished | legs in contact, max normal impulse [N s], dvz [m/s] "
             "of the first sub-step)" % (o["policy"], o["clip"])]
    for d in o["by_reset_phase_32"]:
        if d:
            lines.append("    phase %.3f  n %3d  finished %.2f | %.1f legs  %.4f N s  %+.3f m/s" % (
                d["phase"], d["episodes"], d["finished"], d["legs_in_contact"], d["max_normal_impulse"], d["dvz_first_substep"]))
    return "\n".join(lines)


def fmt_trace(o):
    lines = ["  %s on %s: simulated robot relative to its reference, means over the robots still up (dz [m], world-x velocity error [m/s], joint "
             "errors sim - ref [rad] by joint class, front - rear upper-leg error, rms joint error; normal impulses per sub-step front / rear [N s])"
             % (o["policy"], o["clip"])]
    for t in o["trace"]:
        lines.append("    step %3d up %.2f | dz %+.4f dvx %+.3f | hip-x %+.3f upper %+.3f lower %+.3f  f-r %+.3f  rms %.3f | N front %.4f rear %.4f" % (
            t["step"], t["alive"], t["dz"], t["dvx_world"], t["dq_hip_x"], t["dq_upper"], t["dq_lower"], t["dq_front_minus_rear_upper"],
            t["rms_dq"], t["normal_impulse_front"], t["normal_impulse_rear"]))
    return "\n".join(lines)


def fmt(o):
    t = o["terms"]
    return ("%-17s %-23s p=%.1f seed %d: finished %.3f  len %5.1f  r/step %.3f  [pose %.2f vel %.2f ee %.2f root %.2f rootvel %.2f]  "
            "fall %d pos %d rot %d nan %d | warm-up %s/%d ref-init %s/%d"
            % (o["policy"], o["clip"], o["ref_state_init_prob"], o["seed"], o["finished"], o["len"], o["reward_per_step"],
               t["pose"], t["velocity"], t["end_effector"], t["root_pose"], t["root_velocity"],
               o["reasons"]["fall"], o["reasons"]["root_pos"], o["reasons"]["root_rot"], o["reasons"]["non_finite"],
               "-" if o["warmup"]["finished"] is None else "%.2f" % o["warmup"]["finished"], o["warmup"]["episodes"],
               "-" if o["ref_state_init"]["finished"] is None else "%.3f" % o["ref_state_init"]["finished"], o["ref_state_init"]["episodes"]))


# One-at-a-time variations of the hand-authored (parity-unpinned) entries of the Laikago table and of the engine constants, each across
# a stated plausible interval; "*" marks the shipped value.  ("_build", key) = argument of robots._build, ("model", key) = entry of the
# built table, ("config", key) = orr_config field.  The LAST group are combinations fixed BEFORE the run was looked at:
#   soft_toes     = Bullet's contact stiffness / damping on the toe links with the values pybullet_data's quadruped URDFs are remembered
#                   to carry (<contact><stiffness value="30000"/><damping value="1000"/><lateral_friction value="3.0"/>): from memory of a
#                   file that is not available here, i.e. exactly as unpinned as the rest of the table
#   hip_x_urdf    = hip joints 0.2429 m in front of / behind the base origin (remembered URDF value; the table uses laikago.py:54-59's 0.21)
LAIKAGO_SWEEP = [
    ("foot_friction", ("model", "foot_friction"), [0.5, 0.75, 1.0, 1.5, 2.0, 3.0]),
    ("soft_toe_(k,d)", ("soft", None), [(0.0, 0.0), (30000.0, 1000.0), (30000.0, 300.0), (100000.0, 1000.0), (10000.0, 1000.0), (30000.0, 3000.0)]),
    ("contact_erp", ("config", "contact_erp"), [0.05, 0.1, 0.2, 0.4]),
    ("solver_iters", ("config", "solver_iters"), [5, 9, 20, 50]),
    ("warmstart_factor", ("config", "warmstart_factor"), [0.0, 0.85, 1.0]),
    ("toe_m", ("_build", "toe_m"), [0.005, 0.03, 0.06, 0.12]),
    ("toe_r", ("_build", "toe_r"), [0.02, 0.0265, 0.033]),
    ("hip_z", ("_build", "hip_z"), [-0.07, -0.044, -0.02, 0.0]),
    ("hip_x", ("hip_x", None), [0.19, 0.21, 0.2429]),
    ("up_com_z", ("up_com_z", None), [-0.02, -0.04, -0.08]),
    ("lo_com_z", ("lo_com_z", None), [-0.07, -0.11, -0.15]),
    ("base_mass", ("_build", "base_mass"), [11.0, 13.715, 16.0]),
    ("leg_inertia_scale", ("leg_I", None), [0.5, 1.0, 2.0]),
    ("shank_r", ("_build", "shank_r"), [0.0, 0.02, 0.03]),
    ("joint_limits", ("limits", None), ["table", "continuous"]),
    ("combination", ("combo", None), ["soft_toes", "soft_toes+hip_x_urdf", "soft_toes_mu1", "hip_x_urdf"]),
]
LAIKAGO_SHIPPED = {"foot_friction": 1.0, "soft_toe_(k,d)": (0.0, 0.0), "contact_erp": 0.2, "solver_iters": 9, "warmstart_factor": 0.85,
                   "toe_m": 0.06, "toe_r": 0.0265, "hip_z": -0.044, "hip_x": 0.21, "up_com_z": -0.04, "lo_com_z": -0.11,
                   "base_mass": 13.715, "leg_inertia_scale": 1.0, "shank_r": 0.02, "joint_limits": "table"}


def laikago_variation(kind, key, v):
    """-> (model_over, config_over) of one variation."""
    soft = {"contact_stiffness": 30000.0, "contact_damping": 1000.0}
    if kind == "model":
        return {key: v}, {}
    if kind == "config":
        return {}, {key: v}
    if kind == "_build":
        return {"_build": {key: v}}, {}
    if kind == "soft":
        return {"contact_stiffness": v[0], "contact_damping": v[1]}, {}
    if kind == "hip_x":
        return {"_build": {"hip_xy": [v, 0.1157 - 0.032875]}}, {}
    if kind == "up_com_z":
        return {"_build": {"up_com": [0.0, 0.0, v]}}, {}
    if kind == "lo_com_z":
        return {"_build": {"lo_com": [0.0, 0.0, v]}}, {}
    if kind == "leg_I":
        return {"_build": {"hip_I": [v * x for x in (0.00100, 0.00120, 0.00100)], "up_I": [v * x for x in (0.0078, 0.0081, 0.0012)],
                           "lo_I": [v * x for x in (0.0013, 0.0013, 0.00005)]}}, {}
    if kind == "limits":
        return ({"_build": {"limits": [(-1e9, 1e9)] * 3}} if v == "continuous" else {}), {}
    if kind == "combo":
        m = {}
        if v.startswith("soft_toes"):
            m.update(soft)
            m["foot_friction"] = 1.0 if v.endswith("mu1") else 3.0
        if "hip_x_urdf" in v:
            m["_build"] = {"hip_xy": [0.2429, 0.1157 - 0.032875]}
        return m, {}
    raise ValueError(kind)


def sensitivity(args):
    rows = [r for r in policy_table() if r[3] and r[2] == "laikago"]
    res = {"robots": args.robots, "seed": args.seeds[0], "shipped": LAIKAGO_SHIPPED, "policies": [r[0] for r in rows], "sweep": []}
    print("%-20s %-22s " % ("parameter", "value") + "  ".join("%-16s" % r[0] for r in rows) + "   (finished / mean survival steps / reward per step)")
    for name, (kind, key), values in LAIKAGO_SWEEP:
        for v in values:
            mo, co = laikago_variation(kind, key, v)
            # the sweep is ONE-AT-A-TIME AROUND ROUND 4's TABLE (that is what LAIKAGO_SHIPPED / the "*" marks and the recorded
            # profiles/r04_laikago_sensitivity.* refer to): the round-4 entries go in first, the variation on top
            from openroborl_amd import robots as _robots
            mo = dict(mo)
            mo["_build"] = dict(_robots.LAIKAGO_R04, **mo.get("_build", {}))
            for k_ in ("foot_friction", "contact_stiffness", "contact_damping"):      # table entries that are also _build arguments
                if k_ in mo:
                    mo["_build"][k_] = mo.pop(k_)
            cells = []
            for pol, clip, robot, _ in rows:
                o = run(pol, clip, robot, args.robots, args.seeds[0], model_over=mo, config_over=co)
                cells.append({"policy": pol, "finished": o["finished"], "len": o["len"], "reward_per_step": o["reward_per_step"],
                              "terms": o["terms"], "reasons": o["reasons"]})
            shipped = LAIKAGO_SHIPPED.get(name) == v
            res["sweep"].append({"parameter": name, "value": v, "shipped": shipped, "cells": cells})
            print("%-20s %-22s " % (name, str(v) + ("*" if shipped else "")) +
                  "  ".join("%.2f/%3.0f/%.2f    " % (c["finished"], c["len"], c["reward_per_step"]) for c in cells), flush=True)
    # mini-cheetah: the same contact hypothesis on the table identified in round 3 AND on the round-2 table it replaced (published MIT
    # figures: toe 0.15 kg, shank 0.064 kg with its COM 0.061 m below the knee, hip axis plane through the base COM, thigh COM 0.02 m
    # below the hip, shank sphere 0.012 m at 0.02 m) - did the identification compensate for a missing contact model?
    r02 = {"toe_m": 0.15, "lo_m": 0.064, "lo_com": [0.0, 0.0, -0.061], "lo_I": [0.000245, 0.000248, 0.000006], "hip_z": 0.0,
           "up_com": [0.0, 0.016, -0.02], "shank_r": 0.012, "shank_at": 0.02}
    res["mini_cheetah"] = []
    for table, build in (("identified (round 3)", {}), ("round-2 table", r02)):
        for label, extra in (("rigid, mu 1*", {}), ("rigid, mu 3", {"foot_friction": 3.0}),
                             ("soft toes (30000, 1000), mu 1", {"contact_stiffness": 30000.0, "contact_damping": 1000.0}),
                             ("soft toes (30000, 1000), mu 3", {"contact_stiffness": 30000.0, "contact_damping": 1000.0, "foot_friction": 3.0})):
            mo = dict(extra)
            if build:
                mo["_build"] = build
            o = run("minicheetah_trot", "minicheetah_trot", "mini_cheetah", args.robots, args.seeds[0], model_over=mo)
            res["mini_cheetah"].append({"table": table, "contact": label, "finished": o["finished"], "len": o["len"],
                                        "reward_per_step": o["reward_per_step"], "terms": o["terms"], "reasons": o["reasons"]})
            print("mini_cheetah %-22s %-32s %.2f/%3.0f/%.2f" % (table, label, o["finished"], o["len"], o["reward_per_step"]), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print("written", args.out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sensitivity", action="store_true", help="one-at-a-time sweep of the Laikago table / engine constants over the four Laikago policies")
    ap.add_argument("--robots", type=int, default=1024)
    ap.add_argument("--seeds", type=int, nargs="+", default=[1, 2])
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "policy_probe.json"))
    ap.add_argument("--no-controls", action="store_true")
    args = ap.parse_args()
    if args.sensitivity:
        return sensitivity(args)
    from openroborl_amd import _lib
    res = {"source_hash": _lib.library_hash(), "rows": []}
    for pol, clip, robot, named in policy_table():
        for seed in args.seeds:
            o = run(pol, clip, robot, args.robots, seed)
            o["clip_is_the_zip_name"] = named
            res["rows"].append(o)
            print(fmt(o), flush=True)
            if named and seed == args.seeds[0] and pol in ("minicheetah_trot", "laikago_pace"):
                print(fmt_phase(o), flush=True)
            if named and seed == args.seeds[0]:
                print(fmt_trace(o), flush=True)
        if not args.no_controls and named:
            for p in (1.0, 0.0):
                o = run(pol, clip, robot, args.robots, args.seeds[0], ref_state_init_prob=p)
                o["clip_is_the_zip_name"] = named
                o["control"] = True
                res["rows"].append(o)
                print(fmt(o), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print("written", args.out)


if __name__ == "__main__":
    main()
