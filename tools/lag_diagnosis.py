#!/usr/bin/env python3
"""Where the 0.3 m/s go: `laikago_pace` runs 0.29-0.32 m/s behind its 1.09 m/s clip on the shipped table (VERDICT r5 item 2;
profiles/r05_policy_probe.txt "dvx").  CPU only: float64 oracle + its per-sub-step trace (orc_set_substep_trace).

Two measurements, nothing fitted:

 A. KINEMATICS OF THE CLIP ALONE (no physics, no policy).  The clip is replayed through the table's forward kinematics; a leg is "in stance"
    while its toe sphere is within `--stance-mm` of the ground.  If the robot followed the clip's joint angles and root exactly, its stance
    toes would move over the ground at `skate` m/s: a retargeted dog clip whose stance feet do not stand still.  A robot whose stance feet
    STICK and whose joints follow the clip therefore advances at  v_noslip = v_clip - skate  (per leg; the legs of a pair disagree a little).

 B. THE SIMULATED ROBOT under its policy, per stance phase of each leg (contact sub-steps with a positive normal impulse):
      slip        distance the toe's contact point moves over the ground, forward / sideways [mm per stance]
      at bound    share of contact sub-steps in which a friction row sits at +-mu lambda_n (imitation_task.py:387-414,492-516 price the result)
      N impulse   normal impulse per stance [N s] (and as a share of weight x cycle time / 2: a pace carries the robot on two legs at a time)
      push / brake  forward friction impulse per stance, positive and negative parts [N s]
      need        what the CLIP's root velocity change over that leg's stance asks for: m (v_end - v_start) [N s]
    and per run: mean forward speed, the clip's speed, reward per step and its five terms.
    Variants: the shipped table (mu 0.5, soft toes), mu 1.0, friction anchors on, mu 1.0 + anchors.

usage: python tools/lag_diagnosis.py [--policy laikago_pace] [--robots 8] [--steps 300] [--skip 100] [--out profiles/r06_lag_diagnosis.json]
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from openroborl_amd import config, motion, robots      # noqa: E402
from tests import oracle_lib as ol                      # noqa: E402

LEGS = ("FR", "FL", "RR", "RL")
TERMS = ("pose", "velocity", "end_effector", "root_pose", "root_velocity")


def clip_kinematics(clip_name, model, stance_mm=6.0, samples=400):
    """A: toe positions of the clip replayed through the table's FK, sampled over one cycle."""
    clip = motion.MotionClip(clip_name)
    cfg = config.make_config(1, mode="test", enable_randomizer=False, auto_reset=False)
    orc = ol.OracleEnv(cfg, [model, None, None, None], [clip], 1, robot_type=0, clip_id=0)
    lay = orc.lay
    st = orc.state[0].copy()
    dur = clip.frame_duration * (clip.num_frames - 1)
    ts = np.linspace(0.0, dur, samples + 1)
    toes = np.zeros((samples + 1, 4, 3))
    root = np.zeros((samples + 1, 3))
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
        root[i] = f[:3]
    orc.close()
    dt = ts[1] - ts[0]
    clear = toes[:, :, 2] - model["toe_radius"]
    stance = clear[:-1] < stance_mm * 1e-3                       # [samples, 4]
    vtoe = np.diff(toes[:, :, 0], axis=0) / dt                    # forward toe velocity over the ground
    vroot = np.diff(root[:, 0]) / dt
    v_clip = (root[-1, 0] - root[0, 0]) / dur
    rows = []
    for leg in range(4):
        m = stance[:, leg]
        if not m.any():
            rows.append(None)
            continue
        # root velocity change across the (single, possibly wrapped) stance interval
        idx = np.flatnonzero(m)
        starts = [i for i in idx if not m[i - 1]]
        ends = [i for i in idx if not m[(i + 1) % len(m)]]
        dv = float(np.mean([vroot[e] - vroot[s_] for s_, e in zip(starts, ends)])) if starts and len(starts) == len(ends) else 0.0
        rows.append({"leg": LEGS[leg], "duty": float(m.mean()), "stance_ms": float(m.sum() * dt * 1e3 / max(len(starts), 1)),
                     "skate_m_per_s": float(vtoe[m, leg].mean()), "skate_mm_per_stance": float(vtoe[m, leg].sum() * dt * 1e3 / max(len(starts), 1)),
                     "root_v_in_stance": float(vroot[m].mean()), "v_noslip": float((vroot[m] - vtoe[m, leg]).mean()),
                     "root_dv_over_stance": dv})
    return {"clip": clip_name, "duration_s": float(dur), "v_clip": float(v_clip), "stance_mm": stance_mm, "legs": rows,
            "v_noslip_mean": float(np.mean([r["v_noslip"] for r in rows if r])),
            "skate_mean": float(np.mean([r["skate_m_per_s"] for r in rows if r]))}


def simulate(policy, clip_name, model, n, steps, skip, cfg_over=None, seed=1, threads=8):
    """B: the policy on the oracle with the sub-step trace on; statistics over the steps after `skip`."""
    W = np.load(os.path.join(ol.GOLDEN, "policy_%s.npz" % policy))
    w = {k: W[k].astype(np.float64) for k in W.files}
    clip = motion.MotionClip(clip_name)
    cfg = config.make_config(n, sim_params=config.load_sim_params(None), mode="test", enable_randomizer=False, seed=seed, num_procs=1,
                             auto_reset=False, legacy_grid=False)
    for k, v in (cfg_over or {}).items():
        setattr(cfg, k, type(getattr(cfg, k))(v))
    orc = ol.OracleEnv(cfg, [model, None, None, None], [clip], n, robot_type=np.zeros(n, dtype=np.int32), clip_id=np.zeros(n, dtype=np.int32),
                       threads=threads)
    orc.field("FOOT_MU")[:] = model["foot_friction"]
    obs = orc.reset()
    orc.field("FOOT_MU")[:] = model["foot_friction"]
    rep = int(cfg.action_repeat)
    tw = int(orc.L.orc_trace_words())
    trace = np.zeros((n, rep, tw))
    orc.L.orc_set_substep_trace.argtypes = [C.c_void_p, ol.dp]
    orc.L.orc_set_substep_trace(orc.h, ol.P(trace))
    dt = float(cfg.sim_dt) if hasattr(cfg, "sim_dt") else 1e-3
    mu = float(model["foot_friction"]) * float(cfg.plane_friction)
    mass = float(model["base_mass"] + np.sum(model["link_mass"]))
    alive = np.ones(n, dtype=bool)
    rec = []          # per kept step: [n, rep, tw] copies
    terms_sum = np.zeros(5)
    rew_sum = 0.0
    cnt = 0
    vx_sum, vref_sum = 0.0, 0.0
    for s in range(steps):
        h = np.maximum(obs @ w["model__pi_fc0__w_0"] + w["model__pi_fc0__b_0"], 0.0)
        h = np.maximum(h @ w["model__pi_fc1__w_0"] + w["model__pi_fc1__b_0"], 0.0)
        a = np.clip(h @ w["model__pi__w_0"] + w["model__pi__b_0"], -2 * np.pi, 2 * np.pi)
        obs, rew, done = orc.step(a)
        reason = orc.field("DONE_REASON")[:, 0].astype(int)
        alive &= ~(done & ((reason & ~8) != 0))
        if s >= skip:
            rec.append(trace.copy())
            m = alive
            if m.any():
                terms_sum += orc.terms[m].mean(0)
                rew_sum += float(rew[m].mean())
                vx_sum += float(orc.field("LINVEL")[m, 0].mean())
                vref_sum += float(orc.field("REF_VEL")[m, 0].mean())
                cnt += 1
    orc.L.orc_set_substep_trace(orc.h, None)
    orc.close()
    T = np.stack(rec, axis=1).reshape(n, -1, tw)[alive]          # [robots alive to the end, sub-steps, words]
    lam = T[:, :, 0:12].reshape(len(T), -1, 4, 3)                 # (n, t1 = world x, t2 = world y) per leg
    vt = T[:, :, 16:28].reshape(len(T), -1, 4, 3)                 # toe-point world velocity after the solve
    vbase = T[:, :, 28]
    contact = lam[..., 0] > 0
    legs = []
    for leg in range(4):
        c = contact[:, :, leg]
        ln, lx, ly = lam[:, :, leg, 0], lam[:, :, leg, 1], lam[:, :, leg, 2]
        starts = (c[:, 1:] & ~c[:, :-1]).sum() + c[:, 0].sum()
        nst = max(int(starts), 1)
        sat = c & ((np.abs(lx) >= 0.999 * mu * ln) | (np.abs(ly) >= 0.999 * mu * ln))
        legs.append({"leg": LEGS[leg], "stances": int(starts), "duty": float(c.mean()), "stance_ms": float(c.sum() * dt * 1e3 / nst),
                     "slip_fwd_mm_per_stance": float((vt[:, :, leg, 0] * c).sum() * dt * 1e3 / nst),
                     "slip_abs_mm_per_stance": float((np.hypot(vt[:, :, leg, 0], vt[:, :, leg, 1]) * c).sum() * dt * 1e3 / nst),
                     "slip_side_mm_per_stance": float((vt[:, :, leg, 1] * c).sum() * dt * 1e3 / nst),
                     "toe_v_fwd_in_stance": float((vt[:, :, leg, 0] * c).sum() / max(c.sum(), 1)),
                     "friction_at_bound": float(sat.sum() / max(c.sum(), 1)),
                     "normal_impulse_per_stance": float((ln * c).sum() / nst),
                     "push_impulse_per_stance": float((np.maximum(lx, 0) * c).sum() / nst),
                     "brake_impulse_per_stance": float((np.minimum(lx, 0) * c).sum() / nst),
                     "base_v_in_stance": float((vbase * c).sum() / max(c.sum(), 1))})
    total_n = float(lam[..., 0].sum() / max(lam.shape[0] * lam.shape[1], 1) / dt)        # mean normal force [N]
    return {"policy": policy, "finished_window": float(alive.mean()), "robots": n, "steps": steps, "skip": skip, "mu": mu, "mass": mass,
            "v_sim": vx_sum / max(cnt, 1), "v_ref": vref_sum / max(cnt, 1), "reward_per_step": rew_sum / max(cnt, 1),
            "terms": dict(zip(TERMS, (terms_sum / max(cnt, 1)).tolist())), "mean_normal_force_over_weight": total_n / (mass * 10.0),
            "no_contact_share": float((~contact.any(axis=2)).mean()), "legs": legs}


VARIANTS = [
    ("shipped table", {}, {}),
    ("mu 1.0", {"foot_friction": 1.0}, {}),
    ("friction anchors on", {"friction_anchor": 1}, {}),
    ("mu 1.0 + anchors", {"foot_friction": 1.0, "friction_anchor": 1}, {}),
    ("rigid toes (k = d = 0)", {"contact_stiffness": 0.0, "contact_damping": 0.0}, {}),
]


def fmt_kin(k):
    lines = ["A. clip %s alone through the table's FK: cycle %.3f s, v_clip %.3f m/s, stance = toe within %.0f mm of the ground"
             % (k["clip"], k["duration_s"], k["v_clip"], k["stance_mm"]),
             "   leg  duty  stance[ms]  toe skate [m/s]  [mm/stance]  root v in stance  v_noslip = root v - skate   clip root dv over stance [m/s]"]
    for r in k["legs"]:
        if r:
            lines.append("   %-3s  %.2f  %7.0f     %+.3f          %+7.1f      %.3f            %.3f                     %+.3f" % (
                r["leg"], r["duty"], r["stance_ms"], r["skate_m_per_s"], r["skate_mm_per_stance"], r["root_v_in_stance"], r["v_noslip"], r["root_dv_over_stance"]))
    lines.append("   mean skate %+.3f m/s -> a robot with sticking stance feet that follows the clip's joints advances at %.3f m/s (clip: %.3f)"
                 % (k["skate_mean"], k["v_noslip_mean"], k["v_clip"]))
    return "\n".join(lines)


def fmt_sim(name, o, kin):
    t = o["terms"]
    lines = ["B. %-24s mu %.2f: v_sim %.3f m/s, clip %.3f (lag %+.3f), r/step %.3f [pose %.2f vel %.2f ee %.2f root %.2f rootvel %.2f], up %.2f, "
             "mean normal force / weight %.2f, flight share %.2f"
             % (name, o["mu"], o["v_sim"], o["v_ref"], o["v_sim"] - o["v_ref"], o["reward_per_step"], t["pose"], t["velocity"], t["end_effector"],
                t["root_pose"], t["root_velocity"], o["finished_window"], o["mean_normal_force_over_weight"], o["no_contact_share"]),
             "   leg  duty  stance[ms]  slip fwd / side / path [mm/stance]  toe v fwd [m/s]  at bound  N impulse [N s]  push  brake [N s]  need [N s]"]
    for r, kr in zip(o["legs"], kin["legs"]):
        need = o["mass"] * (kr["root_dv_over_stance"] if kr else 0.0)
        lines.append("   %-3s  %.2f  %7.0f     %+6.1f / %+6.1f / %6.1f               %+.3f          %.2f      %6.2f        %+.2f  %+.2f      %+.2f" % (
            r["leg"], r["duty"], r["stance_ms"], r["slip_fwd_mm_per_stance"], r["slip_side_mm_per_stance"], r["slip_abs_mm_per_stance"],
            r["toe_v_fwd_in_stance"], r["friction_at_bound"], r["normal_impulse_per_stance"], r["push_impulse_per_stance"],
            r["brake_impulse_per_stance"], need))
    return "\n".join(lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--policy", default="laikago_pace")
    ap.add_argument("--clip", default=None)
    ap.add_argument("--robots", type=int, default=8)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--skip", type=int, default=100)
    ap.add_argument("--stance-mm", type=float, default=6.0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--table", default="shipped", choices=["shipped", "r05", "r04"], help="r05: round 5's table, on which the diagnosis was first made")
    args = ap.parse_args()
    clip = args.clip or args.policy.rstrip("0")
    tab = {"shipped": {}, "r05": robots.LAIKAGO_R05, "r04": robots.LAIKAGO_R04}[args.table]
    base = robots.laikago(**tab)
    kin = clip_kinematics(clip, base, args.stance_mm)
    print("table: %s" % args.table)
    print(fmt_kin(kin), flush=True)
    res = {"kinematics": kin, "variants": []}
    for name, mo, co in VARIANTS:
        o = simulate(args.policy, clip, robots.laikago(**dict(tab, **mo)), args.robots, args.steps, args.skip, co)
        o["variant"] = name
        res["variants"].append(o)
        print(fmt_sim(name, o, kin), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
