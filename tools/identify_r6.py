#!/usr/bin/env python3
"""Round-6 identification of the hand-authored robot tables against the reference's PyBullet-trained policies: reward-aware, cross-validated
(VERDICT r5 items 1, 3, 4, 5).  Runs on the GPU box; `--backend oracle` runs the same logic on the CPU oracle at toy sizes for the tests.

The reference ships five policies trained in PyBullet on the real URDFs (task/policies/*.zip): the only PyBullet-derived evidence about SURVEY
8a row C.  Round 5 identified the Laikago table on ONE fit / hold-out split with a survival-only criterion and then edited one entry by hand.
This round: every split of the four Laikago policies is held out once, candidates are judged on the reward the policies were trained to
maximise (imitation_task.py:341-516), and what ships is the protocol's own output.  PyBullet is never imported or run.

==== PROTOCOL (fixed in this docstring and committed BEFORE any GPU call of round 6) ====

P0. Test mode of run.py:151-183 (no randomiser, 2 ms latency, 600-step limit, deterministic actions).  Per candidate and policy, R robots:
      F    = fraction of robots whose first termination is the 600-step time limit
      len  = mean steps to the first failure (600 if none)
      J    = mean over robots of (sum of the step rewards until the first failure) / 600: the episode return per nominal step, i.e. what PPO
             maximised in PyBullet (a failure ends the episode and forfeits the remaining reward).  THE CRITERION.
      R    = mean reward per step while alive (tools/policy_probe.py's "r/step"), recorded beside J; the choice under R instead of J is
             reported as a cross-check but decides nothing.
      terms = the five reward terms, dvx = mean world-x velocity error, advx = its mean absolute value (all over alive robot-steps, every
             4th step), reasons of the first failure.
P1. THE BOX (Laikago).  Reference point = round 4's table (robots.LAIKAGO_R04).  FROZEN at it, never varied:
      hip_z -0.044 and toe_r 0.0265 (jointly pinned by in-tree data: the clips' stance toes touch the ground, tools/diag/clip_toe_clearance.py);
      chassis box, hip / knee spheres (termination-only proxies: the round-5 ablation says the fit does not care, imitation_task.py:536-546
      does) and the shank sphere; what the reference fixes (control constants laikago.py:29-71, link lengths and angle conventions
      trans2minicheetah.m:3-9, gravity, time step, solver iterations); the solver constants, settled beforehand by P5.
    VARIED inside the intervals of SPECS below: the round-5 intervals, widened where round 5's accepted mass sat on an edge (toe_m -> 0.25,
    com_x -> 0.06, hip_y -> 0.12, hip_x -> 0.27, lo_m -> 0.55, foot_friction 0.3 .. 3.5).  hip_x / hip_y: laikago.py:54-59 states
    _DEFAULT_HIP_POSITIONS (0.21, 0.1157); that tuple is never read anywhere in the reference and its mini-cheetah twin (0.38, 0.1161,
    mini_cheetah.py:55-60) contradicts the authors' own retargeting script (0.19, 0.049, trans2minicheetah.m:28-30), so it is the reference
    POINT of the interval, not a pin.
P2. SEARCH per run: 30 % of the budget random candidates (half uniform in the box, half Gaussian clouds around the reference point), 70 % local
    search (children of the current top 16, step 0.15 -> 0.03 of the interval widths); objective = min over the FIT policies of J.  No
    candidate of a run ever sees that run's hold-out policies; nothing is seeded from round 5's table (it was fitted on trot + spin).
P3. CHOICE per run: ACCEPTED = F >= 0.8 on every fit policy (F >= 0.9 for the mini-cheetah, as in round 3).  The 16 accepted candidates with
    the highest min-J are re-evaluated alone at 4 x R robots with another env seed; those still accepted are walked in the order of the
    re-evaluated min-J; the first whose +-10 % cloud (32 samples) keeps mean(min-F) >= 0.8 is CHOSEN.  Its normalised distance to the
    reference point is recorded beside it.  If nothing is accepted the best min-J candidate is reported and the run's verdict is NEGATIVE.
P4. CROSS-VALIDATION (Laikago): all SIX splits of {pace, spin, trot, trot0} into two fit and two held-out policies, each an independent run
    of P2-P3 with its own random seed; the hold-out policies are evaluated ONCE per split, on the chosen candidate, at 1024 robots (seeds 1
    and 2), and reported whatever they are.  Output: profiles/r06_laikago_cv.json and the 6 x 4 table (F and J, fit and held out) of
    DESIGN.md section 7.2.  A split "transfers" iff it is accepted and both held-out policies reach F >= 0.5.
P5. ENGINE CONSTANTS, settled BEFORE P4 by a cross-robot rule on the tables shipped by round 5 (no table entry moves in this step):
      LIB = Bullet library defaults (contact_erp 0.2, warmstart 0.85, contact_margin 0.02, friction_erp 0.2) - what rounds 1-5 shipped;
      PYB = config.PYBULLET_REMEMBERED (0.08, 0.1, 0.004, 0.2) - what PyBullet's createEmptyDynamicsWorld is remembered to set.
    1024 robots, seeds 1 and 2, all five policies under both sets.  Direction A: PYB is PREFERRED on the Laikago iff the mean over its four
    policies of J is not lower than under LIB; it VALIDATES on the other robot iff minicheetah_trot loses no more than 0.02 in F and in J.
    Direction B, the mirror image: preferred on minicheetah_trot, validated on the Laikago iff none of the four policies loses more than
    0.02 in F or J.  PYB is ADOPTED (config.make_config defaults) iff both directions hold; otherwise LIB stays.  Record:
    profiles/r06_constants_rule.json.
P6. WHAT SHIPS (Laikago): a seventh run of P2-P3 with ALL FOUR policies as the fit set (no hold-out exists for it: IN SAMPLE, labelled so; the
    out-of-sample evidence is P4's matrix), followed by P7.
P7. SMALLEST TABLE: from the chosen candidate, entries are put back to the reference point one at a time, each time the one whose reversion
    keeps min-J highest, as long as every fit policy stays accepted and min-J stays within 0.01 of the chosen candidate's (1024 robots; the
    seed-to-seed spread of min-J at that size is ~0.003).  The end of that path ships; robots.py then differs from LAIKAGO_R04 only in
    entries whose reversion costs something, each with its effect recorded (profiles/r06_laikago_minimal.json).
P8. MINI-CHEETAH: one policy exists, so no hold-out is possible: IN SAMPLE.  P2-P3 with fit = {minicheetah_trot} under the settled constants;
    reference point = the round-2 table (robots.MINI_CHEETAH_R02); frozen: hip_z +0.011 and toe_r (clip toe clearance agrees with round 3's
    policy-based value), the termination-only proxies (knee radius 0 included); then P7.  Record: profiles/r06_mc_identify.json.

P9. REVISION OF THE LAIKAGO BOX, written after P1-P8 had run and their hold-outs had been seen, committed before any P9 GPU call.  The post-hoc
    diagnostics (tools/diag/spin_hip_x.py, tools/diag/clip_hip_x_slip.py) show that the turning clip `laikago_turn` pins
    hip_x and hip_y KINEMATICALLY at round 4's values (0.21, 0.0828 = laikago.py:54-59 minus the coxa) - policy-free in-tree data of the same
    kind as the toe clearance that froze hip_z in P1 - and that P4's splits without spin lose the spin policy because their hip_x ran to the
    other edge of its interval.  P9 = P2-P4, P6 and P7 again with exactly ONE change: hip_x and hip_y FROZEN at those values (`--freeze-geometry`).
    com_x stays free although the same clip says 0.00 +- 0.01: with it at 0 no table near P6's keeps trot0 walking
    (profiles/r06_geometry_pinned_probe.txt) - it is the open compensation of DESIGN.md section 7.2, flagged, not hidden.  Search seeds 700 + i
    (all-four run: 800), same budgets.  Because P9's design used knowledge of P4's hold-out outcomes, its 6 x 4 matrix is WEAKER evidence than
    P4's and is reported beside it, never instead of it (profiles/r06_laikago_cv_p9.json).  WHAT SHIPS after P9: the end of P7 on P9's all-four
    run iff it is accepted on all four policies and its min-J (1024 robots x 2 seeds) is not more than 0.02 below the P6 / P7 table's (0.646);
    otherwise the P6 / P7 table stays.  Either way both are recorded.

usage:
  python tools/identify_r6.py constants [--robots 1024] [--out gpurun_out/r06_constants_rule.json]
  python tools/identify_r6.py run --robot laikago --fit laikago_trot laikago_spin --holdout laikago_trot0 laikago_pace --minutes 12 --out X.json
  python tools/identify_r6.py cv --minutes 12 --outdir gpurun_out/r06cv            (six `run` children side by side, then the table)
  python tools/identify_r6.py minimal --record X.json --out Y.json
"""
import argparse
import gzip
import itertools
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
GOLDEN = os.path.join(ROOT, "tests", "golden")

TERMS = ("pose", "velocity", "end_effector", "root_pose", "root_velocity")
LAIKAGO_POLICIES = ["laikago_pace", "laikago_spin", "laikago_trot", "laikago_trot0"]


def clip_of(pol):
    return pol.rstrip("0")


# ---- the boxes ---------------------------------------------------------------------------------------------------------------------------
# name: (reference-point value, low, high)
SPECS = {
    "laikago": {
        "accept": 0.8,
        "params": {
            "toe_m":        (0.06, 0.005, 0.25),     # toe link mass [kg]                                 (round 5: .. 0.15, accepted mass at the edge)
            "hip_x":        (0.21, 0.19, 0.27),      # hip joints in front of / behind the hips' centre [m] (laikago.py:54-59: 0.21, unused there; URDF (mem): 0.2429)
            "hip_y":        (0.082825, 0.07, 0.12),  # hip joints left / right of the centre line [m]      (round 5: .. 0.10, edge)
            "com_x":        (0.0, -0.03, 0.06),      # base COM in front of the hips' centre [m]            (round 5: .. 0.03, edge)
            "base_mass":    (13.715, 11.0, 16.5),
            "base_I":       (1.0, 0.6, 1.6),         # scale of the base inertia
            "hip_m":        (1.095, 0.8, 1.4),
            "up_m":         (1.527, 1.1, 1.9),
            "lo_m":         (0.241, 0.15, 0.55),     # (round 5: .. 0.40, edge)
            "leg_I":        (1.0, 0.5, 2.0),         # scale of the leg link inertias
            "hip_com_y":    (0.0, -0.02, 0.04),      # hip link COM outward of the abduction axis [m]
            "up_com_x":     (0.0, -0.02, 0.02),      # thigh COM: forward / outward / below the hip pitch axis [m]
            "up_com_y":     (0.0, -0.01, 0.04),
            "up_com_z":     (-0.04, -0.09, -0.01),
            "lo_com_x":     (0.0, -0.02, 0.02),      # shank COM: forward / below the knee [m]
            "lo_com_z":     (-0.11, -0.16, -0.06),
            "foot_friction": (1.0, 0.3, 3.5),        # toe lateral friction; test mode keeps the table's value ((mem) URDF: 3.0), training draws U[0.5, 1.25]
        },
        # switched features: (reference point, probability of being on in a uniform candidate)
        "switches": {"limits": (1, 0.5), "soft": (0, 0.5), "anchor": (0, 0.4)},
        "soft_k": (1.0e4, 1.0e5), "soft_d": (3.0e2, 3.0e3), "soft_ref": (30000.0, 1000.0),     # (mem) pybullet_data's quadruped URDFs: 30000 / 1000
    },
    "mini_cheetah": {
        "accept": 0.9,
        "params": {
            "toe_m":    (0.15, 0.02, 0.30),      # toe / foot link mass [kg]    (round 3: .. 0.25)
            "lo_m":     (0.064, 0.05, 0.25),     # shank mass [kg]
            "lo_com_z": (-0.061, -0.12, -0.02),  # shank COM below the knee [m] (shank length 0.18)
            "up_com_z": (-0.02, -0.06, 0.0),     # thigh COM below the hip pitch axis [m]
            "shank_r":  (0.012, 0.0, 0.02),      # shank contact sphere (creates forces, unlike the fall proxies): radius, distance below the knee [m]
            "shank_at": (0.02, 0.0, 0.06),
            "foot_friction": (1.0, 0.3, 2.0),
        },
        "switches": {"limits": (0, 0.5), "soft": (0, 0.5)},
        "soft_k": (1.0e4, 1.0e5), "soft_d": (1.0e2, 3.0e3), "soft_ref": (30000.0, 1000.0),
    },
}


# P9: the Laikago box with the two entries the turning clip pins frozen at the reference point (build_model leaves missing entries at round 4's)
SPECS["laikago_g"] = dict(SPECS["laikago"], params={k: v for k, v in SPECS["laikago"]["params"].items() if k not in ("hip_x", "hip_y")})


# exploration only (never part of what ships): com_x frozen at the clip-pinned 0.00 as well
SPECS["laikago_gc"] = dict(SPECS["laikago"], params={k: v for k, v in SPECS["laikago"]["params"].items() if k not in ("hip_x", "hip_y", "com_x")})


def spec_of(robot, freeze_geometry=False, freeze_com=False):
    if robot == "laikago" and freeze_com:
        return SPECS["laikago_gc"]
    return SPECS["laikago_g" if (robot == "laikago" and freeze_geometry) else robot]


def names(spec):
    return list(spec["params"])


def reference_theta(spec):
    th = {k: v[0] for k, v in spec["params"].items()}
    th.update({k: v[0] for k, v in spec["switches"].items()})
    th["soft_k"], th["soft_d"] = spec["soft_ref"]
    return th


def random_theta(spec, rng, mode):
    """mode 0: uniform in the box; 1: Gaussian cloud around the reference point (0.25 x interval width)."""
    th = {}
    for k, (v0, lo, hi) in spec["params"].items():
        th[k] = float(lo + rng.rand() * (hi - lo)) if mode == 0 else float(np.clip(v0 + rng.randn() * 0.25 * (hi - lo), lo, hi))
    for k, (v0, p) in spec["switches"].items():
        th[k] = int(rng.rand() < p) if mode == 0 else (int(v0) if rng.rand() < 0.7 else 1 - int(v0))
    th["soft_k"] = float(np.exp(rng.uniform(*np.log(spec["soft_k"]))))
    th["soft_d"] = float(np.exp(rng.uniform(*np.log(spec["soft_d"]))))
    return th


def perturb(spec, th, rng, rel):
    """Local move: every continuous parameter by N(0, rel x interval width), a switch flipped with probability rel."""
    out = dict(th)
    for k, (_, lo, hi) in spec["params"].items():
        out[k] = float(np.clip(th[k] + rng.randn() * rel * (hi - lo), lo, hi))
    for k in spec["switches"]:
        if rng.rand() < rel:
            out[k] = 1 - int(th[k])
    out["soft_k"] = float(np.clip(th["soft_k"] * np.exp(rng.randn() * rel * 2), *spec["soft_k"]))
    out["soft_d"] = float(np.clip(th["soft_d"] * np.exp(rng.randn() * rel * 2), *spec["soft_d"]))
    return out


def cloud10(spec, th, rng, n=32):
    """+-10 % of every continuous entry (a zero entry: +-1 % of its interval), switches kept."""
    out = []
    for _ in range(n):
        t2 = dict(th)
        for k, (_, lo, hi) in spec["params"].items():
            t2[k] = float(np.clip(th[k] * (1.0 + rng.uniform(-0.1, 0.1)) + (abs(th[k]) < 1e-9) * rng.uniform(-0.1, 0.1) * 0.1 * (hi - lo), lo, hi))
        out.append(t2)
    return out


def distance(spec, th):
    """Normalised L2 distance from the reference point (each entry scaled by its interval width, a switched feature counts 1)."""
    d2 = sum(((th[k] - v0) / (hi - lo)) ** 2 for k, (v0, lo, hi) in spec["params"].items())
    d2 += sum(float(int(th[k]) != int(v0)) for k, (v0, _) in spec["switches"].items())
    return float(np.sqrt(d2))


def build_model(robot, th):
    """theta -> robot model table: the reference-point table with the varied entries replaced, everything else frozen (P1 / P8)."""
    from openroborl_amd import robots
    if robot == "laikago":
        kw = dict(robots.LAIKAGO_R04, **robots.laikago_theta_kwargs(th))
        ref = robots.laikago(**robots.LAIKAGO_R04)
    else:
        kw = dict(robots.MINI_CHEETAH_R02, **robots.mini_cheetah_theta_kwargs(th))
        ref = robots.mini_cheetah(**robots.MINI_CHEETAH_R02)
    m = robots.ROBOTS[robot](**kw)
    frozen = ["kp", "kd", "init_motor_angles", "motor_dir", "motor_offset", "joint_of_motor", "init_pos", "init_quat", "toe_radius", "fall_radius"]
    for key in frozen + (["shank_radius", "shank_pos"] if robot == "laikago" else []):
        assert np.array_equal(np.asarray(m[key]), np.asarray(ref[key])), key          # the reference's constants and the frozen entries
    return m


# ---- probes ------------------------------------------------------------------------------------------------------------------------------
def _summ(alive, length, ret, first_reason, tsum, tcnt, dvx, advx, k, R, steps):
    """Per-candidate summaries from per-robot numpy arrays."""
    out = []
    for c in range(k):
        sl = slice(c * R, (c + 1) * R)
        n = max(float(tcnt[sl].sum()), 1.0)
        fr = first_reason[sl]
        out.append({"F": float(alive[sl].mean()), "len": float(length[sl].mean()), "R": float((ret[sl] / np.maximum(length[sl], 1)).mean()),
                    "J": float(ret[sl].mean() / steps), "terms": {t: float(tsum[sl, i].sum() / n) for i, t in enumerate(TERMS)},
                    "dvx": float(dvx[sl].sum() / n), "advx": float(advx[sl].sum() / n),
                    "fall": int(((fr & 1) != 0).sum()), "root_pos": int(((fr & 2) != 0).sum()), "root_rot": int(((fr & 4) != 0).sum()),
                    "non_finite": int(((fr & 16) != 0).sum())})
    return out


class HipProbe(object):
    """One env per (policy, group of <= 32 candidates): the candidates of a group live in the robot-type slots of the device table
    (ORR_MAX_ROBOT_TYPES = 32; robot i is of type i // R), so one launch steps all of them: 32 x 128 robots = one full 4096-robot launch,
    which costs the same 0.21 ms as a 512-robot one."""
    SLOTS = 32

    def __init__(self, robot, robots_per_candidate, seed=1, config_over=None, terms_every=4):
        import torch
        self.torch = torch
        self.robot = robot
        self.R = int(robots_per_candidate)
        self.seed = seed
        self.config_over = dict(config_over or {})
        self.terms_every = terms_every
        self._policies = {}

    def _policy(self, pol, device):
        from openroborl_amd import policy as polmod, ppo
        if pol not in self._policies:
            params = polmod.load_parameters(os.path.join(GOLDEN, "policy_%s.npz" % pol))
            self._policies[pol] = ppo.ActorCritic(device, params=params).enable_fused()
        return self._policies[pol]

    def run_group(self, pol, thetas, steps=600, robots_per_candidate=None, seed=None):
        import ctypes as C
        import policy_probe
        from openroborl_amd import _lib, robots
        from openroborl_amd.env import VecQuadrupedEnv
        torch = self.torch
        k, R = len(thetas), int(robots_per_candidate or self.R)
        n = k * R
        env = VecQuadrupedEnv(num_robot=n, seed=self.seed if seed is None else seed, robot=self.robot, motion_file=clip_of(pol), mode="test",
                              enable_randomizer=False, auto_reset=False, config_overrides=self.config_over)
        models = [build_model(self.robot, th) for th in thetas]
        for t, m in enumerate(models):
            _lib.check(env.L.orr_set_model(env.h, t, C.byref(robots.to_struct(m))), env.L)
        typ = torch.arange(n, device=env.device, dtype=torch.int32) // R
        env.field_int("ROBOT_TYPE")[:, 0] = typ
        mu = torch.tensor([float(m["foot_friction"]) for m in models], device=env.device)
        env.field("FOOT_MU")[:, 0] = mu[typ.long()]
        model = self._policy(pol, env.device)
        obs = env.reset()
        dev = env.device
        alive = torch.ones(n, dtype=torch.bool, device=dev)
        length = torch.zeros(n, device=dev)
        ret = torch.zeros(n, device=dev)
        tsum = torch.zeros(n, 5, device=dev)
        tcnt = torch.zeros(n, device=dev)
        dvx = torch.zeros(n, device=dev)
        advx = torch.zeros(n, device=dev)
        first_reason = torch.zeros(n, dtype=torch.int32, device=dev)
        reason_f = env.field_int("DONE_REASON")[:, 0]
        for s in range(steps):
            act, _, _ = model.act(obs, deterministic=True)
            probe = self.terms_every and s % self.terms_every == 0
            if probe:
                rp, rv = env.field("REF_POSE").clone(), env.field("REF_VEL").clone()
            obs, rew, done, _ = env.step(act.contiguous())
            a = alive.float()
            ret += rew * a
            length += a
            if probe:
                tsum += policy_probe.reward_terms(torch, env, rp, rv, rew) * a[:, None]
                d = env.field("LINVEL")[:, 0] - env.field("REF_VEL")[:, 0]
                dvx += d * a
                advx += d.abs() * a
                tcnt += a
            failed = done.bool() & ((reason_f & ~8) != 0)
            first_reason = torch.where(alive & failed, reason_f, first_reason)
            alive &= ~failed
            if s % 50 == 49 and not bool(alive.any()):
                break
        out = _summ(alive.float().cpu().numpy(), length.cpu().numpy(), ret.cpu().numpy(), first_reason.cpu().numpy(), tsum.cpu().numpy(),
                    tcnt.cpu().numpy(), dvx.cpu().numpy(), advx.cpu().numpy(), k, R, steps)
        env.close()
        return out


class OracleProbe(object):
    """The same protocol on the float64 CPU oracle (tests / toy sizes only)."""
    SLOTS = 4

    def __init__(self, robot, robots_per_candidate, seed=1, config_over=None, terms_every=4):
        self.robot = robot
        self.R = int(robots_per_candidate)
        self.seed = seed
        self.config_over = dict(config_over or {})

    def run_group(self, pol, thetas, steps=600, robots_per_candidate=None, seed=None):
        from openroborl_amd import _abi, config, motion
        from tests import oracle_lib as ol
        k, R = len(thetas), int(robots_per_candidate or self.R)
        n = k * R
        W = np.load(os.path.join(GOLDEN, "policy_%s.npz" % pol))
        w = {kk: W[kk].astype(np.float64) for kk in W.files}
        cfg = config.make_config(n, sim_params=config.load_sim_params(None), mode="test", enable_randomizer=False,
                                 seed=self.seed if seed is None else seed, num_procs=1, auto_reset=False)
        for kk, v in self.config_over.items():
            setattr(cfg, kk, type(getattr(cfg, kk))(v))
        models = [build_model(self.robot, th) for th in thetas] + [None] * (_abi.MAX_ROBOT_TYPES - k)
        typ = np.arange(n, dtype=np.int32) // R
        orc = ol.OracleEnv(cfg, models, [motion.MotionClip(clip_of(pol))], n, robot_type=typ, clip_id=np.zeros(n, dtype=np.int32), threads=8)
        mu = np.array([m["foot_friction"] for m in models[:k]])[typ]
        orc.field("FOOT_MU")[:, 0] = mu
        obs = orc.reset()
        orc.field("FOOT_MU")[:, 0] = mu
        alive = np.ones(n, dtype=bool)
        length, ret, tcnt, dvx, advx = np.zeros(n), np.zeros(n), np.zeros(n), np.zeros(n), np.zeros(n)
        tsum = np.zeros((n, 5))
        reasons = np.zeros(n, dtype=int)
        for s in range(steps):
            h = np.maximum(obs @ w["model__pi_fc0__w_0"] + w["model__pi_fc0__b_0"], 0.0)
            h = np.maximum(h @ w["model__pi_fc1__w_0"] + w["model__pi_fc1__b_0"], 0.0)
            a = np.clip(h @ w["model__pi__w_0"] + w["model__pi__b_0"], -2 * np.pi, 2 * np.pi)
            obs, rew, done = orc.step(a)
            length += alive
            ret += rew * alive
            tsum += orc.terms * alive[:, None]
            d = orc.field("LINVEL")[:, 0] - orc.field("REF_VEL")[:, 0]
            dvx += d * alive
            advx += np.abs(d) * alive
            tcnt += alive
            reason = orc.field("DONE_REASON")[:, 0].astype(int)
            failed = done & ((reason & ~8) != 0)
            reasons = np.where(alive & failed, reason, reasons)
            alive &= ~failed
            if not alive.any():
                break
        orc.close()
        return _summ(alive.astype(float), length, ret, reasons, tsum, tcnt, dvx, advx, k, R, steps)


def make_probe(args, robot):
    cls = HipProbe if args.backend == "hip" else OracleProbe
    return cls(robot, args.robots, config_over=constants_set(args.constants))


def constants_set(name):
    from openroborl_amd import config
    if name in (None, "default"):
        return {}
    if name == "lib":
        return dict(config.BULLET_LIBRARY_DEFAULTS)
    if name == "pyb":
        return dict(config.PYBULLET_REMEMBERED)
    raise ValueError(name)


def evaluate(probe, thetas, policies, steps=600, robots_per_candidate=None, seed=None):
    """-> per candidate {policy: result}; candidates in groups of probe.SLOTS."""
    res = [dict() for _ in thetas]
    for g0 in range(0, len(thetas), probe.SLOTS):
        grp = thetas[g0:g0 + probe.SLOTS]
        for pol in policies:
            for i, o in enumerate(probe.run_group(pol, grp, steps, robots_per_candidate, seed)):
                res[g0 + i][pol] = o
    return res


def min_j(r, policies):
    return min(r[p]["J"] for p in policies)


def min_f(r, policies):
    return min(r[p]["F"] for p in policies)


def min_r(r, policies):
    return min(r[p]["R"] for p in policies)


def brief(r, policies):
    return "  ".join("%s F %.3f J %.3f R %.3f dvx %+.2f" % (p.replace("laikago_", "").replace("minicheetah_", "mc_"), r[p]["F"], r[p]["J"], r[p]["R"], r[p]["dvx"])
                     for p in policies)


# ---- one run of P2-P3 (+ the once-only hold-out of P4) ---------------------------------------------------------------------------------------
def run(args):
    spec = spec_of(args.robot, args.freeze_geometry, args.freeze_com)
    probe = make_probe(args, args.robot)
    fit, holdout = list(args.fit), list(args.holdout or [])
    assert not set(fit) & set(holdout)
    rng = np.random.RandomState(args.seed)
    acc_f = spec["accept"]
    t0 = time.time()
    budget = args.minutes * 60.0
    res = {"protocol": __doc__.split("==== PROTOCOL")[1].split("usage:")[0].strip(), "robot": args.robot, "params": spec["params"],
           "switches": spec["switches"], "robots": args.robots, "steps": args.steps, "backend": args.backend, "fit": fit, "holdout": holdout,
           "seed": args.seed, "constants": args.constants, "accept_F": acc_f, "freeze_geometry": bool(args.freeze_geometry), "freeze_com": bool(args.freeze_com)}
    if args.backend == "hip":
        from openroborl_amd import _lib
        res["source_hash"] = _lib.library_hash()
    base = reference_theta(spec)
    r0 = evaluate(probe, [base], fit, args.steps)[0]
    res["reference_point_fit"] = r0
    print("reference point (fit policies only): " + brief(r0, fit), flush=True)
    cands = []

    def run_batch(thetas, tag):
        for th, r in zip(thetas, evaluate(probe, thetas, fit, args.steps)):
            cands.append({"theta": th, "fit": r, "J": min_j(r, fit), "F": min_f(r, fit), "R": min_r(r, fit), "dist": distance(spec, th), "stage": tag})

    def status(tag):
        best = max(cands, key=lambda c: c["J"])
        print("%s: %d candidates, best min-J %.3f (min-F %.3f), accepted %d, %.0f s" % (
            tag, len(cands), best["J"], best["F"], sum(c["F"] >= acc_f for c in cands), time.time() - t0), flush=True)

    nb = 0
    while time.time() - t0 < 0.3 * budget or not cands:          # P2 stage 1 (at least one batch, whatever the budget)
        run_batch([random_theta(spec, rng, (nb + i) % 2) for i in range(max(16, probe.SLOTS))], "random")
        nb += 1
        if nb % 16 == 0:
            status("random")
    gen = 0
    while time.time() - t0 < budget:                # P2 stage 2
        frac = min(1.0, (time.time() - t0 - 0.3 * budget) / max(0.7 * budget, 1e-9))
        rel = 0.15 * (1.0 - frac) + 0.03 * frac
        top = sorted(cands, key=lambda c: c["J"], reverse=True)[:16]
        run_batch([perturb(spec, top[rng.randint(len(top))]["theta"], rng, rel) for _ in range(max(16, probe.SLOTS))], "local")
        gen += 1
        if gen % 16 == 0:
            status("local (step %.3f)" % rel)
    accepted = sorted((c for c in cands if c["F"] >= acc_f), key=lambda c: c["J"], reverse=True)
    res["search"] = {"candidates": len(cands), "random": sum(c["stage"] == "random" for c in cands), "accepted": len(accepted),
                     "elapsed_s": time.time() - t0, "top_by_J": sorted(cands, key=lambda c: c["J"], reverse=True)[:24]}
    print("search done: %d candidates, %d accepted" % (len(cands), len(accepted)), flush=True)
    if args.dump_all:
        with gzip.open(args.dump_all, "wt") as f:
            for c in cands:
                f.write(json.dumps({"theta": c["theta"], "fit": {p: {k: c["fit"][p][k] for k in ("F", "len", "J", "R", "dvx", "advx", "terms")} for p in fit},
                                    "dist": c["dist"], "stage": c["stage"]}) + "\n")
    # P3: re-evaluate the top 16 accepted alone, bigger and on another env seed; walk them by the re-evaluated min-J; cloud check
    chosen, verdict = None, "negative"
    short = accepted[:16]
    big = 4 * args.robots
    for c, r in zip(short, evaluate(probe, [c["theta"] for c in short], fit, args.steps, big, 2) if short else []):
        c["recheck"] = {"fit": r, "J": min_j(r, fit), "F": min_f(r, fit), "R": min_r(r, fit), "robots": big, "env_seed": 2}
    res["shortlist"] = short
    walk = sorted((c for c in short if c["recheck"]["F"] >= acc_f), key=lambda c: c["recheck"]["J"], reverse=True)
    for c in walk:
        sc = [min_f(r, fit) for r in evaluate(probe, cloud10(spec, c["theta"], rng), fit, args.steps)]
        c["robustness"] = {"mean_min_F": float(np.mean(sc)), "min_min_F": float(np.min(sc)), "n": len(sc)}
        print("cloud of the shortlisted candidate with re-evaluated min-J %.3f (distance %.2f): mean min-F %.3f, min %.3f" % (
            c["recheck"]["J"], c["dist"], np.mean(sc), np.min(sc)), flush=True)
        if np.mean(sc) >= 0.8:
            chosen, verdict = c, "accepted"
            break
    if chosen is None:
        chosen = dict(max(cands, key=lambda c: c["J"]))
    # cross-check only: what the same walk would have taken under R (reward per step while alive) in place of J
    alt = sorted((c for c in short if c["recheck"]["F"] >= acc_f), key=lambda c: c["recheck"]["R"], reverse=True)
    res["choice_under_R_instead_of_J"] = ({"theta": alt[0]["theta"], "recheck": alt[0]["recheck"], "same_as_chosen": alt[0] is chosen} if alt else None)
    n_fin = 1024 if args.backend == "hip" else args.robots
    chosen["fit_final"] = {"seed_%d" % s: evaluate(probe, [chosen["theta"]], fit, args.steps, n_fin, s)[0] for s in (1, 2)}
    res.update(chosen=chosen, verdict=verdict, elapsed_s=time.time() - t0)
    print("CHOSEN (%s): distance %.2f  %s" % (verdict, chosen["dist"], brief(chosen["fit_final"]["seed_1"], fit)), flush=True)
    print("  theta: " + json.dumps(chosen["theta"]), flush=True)
    if holdout:                                     # P4: ONCE, whatever it is
        chosen["holdout"] = {"seed_%d" % s: evaluate(probe, [chosen["theta"]], holdout, args.steps, n_fin, s)[0] for s in (1, 2)}
        res["transfers"] = bool(verdict == "accepted" and all(chosen["holdout"][s][p]["F"] >= 0.5 for s in chosen["holdout"] for p in holdout))
        print("  HOLD-OUT: " + brief(chosen["holdout"]["seed_1"], holdout) + "   transfers: %s" % res["transfers"], flush=True)
    res["elapsed_s"] = time.time() - t0
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)


# ---- P4 driver: six children side by side ------------------------------------------------------------------------------------------------
def splits():
    out = []
    for fit in itertools.combinations(LAIKAGO_POLICIES, 2):
        out.append((list(fit), [p for p in LAIKAGO_POLICIES if p not in fit]))
    return out


def cv(args):
    """Spawns the six runs as child processes (this process never touches the GPU), waits, collects."""
    os.makedirs(args.outdir, exist_ok=True)
    procs = []
    todo = [(i, s) for i, s in enumerate(splits()) if args.only is None or i in args.only]
    for i, (fit, hold) in todo:
        out = os.path.join(args.outdir, "split%d.json" % i)
        cmd = [sys.executable, os.path.abspath(__file__), "run", "--robot", "laikago", "--fit"] + fit + ["--holdout"] + hold + [
            "--minutes", str(args.minutes), "--robots", str(args.robots), "--steps", str(args.steps), "--seed", str(args.seed_base + i), "--backend", args.backend,
            "--constants", args.constants, "--out", out, "--dump-all", os.path.join(args.outdir, "split%d_candidates.jsonl.gz" % i)] + (
            ["--freeze-geometry"] if args.freeze_geometry else []) + (["--freeze-com"] if args.freeze_com else [])
        log = open(os.path.join(args.outdir, "split%d_log.txt" % i), "w")
        procs.append((i, subprocess.Popen(cmd, stdout=log, stderr=subprocess.STDOUT), log))
        if args.sequential:      # full 4096-robot launches fill the GPU: side by side gains nothing
            while procs[-1][1].poll() is None:
                time.sleep(20)
                print("cv: split %d running, log %d bytes" % (i, os.path.getsize(log.name)), flush=True)
    t0 = time.time()
    while any(p.poll() is None for _, p, _ in procs):
        time.sleep(30)
        print("cv: %.0f s, running %s" % (time.time() - t0, [i for i, p, _ in procs if p.poll() is None]), flush=True)
    rc = 0
    for i, p, log in procs:
        log.close()
        if p.returncode != 0:
            rc = 1
            print("split %d FAILED (rc %d)" % (i, p.returncode), flush=True)
    collect(args)
    return rc


def collect(args):
    recs = []
    for i, (fit, hold) in enumerate(splits()):
        path = os.path.join(args.outdir, "split%d.json" % i)
        if os.path.exists(path):
            r = json.load(open(path))
            r["search"].pop("top_by_J", None)
            r.pop("shortlist", None)
            r["split"] = i
            recs.append(r)
    out = {"protocol": recs[0]["protocol"] if recs else None, "splits": recs, "table": table_rows(recs)}
    with open(os.path.join(args.outdir, "cv.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(fmt_table(out["table"]), flush=True)


def table_rows(recs):
    rows = []
    for r in recs:
        ch = r["chosen"]
        row = {"split": r["split"], "fit": r["fit"], "holdout": r["holdout"], "verdict": r["verdict"], "transfers": r.get("transfers"),
               "candidates": r["search"]["candidates"], "accepted": r["search"]["accepted"], "dist": ch["dist"], "cells": {}}
        for p in LAIKAGO_POLICIES:
            src = ch["fit_final"] if p in r["fit"] else ch.get("holdout", {})
            vals = [src[s][p] for s in sorted(src) if p in src[s]]
            if vals:
                row["cells"][p] = {"role": "fit" if p in r["fit"] else "held out", "F": float(np.mean([v["F"] for v in vals])),
                                   "J": float(np.mean([v["J"] for v in vals])), "R": float(np.mean([v["R"] for v in vals])),
                                   "dvx": float(np.mean([v["dvx"] for v in vals]))}
        rows.append(row)
    return rows


def fmt_table(rows):
    lines = ["split  fit set -> held out                 " + "  ".join("%-26s" % p for p in LAIKAGO_POLICIES) + "  verdict / transfers, dist, accepted / candidates",
             "                                            " + "  ".join("%-26s" % "role  F     J     R" for _ in LAIKAGO_POLICIES)]
    for r in rows:
        cells = []
        for p in LAIKAGO_POLICIES:
            c = r["cells"].get(p)
            cells.append("%-26s" % ("%-8s %.3f %.3f %.3f" % (c["role"], c["F"], c["J"], c["R"]) if c else "-"))
        lines.append("%d      %-36s " % (r["split"], "+".join(x.replace("laikago_", "") for x in r["fit"]) + " -> " + "+".join(x.replace("laikago_", "") for x in r["holdout"]))
                     + "  ".join(cells) + "  %s / %s, %.2f, %d / %d" % (r["verdict"], r["transfers"], r["dist"], r["accepted"], r["candidates"]))
    return "\n".join(lines)


# ---- P7: smallest table ----------------------------------------------------------------------------------------------------------------------
def minimal(args):
    rec = json.load(open(args.record))
    robot, fit = rec["robot"], rec["fit"]
    spec = spec_of(robot, rec.get("freeze_geometry", False), rec.get("freeze_com", False))
    args.constants = rec.get("constants", args.constants)
    probe = make_probe(args, robot)
    acc_f = spec["accept"]
    base = reference_theta(spec)
    th = rec["chosen"]["theta"]
    n = 1024 if args.backend == "hip" else args.robots
    keys = [k for k in list(spec["params"]) + list(spec["switches"]) if th[k] != base[k]]

    def reverted(theta, k):
        t2 = dict(theta)
        t2[k] = base[k]
        if k == "soft":
            t2["soft_k"], t2["soft_d"] = base["soft_k"], base["soft_d"]
        return t2

    def ev(thetas):
        a = evaluate(probe, thetas, fit, args.steps, n, 1)
        b = evaluate(probe, thetas, fit, args.steps, n, 2)
        return [{p: {k: 0.5 * (x[p][k] + y[p][k]) for k in ("F", "J", "R", "len", "dvx")} for p in fit} for x, y in zip(a, b)]
    t0 = time.time()
    r_ch = ev([th])[0]
    j_ch = min_j(r_ch, fit)
    out = {"of": args.record, "robot": robot, "fit": fit, "robots": n, "env_seeds": [1, 2], "chosen": {"fit": r_ch, "min_J": j_ch, "dist": distance(spec, th)},
           "tolerance_J": 0.01, "accept_F": acc_f}
    print("chosen: min-J %.4f  %s" % (j_ch, brief(r_ch, fit)), flush=True)
    single = ev([reverted(th, k) for k in keys])
    out["single_reverted"] = {k: {"fit": r, "min_J": min_j(r, fit), "min_F": min_f(r, fit)} for k, r in zip(keys, single)}
    for k in keys:
        s = out["single_reverted"][k]
        print("  %-14s alone put back: min-J %.4f (%+.4f)  min-F %.3f" % (k, s["min_J"], s["min_J"] - j_ch, s["min_F"]), flush=True)
    cur, left, path = dict(th), list(keys), []
    while left:
        trial = [reverted(cur, k) for k in left]
        rs = ev(trial)
        ok = [i for i, r in enumerate(rs) if min_f(r, fit) >= acc_f and min_j(r, fit) >= j_ch - 0.01]
        if not ok:
            break
        b = max(ok, key=lambda i: min_j(rs[i], fit))
        cur = trial[b]
        path.append({"reverted": left[b], "min_J": min_j(rs[b], fit), "min_F": min_f(rs[b], fit), "fit": rs[b]})
        print("  greedy: %-14s back to the reference point -> min-J %.4f min-F %.3f (%d entries still moved)" % (left[b], min_j(rs[b], fit), min_f(rs[b], fit), len(left) - 1), flush=True)
        left.pop(b)
    fin = ev([cur])[0]
    # what each entry that stays moved is worth, at the end point
    worth = ev([reverted(cur, k) for k in left]) if left else []
    out["minimal"] = {"path": path, "still_moved": left, "theta": cur, "dist": distance(spec, cur), "fit": fin, "min_J": min_j(fin, fit), "min_F": min_f(fin, fit),
                      "effect_of_each_moved_entry": {k: {"min_J_if_put_back": min_j(r, fit), "min_F_if_put_back": min_f(r, fit)} for k, r in zip(left, worth)}}
    out["elapsed_s"] = time.time() - t0
    print("minimal table: %d entries differ from the reference point: %s; distance %.2f (chosen %.2f); %s" % (
        len(left), left, distance(spec, cur), distance(spec, th), brief(fin, fit)), flush=True)
    for k in left:
        e = out["minimal"]["effect_of_each_moved_entry"][k]
        print("  %-14s = %s: put back -> min-J %.4f min-F %.3f" % (k, cur[k] if k != "soft" else (cur["soft_k"], cur["soft_d"]), e["min_J_if_put_back"], e["min_F_if_put_back"]), flush=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)


# ---- P5: engine constants ----------------------------------------------------------------------------------------------------------------
def constants(args):
    import policy_probe
    from openroborl_amd import config, _lib
    sets = {"LIB": dict(config.BULLET_LIBRARY_DEFAULTS), "PYB": dict(config.PYBULLET_REMEMBERED)}
    pols = [(p, clip_of(p), "laikago") for p in LAIKAGO_POLICIES] + [("minicheetah_trot", "minicheetah_trot", "mini_cheetah")]
    cells = {}
    for name, co in sets.items():
        for pol, clip, robot in pols:
            rs = []
            for seed in (1, 2):
                n, steps = args.robots, 600
                o = policy_probe.run(pol, clip, robot, n, seed, config_over=co, raw=True)
                ln = o["_raw"]["len"]
                rs.append({"F": o["finished"], "R": o["reward_per_step"], "J": float(o["return_per_nominal_step"]), "len": float(ln.mean())})
            cells[(name, pol)] = {k: float(np.mean([r[k] for r in rs])) for k in rs[0]}
            print("%-4s %-17s F %.3f  J %.3f  R %.3f  len %.1f" % (name, pol, cells[(name, pol)]["F"], cells[(name, pol)]["J"], cells[(name, pol)]["R"], cells[(name, pol)]["len"]), flush=True)
    lk = LAIKAGO_POLICIES
    mean_j = {s: float(np.mean([cells[(s, p)]["J"] for p in lk])) for s in sets}
    loss = {p: {k: cells[("LIB", p)][k] - cells[("PYB", p)][k] for k in ("F", "J")} for p, _, _ in pols}
    a_pref = mean_j["PYB"] >= mean_j["LIB"]
    a_valid = loss["minicheetah_trot"]["F"] <= 0.02 and loss["minicheetah_trot"]["J"] <= 0.02
    b_pref = cells[("PYB", "minicheetah_trot")]["J"] >= cells[("LIB", "minicheetah_trot")]["J"]
    b_valid = all(loss[p]["F"] <= 0.02 and loss[p]["J"] <= 0.02 for p in lk)
    adopt = bool(a_pref and a_valid and b_pref and b_valid)
    out = {"rule": __doc__.split("P5.")[1].split("P6.")[0].strip(), "sets": sets, "robots": args.robots, "env_seeds": [1, 2], "source_hash": _lib.library_hash(),
           "cells": {"%s/%s" % k: v for k, v in cells.items()}, "laikago_mean_J": mean_j, "loss_LIB_minus_PYB": loss,
           "direction_A": {"preferred_on_laikago": bool(a_pref), "validates_on_mini_cheetah": bool(a_valid)},
           "direction_B": {"preferred_on_mini_cheetah": bool(b_pref), "validates_on_laikago": bool(b_valid)}, "adopt_PYB": adopt}
    print("direction A: preferred on the Laikago %s (mean J %.4f vs %.4f), validates on the mini-cheetah %s" % (a_pref, mean_j["PYB"], mean_j["LIB"], a_valid))
    print("direction B: preferred on the mini-cheetah %s, validates on the Laikago %s" % (b_pref, b_valid))
    print("ADOPT PYBULLET_REMEMBERED: %s" % adopt, flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)

    def common(p):
        p.add_argument("--freeze-geometry", action="store_true", help="P9: hip_x / hip_y frozen at the values the turning clip pins (Laikago)")
        p.add_argument("--freeze-com", action="store_true", help="EXPLORATION, ships nothing: com_x frozen at the clip-pinned 0.00 too (with hip_x / hip_y)")
        p.add_argument("--robots", type=int, default=128)
        p.add_argument("--steps", type=int, default=600)
        p.add_argument("--backend", default="hip", choices=["hip", "oracle"])
        p.add_argument("--constants", default="default", choices=["default", "lib", "pyb"], help="orr_config solver constants: make_config's defaults, or a named set")
    p = sub.add_parser("run"); common(p)
    p.add_argument("--robot", default="laikago", choices=list(SPECS))
    p.add_argument("--fit", nargs="+", required=True)
    p.add_argument("--holdout", nargs="*", default=[])
    p.add_argument("--minutes", type=float, default=12.0)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--out", required=True)
    p.add_argument("--dump-all", default=None)
    p = sub.add_parser("cv"); common(p)
    p.add_argument("--minutes", type=float, default=12.0)
    p.add_argument("--outdir", default=os.path.join(ROOT, "gpurun_out", "r06cv"))
    p.add_argument("--only", type=int, nargs="*", default=None, help="split numbers to run (default: all six)")
    p.add_argument("--sequential", action="store_true", help="one child after another instead of side by side")
    p.add_argument("--seed-base", type=int, default=100, help="split i searches with random seed seed_base + i (the recorded cross-validation: 100; its replication: 500)")
    p = sub.add_parser("collect"); common(p)
    p.add_argument("--outdir", default=os.path.join(ROOT, "gpurun_out", "r06cv"))
    p = sub.add_parser("minimal"); common(p)
    p.add_argument("--record", required=True)
    p.add_argument("--out", required=True)
    p = sub.add_parser("constants")
    p.add_argument("--robots", type=int, default=1024)
    p.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06_constants_rule.json"))
    args = ap.parse_args()
    return {"run": run, "cv": cv, "collect": collect, "minimal": minimal, "constants": constants}[args.cmd](args)


if __name__ == "__main__":
    sys.exit(main() or 0)
