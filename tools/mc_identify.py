#!/usr/bin/env python3
"""Mini-cheetah model identification against the shipped `minicheetah_trot` policy (runs on the GPU box).

The policy in the reference's policies/minicheetah_trot.zip was trained in PyBullet on the real URDF: it is the only
PyBullet-derived artefact for BASELINE configs[2].  The mini-cheetah URDF is not available here, so the inertial / collision
entries of openroborl_amd/robots.py:mini_cheetah() are hand-authored; this tool asks which values INSIDE stated plausible
intervals make the PyBullet-trained policy behave on this engine as it must have in PyBullet (walk the 600-step episode).
Never varied: the reference's control constants (robots/mini_cheetah.py:49-67), link lengths and hip positions
(trans2minicheetah.m:28-30), base and thigh masses, the Laikago table.

CRITERION (fixed before the sweep was run):
  F = fraction of 256 robots (test mode: no randomiser, 2 ms latency, 600-step limit, deterministic policy, env seed 1) that finish
      the 600-step episode;  R = mean reward per step while alive.
  1. accept candidates with F >= 0.90 whose parameters all lie inside the intervals below;
  2. among the accepted take the one CLOSEST to the round-2 table (normalised L2 distance over the varied parameters, each scaled by
     its interval width): the smallest change that explains the policy, not the best score;
  3. it must not be a knife edge: 64 random +-10 % perturbations of its varied parameters (clipped to the intervals) keep mean F >= 0.80;
  4. if nothing is accepted: report the best candidate, its F / R / reward terms, and why it still falls - a negative result.

usage: python tools/mc_identify.py [--candidates 1500] [--robots 256] [--out gpurun_out/mc_identify.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# name: (round-2 table value, low, high)  -- physically plausible intervals for a 9 kg, 0.39 m-leg robot
PARAMS = {
    "toe_m":    (0.15, 0.02, 0.25),      # toe / foot link mass [kg]
    "lo_m":     (0.064, 0.05, 0.25),     # shank mass [kg]
    "lo_com_z": (-0.061, -0.12, -0.02),  # shank COM below the knee [m] (shank length 0.18)
    "toe_r":    (0.0175, 0.010, 0.025),  # toe sphere radius [m]
    "hip_z":    (0.0, -0.02, 0.02),      # hip axis plane relative to the base COM [m]
    "foot_mu":  (1.0, 0.5, 2.0),         # toe lateral friction (x plane 1.0)
    "shank_r":  (0.012, 0.0, 0.02),      # shank contact sphere radius [m]
    "shank_at": (0.02, 0.0, 0.06),       # ... its distance below the knee [m]
    "up_com_z": (-0.02, -0.06, 0.0),     # thigh COM below the hip pitch axis [m]
    "limits":   (0.0, 0.0, 1.0),         # < 0.5: continuous joints (table); >= 0.5: approximate MIT actuator ranges
}
MIT_LIMITS = [(-1.05, 1.05), (-3.6, 1.6), (0.05, 2.77)]     # abad, hip pitch, knee (motor convention; approximate published ranges)
NAMES = list(PARAMS)


def build_model(theta):
    from openroborl_amd import robots
    p = dict(zip(NAMES, theta))
    import inspect
    src = robots.mini_cheetah
    # rebuild through _build with the varied entries replaced
    kw = dict(
        name="mini_cheetah", init_pos=[0, 0, 0.28], init_quat=[0.0, 0.0, 0.0, 1.0], init_motor_angles=[0, -0.78, 1.74] * 4,
        motor_dir=[1] * 12, motor_offset=[0.0] * 12, joint_of_motor=[3, 4, 5, 9, 10, 11, 0, 1, 2, 6, 7, 8],
        kp=[80.0] * 12, kd=[0.1, 1.0, 1.0] * 4, base_mass=3.3, base_inertia=[0.011253, 0.036203, 0.042673],
        hip_xy=[0.19, 0.049], hip_z=p["hip_z"], coxa=0.062, femur=0.209, tibia=0.18, pitch_axis=[0.0, -1.0, 0.0],
        hip_m=0.54, hip_com=[0.0, 0.036, 0.0], hip_I=[0.000381, 0.000560, 0.000444],
        up_m=0.634, up_com=[0.0, 0.016, p["up_com_z"]], up_I=[0.001983, 0.002103, 0.000408],
        lo_m=p["lo_m"], lo_com=[0.0, 0.0, p["lo_com_z"]],
        # a slender rod of the candidate's mass (the table's 0.000245 is that of a 0.064 kg, 0.21 m rod)
        lo_I=[p["lo_m"] * 0.18 ** 2 / 12.0 + 0.00007, p["lo_m"] * 0.18 ** 2 / 12.0 + 0.00007, 0.000006],
        toe_m=p["toe_m"], toe_r=p["toe_r"],
        limits=MIT_LIMITS if p["limits"] >= 0.5 else [(-1e9, 1e9)] * 3,
        chassis_half=[0.19, 0.049, 0.05], hip_r=0.04, knee_r=0.0, foot_friction=p["foot_mu"], shank_r=p["shank_r"], shank_at=p["shank_at"])
    ref = src()
    m = robots._build(**kw)
    assert np.allclose(m["kp"], ref["kp"]) and np.allclose(m["init_motor_angles"], ref["init_motor_angles"])     # control constants untouched
    return m


class Probe(object):
    def __init__(self, n, seed=1):
        import torch
        from openroborl_amd import policy as pol, ppo
        from openroborl_amd.env import VecQuadrupedEnv
        self.torch = torch
        self.env = VecQuadrupedEnv(num_robot=n, seed=seed, robot="mini_cheetah", motion_file="minicheetah_trot", mode="test",
                                   enable_randomizer=False, auto_reset=False)
        params = pol.load_parameters(os.path.join(ROOT, "tests", "golden", "policy_minicheetah_trot.npz"))
        self.model = ppo.ActorCritic(self.env.device, params=params).enable_fused()
        self.n = n

    def run(self, theta, steps=600, terms=False):
        import ctypes as C
        from openroborl_amd import _lib, robots
        torch, env = self.torch, self.env
        m = build_model(theta)
        _lib.check(env.L.orr_set_model(env.h, 1, C.byref(robots.to_struct(m))), env.L)
        env.field("FOOT_MU")[:] = float(m["foot_friction"])
        obs = env.reset()
        alive = torch.ones(self.n, dtype=torch.bool, device=env.device)
        length = torch.zeros(self.n, device=env.device)
        ret = torch.zeros(self.n, device=env.device)
        first_reason = torch.zeros(self.n, dtype=torch.int32, device=env.device)
        reason_f = env.field_int("DONE_REASON")[:, 0]
        for _ in range(steps):
            act, _, _ = self.model.act(obs, deterministic=True)
            obs, rew, done, _ = env.step(act.contiguous())
            ret += rew * alive
            length += alive.float()
            # an episode that ends by the 600-step time limit alone is a FINISHED episode; anything else (fall contact, root position /
            # rotation error, non-finite state) is a failure.  Robots keep being stepped after they failed (no auto-reset): only their
            # first failure counts
            failed = done.bool() & ((reason_f & ~8) != 0)
            first_reason = torch.where(alive & failed, reason_f, first_reason)
            alive &= ~failed
        out = {"F": float(alive.float().mean()), "len": float(length.mean()), "R": float((ret / length.clamp(min=1)).mean())}
        if terms:
            r = first_reason.cpu().numpy()
            out["reasons"] = {"finished": int(alive.sum()), "fall": int((r & 1 != 0).sum()), "root_pos": int((r & 2 != 0).sum()),
                              "root_rot": int((r & 4 != 0).sum()), "non_finite": int((r & 16 != 0).sum())}
        return out


def dist(theta, base, width):
    return float(np.sqrt((((np.asarray(theta) - base) / width) ** 2).sum()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--candidates", type=int, default=1500)
    ap.add_argument("--robots", type=int, default=256)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "mc_identify.json"))
    args = ap.parse_args()
    base = np.array([PARAMS[k][0] for k in NAMES])
    lo = np.array([PARAMS[k][1] for k in NAMES])
    hi = np.array([PARAMS[k][2] for k in NAMES])
    width = hi - lo
    probe = Probe(args.robots)
    t0 = time.time()
    res = {"criterion": __doc__.split("CRITERION")[1].split("usage:")[0].strip(), "params": {k: PARAMS[k] for k in NAMES}, "robots": args.robots}
    res["table_r02"] = probe.run(base, terms=True)
    print("round-2 table:", res["table_r02"], flush=True)
    # 1. one-at-a-time sensitivity (9 values per parameter across its interval, the others at the table)
    sens = {}
    for i, k in enumerate(NAMES):
        rows = []
        for v in ([0.0, 1.0] if k == "limits" else np.linspace(lo[i], hi[i], 9)):
            th = base.copy(); th[i] = v
            r = probe.run(th)
            rows.append({"value": float(v), **r})
        sens[k] = rows
        print("sens %-9s" % k, " ".join("%.3g:%.2f/%.0f" % (r["value"], r["F"], r["len"]) for r in rows), flush=True)
    res["sensitivity"] = sens
    # 2. random search in the box (half of the candidates uniformly, half as Gaussian clouds around the table scaled by 0.3 width)
    rng = np.random.RandomState(0)
    cands = []
    for c in range(args.candidates):
        if c % 2 == 0:
            th = lo + rng.rand(len(NAMES)) * width
        else:
            th = np.clip(base + rng.randn(len(NAMES)) * 0.3 * width, lo, hi)
        th[NAMES.index("limits")] = float(rng.rand() < 0.5)
        r = probe.run(th)
        cands.append({"theta": th.tolist(), "dist": dist(th, base, width), **r})
        if c % 100 == 99:
            acc = [x for x in cands if x["F"] >= 0.9]
            print("random %d / %d: accepted %d, best F %.3f, elapsed %.0f s" % (c + 1, args.candidates, len(acc), max(x["F"] for x in cands), time.time() - t0), flush=True)
    res["random_search"] = {"n": len(cands), "accepted": sum(x["F"] >= 0.9 for x in cands),
                            "top_by_F": sorted(cands, key=lambda x: -x["F"])[:20]}
    accepted = sorted((x for x in cands if x["F"] >= 0.9), key=lambda x: x["dist"])
    # 3. shrink the closest accepted candidates towards the table while they stay accepted (criterion 2: smallest change)
    best = None
    if accepted:
        pool = accepted[:8]
        refined = []
        for x in pool:
            th = np.array(x["theta"])
            cur = x
            for frac in (0.75, 0.5, 0.25):
                t2 = base + (th - base) * frac
                t2[NAMES.index("limits")] = th[NAMES.index("limits")]
                r = probe.run(t2)
                if r["F"] >= 0.9:
                    cur = {"theta": t2.tolist(), "dist": dist(t2, base, width), **r}
            refined.append(cur)
        refined.sort(key=lambda x: x["dist"])
        best = refined[0]
        res["refined"] = refined
    else:
        best = max(cands, key=lambda x: (x["F"], x["len"]))
    best["detail"] = probe.run(np.array(best["theta"]), terms=True)
    # 4. knife-edge check: +-10 % perturbations of the varied parameters
    th = np.array(best["theta"])
    cloud = []
    for _ in range(64):
        t2 = np.clip(th * (1.0 + rng.uniform(-0.1, 0.1, len(th))) + rng.uniform(-0.1, 0.1, len(th)) * (np.abs(th) < 1e-9) * width * 0.1, lo, hi)
        t2[NAMES.index("limits")] = th[NAMES.index("limits")]
        cloud.append(probe.run(t2)["F"])
    best["robustness"] = {"mean_F": float(np.mean(cloud)), "min_F": float(np.min(cloud)), "n": len(cloud)}
    best["params"] = dict(zip(NAMES, best["theta"]))
    res["best"] = best
    res["verdict"] = ("accepted" if best["detail"]["F"] >= 0.9 and best["robustness"]["mean_F"] >= 0.8 else "negative")
    res["elapsed_s"] = time.time() - t0
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)
    print("BEST", json.dumps(best["params"]), best["detail"], best["robustness"], res["verdict"], flush=True)


if __name__ == "__main__":
    main()
