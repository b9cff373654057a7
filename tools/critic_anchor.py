#!/usr/bin/env python3
"""A second PyBullet-derived anchor for the physics (row C): the CRITICS inside the reference's shipped policy zips.

The value heads of policies/laikago_pace.zip and minicheetah_trot.zip were fitted (TD(lambda), gamma = 0.95, run.py:113,120) to the rewards
their policies collected IN PYBULLET.  Under the same policy, V(s_t) predicts the discounted return G_t = sum_k gamma^k r_{t+k} - so running
the shipped policy on THIS engine and comparing G_t with the shipped critic's V(s_t) says how the imitation reward this engine produces
compares with what PyBullet produced, state by state: (1 - gamma) * mean V is the reward per step the policy earned in PyBullet.
Independent of the survival criterion that tools/mc_identify.py optimised.  usage (GPU box): python tools/critic_anchor.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
GAMMA = 0.95


def run(robot, policy_npz, n=1024, mode="train", model_override=None, stochastic=True, seed=3):
    import ctypes as C
    import torch
    from openroborl_amd import _lib, policy as pol, ppo, robots
    from openroborl_amd.env import VecQuadrupedEnv
    clip = {"laikago": "laikago_pace", "mini_cheetah": "minicheetah_trot"}[robot]
    # the critic was trained in train mode (randomiser on, policy noise 0.125); the 20-step curriculum start is lifted so that
    # discounted returns are not truncated by the time limit
    env = VecQuadrupedEnv(num_robot=n, seed=seed, robot=robot, motion_file=clip, mode=mode, enable_randomizer=(mode == "train"), auto_reset=True,
                          config_overrides=dict(ep_len_start=600))
    if model_override is not None:
        t = robots.ROBOT_TYPE_ID[robot]
        _lib.check(env.L.orr_set_model(env.h, t, C.byref(robots.to_struct(model_override))), env.L)
        env.field("FOOT_MU")[:] = float(model_override["foot_friction"])
    model = ppo.ActorCritic(env.device, params=pol.load_parameters(os.path.join(ROOT, "tests", "golden", policy_npz))).enable_fused()
    g = torch.Generator(device=env.device).manual_seed(0)
    obs = env.reset()
    T = 400
    R, V, D = [], [], []
    for _ in range(T):
        act, _, val = model.act(obs, deterministic=not stochastic, generator=g)
        V.append(val.clone())
        obs, rew, done, _ = env.step(act.contiguous())
        R.append(rew.clone()); D.append(done.bool().clone())
    R, V, D = torch.stack(R), torch.stack(V), torch.stack(D)
    # discounted return inside the episode (0 beyond its end, like the critic's targets)
    G = torch.zeros_like(R)
    acc = torch.zeros(n, device=env.device)
    for k in range(T - 1, -1, -1):
        acc = R[k] + GAMMA * acc * (~D[k]).float()
        G[k] = acc
    # only states whose 100-step horizon lies inside the recorded window (gamma^100 = 0.6 %)
    sel = slice(0, T - 100)
    v, gret, r = V[sel].reshape(-1), G[sel].reshape(-1), R[sel].reshape(-1)
    out = {"robot": robot, "mode": mode, "stochastic": stochastic, "states": int(v.numel()),
           "reward_per_step_here": float(r.mean()), "mean_G_here": float(gret.mean()), "mean_V_shipped_critic": float(v.mean()),
           "reward_per_step_implied_by_critic_(1-gamma)V": float((1 - GAMMA) * v.mean()),
           "rmse_V_minus_G": float(((v - gret) ** 2).mean().sqrt()), "std_G": float(gret.std()),
           "corr_V_G": float(torch.corrcoef(torch.stack([v, gret]))[0, 1]), "episode_ends_per_robot_step": float(D.float().mean())}
    env.close()
    return out


def main():
    import mc_identify as mi
    rows = []
    for stochastic, mode in ((True, "train"), (False, "test")):
        rows.append(dict(run("laikago", "policy_laikago_pace.npz", mode=mode, stochastic=stochastic), table="robots.py"))
        rows.append(dict(run("mini_cheetah", "policy_minicheetah_trot.npz", mode=mode, stochastic=stochastic), table="robots.py (identified, round 3)"))
        old = mi.build_model(np.array([mi.PARAMS[k][0] for k in mi.NAMES]))
        rows.append(dict(run("mini_cheetah", "policy_minicheetah_trot.npz", mode=mode, stochastic=stochastic, model_override=old), table="round-2 table"))
    for r in rows:
        print(json.dumps(r))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "critic_anchor.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
