#!/usr/bin/env python3
"""Golden vectors for the rollout post-processing next to the env path:
`add_vtarg_and_adv` (agents/ppo_imitation.py:68-93) and the per-robot advantage standardisation inside
PPOImitation.learn (agents/ppo_imitation.py:329-338).

Run ONLY in the build container:   PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ppo.py

agents/ppo_imitation.py imports tensorflow / mpi4py / stable_baselines at module level, so the module cannot be imported
here.  The two pieces are plain numpy: this script parses the file with `ast`, takes the source segment of the function
(resp. of the four statements of the standardisation block) and executes THAT text in a namespace holding numpy (and a
stand-in `self` with num_robot / timesteps_per_actorbatch).  Nothing of the reference is written to the repo; only the
inputs and outputs are (tests/golden/ppo_gae.npz).

Layout of the reference's segment arrays: flat, index = step * num_robot + robot (imitation_runners.py:128-136).
Cases: num_robot = 1 (where ppo_imitation.py:88's `episode_starts[(step*num_robot+i) + (1+i)]` is the next step's flag
of the same robot) and num_robot = 3 (where it reads a neighbouring robot's flag: kept as a fixture of the quirk).
"""
import ast
import os
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/OpenRoboRL/agents/ppo_imitation.py"


def _extract():
    text = open(SRC).read()
    tree = ast.parse(text)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "add_vtarg_and_adv")
    ns = {"np": np}
    exec(compile(ast.get_source_segment(text, fn), SRC, "exec"), ns)

    block = None
    for node in ast.walk(tree):
        body = getattr(node, "body", None)
        if not isinstance(body, list):
            continue
        for k, st in enumerate(body):
            if isinstance(st, ast.Assign) and getattr(st.targets[0], "id", None) == "temp_atarg":
                block = body[k:k + 4]        # temp_atarg = ...; for (gather); for (standardise); for (scatter)
    assert block is not None and all(isinstance(b, ast.For) for b in block[1:])
    mod = ast.Module(body=block, type_ignores=[])
    code = compile(mod, SRC, "exec")

    def normalize(atarg, num_robot):
        env = {"np": np, "atarg": atarg.copy(),
               "self": types.SimpleNamespace(num_robot=num_robot, timesteps_per_actorbatch=len(atarg))}
        exec(code, env)
        return env["atarg"]
    return ns["add_vtarg_and_adv"], normalize


def make_segment(rng, T, n, p_done):
    """A segment as imitation_runners.py:128-188 would fill it: per-robot done flags, next values zeroed at episode ends
    and at the end of the segment (`last_vpred = 0.0`, :98-100,187-188)."""
    rew = rng.uniform(0, 1, (T, n)).astype(np.float32)
    vpred = rng.randn(T, n).astype(np.float32)
    done = rng.rand(T, n) < p_done
    starts = np.zeros((T, n), dtype=bool)
    starts[0] = True
    starts[1:] = done[:-1]                                  # episode_start[j] = done[j]  (:178)
    nextv = np.zeros((T, n), dtype=np.float32)
    nextv[:-1] = np.where(done[:-1], 0.0, vpred[1:])
    return rew, vpred, done, starts, nextv


def main():
    add_vtarg_and_adv, normalize = _extract()
    rng = np.random.RandomState(42)
    out = {}
    cases = [("n1_T64", 64, 1, 0.08), ("n1_T256", 256, 1, 0.03), ("n1_T7", 7, 1, 0.3), ("n3_T40", 40, 3, 0.1)]
    for name, T, n, p in cases:
        rew, vpred, done, starts, nextv = make_segment(rng, T, n, p)
        seg = {"episode_starts": starts.reshape(-1), "vpred": vpred.reshape(-1), "nextvpreds": nextv.reshape(-1),
               "rewards": rew.reshape(-1)}
        add_vtarg_and_adv(seg, n, 0.95, 0.95)               # run.py:113,120 gamma = lam = 0.95
        adv = seg["adv"].copy()
        out[name + "/rewards"], out[name + "/vpred"], out[name + "/dones"] = rew, vpred, done
        out[name + "/adv"] = adv.reshape(T, n)
        out[name + "/tdlamret"] = seg["tdlamret"].reshape(T, n)
        out[name + "/adv_normalized"] = normalize(adv, n).reshape(T, n)
    out["cases"] = np.array([c[0] for c in cases])
    np.savez_compressed(os.path.join(HERE, "ppo_gae.npz"), **out)
    print("written ppo_gae.npz:", [c[0] for c in cases])


if __name__ == "__main__":
    main()
