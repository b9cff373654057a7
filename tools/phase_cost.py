#!/usr/bin/env python3
"""Kernel time vs run-time knobs (solver iterations, randomiser) to apportion the step kernel's time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openroborl_amd import config as cfgmod
from openroborl_amd.env import VecQuadrupedEnv

orig = cfgmod.make_config
for iters, rnd in ((9, True), (1, True), (9, False), (1, False), (0, False)):
    def mk(*a, **k):
        c = orig(*a, **k)
        c.solver_iters = max(iters, 1)
        return c
    cfgmod.make_config = mk
    env = VecQuadrupedEnv(task_name="imitation_learning_laikago", num_robot=4096, mode="train", enable_randomizer=rnd, seed=0)
    if iters == 0:
        pass
    obs = env.reset()
    act = torch.zeros(4096, 12, device=env.device)
    for _ in range(50):
        env.step(act)
    torch.cuda.synchronize()
    ms = env.time_steps(act, 100) / 100
    print("solver_iters=%d randomizer=%s  kernel %.3f ms" % (max(iters, 1), rnd, ms), flush=True)
    env.close()
