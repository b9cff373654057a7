#!/usr/bin/env python3
"""Digest of the device state after a seeded run (development aid): two builds of the library that claim to compute the same bits
print the same line.   usage (GPU box): ORR_LIB_PATH=... ORR_ALLOW_STALE_LIB=1 python3 tools/diag/state_digest.py [robots] [steps] [task]"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openroborl_amd.env import build_env  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
task = sys.argv[3] if len(sys.argv) > 3 else "imitation_learning_laikago"
env = build_env(task, num_robot=n, mode="train", seed=7)
obs = env.reset()
g = torch.Generator(device="cuda").manual_seed(11)
h = hashlib.sha256()
for k in range(steps):
    act = 0.4 * torch.randn(n, 12, device="cuda", generator=g)
    obs, rew, done, _ = env.step(act)
    if k % 50 == 49 or k == steps - 1:
        h.update(obs.cpu().numpy().tobytes()); h.update(rew.cpu().numpy().tobytes()); h.update(done.cpu().numpy().tobytes())
sd = env.state_dict()
for key in sorted(sd):
    v = sd[key]
    if torch.is_tensor(v):
        h.update(v.cpu().numpy().tobytes())
print(os.path.basename(os.environ.get("ORR_LIB_PATH", "product")), task, n, steps, h.hexdigest()[:32])
