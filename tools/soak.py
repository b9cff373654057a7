"""Long soak on the GPU box (development aid): 30000 env steps x 4096 robots with large random actions (sigma 0.3 rad), auto-reset on;
asserts that observations and rewards stay finite and prints env.stats().  usage: python tools/soak.py"""
import sys, torch
sys.path.insert(0, '.')
from openroborl_amd.env import VecQuadrupedEnv
from openroborl_amd import _abi
env = VecQuadrupedEnv(task_name="imitation_learning_laikago", num_robot=4096, mode="train", auto_reset=True, seed=7)
obs = env.reset()
g = torch.Generator(device=env.device).manual_seed(0)
bad = 0; rsum = 0.0
for k in range(30000):
    act = obs[:, 91:103] * 0 + torch.randn(4096, 12, device=env.device, generator=g) * 0.3
    obs, r, d, _ = env.step(act)
    if k % 1000 == 999:
        assert torch.isfinite(obs).all() and torch.isfinite(r).all()
        rsum += float(r.mean())
st = env.stats()
print({k: (v if not isinstance(v, dict) else v) for k, v in st.items()})
print('mean reward samples', rsum / 30)
