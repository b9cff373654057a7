"""Long soak on the GPU box (development aid): 30000 env steps x N robots with large random actions (sigma 0.3 rad), auto-reset on;
asserts that observations and rewards stay finite and prints env.stats().  usage: python tools/soak.py [robots=4096] [steps=30000] [task]
(robots > 4096 run the two-waves-per-SIMD build of the step kernel; ORR_SOAK_ANCHOR=1: Laikago toes with friction anchors = the anchor
variant of the step kernel)"""
import sys, torch
sys.path.insert(0, '.')
from openroborl_amd.env import VecQuadrupedEnv
from openroborl_amd import _abi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
TASK = sys.argv[3] if len(sys.argv) > 3 else "imitation_learning_laikago"
import os
OVER = {"laikago": {"friction_anchor": 1}} if os.environ.get("ORR_SOAK_ANCHOR") == "1" else None
env = VecQuadrupedEnv(task_name=TASK, num_robot=N, mode="train", auto_reset=True, seed=7, model_overrides=OVER)
obs = env.reset()
g = torch.Generator(device=env.device).manual_seed(0)
bad = 0; rsum = 0.0
for k in range(STEPS):
    act = obs[:, 91:103] * 0 + torch.randn(N, 12, device=env.device, generator=g) * 0.3
    obs, r, d, _ = env.step(act)
    if k % 1000 == 999:
        assert torch.isfinite(obs).all() and torch.isfinite(r).all()
        rsum += float(r.mean())
st = env.stats()
print({k: (v if not isinstance(v, dict) else v) for k, v in st.items()})
print('robots', N, 'steps', STEPS, 'task', TASK, 'friction anchors', bool(OVER), 'mean reward samples', rsum / max(STEPS // 1000, 1))
if OVER:
    print('toes holding a cached contact point at the end:', float(env.field_int("ANCHOR_VALID").float().mean()))
