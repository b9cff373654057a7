"""How long is a launch of the step kernel when no robot is reset in it?  (development aid, GPU box)
4096 Laikago robots driven by a trained policy (deterministic actions) in test mode: every robot starts together and runs the full
600 steps, so the launches of steps 100..500 contain no reset at all; the same policy in train mode (20-step curriculum episodes,
~200 inline resets per launch) is the comparison.  The step kernel alone is timed (HIP events around env.step).
usage: python tools/diag/no_reset_timing.py <policy.zip> [robots=4096]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from openroborl_amd import policy as pol, ppo  # noqa: E402
from openroborl_amd.env import VecQuadrupedEnv  # noqa: E402

n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device("cuda", 0)
model = ppo.ActorCritic(dev, params=pol.load_parameters(sys.argv[1])).enable_fused()


def run(mode, lo, hi, total):
    env = VecQuadrupedEnv(task_name="imitation_learning_laikago", num_robot=n, mode=mode, auto_reset=True, seed=0, device=dev)
    warm = VecQuadrupedEnv(task_name="imitation_learning_laikago", num_robot=n, mode="train", auto_reset=True, seed=1, device=dev)
    z = torch.zeros(n, 12, device=dev)
    warm.reset()
    for _ in range(6000):          # clocks of a fresh box
        warm.step(z)
    obs = env.reset()
    ev, dones = [], torch.zeros((), dtype=torch.int64, device=dev)
    for k in range(total):
        act, _, _ = model.act(obs, deterministic=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        obs, _, d, _ = env.step(act)
        e1.record()
        if lo <= k < hi:
            ev.append((e0, e1))
            dones += d.sum()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    print("%s mode, steps %d..%d: step kernel median %.4f ms, mean %.4f; done flags per step %.2f" % (mode, lo, hi, ms[len(ms) // 2], sum(ms) / len(ms), float(dones) / len(ev)))
    env.close(); warm.close()


run("test", 100, 500, 520)
run("train", 100, 500, 520)
run("test", 100, 500, 520)
run("train", 100, 500, 520)
