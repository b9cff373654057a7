"""Where one train.py iteration spends its time (development aid, GPU box): rollout / GAE / PPO update by HIP events, and the pieces of
one minibatch update (forward, backward, optimiser) timed separately.

usage:  python tools/train_breakdown.py [robots=4096] [horizon=32] [minibatch=16384] [epochs=2] [torch-learner]
        rocprofv3 --kernel-trace --stats -- python3 tools/train_breakdown.py ... update-only      (per-kernel view of the update alone)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from openroborl_amd import learner_hip, ppo, rollout  # noqa: E402
from openroborl_amd.env import VecQuadrupedEnv  # noqa: E402

argv = [a for a in sys.argv[1:] if a not in ("update-only", "torch-learner")]
update_only = "update-only" in sys.argv
torch_learner = "torch-learner" in sys.argv
n = int(argv[0]) if len(argv) > 0 else 4096
T = int(argv[1]) if len(argv) > 1 else 32
mb = int(argv[2]) if len(argv) > 2 else 16384
epochs = int(argv[3]) if len(argv) > 3 else 2
dev = torch.device("cuda", 0)


def ms(f, reps):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


model = ppo.ActorCritic(dev, seed=0).enable_fused()
learner = ppo.PPO(model, lr=1e-4, minibatch=mb) if torch_learner else learner_hip.FusedPPO(model, lr=1e-4, minibatch=mb)
gen = torch.Generator(device=dev).manual_seed(0)
B = T * n
obs = torch.randn(B, 160, device=dev)
act = torch.randn(B, 12, device=dev) * 0.2
adv = torch.randn(B, device=dev)
ret = torch.randn(B, device=dev)
logp = torch.randn(B, device=dev) * 0.1 - 10.0

t_update = ms(lambda: learner.update(obs, act, adv, ret, old_logp=logp, epochs=epochs, generator=gen), 10 if not update_only else 30)
print(("torch learner " if torch_learner else "fused learner ") + "update: %.2f ms per iteration (%d samples, minibatch %d, %d epochs = %d optimiser steps: %.3f ms each)"
      % (t_update, B, mb, epochs, epochs * (B // mb), t_update / (epochs * (B // mb))))
if update_only:
    sys.exit(0)

# pieces of one minibatch
o, a = obs[:mb], act[:mb]


def fwd():
    with torch.no_grad():
        pm.log_prob(o, a)
        pm.value(o)


def fwd_bwd():
    lp = pm.log_prob(o, a)
    ratio = torch.exp(lp - logp[:mb])
    surr = -torch.min(ratio * adv[:mb], torch.clamp(ratio, 0.8, 1.2) * adv[:mb]).mean()
    vf = ((pm.value(o) - ret[:mb]) ** 2).mean()
    ref.opt.zero_grad(set_to_none=True)
    (surr + vf).backward()


# the autograd path (ppo.PPO), whichever learner is being measured above
ref = learner if torch_learner else ppo.PPO(ppo.ActorCritic(dev, seed=0), lr=1e-4, minibatch=mb)
pm = ref.model
t_f = ms(fwd, 50)
t_fb = ms(fwd_bwd, 50)
t_opt = ms(lambda: ref.opt.step(), 50)
t_perm = ms(lambda: (torch.randperm(B, device=dev, generator=gen), obs[torch.arange(B, device=dev)]), 20)
flops = 2.0 * mb * (2 * (160 * 512 + 512 * 256) + 256 * 13)
print("autograd path, one minibatch of %d: forward %.3f ms (%.1f TFLOP/s), forward+backward %.3f ms (%.1f TFLOP/s over 3x forward flops), Adam %.3f ms; "
      "per-epoch permutation + obs gather %.3f ms" % (mb, t_f, flops / t_f * 1e-9, t_fb, 3 * flops / t_fb * 1e-9, t_opt, t_perm))

env = VecQuadrupedEnv(task_name="imitation_learning_laikago", num_robot=n, mode="train", auto_reset=True, seed=0, device=dev)
state = {"obs": env.reset()}
for _ in range(6000):    # past the start-up transient of a fresh box (bench.py uses the same floor)
    env.step(torch.zeros(n, 12, device=dev))


def roll():
    buf = rollout.collect_rollout(env, model, T, obs=state["obs"], generator=gen)
    state["obs"] = buf["last_obs"]
    state["buf"] = buf


t_roll_eager = ms(roll, 10)
collector = rollout.GraphRollout(env, model, T)


def roll_graph():
    buf = collector.collect(state["obs"], generator=gen)
    state["obs"] = buf["last_obs"]
    state["buf"] = buf


t_roll = ms(roll_graph, 10)
buf = state["buf"]
t_env = ms(lambda: env.step(buf["actions"][0]), 100)
t_act = ms(lambda: model.act(state["obs"], noise=buf["actions"][0]), 100)
boot = model.value(state["obs"]).detach()
t_gae = ms(lambda: rollout.gae_fused(buf["rewards"], buf["vpred"], buf["dones"], 0.95, 0.95, bootstrap=boot, normalize=True, eps=1e-8), 50)
print("rollout of %d steps: %.2f ms as one hipGraph (%.3f per step), %.2f ms launched step by step; env.step alone %.3f, act alone %.3f; GAE %.3f ms"
      % (T, t_roll, t_roll / T, t_roll_eager, t_env, t_act, t_gae))
tot = t_roll + t_gae + t_update
print("iteration: %.2f ms -> %.2f M samples/s (rollout %.0f %%, update %.0f %%)" % (tot, B / tot * 1e-3, 100 * t_roll / tot, 100 * t_update / tot))
env.close()
