"""The PPO learner on the device against a float64 CPU evaluation of the same loss, gradient and
first Adam step (agents/ppo_imitation.py:156-258 loss terms; stable_baselines/common/mpi_adam.py:40-62 update rule)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _loss64(p, obs, act, adv, ret, old_logp, clip=0.2, std=0.125):
    import torch

    def mlp(net):
        h = torch.relu(obs @ p["model/%s_fc0/w:0" % net] + p["model/%s_fc0/b:0" % net])
        h = torch.relu(h @ p["model/%s_fc1/w:0" % net] + p["model/%s_fc1/b:0" % net])
        return h @ p["model/%s/w:0" % net] + p["model/%s/b:0" % net]
    var = std * std
    logp = (-0.5 * ((act - mlp("pi")) ** 2) / var - 0.5 * math.log(2.0 * math.pi * var)).sum(dim=1)
    ratio = torch.exp(logp - old_logp)
    surr = -torch.min(ratio * adv, torch.clamp(ratio, 1.0 - clip, 1.0 + clip) * adv).mean()
    vf = ((mlp("vf")[:, 0] - ret) ** 2).mean()
    return surr + vf, surr, vf


def test_one_ppo_update_matches_float64_cpu():
    import torch
    from openroborl_amd import ppo
    dev = torch.device("cuda:0")
    B = 16384                                   # engages the split-K weight-gradient path (ppo._wgrad)
    g = torch.Generator().manual_seed(0)
    obs = torch.randn(B, 160, generator=g)
    model = ppo.ActorCritic(dev, seed=4)
    with torch.no_grad():
        mu = model.mean(obs.to(dev)).cpu()
    act = mu + 0.125 * torch.randn(B, 12, generator=g)
    # old policy = current policy shifted a little, so that the ratio leaves [0.8, 1.2] for part of the batch (clip active)
    old_logp = (-0.5 * ((act - (mu + 0.04 * torch.randn(B, 12, generator=g))) ** 2) / 0.125 ** 2
                - 0.5 * math.log(2.0 * math.pi * 0.125 ** 2)).sum(dim=1)
    adv = torch.randn(B, generator=g)
    ret = torch.randn(B, generator=g)

    p64 = {k: v.detach().cpu().double().requires_grad_(True) for k, v in model.p.items()}
    loss64, surr64, vf64 = _loss64(p64, obs.double(), act.double(), adv.double(), ret.double(), old_logp.double())
    loss64.backward()

    # device: the same minibatch through the learner's own code path (custom autograd Functions, fused epilogues)
    o, a, ad, r, ol = (t.to(dev) for t in (obs, act, adv, ret, old_logp))
    logp = model.log_prob(o, a)
    ratio = torch.exp(logp - ol)
    surr = -torch.min(ratio * ad, torch.clamp(ratio, 0.8, 1.2) * ad).mean()
    vf = ((model.value(o) - r) ** 2).mean()
    (surr + vf).backward()
    clipped = ((ratio < 0.8) | (ratio > 1.2)).float().mean().item()
    assert 0.02 < clipped < 0.9, clipped
    assert abs(surr.item() - surr64.item()) < 2e-4 * max(1.0, abs(surr64.item()))
    assert abs(vf.item() - vf64.item()) < 2e-4 * max(1.0, abs(vf64.item()))
    for k in sorted(model.p):
        g32, g64 = model.p[k].grad.detach().cpu().double(), p64[k].grad
        scale = g64.abs().max().item() + 1e-12
        assert (g32 - g64).abs().max().item() < 2e-3 * scale, (k, (g32 - g64).abs().max().item(), scale)
        # cosine similarity of the whole gradient tensor
        cs = (g32 * g64).sum() / (g32.norm() * g64.norm() + 1e-30)
        assert cs.item() > 0.99999, (k, cs.item())

    # one optimiser step through PPO.update (fused Adam) vs the closed form of the first Adam step
    before = {k: v.detach().clone() for k, v in model.p.items()}
    for v in model.p.values():
        v.grad = None
    learner = ppo.PPO(model, lr=1e-4, adam_eps=1e-5, minibatch=B)
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    learner.update(o, a, ad, r, old_logp=ol, epochs=1, generator=gen)
    for k in sorted(model.p):
        g64 = p64[k].grad
        step64 = -1e-4 * g64 / (g64.abs() + 1e-5)              # m_hat = g, v_hat = g^2 at t = 1
        step = (model.p[k].detach() - before[k]).cpu().double()
        big = g64.abs() > 1e-3                                   # away from the eps-dominated region
        assert big.any()
        assert (step - step64)[big].abs().max().item() < 2e-6, k
        assert step.abs().max().item() <= 1e-4 * (1 + 1e-3)


def test_seed_reseeds_the_counter_based_rng():
    """quadruped_gym_env.py:59-61 seed(): a new seed changes the episodes that start afterwards; the old one restores them."""
    import torch
    from openroborl_amd.env import VecQuadrupedEnv
    env = VecQuadrupedEnv(task_name="imitation_learning_laikago", num_robot=64, mode="train", seed=5, auto_reset=False)
    assert env.seed() == [5]
    a = env.reset().clone()
    ep = env.field_int("EPISODE_IDX").clone()
    assert env.seed(6) == [6]
    env.field_int("EPISODE_IDX").copy_(ep - 1)     # replay the same episode index with the new key
    b = env.reset().clone()
    assert not torch.equal(a, b)
    env.seed(5)
    env.field_int("EPISODE_IDX").copy_(ep - 1)
    c = env.reset().clone()
    assert torch.equal(a, c)
    env.close()
