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


# ---- the hand-written part of the update (include/openroborl_learner.h) against plain PyTorch fp32 / float64 ----------------------

def _ptr(x):
    return x.data_ptr()


def _stream(dev):
    import ctypes as C
    import torch
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _synthetic_batch(dev, B, seed=0, model=None):
    """Samples whose probability ratio leaves [0.8, 1.2] for part of the batch (clip active) and whose advantages include zeros."""
    import torch
    g = torch.Generator().manual_seed(seed)
    obs = torch.randn(B, 160, generator=g).to(dev)
    with torch.no_grad():
        mu = model.mean(obs) if model is not None else torch.randn(B, 12, generator=g).to(dev)
    act = mu + 0.125 * torch.randn(B, 12, generator=g).to(dev)
    shifted = mu + 0.04 * torch.randn(B, 12, generator=g).to(dev)
    old = (-0.5 * ((act - shifted) ** 2) / 0.125 ** 2 - 0.5 * math.log(2.0 * math.pi * 0.125 ** 2)).sum(dim=1)
    adv = torch.randn(B, generator=g).to(dev)
    adv[::97] = 0.0
    ret = torch.randn(B, generator=g).to(dev)
    return obs, mu, act, old, adv, ret


@pytest.mark.parametrize("m", [16384, 1000])
def test_loss_head_kernel_matches_autograd(m):
    import torch
    from openroborl_amd import _lib
    L = _lib.load()
    dev = torch.device("cuda:0")
    _, mu, act, old, adv, ret = _synthetic_batch(dev, m, seed=3)
    value = torch.randn(m, device=dev)
    batch = torch.zeros(m, 16, device=dev)
    batch[:, :12], batch[:, 12], batch[:, 13], batch[:, 14] = act, old, adv, ret
    gm, gv = torch.empty(m, 12, device=dev), torch.empty(m, device=dev)
    gb_m, gb_v, stats = torch.empty(12, device=dev), torch.empty(1, device=dev), torch.empty(2, device=dev)
    ws = torch.empty(int(L.orr_learner_workspace_floats(m, 512)), device=dev)
    _lib.check(L.orr_ppo_head(_ptr(mu), _ptr(value), _ptr(batch), m, 0.125, 0.2, 0.5, _ptr(gm), _ptr(gv), _ptr(gb_m), _ptr(gb_v), _ptr(stats), _ptr(ws),
                              _stream(dev)), L)
    # float64 autograd of the same loss (the formula of ppo.PPO.update)
    mu64, v64 = mu.double().requires_grad_(True), value.double().requires_grad_(True)
    logp = (-0.5 * ((act.double() - mu64) ** 2) / 0.125 ** 2 - 0.5 * math.log(2.0 * math.pi * 0.125 ** 2)).sum(dim=1)
    ratio = torch.exp(logp - old.double())
    surr = -torch.min(ratio * adv.double(), torch.clamp(ratio, 0.8, 1.2) * adv.double()).mean()
    vf = ((v64 - ret.double()) ** 2).mean()
    (surr + 0.5 * vf).backward()
    clipped = ((ratio < 0.8) | (ratio > 1.2)).double().mean().item()
    assert 0.02 < clipped < 0.9, clipped
    # the ratio of a float32 log-probability difference of O(10): relative error ~1e-6 per sample
    scale = mu64.grad.abs().max().item()
    assert (gm.double() - mu64.grad).abs().max().item() < 2e-5 * scale
    assert (gv.double() - v64.grad).abs().max().item() < 1e-6 * v64.grad.abs().max().item()
    assert (gb_m.double() - mu64.grad.sum(dim=0)).abs().max().item() < 2e-5 * scale * math.sqrt(m)
    assert abs(gb_v.item() - v64.grad.sum().item()) < 1e-5
    assert abs(stats[0].item() - surr.item()) < 2e-5 * max(1.0, abs(surr.item()))
    assert abs(stats[1].item() - vf.item()) < 2e-5 * vf.item()
    # samples outside the clip range on the clipped side, and zero advantages, get exactly zero gradient (as autograd gives them)
    dead = (mu64.grad.abs().sum(dim=1) == 0)
    assert dead.any() and bool((gm[dead] == 0).all())


@pytest.mark.parametrize("m,c", [(16384, 512), (16384, 256), (1000, 512), (33, 64)])
def test_relu_backward_kernel_matches_torch(m, c):
    import torch
    from openroborl_amd import _lib
    L = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(m + c)
    gh = torch.randn(m, c, device=dev, generator=g)
    h = torch.relu(torch.randn(m, c, device=dev, generator=g))
    want = gh * (h > 0).float()
    got, gb = gh.clone(), torch.empty(c, device=dev)
    ws = torch.empty(int(L.orr_learner_workspace_floats(m, c)), device=dev)
    _lib.check(L.orr_relu_backward(_ptr(got), _ptr(h), m, c, _ptr(gb), _ptr(ws), _stream(dev)), L)
    assert torch.equal(got, want)                                     # a select: bit-exact
    ref = want.double().sum(dim=0)
    assert (gb.double() - ref).abs().max().item() < 1e-5 * math.sqrt(m)
    gb2 = torch.empty(c, device=dev)                                  # fixed summation order: the same bits every time
    got2 = gh.clone()
    _lib.check(L.orr_relu_backward(_ptr(got2), _ptr(h), m, c, _ptr(gb2), _ptr(ws), _stream(dev)), L)
    assert torch.equal(gb, gb2)


@pytest.mark.parametrize("m,k", [(16384, 12), (16384, 1), (999, 12), (31, 1)])
def test_head_backward_kernel_matches_torch(m, k):
    import ctypes as C
    import torch
    from openroborl_amd import _abi, _lib
    L = _lib.load()
    dev = torch.device("cuda:0")
    c = 256
    g = torch.Generator(device=dev).manual_seed(m + k)
    gy = torch.randn(m, k, device=dev, generator=g)
    w = torch.randn(c, k, device=dev, generator=g)
    h = torch.relu(torch.randn(m, c, device=dev, generator=g))
    want = (gy.double() @ w.double().t()) * (h > 0).double()
    want_gw = h.double().t() @ gy.double()
    gz, gb, gw = torch.empty(m, c, device=dev), torch.empty(c, device=dev), torch.empty(c, k, device=dev)
    ws = torch.empty(int(L.orr_learner_workspace_floats(m, c)), device=dev)
    _lib.check(L.orr_head_backward(_ptr(gy), k, _ptr(w), _ptr(h), m, c, _ptr(gz), _ptr(gb), _ptr(gw), _ptr(ws), _stream(dev)), L)
    assert (gz.double() - want).abs().max().item() < 1e-5
    assert bool((gz[h == 0] == 0).all())
    assert (gb.double() - want.sum(dim=0)).abs().max().item() < 1e-5 * math.sqrt(m) * math.sqrt(k)
    assert (gw.double() - want_gw).abs().max().item() < 2e-5 * math.sqrt(m)
    # deferred: the same sums through orr_colsum_finish, bit for bit
    rows = int(L.orr_learner_partial_rows(m))
    gz2, gb2, gw2 = torch.empty_like(gz), torch.empty_like(gb), torch.empty_like(gw)
    ws2 = torch.empty_like(ws)
    _lib.check(L.orr_head_backward(_ptr(gy), k, _ptr(w), _ptr(h), m, c, _ptr(gz2), None, None, _ptr(ws2), _stream(dev)), L)
    jobs = (_abi.OrrColsumJob * 2)(_abi.OrrColsumJob(_ptr(ws2), _ptr(gb2), rows, c), _abi.OrrColsumJob(_ptr(ws2) + 4 * rows * c, _ptr(gw2), rows, c * k))
    _lib.check(L.orr_colsum_finish(jobs, 2, _stream(dev)), L)
    assert torch.equal(gz, gz2) and torch.equal(gb, gb2) and torch.equal(gw, gw2)


@pytest.mark.parametrize("n", [434701, 8, 3])
def test_adam_kernel_matches_torch_adam_and_the_reference_form(n):
    import torch
    from openroborl_amd import _lib
    L = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(n)
    pad = (n + 3) // 4 * 4
    p0 = torch.randn(pad, device=dev, generator=g)
    grads = [torch.randn(pad, device=dev, generator=g) * (10.0 ** -(i % 4)) for i in range(6)]
    # torch.optim.Adam
    pt = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pt], lr=1e-3, eps=1e-5)
    p, m, v = p0.clone(), torch.zeros(pad, device=dev), torch.zeros(pad, device=dev)
    state = torch.zeros(2, dtype=torch.int32, device=dev)
    # the reference's MpiAdam (mpi_adam.py:55-62) in float64, gradient averaged over 4 ranks
    p64, m64, v64 = p0.double().clone(), torch.zeros(pad, device=dev, dtype=torch.float64), torch.zeros(pad, device=dev, dtype=torch.float64)
    pm, mm, vm = p0.clone(), torch.zeros(pad, device=dev), torch.zeros(pad, device=dev)
    state_m = torch.zeros(2, dtype=torch.int32, device=dev)
    for t, gr in enumerate(grads, 1):
        pt.grad = gr.clone()
        opt.step()
        _lib.check(L.orr_adam_step(_ptr(p), _ptr(gr), _ptr(m), _ptr(v), n, 1e-3, 0.9, 0.999, 1e-5, 1.0, 0, _ptr(state), _stream(dev)), L)
        assert (p[:n] - pt.detach()[:n]).abs().max().item() < 2e-6
        g64 = gr.double() / 4.0
        step_size = 1e-3 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        m64 = 0.9 * m64 + 0.1 * g64
        v64 = 0.999 * v64 + 0.001 * g64 * g64
        p64 = p64 - step_size * m64 / (torch.sqrt(v64) + 1e-5)
        _lib.check(L.orr_adam_step(_ptr(pm), _ptr(gr), _ptr(mm), _ptr(vm), n, 1e-3, 0.9, 0.999, 1e-5, 0.25, 1, _ptr(state_m), _stream(dev)), L)
        assert (pm[:n].double() - p64[:n]).abs().max().item() < 2e-6
    assert state.tolist() == [6, 0] and state_m.tolist() == [6, 0]
    if pad > n:                                             # nothing beyond n is touched
        assert torch.equal(p[n:], p0[n:]) and bool((m[n:] == 0).all())


def test_fused_gradient_matches_float64_cpu():
    """The hand-written backward (library GEMMs + the four HIP launches) against float64 autograd of the same loss."""
    import torch
    from openroborl_amd import learner_hip, ppo
    dev = torch.device("cuda:0")
    B = 16384
    model = ppo.ActorCritic(dev, seed=4)
    obs, _, act, old, adv, ret = _synthetic_batch(dev, B, seed=0, model=model)
    p64 = {k: v.detach().cpu().double().requires_grad_(True) for k, v in model.p.items()}
    loss64, surr64, vf64 = _loss64(p64, obs.cpu().double(), act.cpu().double(), adv.cpu().double(), ret.cpu().double(), old.cpu().double())
    loss64.backward()
    learner = learner_hip.FusedPPO(model, lr=1e-4, minibatch=B, use_graph=False)
    P = learner._plan(B, B)
    with torch.no_grad():
        P["obs"].copy_(obs)
        P["aux"][:, :12], P["aux"][:, 12], P["aux"][:, 13], P["aux"][:, 14] = act, old, adv, ret
        P["perm"].copy_(torch.arange(B, device=dev))
        learner._minibatch(P, 0)
    assert abs(P["stats"][0, 0].item() - surr64.item()) < 2e-4 * max(1.0, abs(surr64.item()))
    assert abs(P["stats"][0, 1].item() - vf64.item()) < 2e-4 * max(1.0, abs(vf64.item()))
    for k in sorted(model.p):
        g32, g64 = learner.g[k].detach().cpu().double(), p64[k].grad
        scale = g64.abs().max().item() + 1e-12
        assert (g32 - g64).abs().max().item() < 2e-3 * scale, (k, (g32 - g64).abs().max().item(), scale)
        cs = (g32 * g64).sum() / (g32.norm() * g64.norm() + 1e-30)
        assert cs.item() > 0.99999, (k, cs.item())


@pytest.mark.parametrize("graph,B,M", [(False, 32768, 4096), (True, 32768, 4096), (True, 8192, 1024)])
def test_fused_update_matches_the_torch_learner(graph, B, M):
    """Two epochs of eight minibatches through ppo.PPO (autograd + torch Adam, the fp32 reference) and through FusedPPO from the
    same parameters, data and permutations: same losses, same parameters (Adam moves every weight by at most lr per step)."""
    import torch
    from openroborl_amd import learner_hip, ppo
    dev = torch.device("cuda:0")
    lr = 1e-4                                    # M = 1024: the weight gradients as plain GEMMs (no 16-way split over the batch)
    ref_model, model = ppo.ActorCritic(dev, seed=2), ppo.ActorCritic(dev, seed=2)
    obs, _, act, old, adv, ret = _synthetic_batch(dev, B, seed=1, model=ref_model)
    before = {k: v.detach().clone() for k, v in model.p.items()}
    ref = ppo.PPO(ref_model, lr=lr, minibatch=M)
    fused = learner_hip.FusedPPO(model, lr=lr, minibatch=M, use_graph=graph)
    for k in before:
        assert torch.equal(model.p[k].detach(), before[k])         # re-homing the parameters in the flat buffer keeps their values
    for it in range(2):
        g1, g2 = torch.Generator(device=dev), torch.Generator(device=dev)
        g1.manual_seed(10 + it); g2.manual_seed(10 + it)
        s_ref = ref.update(obs, act, adv, ret, old_logp=old, epochs=2, generator=g1)
        s_fused = fused.update(obs, act, adv, ret, old_logp=old, epochs=2, generator=g2)
        np.testing.assert_allclose(s_fused, s_ref, rtol=2e-4, atol=2e-5)
    assert fused.steps_taken() == 2 * 2 * (B // M)                   # the capture's warm-up steps were undone
    moved = 0.0
    for k in sorted(before):
        a, b = ref_model.p[k].detach(), model.p[k].detach()
        moved = max(moved, (a - before[k]).abs().max().item())
        # 32 Adam steps of at most lr each; the two learners may differ where a gradient is within float32 rounding of zero
        assert (a - b).abs().max().item() < 0.02 * 32 * lr, (k, (a - b).abs().max().item())
        assert (a - b).abs().mean().item() < 2e-3 * 32 * lr, (k, (a - b).abs().mean().item())
    assert moved > 10 * lr
    # the rollout's fused forward pass sees the updated weights (views of the flat buffer)
    model.enable_fused()
    with torch.no_grad():
        a_fused, _, v_fused = model.act(obs[:256], deterministic=True)
        np.testing.assert_allclose(a_fused.cpu().numpy(), model.mean(obs[:256]).clamp(-2 * math.pi, 2 * math.pi).cpu().numpy(), atol=2e-5)


def test_fused_update_is_reproducible_bit_for_bit():
    import torch
    from openroborl_amd import learner_hip, ppo
    dev = torch.device("cuda:0")
    B, M = 16384, 4096
    outs = []
    for _ in range(2):
        model = ppo.ActorCritic(dev, seed=3)
        obs, _, act, old, adv, ret = _synthetic_batch(dev, B, seed=5, model=model)
        fused = learner_hip.FusedPPO(model, lr=1e-4, minibatch=M)
        gen = torch.Generator(device=dev); gen.manual_seed(0)
        fused.update(obs, act, adv, ret, old_logp=old, epochs=2, generator=gen)
        outs.append(fused.flat_p.clone())
    assert torch.equal(outs[0], outs[1])
