"""Fused policy / value forward pass (csrc/orr_policy.hip through the C-ABI) against a plain PyTorch float32 reference
of the same MLPs, and against the golden outputs of the reference's shipped policy.  GPU only."""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _torch_forward(p, obs):
    import torch
    hi = torch.float64          # reference in float64: the kernel and torch-f32 are both compared against it

    @torch.no_grad()
    def mlp(net):
        h = torch.relu(obs.to(hi) @ p["model/%s_fc0/w:0" % net].to(hi) + p["model/%s_fc0/b:0" % net].to(hi))
        h = torch.relu(h @ p["model/%s_fc1/w:0" % net].to(hi) + p["model/%s_fc1/b:0" % net].to(hi))
        return h @ p["model/%s/w:0" % net].to(hi) + p["model/%s/b:0" % net].to(hi)
    return mlp("pi"), mlp("vf")[:, 0]


@pytest.mark.parametrize("n", [4096, 1000, 16, 3])
def test_fused_forward_matches_torch(n):
    import torch
    from openroborl_amd import policy_hip, ppo
    dev = torch.device("cuda", 0)
    model = ppo.ActorCritic(dev, seed=3)
    with torch.no_grad():
        for k, v in model.p.items():          # non-trivial biases and a head that is not tiny
            if k.endswith("/b:0"):
                v.copy_(torch.randn(v.shape, generator=torch.Generator().manual_seed(len(k))).to(dev) * 0.1)
        model.p["model/pi/w:0"].mul_(30.0)
    fused = policy_hip.FusedActorCritic(model.p, dev, std=0.125)
    g = torch.Generator(device=dev).manual_seed(n)
    obs = torch.randn(n, 160, generator=g, device=dev) * 2.0
    noise = torch.randn(n, 12, generator=g, device=dev)
    act, raw, val, mean = fused.forward(obs, noise, want_mean=True)
    mu64, v64 = _torch_forward(model.p, obs)
    scale = float(mu64.abs().max())
    # f32 MFMA = k-ordered fmaf chain: error ~1e-7 * sum |a b| (tolerance: 2e-5 of the output scale)
    assert float((mean.double() - mu64).abs().max()) < 2e-5 * max(1.0, scale)
    assert float((val.double() - v64).abs().max()) < 2e-5 * max(1.0, float(v64.abs().max()))
    np.testing.assert_allclose(raw.cpu().numpy(), (mean + 0.125 * noise).cpu().numpy(), atol=1e-6)
    np.testing.assert_allclose(act.cpu().numpy(), np.clip(raw.cpu().numpy(), -2 * math.pi, 2 * math.pi), atol=0)
    # deterministic mode
    act_d, raw_d, _, _ = fused.forward(obs, None)
    np.testing.assert_allclose(raw_d.cpu().numpy(), mean.cpu().numpy(), atol=0)
    # and no worse than torch's own float32 path
    with torch.no_grad():
        mu32 = model.mean(obs)
    assert float((mean.double() - mu64).abs().max()) <= 4.0 * float((mu32.double() - mu64).abs().max()) + 1e-6


def test_fused_forward_refresh_and_shipped_policy():
    """The reference's shipped laikago_pace actor (tests/golden/policy_laikago_pace.npz, weights taken from
    policies/laikago_pace.zip) through the fused kernel vs float64; then refresh() after a parameter change."""
    import torch
    from openroborl_amd import policy as pol, policy_hip, ppo
    dev = torch.device("cuda", 0)
    gold = np.load(os.path.join(GOLD, "policy_laikago_pace.npz"))
    params = {pol._norm_key(k): gold[k] for k in gold.files}
    model = ppo.ActorCritic(dev, params=params)
    fused = policy_hip.FusedActorCritic(model.p, dev, std=0.125)
    g = torch.Generator(device=dev).manual_seed(5)
    obs = (torch.rand(2048, 160, generator=g, device=dev) * 2.0 - 1.0) * 3.0
    _, raw, val, _ = fused.forward(obs, None)
    mu64, v64 = _torch_forward(model.p, obs)
    assert float((raw.double() - mu64).abs().max()) < 2e-5 * max(1.0, float(mu64.abs().max()))
    assert float((val.double() - v64).abs().max()) < 2e-5 * max(1.0, float(v64.abs().max()))
    with torch.no_grad():
        model.p["model/pi/b:0"].add_(1.0)
    fused.refresh()
    _, raw2, _, _ = fused.forward(obs, None)
    np.testing.assert_allclose(raw2.cpu().numpy(), raw.cpu().numpy() + 1.0, atol=1e-5)


def test_rollout_fused_equals_torch_policy():
    """collect_rollout through the fused kernel vs through plain torch, same env seed and same noise."""
    import torch
    from openroborl_amd import ppo, rollout
    from openroborl_amd.env import VecQuadrupedEnv
    dev = torch.device("cuda", 0)
    out = []
    for fused in (False, True):
        env = VecQuadrupedEnv(task_name="imitation_learning_laikago", num_robot=256, seed=4, device=dev)
        model = ppo.ActorCritic(dev, seed=1)
        if fused:
            model.enable_fused()
        obs = env.reset()
        buf = rollout.collect_rollout(env, model, 3, obs=obs, deterministic=True)
        out.append({k: v.clone() for k, v in buf.items()})
        env.close()
    a, b = out
    # step 0 sees identical observations: actions / values agree to float32 rounding
    np.testing.assert_allclose(a["actions"][0].cpu().numpy(), b["actions"][0].cpu().numpy(), atol=2e-5)
    np.testing.assert_allclose(a["vpred"][0].cpu().numpy(), b["vpred"][0].cpu().numpy(), atol=2e-5)
    # later steps differ only through the chaotic amplification of that rounding
    # (a robot that sits within rounding of a termination threshold may end its episode on one side only: nearly all agree)
    dr = np.abs(a["rewards"].cpu().numpy() - b["rewards"].cpu().numpy())
    assert np.median(dr) < 1e-4 and np.percentile(dr, 99) < 5e-3, (np.median(dr), np.percentile(dr, 99))
    assert float((a["dones"] == b["dones"]).float().mean()) > 0.99


@pytest.mark.parametrize("shape", [(32, 4096), (7, 33), (1, 5)])
def test_fused_gae_matches_torch(shape):
    """orr_gae (one launch) vs rollout.gae + normalize_per_robot (the plain PyTorch restatement of
    agents/ppo_imitation.py:68-93,329-338)."""
    import torch
    from openroborl_amd import rollout
    dev = torch.device("cuda", 0)
    T, n = shape
    g = torch.Generator(device=dev).manual_seed(T * 1000 + n)
    rew = torch.rand(T, n, generator=g, device=dev)
    vp = torch.randn(T, n, generator=g, device=dev)
    dones = torch.rand(T, n, generator=g, device=dev) < 0.1
    boot = torch.randn(n, generator=g, device=dev)
    for bootstrap in (None, boot):
        a0, r0 = rollout.gae(rew, vp, dones, 0.95, 0.95, bootstrap=bootstrap)
        a1, r1 = rollout.gae_fused(rew, vp, dones, 0.95, 0.95, bootstrap=bootstrap, normalize=False)
        np.testing.assert_allclose(a1.cpu().numpy(), a0.cpu().numpy(), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(r1.cpu().numpy(), r0.cpu().numpy(), rtol=1e-5, atol=1e-5)
        if T > 1:
            a2, r2 = rollout.gae_fused(rew, vp, dones, 0.95, 0.95, bootstrap=bootstrap, normalize=True, eps=1e-8)
            np.testing.assert_allclose(a2.cpu().numpy(), rollout.normalize_per_robot(a0, eps=1e-8).cpu().numpy(), rtol=2e-4, atol=2e-5)
            np.testing.assert_allclose(r2.cpu().numpy(), r0.cpu().numpy(), rtol=1e-5, atol=1e-5)


def test_rollout_logp_matches_learner_log_prob():
    """The log-probabilities derived from the rollout noise equal ActorCritic.log_prob at the sampling parameters."""
    import torch
    from openroborl_amd import ppo, rollout
    from openroborl_amd.env import VecQuadrupedEnv
    dev = torch.device("cuda", 0)
    env = VecQuadrupedEnv(task_name="imitation_learning_laikago", num_robot=128, seed=2, device=dev)
    model = ppo.ActorCritic(dev, seed=3).enable_fused()
    g = torch.Generator(device=dev).manual_seed(0)
    buf = rollout.collect_rollout(env, model, 4, obs=env.reset(), generator=g)
    with torch.no_grad():
        lp = model.log_prob(buf["obs"].reshape(-1, 160), buf["actions"].reshape(-1, 12))
    np.testing.assert_allclose(buf["logp"].reshape(-1).cpu().numpy(), lp.cpu().numpy(), atol=2e-3, rtol=1e-4)
    env.close()


def test_graph_rollout_equals_the_eager_collector():
    """rollout.GraphRollout (static buffers, env.step_into, one hipGraph per segment) against rollout.collect_rollout on a twin env with
    the same seed and the same exploration noise: every buffer bit for bit, over an eager first segment, the capture and two replays,
    with the weights changed between segments (the captured re-pack must pick the new ones up)."""
    import torch
    from openroborl_amd import ppo, rollout
    from openroborl_amd.env import VecQuadrupedEnv
    dev = torch.device("cuda:0")
    n, T = 256, 8
    envs = [VecQuadrupedEnv(task_name="imitation_learning_laikago", num_robot=n, mode="train", auto_reset=True, seed=11, device=dev) for _ in range(2)]
    models = [ppo.ActorCritic(dev, seed=1).enable_fused() for _ in range(2)]
    collector = rollout.GraphRollout(envs[1], models[1], T)
    obs = [e.reset() for e in envs]
    gen = torch.Generator(device=dev).manual_seed(0)
    ended = 0
    for seg in range(4):
        noise = torch.randn(T, n, 12, device=dev, generator=gen)
        a = rollout.collect_rollout(envs[0], models[0], T, obs=obs[0], noise=noise)
        b = collector.collect(obs[1], noise=noise)
        for k in ("obs", "actions", "rewards", "dones", "vpred", "last_obs"):
            assert torch.equal(a[k], b[k]), (seg, k)
        np.testing.assert_allclose(b["logp"].cpu().numpy(), a["logp"].cpu().numpy(), rtol=1e-6, atol=1e-5)
        assert envs[0].env_step_counter == envs[1].env_step_counter == (seg + 1) * T
        ended += int(b["dones"].sum()) if seg >= 1 else 0
        obs = [a["last_obs"], b["last_obs"]]
        with torch.no_grad():                       # "the learner" moves the weights; both policies are told so
            for m in models:
                for k2 in sorted(m.p):
                    m.p[k2].mul_(1.0 + 0.01 * (seg + 1))
                m.mark_updated()
    assert collector.graph is not None and ended >= n                  # episodes ended and restarted inside the replayed segments
    # by-value launch arguments frozen into the captured graph: a new exploration std and a new env seed after the capture -
    # the collector must capture again, and the new graph must SAMPLE with the new std (it is handed on to the fused forward pass by
    # ActorCritic.std's setter), not only compute the log-probabilities with it
    for seg, change in ((4, "std"), (5, "seed"), (6, None)):
        if change == "std":
            for m in models:
                m.std = 0.25
        elif change == "seed":
            for e in envs:
                e.seed(12345)
        noise = torch.randn(T, n, 12, device=dev, generator=gen)
        a = rollout.collect_rollout(envs[0], models[0], T, obs=obs[0], noise=noise)
        b = collector.collect(obs[1], noise=noise)
        for k in ("obs", "actions", "rewards", "dones", "vpred", "last_obs"):
            assert torch.equal(a[k], b[k]), (seg, change, k)
        np.testing.assert_allclose(b["logp"].cpu().numpy(), a["logp"].cpu().numpy(), rtol=1e-6, atol=1e-5)
        if change == "std":      # the sampled actions really moved with the std: raw action - mean = std * noise
            assert float((b["actions"][0] - a["actions"][0]).abs().max()) == 0.0 and models[1].fused.std == 0.25
        obs = [a["last_obs"], b["last_obs"]]
    stats = [e.stats() for e in envs]
    assert stats[0] == stats[1]
    for e in envs:
        e.close()
