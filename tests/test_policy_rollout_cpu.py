"""CPU tests of the "next" rows: batched policy MLP and device-side GAE."""
import os

import numpy as np
import torch

from openroborl_amd import policy as pol, rollout
from tests import oracle_lib as ol


def test_policy_matches_numpy_mlp_on_shipped_weights():
    path = os.path.join(ol.GOLDEN, "policy_laikago_pace.npz")
    p = pol.MLPPolicy.from_file(path, "cpu")
    W = np.load(path)
    obs = np.random.RandomState(0).randn(17, 160).astype(np.float32)
    h = np.maximum(obs @ W["model__pi_fc0__w_0"] + W["model__pi_fc0__b_0"], 0)
    h = np.maximum(h @ W["model__pi_fc1__w_0"] + W["model__pi_fc1__b_0"], 0)
    mu = h @ W["model__pi__w_0"] + W["model__pi__b_0"]
    a, raw, v = p.act(torch.from_numpy(obs), deterministic=True)
    np.testing.assert_allclose(raw.numpy(), mu, atol=2e-4, rtol=1e-4)
    assert a.abs().max() <= 2 * np.pi + 1e-6 and v.shape == (17,)
    g = torch.Generator().manual_seed(0)
    a2, raw2, _ = p.act(torch.from_numpy(obs), generator=g)
    assert abs((raw2 - raw).std().item() - pol.PI_STD) < 0.02          # fixed std 0.125 (imitation_policies.py:106)
    lp = p.log_prob(torch.from_numpy(obs), raw)
    np.testing.assert_allclose(lp.numpy(), 12 * (-0.5 * np.log(2 * np.pi * pol.PI_STD ** 2)), rtol=1e-5)


def test_gae_matches_reference_loop():
    rng = np.random.RandomState(1)
    T, n, gamma, lam = 40, 5, 0.95, 0.95
    rew = rng.rand(T, n).astype(np.float32)
    vp = rng.randn(T, n).astype(np.float32)
    done = rng.rand(T, n) < 0.1
    adv, ret = rollout.gae(torch.from_numpy(rew), torch.from_numpy(vp), torch.from_numpy(done), gamma, lam)
    exp = np.zeros((T, n), dtype=np.float64)
    for i in range(n):
        last = 0.0
        for k in reversed(range(T)):
            nonterm = 0.0 if done[k, i] else 1.0
            nxt = (vp[k + 1, i] if k + 1 < T else 0.0) * nonterm      # 0 after an episode end and at the segment end
            delta = rew[k, i] + gamma * nxt - vp[k, i]
            last = delta + gamma * lam * nonterm * last
            exp[k, i] = last
    np.testing.assert_allclose(adv.numpy(), exp, atol=1e-4)
    np.testing.assert_allclose(ret.numpy(), exp + vp, atol=1e-4)
    nrm = rollout.normalize_per_robot(adv)
    np.testing.assert_allclose(nrm.mean(dim=0).numpy(), 0, atol=1e-5)
    np.testing.assert_allclose(nrm.std(dim=0, unbiased=False).numpy(), 1, atol=1e-4)


def test_collect_rollout_with_a_fake_env():
    class FakeEnv(object):
        torch = torch
        num_robot = 3
        device = torch.device("cpu")

        def __init__(self):
            self.k = 0

        def reset(self):
            return torch.zeros(3, 160)

        def step(self, a):
            self.k += 1
            return torch.full((3, 160), float(self.k)), torch.full((3,), 0.5), torch.tensor([0, 1, 0], dtype=torch.uint8), {}
    p = pol.MLPPolicy.from_file(os.path.join(ol.GOLDEN, "policy_laikago_pace.npz"), "cpu")
    buf = rollout.collect_rollout(FakeEnv(), p, horizon=4, deterministic=True)
    assert buf["obs"].shape == (4, 3, 160) and buf["actions"].shape == (4, 3, 12)
    assert torch.equal(buf["obs"][2], torch.full((3, 160), 2.0)) and buf["dones"][:, 1].all() and not buf["dones"][:, 0].any()
    assert torch.equal(buf["last_obs"], torch.full((3, 160), 4.0))


def test_stable_baselines_zip_round_trip(tmp_path):
    from openroborl_amd import ppo
    m = ppo.ActorCritic("cpu", seed=3)
    path = str(tmp_path / "model.zip")
    pol.save_parameters_zip(path, m.state_dict())
    import json, zipfile
    with zipfile.ZipFile(path) as z:
        assert set(z.namelist()) == {"data", "parameter_list", "parameters"}
        # the reference's load_parameters(exact_match=True) needs every variable of its graph: same names, order and shapes
        # as the shipped laikago_pace.zip (fixture from tests/golden/make_golden.py), including the unused q head
        want = json.load(open(os.path.join(ol.GOLDEN, "policy_parameter_list.json")))
        assert json.loads(z.read("parameter_list")) == [k for k, _ in want]
        import io
        saved = np.load(io.BytesIO(z.read("parameters")))
        assert [list(saved[k].shape) for k, _ in want] == [shape for _, shape in want]
        assert not saved["model/q/w:0"].any()            # synthesised: this learner has no q head
    w = pol.load_parameters(path)
    for k, v in m.state_dict().items():
        np.testing.assert_array_equal(w[k], v)
    p = pol.MLPPolicy.from_file(path, "cpu")
    obs = torch.randn(4, 160)
    np.testing.assert_allclose(p.mean(obs).numpy(), m.mean(obs).detach().numpy(), atol=1e-6)
    np.testing.assert_allclose(p.value(obs).numpy(), m.value(obs).detach().numpy(), atol=1e-6)
    # a policy warm-started from a reference-style zip keeps that zip's q head when it is saved again
    w["model/q/w:0"] = np.full((256, 12), 0.25, dtype=np.float32)
    m2 = ppo.ActorCritic("cpu", params=w)
    path2 = str(tmp_path / "model2.zip")
    pol.save_parameters_zip(path2, m2.state_dict())
    assert (pol.load_parameters(path2)["model/q/w:0"] == 0.25).all()


def test_graph_replayed_paths_refuse_to_run_without_a_gpu():
    """learner_hip.FusedPPO and rollout.GraphRollout are HIP-only: on a CPU model / without the fused policy they raise instead of
    falling back to the plain PyTorch paths (ppo.PPO, rollout.collect_rollout), which stay available under their own names."""
    import pytest
    import torch
    from openroborl_amd import learner_hip, ppo, rollout
    model = ppo.ActorCritic(torch.device("cpu"), seed=0)
    with pytest.raises(RuntimeError, match="needs a GPU"):
        learner_hip.FusedPPO(model)

    class FakeEnv(object):
        torch = __import__("torch")
        num_robot, device = 4, torch.device("cpu")
    with pytest.raises(RuntimeError, match="fused policy"):
        rollout.GraphRollout(FakeEnv(), model, 8)
    # the plain learner still works on the CPU (the reference path of the GPU tests)
    learner = ppo.PPO(model, lr=1e-4, minibatch=32)
    g = torch.Generator().manual_seed(0)
    obs, act = torch.randn(64, 160, generator=g), torch.randn(64, 12, generator=g) * 0.1
    s = learner.update(obs, act, torch.randn(64, generator=g), torch.randn(64, generator=g), epochs=1, generator=g)
    assert s.shape == (2,) and bool((s == s).all())
