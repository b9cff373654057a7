"""CPU tests of the PyTorch PPO learner, incl. the gradient all-reduce with gloo, world size 2."""
import os
import socket
import subprocess
import sys

import numpy as np
import torch

from openroborl_amd import ppo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_update_improves_surrogate_and_value_fit():
    torch.manual_seed(0)
    m = ppo.ActorCritic("cpu", seed=1)
    learner = ppo.PPO(m, lr=3e-4, minibatch=256)
    obs = torch.randn(1024, 160)
    with torch.no_grad():
        _, actions, _ = m.act(obs, generator=torch.Generator().manual_seed(2))
    adv = (actions[:, 0] - m.mean(obs)[:, 0].detach())            # reward pushing joint 0 up
    adv = (adv - adv.mean()) / adv.std()
    ret = obs[:, 0] * 0.5
    v0 = ((m.value(obs) - ret) ** 2).mean().item()
    mu0 = m.mean(obs)[:, 0].mean().item()
    for _ in range(20):
        learner.update(obs, actions, adv, ret)
    assert ((m.value(obs) - ret) ** 2).mean().item() < v0
    lp = m.log_prob(obs, actions)
    assert torch.isfinite(lp).all()


def test_parameter_names_match_stable_baselines_zip_layout():
    m = ppo.ActorCritic("cpu")
    sd = m.state_dict()
    for k, shape in (("model/pi_fc0/w:0", (160, 512)), ("model/pi_fc1/w:0", (512, 256)), ("model/pi/w:0", (256, 12)),
                     ("model/vf_fc0/w:0", (160, 512)), ("model/vf/w:0", (256, 1)), ("model/pi/b:0", (12,))):
        assert sd[k].shape == shape
    # warm start from the shipped policy weights (run.py:220-221 load_parameters)
    from openroborl_amd import policy as pol
    from tests import oracle_lib as ol
    w = pol.load_parameters(os.path.join(ol.GOLDEN, "policy_laikago_pace.npz"))
    m2 = ppo.ActorCritic("cpu", params=w)
    np.testing.assert_array_equal(m2.state_dict()["model/pi_fc0/w:0"], w["model/pi_fc0/w:0"])


WORKER = r"""
import os, sys, torch
sys.path.insert(0, %r)
from openroborl_amd import dist as odist, ppo
rank, world, local = odist.init_from_env(backend="gloo")
m = ppo.ActorCritic("cpu", seed=5)                    # identical initial weights on both ranks
learner = ppo.PPO(m, lr=1e-3, minibatch=64)
g = torch.Generator().manual_seed(100 + rank)         # different data per rank
obs = torch.randn(128, 160, generator=g)
act = torch.randn(128, 12, generator=g) * 0.1
adv = torch.randn(128, generator=g)
ret = torch.randn(128, generator=g)
learner.update(obs, act, adv, ret, generator=torch.Generator().manual_seed(7))
flat = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
out = [torch.empty_like(flat) for _ in range(world)]
torch.distributed.all_gather(out, flat)
assert torch.allclose(out[0], out[1]), (out[0] - out[1]).abs().max()   # averaged gradients keep the replicas in sync
torch.distributed.destroy_process_group()
print("rank", rank, "ok")
""" % (ROOT,)


def test_gradient_allreduce_keeps_replicas_identical(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
