"""The rollout-boundary collective through RCCL (backend "nccl") on the one GPU this box has: a one-rank process group
(ORR_FORCE_DIST=1), so that the code path the multi-GPU runs take -- set_device before init, float64 all_gather of the
device-packed payload, describe() -- is exercised on real RCCL and not only on gloo (tests/test_dist_cpu.py)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, torch
sys.path.insert(0, %r)
from openroborl_amd import dist as odist
from openroborl_amd.env import VecQuadrupedEnv
rank, world, local = odist.init_from_env()
assert torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl"
info = odist.describe()
assert info == {"backend": "nccl", "world": 1, "device_of_rank": [0]}, info
env = VecQuadrupedEnv(task_name="imitation_learning_laikago", num_robot=256, mode="train", seed=0, device="cuda:0")
env.reset()
a = torch.zeros(256, 12, device="cuda:0")
ndone = 0
for _ in range(25):
    _, _, d, _ = env.step(a)
    ndone += int(d.sum())
stats = odist.gather_env_episodes(env, 25, capacity=64)
rets, lens, ts, dropped = stats
assert stats.sums[0] == ndone and ts == 25 * 256 and rets.numel() == min(ndone, 64) and dropped == max(ndone - 64, 0), (stats.sums, ndone, dropped)
assert abs(stats.mean_length - float(stats.sums[2]) / max(ndone, 1)) < 1e-9
torch.distributed.barrier()
torch.distributed.destroy_process_group()
env.close()
print("rccl ok", ndone)
""" % (ROOT,)


def test_episode_gather_through_rccl_one_rank(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ORR_FORCE_DIST="1")
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "rccl ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
