"""The rollout-boundary collective on CPU: world_size 2, gloo (covers the N > 1 path of bench.py / dist.py)."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, torch
sys.path.insert(0, %r)
from openroborl_amd import dist as odist
rank, world, local = odist.init_from_env(backend="gloo")
assert world == 2
lo, hi = odist.shard_range(8192, rank, world)
assert (lo, hi) == (rank * 4096, (rank + 1) * 4096)
k = 3 + 2 * rank
rets = torch.arange(k, dtype=torch.float32) + 100 * rank
lens = torch.full((k,), 20.0 + rank)
all_r, all_l, ts, dropped = odist.allgather_episode_stats(rets, lens, total_timesteps=1000 * (rank + 1), dropped=rank, capacity=16)
assert all_r.numel() == 3 + 5, all_r
assert torch.equal(all_r, torch.cat([torch.arange(3.), torch.arange(5.) + 100]))
assert torch.equal(all_l, torch.cat([torch.full((3,), 20.), torch.full((5,), 21.)]))
assert ts == 3000 and dropped == 1
# overflow is counted, not silently lost
big = torch.ones(40)
r2, l2, ts2, dr2 = odist.allgather_episode_stats(big, big, 0, 0, capacity=16)
assert r2.numel() == 32 and dr2 == 2 * (40 - 16)
torch.distributed.destroy_process_group()
print("rank", rank, "ok")
""" % (ROOT,)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_allgather_episode_stats_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "ok" in o


def test_single_process_path_needs_no_group():
    import torch
    from openroborl_amd import dist as odist
    r, l, ts, dr = odist.allgather_episode_stats(torch.tensor([1.0, 2.0]), torch.tensor([20.0, 21.0]), 77, 0, capacity=8)
    assert r.tolist() == [1.0, 2.0] and l.tolist() == [20.0, 21.0] and ts == 77 and dr == 0
    buf = odist.pack_episode_stats(torch.tensor([1.0, 2.0]), torch.tensor([20.0, 21.0]), 77, 0, 8)
    assert buf.numel() == odist.HEADER + 16
