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


def test_bench_self_launch_dry_run_argv():
    """`python bench.py --gpus N` with no launcher around it starts torch.distributed.run itself: the argv."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--launch-dry-run"],
                         capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr
    d = json.loads(out.stdout)
    argv = d["argv"]
    assert d["ranks"] == 2 and argv[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert argv[argv.index("--nproc-per-node") + 1] == "2" and argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    k = argv.index(os.path.join(ROOT, "bench.py"))
    assert argv[k + 1:] == ["--gpus", "2", "--steps", "20", "--warmup", "5"]     # same arguments, launcher-only flags removed
    assert 1024 < int(argv[argv.index("--master-port") + 1]) < 65536


def test_bench_self_launch_runs_ranks_and_propagates_failure():
    """The launcher really starts N fresh rank processes and exits non-zero when they fail.  Without a GPU every rank dies in
    VecQuadrupedEnv ("no HIP device": there is no CPU fallback), after torch.distributed (gloo) came up with world size 2."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-side check of the failure path")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["ORR_DIST_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode != 0
    assert out.stdout.strip() == ""                    # no result line
    assert "ChildFailedError" in out.stderr or "exitcode" in out.stderr, out.stderr[-2000:]


SHARD_WORKER = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
from openroborl_amd import _abi, config, dist as odist, motion, robots
from tests import oracle_lib as ol
rank, world, local = odist.init_from_env(backend="gloo")
TOTAL, STEPS, CAP = 32, 22, 64


def run(n, offset, num_procs):
    cfg = config.make_config(n, mode="train", enable_randomizer=True, auto_reset=True, seed=9, num_procs=num_procs)
    env = ol.OracleEnv(cfg, [robots.laikago(), None, None, None], [motion.MotionClip("laikago_pace")], n, robot_type=0, clip_id=0,
                       robot_index=np.arange(offset, offset + n), threads=2, ep_log_capacity=256)
    obs = env.reset()
    noise = np.random.RandomState(5).randn(STEPS, TOTAL, 12)[:, offset:offset + n] * 0.125      # robot-local: one global table
    m = env.models[0]
    outs = []
    for k in range(STEPS):
        tar = obs[:, 84 + 7:84 + 19][:, m["joint_of_motor"]]
        a = np.clip((tar - m["motor_offset"]) * m["motor_dir"] - m["init_motor_angles"] + noise[k], -2 * np.pi, 2 * np.pi)
        obs, rew, done = env.step(a)
        outs.append((obs.copy(), rew.copy(), done.copy()))
    cnt = int(env.counters[_abi.CNT_EPISODES])
    log = env.ep_log[:cnt].copy()
    env.close()
    return outs, log


lo, hi = odist.shard_range(TOTAL, rank, world)
outs, log = run(hi - lo, lo, world)
stats = odist.allgather_episode_stats(torch.from_numpy(log[:, 0]).float(), torch.from_numpy(log[:, 1]).float(), STEPS * (hi - lo), 0, capacity=CAP)
if rank == 0:
    big, biglog = run(TOTAL, 0, 1)
    for k in range(STEPS):                      # this rank's shard IS rows lo..hi of the one big env, bit for bit (float64 on both sides)
        for a, b in zip(outs[k], big[k]):
            assert np.array_equal(a, b[lo:hi]), k
    rs, ls = stats[0].numpy().astype(np.float64), stats[1].numpy()
    rb, lb = biglog[:, 0].astype(np.float32).astype(np.float64), biglog[:, 1].astype(np.float32)
    assert len(rs) == len(rb) >= TOTAL and stats[2] == STEPS * TOTAL
    o1, o2 = np.lexsort((rs, ls)), np.lexsort((rb, lb))
    assert np.array_equal(ls[o1], lb[o2]) and np.array_equal(rs[o1], rb[o2])     # the gathered payload = the big env's episode log (as a multiset)
torch.distributed.barrier()
torch.distributed.destroy_process_group()
print("rank", rank, "ok")
""" % (ROOT,)


def test_two_oracle_shards_gathered_over_gloo_are_the_one_big_env(tmp_path):
    """The N > 1 path end to end on CPU: two processes, each stepping its shard of the robots (global indices by
    shard_range, per-rank curriculum counter with num_procs = 2) on the oracle, ONE all_gather of the episode payload over gloo -
    and rank 0 checks that its shard is, bit for bit, its rows of the single 32-robot env and that the gathered payload is that env's
    episode log.  The device counterpart (one GPU, eight shards one after another): tests/test_gpu_shards.py."""
    script = tmp_path / "shard_worker.py"
    script.write_text(SHARD_WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
        assert "ok" in o
