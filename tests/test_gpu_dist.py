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


LEARNER_WORKER = r"""
import os, sys, math, torch
sys.path.insert(0, %r)
import torch.distributed as dist
from openroborl_amd import learner_hip, ppo
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
B, M = 16384, 4096
def data(seed):
    g = torch.Generator().manual_seed(seed)
    obs = torch.randn(B, 160, generator=g).to(dev)
    act = (torch.randn(B, 12, generator=g) * 0.2).to(dev)
    old = (torch.randn(B, generator=g) * 0.1 - 10.0).to(dev)
    return obs, act, torch.randn(B, generator=g).to(dev), torch.randn(B, generator=g).to(dev), old
def run(seed):
    model = ppo.ActorCritic(dev, seed=7)
    obs, act, adv, ret, old = data(seed)
    with torch.no_grad():
        old = model.log_prob(obs, act) + 0.05 * adv        # ratios around 1
    learner = learner_hip.FusedPPO(model, lr=1e-4, minibatch=M)
    gen = torch.Generator(device=dev); gen.manual_seed(3)
    stats = learner.update(obs, act, adv, ret, old_logp=old, epochs=2, generator=gen)
    assert learner.steps_taken() == 2 * (B // M)
    return learner.flat_p.clone(), stats
# the same shard on every rank: the averaged gradient IS each rank's gradient, so the result is the one-rank result bit for bit
p_same, _ = run(100)
torch.save(p_same.cpu(), os.path.join(%r, "same_%%d_of_%%d.pt" %% (rank, world)))
if world > 1:
    # different shards: every rank ends with the same parameters
    p_diff, _ = run(200 + rank)
    mine = p_diff.cpu()
    other = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(other, mine)
    assert all(torch.equal(o, other[0]) for o in other)
    assert not torch.equal(mine, p_same.cpu())
    dist.barrier()
    dist.destroy_process_group()
print("learner ok")
"""


def test_fused_learner_two_ranks_gloo_on_one_gpu(tmp_path):
    """The several-ranks path of learner_hip.FusedPPO (one hipGraph per minibatch, all-reduce of the flat gradient and Adam between
    replays) rehearsed with two gloo ranks on this box's one GPU, against the one-rank path (one graph per epoch)."""
    import torch
    script = tmp_path / "worker.py"
    script.write_text(LEARNER_WORKER % (ROOT, str(tmp_path)))
    base = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = subprocess.run([sys.executable, str(script)], env=dict(base, RANK="0", WORLD_SIZE="1"), capture_output=True, text=True, timeout=300)
    assert one.returncode == 0 and "learner ok" in one.stdout, one.stdout[-2000:] + one.stderr[-3000:]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(base, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0 and "learner ok" in so, so[-2000:] + se[-3000:]
    ref = torch.load(tmp_path / "same_0_of_1.pt")
    for r in range(2):
        assert torch.equal(torch.load(tmp_path / ("same_%d_of_2.pt" % r)), ref)


def test_train_loop_two_ranks_gloo_on_one_gpu(tmp_path):
    """train.py end to end with two ranks (gloo, both on this box's one GPU): graph-replayed rollout, per-minibatch graphs with the
    gradient all-reduce between replays, the episode all-gather, and the replica check of MpiAdam.check_synced every 100 iterations."""
    import json
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    base = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                ORR_DIST_BACKEND="gloo", ORR_BENCH_SINGLE_DEVICE="1")
    cmd = [sys.executable, os.path.join(ROOT, "train.py"), "--num-robot", "256", "--horizon", "8", "--minibatch", "1024", "--iters", "101",
           "--save", str(tmp_path / "p.zip")]
    procs = [subprocess.Popen(cmd, env=dict(base, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-2000:] + se[-3000:]
    recs = [json.loads(ln) for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert recs and recs[-1]["iter"] == 100 and recs[-1]["samples"] == 101 * 8 * 256 * 2      # both shards counted
    assert not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]                    # only rank 0 logs
    assert recs[-1]["episodes"] > 0 and os.path.exists(tmp_path / "p.zip")


RCCL_LEARNER_WORKER = r"""
import os, sys, torch
sys.path.insert(0, %r)
import torch.distributed as dist
from openroborl_amd import dist as odist, learner_hip, ppo
dev = torch.device("cuda:0")
B, M = 16384, 4096
def run():
    model = ppo.ActorCritic(dev, seed=7)
    g = torch.Generator().manual_seed(100)
    obs = torch.randn(B, 160, generator=g).to(dev)
    act = (torch.randn(B, 12, generator=g) * 0.2).to(dev)
    adv, ret = torch.randn(B, generator=g).to(dev), torch.randn(B, generator=g).to(dev)
    with torch.no_grad():
        old = model.log_prob(obs, act) + 0.05 * adv
    learner = learner_hip.FusedPPO(model, lr=1e-4, minibatch=M)
    learner.sync()
    gen = torch.Generator(device=dev); gen.manual_seed(3)
    learner.update(obs, act, adv, ret, old_logp=old, epochs=2, generator=gen)
    learner.check_synced()
    # a new learning rate must reach the next update although the graphs were captured with the old one (ADVICE r3)
    learner.lr = 3e-4
    learner.update(obs, act, adv, ret, old_logp=old, epochs=1, generator=gen)
    return learner, learner.flat_p.clone()
if os.environ.get("ORR_FORCE_DIST"):
    rank, world, local = odist.init_from_env()
    assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
    learner, p = run()
    assert learner._several() and len(next(iter(learner._plans.values()))["graphs"]) == B // M      # one graph per minibatch
    dist.barrier(); dist.destroy_process_group()
else:
    learner, p = run()
    assert not learner._several() and len(next(iter(learner._plans.values()))["graphs"]) == 1       # one graph per epoch
torch.save(p.cpu(), sys.argv[1])
print("learner ok")
"""


def test_fused_learner_several_ranks_path_through_rccl_one_rank(tmp_path):
    """The several-ranks path of learner_hip.FusedPPO - per-minibatch graphs, all-reduce of the flat gradient IN PLACE ON THE DEVICE,
    Adam between replays, the broadcast behind sync() / check_synced() - on the real RCCL backend with a one-rank group
    (ORR_FORCE_DIST=1; the gloo rehearsals stage the gradient through the host and never touch these calls).  An all-reduce over one
    rank is the identity, so the parameters must equal the no-group path (one graph per epoch) bit for bit, including after a
    change of the learning rate, which both paths must pick up although their graphs were captured before it."""
    import torch
    script = tmp_path / "worker.py"
    script.write_text(RCCL_LEARNER_WORKER % (ROOT,))
    base = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    base.pop("ORR_FORCE_DIST", None)
    plain = subprocess.run([sys.executable, str(script), str(tmp_path / "plain.pt")], env=base, capture_output=True, text=True, timeout=300)
    assert plain.returncode == 0 and "learner ok" in plain.stdout, plain.stdout[-2000:] + plain.stderr[-3000:]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(base, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ORR_FORCE_DIST="1")
    rccl = subprocess.run([sys.executable, str(script), str(tmp_path / "rccl.pt")], env=env, capture_output=True, text=True, timeout=300)
    assert rccl.returncode == 0 and "learner ok" in rccl.stdout, rccl.stdout[-2000:] + rccl.stderr[-3000:]
    assert torch.equal(torch.load(tmp_path / "plain.pt"), torch.load(tmp_path / "rccl.pt"))


def test_train_loop_through_rccl_one_rank(tmp_path):
    """train.py, 20 iterations, as the multi-GPU run executes it (process group on nccl, sync(), per-minibatch graphs + device all-reduce,
    check_synced, the episode all_gather of device buffers) with a one-rank group, against the same run without a group: the same
    log lines and the same saved weights."""
    import json
    import torch
    from openroborl_amd import policy as pol
    cmd = [sys.executable, os.path.join(ROOT, "train.py"), "--num-robot", "256", "--horizon", "8", "--minibatch", "1024", "--iters", "20",
           "--sync-check-every", "10"]
    base = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    base.pop("ORR_FORCE_DIST", None)
    plain = subprocess.run(cmd + ["--save", str(tmp_path / "plain.zip")], env=base, capture_output=True, text=True, timeout=600)
    assert plain.returncode == 0, plain.stdout[-2000:] + plain.stderr[-3000:]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(base, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ORR_FORCE_DIST="1")
    rccl = subprocess.run(cmd + ["--save", str(tmp_path / "rccl.zip")], env=env, capture_output=True, text=True, timeout=600)
    assert rccl.returncode == 0, rccl.stdout[-2000:] + rccl.stderr[-3000:]
    strip = lambda out: [{k: v for k, v in json.loads(ln).items() if k != "sec"} for ln in out.splitlines() if ln.startswith("{")]   # noqa: E731
    a, b = strip(plain.stdout), strip(rccl.stdout)
    assert a and a[-1]["iter"] == 19 and a == b
    pa, pb = pol.load_parameters(str(tmp_path / "plain.zip")), pol.load_parameters(str(tmp_path / "rccl.zip"))
    assert sorted(pa) == sorted(pb) and all((pa[k] == pb[k]).all() for k in pa)


def test_bench_launcher_through_rccl_one_rank():
    """bench.py --spawn --gpus 1 with ORR_FORCE_DIST=1: launcher -> fresh torchrun child -> nccl init -> describe() -> barriers,
    the MAX all-reduce of the elapsed time and the episode all_gather on RCCL - the chain the driver's 8-GPU run takes, with one rank."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", ORR_FORCE_DIST="1", ORR_BENCH_WARMUP_FLOOR="200")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--spawn", "--gpus", "1", "--steps", "40", "--warmup", "5",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["dist"] == {"backend": "nccl", "world": 1, "device_of_rank": [0]} and d["n_gpus"] == 1 and d["value"] > 1e6
    assert d["config"]["episodes_gathered"] >= 0 and d["roofline"]["kernel_ms"] > 0
