"""Multi-GPU sharding of the env and the one collective on its path.

Robots are independent, so a job of R robots on G GPUs is G processes (one per GPU, torchrun) each
owning R/G robots with global indices [rank*R/G, (rank+1)*R/G) (RNG streams and grid slots are keyed by
the global index, so the union over ranks equals one big env).  Nothing is exchanged inside
reset/step.  The only exchange is the rollout-boundary gather that replaces
  lrlocal = (seg["ep_lens"], seg["ep_rets"]); MPI.COMM_WORLD.allgather(lrlocal)     (agents/ppo_imitation.py:405-408)
  MPI.COMM_WORLD.allreduce(seg["total_timestep"])                                   (agents/ppo_imitation.py:421)
with ONE fixed-size all_gather (RCCL over xGMI when the backend is "nccl"; gloo on CPU in tests) of
  [n_episodes, total_timesteps, n_dropped, ret_0..ret_{K-1}, len_0..len_{K-1}]  (float32, K = capacity).
The payload is a few KB per rank: latency-bound, ring vs tree does not matter.
"""
import os

HEADER = 3


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment (RANK / WORLD_SIZE / MASTER_*).  Returns
    (rank, world_size, local_rank).  World size 1 needs no process group."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("ORR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(total_robots, rank, world):
    """Contiguous shard [lo, hi) of the global robot index space owned by `rank`."""
    if total_robots % world != 0:
        raise ValueError("total_robots (%d) must be divisible by the number of ranks (%d)" % (total_robots, world))
    per = total_robots // world
    return rank * per, (rank + 1) * per


def pack_episode_stats(returns, lengths, total_timesteps, dropped, capacity):
    """[count, total_timesteps, dropped, ret..., len...] padded to capacity (float32, same device as returns)."""
    import torch
    k = min(int(returns.numel()), capacity)
    buf = torch.zeros(HEADER + 2 * capacity, dtype=torch.float32, device=returns.device)
    buf[0] = float(k)
    buf[1] = float(total_timesteps)
    buf[2] = float(dropped) + float(max(int(returns.numel()) - capacity, 0))
    buf[HEADER:HEADER + k] = returns[:k]
    buf[HEADER + capacity:HEADER + capacity + k] = lengths[:k]
    return buf


def unpack_episode_stats(gathered, capacity):
    """list of per-rank buffers -> (all_returns, all_lengths, total_timesteps, dropped)."""
    import torch
    rets, lens, ts, dr = [], [], 0, 0
    for buf in gathered:
        k = int(buf[0].item())
        ts += int(buf[1].item())
        dr += int(buf[2].item())
        rets.append(buf[HEADER:HEADER + k])
        lens.append(buf[HEADER + capacity:HEADER + capacity + k])
    return torch.cat(rets), torch.cat(lens), ts, dr


def allgather_episode_stats(returns, lengths, total_timesteps, dropped=0, capacity=4096, group=None):
    """The rollout-boundary collective.  Works without a process group (world size 1)."""
    import torch
    import torch.distributed as dist
    buf = pack_episode_stats(returns, lengths, total_timesteps, dropped, capacity)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return unpack_episode_stats([buf], capacity)
    dev = buf.device
    if dist.get_backend(group) == "gloo" and buf.is_cuda:   # rehearsal on a one-GPU box: stage through the host
        buf = buf.cpu()
    out = [torch.empty_like(buf) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, buf, group=group)
    return unpack_episode_stats([o.to(dev) for o in out], capacity)


def gather_env_episodes(env, steps_since_last, capacity=4096, group=None):
    """Drain the env's device-side episode log and all-gather it across ranks."""
    rets, lens = env.episode_log()
    return allgather_episode_stats(rets, lens, steps_since_last * env.num_robot, 0, capacity, group)
