"""Multi-GPU sharding of the env and the one collective on its path.

Robots are independent, so a job of R robots on G GPUs is G processes (one per GPU, torchrun) each
owning R/G robots with global indices [rank*R/G, (rank+1)*R/G) (RNG streams and grid slots are keyed by
the global index, so the union over ranks equals one big env).  Nothing is exchanged inside
reset/step.  The only exchange is the rollout-boundary gather that replaces
  lrlocal = (seg["ep_lens"], seg["ep_rets"]); MPI.COMM_WORLD.allgather(lrlocal)     (agents/ppo_imitation.py:405-408)
  MPI.COMM_WORLD.allreduce(seg["total_timestep"])                                   (agents/ppo_imitation.py:421)
with ONE fixed-size all_gather (RCCL over xGMI when the backend is "nccl"; gloo on CPU in tests) of
  [n_listed, total_timesteps, n_dropped, n_episodes, sum_ret, sum_len, ret_0..ret_{K-1}, len_0..len_{K-1}]
(float64: counts stay exact up to 2^53; K = capacity).  n_episodes / sum_ret / sum_len cover EVERY episode the rank
logged, so the means are exact even when more than K episodes finished; only the per-episode list is truncated
(n_dropped says by how much).  The payload is KBs per rank: latency-bound, ring vs tree does not matter.

The pack runs on the device without a host sync (counts stay device scalars); the one sync of a rollout
boundary is the copy of the gathered buffer to the host in unpack_episode_stats.
"""
import os

HEADER = 6
_PINNED = {}      # pinned host staging buffers of unpack_episode_stats, by (shape, dtype)


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment (RANK / WORLD_SIZE / MASTER_*).  Returns
    (rank, world_size, local_rank).  World size 1 needs no process group."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # ORR_FORCE_DIST=1: build the process group even for one rank (a one-GPU smoke test of the RCCL code path)
    if (world > 1 or os.environ.get("ORR_FORCE_DIST", "0") == "1") and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("ORR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)       # before any other GPU call of this process
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def describe():
    """What torch.distributed actually set up (bench.py prints it, so a scaling run proves RCCL saw N ranks)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return {"backend": None, "world": 1, "device_of_rank": [torch.cuda.current_device() if torch.cuda.is_available() else None]}
    world = dist.get_world_size()
    backend = dist.get_backend()
    mine = torch.tensor([torch.cuda.current_device() if torch.cuda.is_available() else -1], dtype=torch.int64,
                        device="cuda" if backend == "nccl" else "cpu")
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    return {"backend": backend, "world": world, "device_of_rank": [int(o.item()) for o in out]}


def shard_range(total_robots, rank, world):
    """Contiguous shard [lo, hi) of the global robot index space owned by `rank`."""
    if total_robots % world != 0:
        raise ValueError("total_robots (%d) must be divisible by the number of ranks (%d)" % (total_robots, world))
    per = total_robots // world
    return rank * per, (rank + 1) * per


def pack_episode_stats(returns, lengths, total_timesteps, dropped, capacity, count=None):
    """One rank's payload (float64, on the device of `returns`).  `returns` / `lengths` hold the logged episodes in
    their first `count` rows (count: python int or a 0-d integer tensor = no host sync; default: all rows);
    `dropped` (int or 0-d tensor) = episodes the producer could not even log."""
    import torch
    dev = returns.device
    f64 = torch.float64
    rows = int(returns.shape[0])
    cnt = torch.as_tensor(rows if count is None else count, device=dev).to(f64).clamp(max=float(rows))
    idx = torch.arange(rows, device=dev, dtype=f64)
    live = idx < cnt
    r = torch.where(live, returns.to(f64), torch.zeros((), dtype=f64, device=dev))
    ln = torch.where(live, lengths.to(f64), torch.zeros((), dtype=f64, device=dev))
    listed = cnt.clamp(max=float(capacity))
    buf = torch.zeros(HEADER + 2 * capacity, dtype=f64, device=dev)
    k = min(rows, capacity)
    buf[0] = listed
    buf[1] = float(total_timesteps)
    buf[2] = torch.as_tensor(dropped, device=dev).to(f64) + (cnt - listed)
    buf[3] = cnt
    buf[4] = r.sum()
    buf[5] = ln.sum()
    buf[HEADER:HEADER + k] = r[:k]
    buf[HEADER + capacity:HEADER + capacity + k] = ln[:k]
    return buf


def unpack_episode_stats(gathered, capacity):
    """Per-rank payloads -> (all_returns, all_lengths, total_timesteps, dropped); .sums = (n_episodes, sum_ret, sum_len)
    over every logged episode of every rank.  ONE device->host copy (the rollout boundary's only sync)."""
    import torch
    dev = torch.stack([g.reshape(-1) for g in gathered]) if len(gathered) > 1 else gathered[0].reshape(1, -1)
    if dev.is_cuda:
        # one asynchronous copy into a cached pinned buffer + one stream synchronisation (a pageable .cpu() costs 2-3x as much, and this
        # copy is the only host synchronisation of a rollout boundary)
        key = (tuple(dev.shape), dev.dtype)
        host = _PINNED.get(key)
        if host is None:
            host = _PINNED[key] = torch.empty(dev.shape, dtype=dev.dtype).pin_memory()
        host.copy_(dev, non_blocking=True)
        torch.cuda.current_stream(dev.device).synchronize()
    else:
        host = dev
    rets, lens, ts, dr, n, sr, sl = [], [], 0, 0, 0, 0.0, 0.0
    head = host[:, :HEADER].tolist()        # one conversion for all the scalars (indexing a tensor element by element costs ~5 us each)
    for buf, h in zip(host, head):
        k = int(h[0])
        ts += int(h[1])
        dr += int(h[2])
        n += int(h[3])
        sr += h[4]
        sl += h[5]
        rets.append(buf[HEADER:HEADER + k])
        lens.append(buf[HEADER + capacity:HEADER + capacity + k])
    out = EpisodeStats(((rets[0] if len(rets) == 1 else torch.cat(rets)).float(), (lens[0] if len(lens) == 1 else torch.cat(lens)).float(), ts, dr))
    out.sums = (n, sr, sl)
    return out


class EpisodeStats(tuple):
    """(returns, lengths, total_timesteps, dropped) + .sums; mean_return / mean_length use the exact sums."""
    sums = (0, 0.0, 0.0)

    @property
    def mean_return(self):
        return self.sums[1] / max(self.sums[0], 1)

    @property
    def mean_length(self):
        return self.sums[2] / max(self.sums[0], 1)


def allgather_episode_stats(returns, lengths, total_timesteps, dropped=0, capacity=4096, group=None, count=None):
    """The rollout-boundary collective.  Works without a process group (world size 1)."""
    return allgather_packed(pack_episode_stats(returns, lengths, total_timesteps, dropped, capacity, count), capacity, group)


def allgather_packed(buf, capacity, group=None):
    """all_gather of one packed payload per rank (pack_episode_stats layout) + unpack."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and os.environ.get("ORR_FORCE_DIST", "0") != "1"):
        return unpack_episode_stats([buf], capacity)
    if dist.get_backend(group) == "gloo" and buf.is_cuda:   # rehearsal on a one-GPU box: stage through the host
        buf = buf.cpu()
    out = [torch.empty_like(buf) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, buf, group=group)
    return unpack_episode_stats(out, capacity)


def gather_env_episodes(env, steps_since_last, capacity=4096, group=None):
    """Drain the env's device-side episode log and all-gather it across ranks (no host sync before the collective)."""
    if hasattr(env, "episode_stats_packed"):      # one HIP launch packs the payload and clears the log (orr_episode_stats)
        return allgather_packed(env.episode_stats_packed(steps_since_last * env.num_robot, capacity), capacity, group)
    log, count, dropped = env.episode_log_device()
    return allgather_episode_stats(log[:, 0], log[:, 1], steps_since_last * env.num_robot, dropped, capacity, group, count)
