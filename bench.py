#!/usr/bin/env python3
"""Headline benchmark: env steps/sec at N parallel robots (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): imitation_learning_laikago, 4096 robots per GPU, laikago_pace clip,
train semantics (domain randomiser on, 20->600 step curriculum, per-robot auto-reset).  A "step" is one
env.step() of all robots of a GPU = one launch of the fused HIP kernel = 33 physics sub-steps +
observation + reward + termination (+ auto-reset) per robot.  Actions are the policy-free stress input of
SURVEY.md section 8d: reference joint pose one control step ahead (taken from the observation), in
motor space, plus N(0, 0.125^2) noise, generated on the device.  N > 1: independent shards, one process
per GPU, and the rollout-boundary all_gather of episode returns (RCCL) every 256 steps and at the end.

Prints ONE JSON line (rank 0).  The roofline object prices the step kernel against HBM bandwidth as the
north star asks; DESIGN.md section 6 explains why the kernel is VALU-latency bound and nowhere near it.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ROBOTS_PER_GPU = 4096
ROLLOUT = 256
GRAPH_LEN = 64      # env steps per hipGraph replay (divides ROLLOUT; the action-noise pool has 64 entries)
# algorithmic HBM bytes per robot-step of the step kernel: actions 48 + obs 640 +
# reward 4 + done 1 = 693; state head 307 words read + written = 2456; latency ring 33 entries written
# (2640) + 35 distinct entries read (2660).  Model tables and clip frames are shared and L2-resident.
B_ALG = 693 + 2456 + 2640 + 2660
HBM_PEAK_GBS = 8000.0


def cpu_baseline(env, seconds_target=12.0):
    """Time the CPU oracle (kind "port") on the host cores on a bounded sample of the same workload."""
    import numpy as np
    from tests import oracle_lib as ol
    # a one-GPU box owns a 16-core share of the host (more threads only oversubscribe it)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, int(os.environ.get("ORR_CPU_BASELINE_THREADS", "16"))))
    n = 32 * cores
    orc = ol.OracleEnv(env.cfg, env.models, env.clips, n, robot_type=0, clip_id=0, threads=cores)
    obs = orc.reset()
    jom = env.models[0]["joint_of_motor"]
    m = env.models[0]
    rng = np.random.RandomState(0)

    def act(o):
        tar = o[:, 84 + 7:84 + 19]
        a = (tar[:, jom] - m["motor_offset"]) * m["motor_dir"] - m["init_motor_angles"] + rng.randn(n, 12) * 0.125
        return np.clip(a, -2 * np.pi, 2 * np.pi)
    for _ in range(2):
        obs, _, _ = orc.step(act(obs))
    t0 = time.time()
    steps = 0
    while time.time() - t0 < seconds_target:
        obs, _, _ = orc.step(act(obs))
        steps += 1
    dt = time.time() - t0
    orc.close()
    return {"value": n * steps / dt, "unit": "env steps/s", "cores": cores, "kind": "port",
            "sample": "%d robots x %d env steps of the same workload, oracle/orr_oracle.c with %d OpenMP threads" % (n, steps, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--warmup", type=int, default=6000)   # the first process on a fresh box runs ~6 % slower for its first seconds
    ap.add_argument("--robots-per-gpu", type=int, default=ROBOTS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    from openroborl_amd import dist as odist
    from openroborl_amd.env import VecQuadrupedEnv

    rank, world, local = odist.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus > 1 launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N")
    # ORR_BENCH_SINGLE_DEVICE=1 (+ ORR_DIST_BACKEND=gloo): multi-rank rehearsal on a one-GPU box, never a measurement
    dev = torch.device("cuda", 0 if os.environ.get("ORR_BENCH_SINGLE_DEVICE") else local)
    torch.cuda.set_device(dev)
    n = args.robots_per_gpu
    env = VecQuadrupedEnv(task_name="imitation_learning_laikago", num_robot=n, mode="train", enable_randomizer=True,
                          auto_reset=True, seed=0, device=dev, num_procs=world, robot_index_offset=rank * n)
    m = env.models[0]
    jom = torch.tensor(m["joint_of_motor"], dtype=torch.long, device=dev)
    off = torch.tensor(m["motor_offset"], dtype=torch.float32, device=dev)
    mdir = torch.tensor(m["motor_dir"], dtype=torch.float32, device=dev)
    init = torch.tensor(m["init_motor_angles"], dtype=torch.float32, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    # action = clip((target motor pose - offset) * dir - init + noise): noise and the constant terms are pre-combined
    noise_pool = torch.randn(64, n, 12, generator=gen, device=dev) * 0.125 - (off * mdir + init)
    two_pi = 2.0 * 3.141592653589793

    # the joint -> motor permutation and the motor direction signs as one 12x12 matrix: two kernels per step (GEMM, clamp)
    perm = torch.zeros(12, 12, dtype=torch.float32, device=dev)
    perm[jom, torch.arange(12, device=dev)] = mdir

    def make_action(obs, k):
        tar = obs[:, 84 + 7:84 + 19]
        if os.environ.get("ORR_BENCH_INDEX_SELECT"):
            return torch.addcmul(noise_pool[k & 63], tar.index_select(1, jom), mdir).clamp_(-two_pi, two_pi)
        # |reference pose - init + noise| stays far below the 2 pi action bound, so the runner's clip
        # (imitation_runners.py:140-143) is a no-op here and is left out: one GEMM launch per step
        return torch.addmm(noise_pool[k & 63], tar, perm)

    def sync_all():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    # Default: eager launches (three small torch kernels for the synthetic actions + one step launch per env step; the
    # host keeps ahead of a 0.46 ms kernel).  ORR_BENCH_GRAPH=1 captures GRAPH_LEN consecutive env steps in a hipGraph and
    # replays it (remainder eagerly); measured slower on ROCm 7.2 (0.548 vs 0.498 ms per step), kept for comparison.
    def eager_steps(k0, count):
        for k in range(k0, k0 + count):
            env.step(make_action(env.obs, k))

    obs = env.reset()
    graph = None
    pre = min(args.warmup, 8)
    eager_steps(0, pre)                      # allocator / clocks before the capture
    if os.environ.get("ORR_BENCH_GRAPH"):
        try:
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                eager_steps(0, GRAPH_LEN)    # recorded, not executed
            env._env_step_counter -= GRAPH_LEN
        except Exception as e:               # noqa: BLE001  (report and measure eagerly)
            sys.stderr.write("bench: hipGraph capture failed (%r), running eagerly\n" % (e,))
            graph = None

    def run_steps(count, on_block=None):
        """Exactly `count` env steps: graph replays of GRAPH_LEN steps, then an eager remainder."""
        done_steps = 0
        while graph is not None and count - done_steps >= GRAPH_LEN:
            graph.replay()
            env._env_step_counter += GRAPH_LEN
            done_steps += GRAPH_LEN
            if on_block:
                on_block(GRAPH_LEN, done_steps == count)
        while done_steps < count:
            eager_steps(done_steps, 1)
            done_steps += 1
            if on_block:
                on_block(1, done_steps == count)

    run_steps(args.warmup - pre)
    env.episode_log()
    sync_all()
    t0 = time.perf_counter()
    acc = {"since": 0, "n_eps": 0}

    def on_block(nsteps, last):
        acc["since"] += nsteps
        if acc["since"] >= ROLLOUT or last:
            rets, lens, ts, dropped = odist.gather_env_episodes(env, acc["since"])
            acc["n_eps"] += int(rets.numel())
            acc["since"] = 0
    run_steps(args.steps, on_block)
    n_eps = acc["n_eps"]
    obs = env.obs
    sync_all()
    elapsed = time.perf_counter() - t0
    gloo = world > 1 and torch.distributed.get_backend() == "gloo"
    el = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if gloo else dev)
    if world > 1:
        torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(el.item())

    # dominant kernel: average launch duration with hipEvents on the launch stream (same workload state)
    act = make_action(obs, 0).contiguous()
    torch.cuda.synchronize(dev)
    kern_ms = env.time_steps(act, 50) / 50.0
    achieved = B_ALG * n / (kern_ms * 1e-3) / 1e9

    if rank == 0:
        # PMC numbers come from a separate rocprofv3 --pmc run of this same command (tools/profile_gpu.sh), committed
        # under profiles/; FETCH_SIZE is taken as reported (4-byte-per-lane rows, the x2 wide-read correction does not apply)
        traffic, valu = None, {}
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
        if os.path.exists(pmc):
            try:
                d = json.load(open(pmc))
                traffic = d.get("hbm_bytes_per_launch")
                waves = d["SQ_WAVES"]["mean_per_launch"]
                valu = {"valu_insts_per_robot_step": d["SQ_INSTS_VALU"]["mean_per_launch"] / n,
                        "valu_active_frac_of_wave_cycles": d["SQ_ACTIVE_INST_VALU"]["mean_per_launch"] / d["SQ_WAVE_CYCLES"]["mean_per_launch"],
                        "waves_per_simd": waves / 1024.0}
            except Exception:
                traffic, valu = None, {}
        out = {
            "metric": "env steps/sec at N parallel robots", "value": world * n * args.steps / elapsed, "unit": "env steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "imitation_learning_laikago, %d parallel robots per GPU, laikago_pace motion_file "
                                   "(BASELINE configs[1]; configs[3] when n_gpus=8)" % n,
                       "robots_per_gpu": n, "total_robots": world * n, "substeps_per_step": 33, "solver_iters": 9,
                       "randomizer": True, "auto_reset": True, "actions": "reference pose + N(0,0.125^2), on device (one GEMM launch per step)",
                       "launch": ("hipGraph of %d env steps per replay" % GRAPH_LEN) if graph is not None else "eager",
                       "collective": "all_gather of episode returns every %d steps" % ROLLOUT,
                       "episodes_gathered": n_eps},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "orr_step_kernel<0>", "kernel_ms": kern_ms, "alg_bytes_per_robot_step": B_ALG,
                         "pmc": valu,
                         "note": "instruction-issue-bound serial chain of a lone wave (33 x (leg dynamics + rows + 9 PGS sweeps)); HBM fraction is reported "
                                 "because the north star asks for it, see DESIGN.md section 6"},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(env)
        print(json.dumps(out))
    env.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
