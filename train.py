#!/usr/bin/env python3
"""End-to-end motion-imitation training on the device: env (HIP kernels) + batched policy + GAE + PPO, nothing
leaves the GPU.  Counterpart of `python3 OpenRoboRL/run.py --task imitation_learning_laikago` in train mode
(run.py:186-237), restricted to what SURVEY.md section 8f lists as "next" rows.

  python train.py --task imitation_learning_laikago --iters 200
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="imitation_learning_laikago")
    ap.add_argument("--num-robot", type=int, default=4096)
    ap.add_argument("--horizon", type=int, default=32)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--minibatch", type=int, default=16384)
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--model-file", default="")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--log", default="")
    ap.add_argument("--save", default="", help="write the trained weights as a stable-baselines style zip")
    ap.add_argument("--torch-policy", action="store_true", help="rollout policy through plain torch instead of the fused HIP kernel")
    ap.add_argument("--torch-learner", action="store_true",
                    help="PPO update through autograd + torch.optim.Adam (ppo.PPO, the fp32 reference) instead of the hand-written backward "
                         "replayed as a hipGraph (learner_hip.FusedPPO)")
    ap.add_argument("--eval", default="", help="evaluate a policy zip instead of training: deterministic actions, test mode "
                                                 "(no randomiser, full-length episodes), like `run.py --mode test`")
    ap.add_argument("--legacy-gae-index", action="store_true",
                    help="reproduce the reference's neighbouring-robot episode-start index in the GAE recursion (ppo_imitation.py:88); "
                         "only for curve-by-curve comparisons with the reference at num_robot > 1")
    ap.add_argument("--mpi-adam", action="store_true",
                    help="the reference's MpiAdam arithmetic (stable_baselines/common/mpi_adam.py:53-58: epsilon outside the bias correction, i.e. "
                         "NOT scaled by sqrt(1 - beta2^t)) instead of torch.optim.Adam's; with adam_epsilon 1e-5 (run.py:118) the reference's "
                         "effective epsilon is ~30x larger in the first steps.  Default off: the recorded training runs used torch's form")
    ap.add_argument("--sync-check-every", type=int, default=100, help="replica consistency check (MpiAdam.check_synced, mpi_adam.py:47-48: every 100 updates)")
    ap.add_argument("--tune-gemms", action="store_true", help="let PyTorch's TunableOp pick the learner's GEMM kernels (graph-replayed learner: 16.8 -> 16.3 ms per iteration after ~3 s of tuning)")
    args = ap.parse_args()

    import torch
    from openroborl_amd import dist as odist, policy as pol, ppo, rollout
    from openroborl_amd.env import VecQuadrupedEnv

    if args.tune_gemms:
        torch.cuda.tunable.enable(True)
    if args.eval:
        return evaluate(args, torch, pol, ppo, VecQuadrupedEnv)
    rank, world, local = odist.init_from_env()
    # ORR_BENCH_SINGLE_DEVICE=1 (+ ORR_DIST_BACKEND=gloo): several ranks rehearsed on a one-GPU box (tests), never a measurement
    dev = torch.device("cuda", 0 if os.environ.get("ORR_BENCH_SINGLE_DEVICE") else local)
    torch.cuda.set_device(dev)
    env = VecQuadrupedEnv(task_name=args.task, num_robot=args.num_robot, mode="train", auto_reset=True, seed=args.seed,
                          device=dev, num_procs=world, robot_index_offset=rank * args.num_robot)
    params = pol.load_parameters(args.model_file) if args.model_file else None     # run.py:220-221
    model = ppo.ActorCritic(dev, params=params, seed=args.seed)                     # same seed -> identical replicas
    if not args.torch_policy:
        model.enable_fused()
    if args.torch_learner:
        learner = ppo.PPO(model, lr=args.lr, minibatch=args.minibatch)
    else:
        from openroborl_amd import learner_hip
        # equal minibatches only (static buffers of the graph-replayed update): the largest size <= --minibatch that divides the
        # segment, e.g. --num-robot 1000 x horizon 32 = 32000 samples run as 2 x 16000 instead of failing on 16384
        mb = learner_hip.FusedPPO.fit_minibatch(args.num_robot * args.horizon, args.minibatch)
        if mb != min(args.minibatch, args.num_robot * args.horizon) and rank == 0:
            print("train.py: minibatch %d -> %d (must divide the %d samples of a segment)" % (args.minibatch, mb, args.num_robot * args.horizon),
                  file=sys.stderr)
        if mb < 256 and args.num_robot * args.horizon >= 4096:
            raise SystemExit("train.py: %d samples per segment have no divisor between 256 and %d; pick another --num-robot / --horizon "
                             "or use --torch-learner (handles a short last minibatch)" % (args.num_robot * args.horizon, args.minibatch))
        learner = learner_hip.FusedPPO(model, lr=args.lr, minibatch=mb, mpi_adam_epsilon=args.mpi_adam)
        learner.sync()                                        # MpiAdam.sync before training (ppo_imitation.py:274)
    gen = torch.Generator(device=dev)
    gen.manual_seed(args.seed * 1000 + rank)
    obs = env.reset()
    t0 = time.time()
    samples = 0
    log = []
    first_starts = None                   # episode_starts of a segment's first step = the last step's done flags of the previous segment
    # static rollout buffers + the segment replayed as one hipGraph (fused policy only); --torch-policy keeps the eager collector
    collector = None if args.torch_policy else rollout.GraphRollout(env, model, args.horizon)
    for it in range(args.iters):
        buf = collector.collect(obs, generator=gen) if collector else rollout.collect_rollout(env, model, args.horizon, obs=obs, generator=gen)
        obs = buf["last_obs"]
        with torch.no_grad():
            boot = model.value(obs)
        # bootstrap with the critic at the segment end (the reference uses 0 there, imitation_runners.py:98-100;
        # with 32-step segments that bias would dominate)
        adv, ret = rollout.gae_fused(buf["rewards"], buf["vpred"], buf["dones"], 0.95, 0.95, bootstrap=boot, normalize=True, eps=1e-8,
                                     legacy_gae_index=args.legacy_gae_index, first_starts=first_starts)
        first_starts = buf["dones"][-1].clone()               # the collector's buffers are overwritten by the next segment
        T, n = buf["rewards"].shape
        surr, vf = learner.update(buf["obs"].reshape(T * n, -1), buf["actions"].reshape(T * n, -1), adv.reshape(-1),
                                  ret.reshape(-1), old_logp=buf["logp"].reshape(-1) if "logp" in buf else None,
                                  epochs=args.epochs, generator=gen)
        samples += T * n * world
        # (> 1 rank, or ORR_FORCE_DIST=1: the one-rank rehearsal of the several-ranks path on RCCL runs the check too)
        if (world > 1 or os.environ.get("ORR_FORCE_DIST", "0") == "1") and args.sync_check_every > 0 and it % args.sync_check_every == args.sync_check_every - 1 and hasattr(learner, "check_synced"):
            learner.check_synced()                            # like MpiAdam every 100 updates (mpi_adam.py:47-48)
        stats = odist.gather_env_episodes(env, args.horizon)   # means come from the exact per-rank sums, not the truncated list
        if rank == 0 and (it % 10 == 0 or it == args.iters - 1):
            rec = {"iter": it, "samples": samples, "sec": round(time.time() - t0, 2),
                   "mean_step_reward": round(float(buf["rewards"].mean()), 4),
                   "ep_len_mean": round(stats.mean_length, 1), "ep_ret_mean": round(stats.mean_return, 2),
                   "episodes": stats.sums[0], "episodes_unlogged": stats[3] - max(stats.sums[0] - int(stats[0].numel()), 0),
                   "max_ep_steps": int(env.field_int("MAX_EP_STEPS").max()), "surr": round(float(surr), 4), "vf": round(float(vf), 4)}
            log.append(rec)
            print(json.dumps(rec), flush=True)
    if rank == 0 and args.save:
        pol.save_parameters_zip(args.save, model.state_dict())
    if rank == 0 and args.log:
        with open(args.log, "w") as f:
            json.dump(log, f, indent=1)
    env.close()
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def evaluate(args, torch, pol, ppo, VecQuadrupedEnv):
    """run.py:151-183 (test): one full episode per robot with the policy mean; prints return / length statistics."""
    dev = torch.device("cuda", 0)
    n = min(args.num_robot, 1024)
    env = VecQuadrupedEnv(task_name=args.task, num_robot=n, mode="test", auto_reset=False, seed=args.seed, device=dev)
    model = ppo.ActorCritic(dev, params=pol.load_parameters(args.eval))
    if not args.torch_policy:
        model.enable_fused()
    obs = env.reset()
    alive = torch.ones(n, dtype=torch.bool, device=dev)
    ret = torch.zeros(n, device=dev)
    length = torch.zeros(n, device=dev)
    limit = int(env.field_int("MAX_EP_STEPS").max())
    for _ in range(limit):
        act, _, _ = model.act(obs, deterministic=True)
        obs, rew, done, _ = env.step(act)
        ret += rew * alive
        length += alive.float()
        alive &= ~done.bool()
        if not bool(alive.any()):
            break
    print(json.dumps({"policy": args.eval, "task": args.task, "robots": n, "episode_limit": limit,
                      "return_mean": round(float(ret.mean()), 2), "return_min": round(float(ret.min()), 2),
                      "length_mean": round(float(length.mean()), 1), "full_length_fraction": round(float((length >= limit).float().mean()), 3),
                      "reward_per_step": round(float((ret / length).mean()), 3)}))
    env.close()


if __name__ == "__main__":
    main()
