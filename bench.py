#!/usr/bin/env python3
"""Headline benchmark: env steps/sec at N parallel robots (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--config laikago4096|minicheetah4096|mixed8192] [--no-randomizer] [--repeats R]
  N > 1 without a launcher around it (WORLD_SIZE unset): this process touches no GPU, starts
  `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...` as a fresh
  child (one rank per GPU over RCCL, like the reference's `mpiexec -n 8 python3 OpenRoboRL/run.py`, README.md:27), forwards rank 0's
  JSON line and exits with the child's return code.  Under an external torchrun (the driver's N > 1 command) it is a rank.

Workload (default = BASELINE.json configs[1]): imitation_learning_laikago, 4096 robots per GPU, laikago_pace clip,
train semantics (domain randomiser on, 20->600 step curriculum, per-robot auto-reset).  --config selects configs[2]
(mini-cheetah, trot clip, 4096 robots) or configs[4] (8192 robots, Laikago and mini-cheetah interleaved in every wave).
A "step" is one env.step() of all robots of a GPU = one launch of the fused HIP kernel = 33 physics sub-steps +
observation + reward + termination (+ auto-reset) per robot.  Actions are the policy-free stress input of SURVEY.md
section 8d: reference joint pose one control step ahead (taken from the observation), in motor space, plus
N(0, 0.125^2) noise, generated on the device (one elementwise launch per step for any mix of robot types: the library's
orr_stress_actions; inside the timed region).  N > 1: independent
shards, one process per GPU, and the rollout-boundary all_gather of episode returns (RCCL) every 256 steps and at the
end of the timed region.

Timing protocol.  Untimed: `warmup_internal` env steps (a floor that does not depend on --warmup: a fresh box needs
~2 s of work before its clocks and code objects are in steady state) + one rollout-boundary gather (its first call
loads code objects) + the W steps of --warmup.  Timed: EXACTLY K steps incl. the action launches and the gathers a real
rollout performs, bracketed by barrier + synchronize.  Every 8th timed launch of the step kernel (every 4th when K <= 64, every one
when K <= 8; never launch 0, whose bracket would hold the host's launch latency on an empty queue) is additionally bracketed by HIP
events on the launch stream; their MEDIAN is the roofline's kernel time (`kernel_ms`; the mean, min and max ride along).

PMC-derived fields (`roofline.traffic`, `.pmc`, `.valu_issue`) come from a committed summary of separate rocprofv3 --pmc passes
(profiles/rNN_<config>_pmc_summary.json, tools/profile_gpu.sh); every summary records the source hash of the library it was taken on,
and a summary taken on other kernel sources than the loaded library's is NOT used: the fields are null and `"pmc_stale": true`.

Prints ONE JSON line (rank 0).  The roofline object prices the step kernel against HBM bandwidth as the
north star asks; DESIGN.md section 6 explains why the kernel is VALU-issue bound and nowhere near it.
"""
import argparse
import json
import os
import re
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ROLLOUT = 256
# untimed env steps before anything else (~2 s of kernel time at 4096 robots).  ORR_BENCH_WARMUP_FLOOR overrides it for the
# rocprofv3 --pmc passes only (tools/profile_gpu.sh): counter collection serialises and records every dispatch, and the per-launch
# counters of the step kernel do not depend on clocks; the value actually used is printed as `warmup_internal`.
WARMUP_FLOOR = int(os.environ.get("ORR_BENCH_WARMUP_FLOOR", "6000"))
# algorithmic HBM bytes per robot-step of the step kernel: actions 48 + obs 640 +
# reward 4 + done 1 = 693; state head 307 words read + written = 2456; latency ring 33 entries written
# (2640) + 35 distinct entries read (2660).  Model tables and clip frames are shared and L2-resident.
B_ALG = 693 + 2456 + 2640 + 2660
# --no-randomizer (the reference's test-mode latency: fixed 2 ms = the entries 2 and 3 sub-steps back, SURVEY.md section 8d second
# row): only the last 4 ring entries of a step are ever read again (4 x 80 B written) and 3 entries of the previous step are read
# (3 x 76 B); the other 29 entries a step pushes are dead stores as far as the algorithm goes (the kernel still makes them: the
# PMC traffic says what that costs)
B_ALG_FIXED_LATENCY = 693 + 2456 + 320 + 228
HBM_PEAK_GBS = 8000.0
# SURVEY.md section 8(d)'s own per-unit figures (the survey session's estimate of the algorithmic traffic, made before there was a layout):
# 6.4 KB per robot-step with the randomiser on, 2.6 KB with the fixed 2 ms latency.  Reported BESIDE the built layout's figure (B_ALG
# above: what the records of this implementation really move) so that a reader can check either convention.
B_ALG_SURVEY = 6400
B_ALG_SURVEY_FIXED_LATENCY = 2600

CONFIGS = {
    # name: (VecQuadrupedEnv keyword arguments, robots per GPU, description for config.workload)
    "laikago4096": (dict(task_name="imitation_learning_laikago"), 4096,
                    "imitation_learning_laikago, %d parallel robots per GPU, laikago_pace motion_file (BASELINE configs[1]; configs[3] when n_gpus=8)"),
    "minicheetah4096": (dict(task_name="imitation_learning_minicheetah"), 4096,
                        "imitation_learning_minicheetah, %d parallel robots per GPU, minicheetah_trot motion_file (BASELINE configs[2])"),
    "mixed8192": (dict(task_name="imitation_learning_laikago", mixed_robots=["laikago", "mini_cheetah"],
                       motion_file=["laikago_pace.txt", "minicheetah_trot.txt"]), 8192,
                  "mixed Laikago + mini-cheetah batch (robot_type = i & 1: both models in every wavefront), %d robots per GPU (BASELINE configs[4])"),
}


def _oracle_rate(ol, env, n, threads, seconds, f32, build_dir):
    import numpy as np
    orc = ol.OracleEnv(env.cfg, env.models, env.clips, n, robot_type=env.robot_type[:n], clip_id=env.clip_id[:n], threads=threads,
                       f32=f32, build_dir=build_dir)
    obs = orc.reset()
    rng = np.random.RandomState(0)
    jom = np.stack([env.models[t]["joint_of_motor"] for t in env.robot_type[:n]])
    off = np.stack([env.models[t]["motor_offset"] for t in env.robot_type[:n]])
    mdir = np.stack([env.models[t]["motor_dir"] for t in env.robot_type[:n]])
    init = np.stack([env.models[t]["init_motor_angles"] for t in env.robot_type[:n]])

    def act(o):
        tar = np.take_along_axis(o[:, 84 + 7:84 + 19], jom, axis=1)
        return np.clip((tar - off) * mdir - init + rng.randn(n, 12) * 0.125, -2 * np.pi, 2 * np.pi)
    for _ in range(2):
        obs, _, _ = orc.step(act(obs))
    t0 = time.time()
    steps = 0
    while time.time() - t0 < seconds:
        obs, _, _ = orc.step(act(obs))
        steps += 1
    dt = time.time() - t0
    orc.close()
    return n * steps / dt, steps


def _cgroup_cpu_quota():
    """CPUs this process may use per the cgroup CPU controller (v2 cpu.max, v1 cfs quota), or None when unlimited / unknown."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


def cpu_baseline(env):
    """Time the CPU oracle (kind "port": PyBullet is not installed) on the host cores on bounded
    samples of the same workload.  Rows: float64 -O2 (the parity oracle) and float32 -O3 -march=native builds of the same
    source, at 1 thread x 1 robot (BASELINE configs[0]), 1 thread x 512 robots, 16 threads and all available cores.  ~30 s in total."""
    from tests import oracle_lib as ol
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    quota = _cgroup_cpu_quota()
    build_dir = None if os.access(os.path.join(ROOT, "oracle"), os.W_OK) else tempfile.mkdtemp()   # None = in oracle/
    rows = []
    # a one-GPU box is a 16-core share of a bigger host whose affinity mask still lists every CPU (256 in round 3, where the
    # "all cores" rows came out SLOWER than the 16-thread rows: they measured oversubscription).  The wide row is sized from the
    # cgroup CPU quota when the box has one; without a quota it runs on the affinity mask and is labelled for what it is
    share = min(16, avail)
    wide = min(avail, int(quota)) if quota and quota >= 1.0 else avail
    f64, f32 = "f64 -O2 -ffp-contract=off", "f32 -O3 -march=native"
    # BASELINE.md section 3.1 asks for N = 1 and N = 4096: the 4096-robot rows run the WHOLE batch of the GPU workload on the box's core
    # share for a bounded number of env steps (~4 s each: a dozen steps of the float64 build, a few dozen of the float32 build)
    plan = [(f64, False, 1, 1, 3.0), (f64, False, 512, 1, 3.0), (f64, False, 32 * share, share, 4.0), (f64, False, 4096, share, 4.0),
            (f32, True, 1, 1, 3.0), (f32, True, 512, 1, 3.0), (f32, True, 32 * share, share, 4.0), (f32, True, 4096, share, 4.0)]
    if wide > share:
        plan += [(f64, False, 16 * wide, wide, 4.0), (f32, True, 16 * wide, wide, 4.0)]
    for build, f32, n, threads, secs in plan:
        try:
            rate, steps = _oracle_rate(ol, env, min(n, env.num_robot), threads, secs, f32, build_dir)
            rows.append({"value": rate, "unit": "env steps/s", "cores": threads, "robots": min(n, env.num_robot), "build": build,
                         "sample": "%d robots x %d env steps" % (min(n, env.num_robot), steps)})
            if threads > share and not quota:
                rows[-1]["label"] = ("threads = CPUs in the affinity mask; this box reports no cgroup CPU quota, so if its real share is "
                                     "smaller this row measures oversubscription, not more cores")
        except Exception as e:      # noqa: BLE001  (e.g. no compiler on the box for the f32 build): report, keep the other rows
            rows.append({"build": build, "cores": threads, "robots": n, "error": repr(e)})
    # the real reference physics, if this box happens to have it (it does not on this pool: SURVEY.md section 8c)
    pyb = "unavailable: neither pybullet nor pybullet_data is installed; the rows above time the CPU restatement instead"
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import pybullet_ref
        if pybullet_ref.available():
            name = "mini_cheetah" if int(env.robot_type[0]) == 1 else "laikago"
            rows += [pybullet_ref.time_path(name, k, seconds=5.0) for k in (1, 16)]
            pyb = "available: rows of kind 'reference' are real PyBullet through tools/pybullet_ref.py"
    except Exception as e:          # noqa: BLE001
        pyb = "probe failed: %r" % (e,)
    best = max((r for r in rows if "value" in r and r.get("kind", "port") == "port"), key=lambda r: r["value"])
    at = {}
    for nr in (1, 4096):          # the N = 1 and the N = 4096 row, fastest build / thread count of each
        cand = [r for r in rows if r.get("robots") == nr and "value" in r and r.get("kind", "port") == "port"]
        if cand:
            b = max(cand, key=lambda r: r["value"])
            at["robots_%d" % nr] = {"value": b["value"], "unit": "env steps/s", "cores": b["cores"], "build": b["build"], "sample": b["sample"]}
    return {"value": best["value"], "unit": "env steps/s", "cores": best["cores"], "kind": "port", "by_batch": at,
            "sample": "%s of the same workload, oracle/orr_oracle.c (%s), %d OpenMP threads; the fastest of the rows below"
                      % (best["sample"], best["build"], best["cores"]),
            "host_cpu_count": os.cpu_count(), "host_cores_available": avail, "cgroup_cpu_quota_cores": quota, "pybullet": pyb,
            "note": "a restatement of the same algorithm (un-tuned articulated-body + PGS code), not PyBullet; never the target",
            "rows": rows}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launcher_argv(n_ranks, child_args, port=None):
    """argv of the child that runs this file as `n_ranks` ranks (one per GPU) under torch.distributed.run."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
            "--master-addr", "127.0.0.1", "--master-port", str(port or _free_port()), os.path.abspath(__file__)] + list(child_args)


def launch_ranks(n_ranks, child_args):
    """Parent side of a self-launched multi-rank run.  Nothing here imports torch or touches the GPU: the ranks are FRESH
    child processes (never an exec of a process that has initialised HIP).  stdout of the children is filtered: the one JSON
    result line goes to stdout, everything else to stderr.  Returns the child's exit code (non-zero if any rank failed; torchrun
    tears the other ranks down) - no retry."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(launcher_argv(n_ranks, child_args), stdout=subprocess.PIPE, env=env, text=True)
    lines = 0
    for line in proc.stdout:
        if line.startswith("{") and '"metric"' in line:
            sys.stdout.write(line)
            sys.stdout.flush()
            lines += 1
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if rc == 0 and lines != 1:
        sys.stderr.write("bench.py launcher: expected one result line from rank 0, got %d\n" % lines)
        rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="laikago4096")
    ap.add_argument("--robots-per-gpu", type=int, default=0, help="override the config's robots per GPU (experiments)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-randomizer", action="store_true",
                    help="second row of SURVEY 8d: domain randomiser off, fixed 2 ms control latency (run.py:205-206, laikago.py:27)")
    ap.add_argument("--repeats", type=int, default=1, help="time R regions of K steps each and report the median (SURVEY 8d: 5)")
    ap.add_argument("--spawn", action="store_true", help="go through the rank launcher even for --gpus 1")
    ap.add_argument("--launch-dry-run", action="store_true", help="print the launcher's argv as JSON and exit")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn or args.launch_dry_run):
        child_args = [a for a in sys.argv[1:] if a not in ("--spawn", "--launch-dry-run")]
        if args.launch_dry_run:
            print(json.dumps({"argv": launcher_argv(args.gpus, child_args), "ranks": args.gpus}))
            return 0
        return launch_ranks(args.gpus, child_args)

    import torch
    from openroborl_amd import dist as odist
    from openroborl_amd.env import VecQuadrupedEnv

    rank, world, local = odist.init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    # ORR_BENCH_SINGLE_DEVICE=1 (+ ORR_DIST_BACKEND=gloo): multi-rank rehearsal on a one-GPU box, never a measurement
    dev = torch.device("cuda", 0 if os.environ.get("ORR_BENCH_SINGLE_DEVICE") else local)
    torch.cuda.set_device(dev)
    env_kw, n, workload = CONFIGS[args.config]
    n = args.robots_per_gpu or n
    env = VecQuadrupedEnv(num_robot=n, mode="train", enable_randomizer=not args.no_randomizer, auto_reset=True, seed=int(os.environ.get("ORR_BENCH_SEED", "0")), device=dev,
                          num_procs=world, robot_index_offset=rank * n, **env_kw)
    # action = clip((target joint pose -> motor space) - init + noise, +-2 pi) with every robot's own joint -> motor table: ONE launch
    # of the library's stress-input kernel per step whatever the mix of robot types (as tensor operations a heterogeneous batch
    # needs a copy + a batched GEMM: three launches, 4.9 % of the 8192-robot step in round 3)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    noise_pool = (torch.randn(64, n, 12, generator=gen, device=dev) * 0.125).contiguous()
    act_buf = torch.zeros(n, 12, dtype=torch.float32, device=dev)

    def make_action(obs, k):
        return env.stress_actions(obs, noise_pool[k & 63], act_buf)

    in_group = torch.distributed.is_available() and torch.distributed.is_initialized()   # world > 1, or ORR_FORCE_DIST=1

    def sync_all():
        if in_group:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    def eager_steps(k0, count):
        for k in range(k0, k0 + count):
            env.step(make_action(env.obs, k))

    # ---- untimed: warm-up floor, one gather (first-call code-object loads, allocations), then the caller's --warmup ----
    env.reset()
    eager_steps(0, WARMUP_FLOOR)
    odist.gather_env_episodes(env, WARMUP_FLOOR)
    eager_steps(0, args.warmup)
    odist.gather_env_episodes(env, args.warmup)
    dist_info = odist.describe()
    # HIP events around the step kernel of every EV_STRIDE-th timed launch (every launch for K <= 8, every 4th for K <= 64): each
    # record is a packet on the launch stream (~6 us of stream time per bracketed launch), so bracketing all launches would itself
    # cost 2-4 % of the measured rate.  Launch 0 is never bracketed (unless K = 1): its opening record lands on an EMPTY queue right
    # after the synchronize, so that bracket holds the host's launch latency, not only the kernel (one such outlier of
    # five biased the 20-step line).  The roofline's kernel time is the MEDIAN of the bracketed launches; the mean rides along.
    ev_stride = int(os.environ.get("ORR_BENCH_EVENT_STRIDE", "0")) or (1 if args.steps <= 8 else (4 if args.steps <= 64 else 8))
    ev_first = 1 if args.steps > 1 else 0
    stream = torch.cuda.current_stream(dev)
    gloo = in_group and torch.distributed.get_backend() == "gloo"

    def timed_region(k0):
        """EXACTLY args.steps env steps (action launch + step kernel) + the rollout-boundary gathers, between barrier + synchronize
        on both sides; returns the MAX over ranks of the elapsed time and rank-local details."""
        ev = {k: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for k in range(ev_first, args.steps, ev_stride)}
        sync_all()
        t0 = time.perf_counter()
        gather_s, since, n_eps = 0.0, 0, 0
        for k in range(args.steps):
            act = make_action(env.obs, k0 + k)
            e = ev.get(k)
            if e is not None:
                e[0].record(stream)
            env.step(act)
            if e is not None:
                e[1].record(stream)
            since += 1
            if since >= ROLLOUT or k == args.steps - 1:
                torch.cuda.synchronize(dev)                         # drain the queued steps first: that wait is kernel time, not gather time
                g0 = time.perf_counter()
                stats = odist.gather_env_episodes(env, since)
                torch.cuda.synchronize(dev)
                q0 = time.perf_counter()
                n_eps += stats.sums[0]
                since = 0
                gather_s += q0 - g0
        sync_all()
        elapsed = time.perf_counter() - t0
        el = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if gloo else dev)
        if in_group:
            torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
        durs = sorted(a.elapsed_time(b) for a, b in ev.values())
        return {"elapsed": float(el.item()), "gather_s": gather_s, "episodes": n_eps, "kern_ms": durs[len(durs) // 2],
                "kern_ms_mean": sum(durs) / len(durs), "kern_ms_min": durs[0], "kern_ms_max": durs[-1], "launches_timed": len(ev)}

    # ---- timed: --repeats regions of exactly --steps env steps each; the reported one is the median by elapsed time ----
    regions = [timed_region(r * args.steps) for r in range(max(1, args.repeats))]
    mid = sorted(regions, key=lambda r: r["elapsed"])[len(regions) // 2]
    elapsed, gather_s, n_eps, n_ev = mid["elapsed"], mid["gather_s"], mid["episodes"], mid["launches_timed"]

    # dominant kernel: HIP events on the launch stream around the bracketed timed launches, median
    kern_ms = mid["kern_ms"]
    kern_total_ms = kern_ms * args.steps
    # cross-check: back-to-back launches without the action kernels in between (C-ABI helper, same stream)
    act = make_action(env.obs, 0).contiguous()
    torch.cuda.synchronize(dev)
    kern_b2b_ms = env.time_steps(act, 50) / 50.0
    b_alg = B_ALG_FIXED_LATENCY if args.no_randomizer else B_ALG
    b_survey = B_ALG_SURVEY_FIXED_LATENCY if args.no_randomizer else B_ALG_SURVEY
    achieved = b_alg * n / (kern_ms * 1e-3) / 1e9
    achieved_survey = b_survey * n / (kern_ms * 1e-3) / 1e9

    if rank == 0:
        # PMC numbers come from separate rocprofv3 --pmc passes over this same command (tools/profile_gpu.sh), committed under
        # profiles/.  FETCH_SIZE x2: gfx950 tallies 128-B read requests at 64 B (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact.
        traffic, valu, issue = None, {}, None
        clock_hz = 1e3 * float(getattr(torch.cuda.get_device_properties(dev), "clock_rate", 2.4e6))   # kHz -> Hz (2.4 GHz)
        tag = args.config + ("_norand" if args.no_randomizer else "")
        if os.environ.get("ORR_STEP_WAVES_PER_EU") == "1" and n > 4 * 4 * torch.cuda.get_device_properties(dev).multi_processor_count:
            tag += "_wpe1"          # a batch that would run the two-waves-per-SIMD variant, forced onto the one-wave kernel (comparison runs)
        if os.environ.get("ORR_STEP_WAVES_PER_EU") == "2" and n <= 4 * 4 * torch.cuda.get_device_properties(dev).multi_processor_count:
            tag += "_wpe2"          # the reverse (a small batch forced onto the two-wave kernel): the committed counters of this config are another kernel's
        lib_hash = env.L.orr_source_hash().decode()
        pmc_stale, pmc_seen = None, []
        for rnd in ("r06", "r05", "r04", "r03", "r02"):
            name = "%s_%s_pmc_summary.json" % (rnd, tag)
            pmc = os.path.join(ROOT, "profiles", name)
            if not os.path.exists(pmc) or n != CONFIGS[args.config][1]:
                continue
            try:
                have = json.load(open(pmc)).get("source_hash")
            except Exception:            # noqa: BLE001
                have = None
            pmc_seen.append({"file": "profiles/" + name, "source_hash": have})
            if have != lib_hash:
                # counters of OTHER kernel sources (or of a summary that does not say which): not this run's kernel, never reported as such
                pmc_stale = True
                continue
            pmc_stale = False
            try:
                d = json.load(open(pmc))
                traffic = d.get("hbm_bytes_per_launch_fetch_x2")
                waves = d["SQ_WAVES"]["mean_per_launch"]
                valu = {"source": "profiles/" + name,
                        "valu_insts_per_robot_step": d["SQ_INSTS_VALU"]["mean_per_launch"] / n,
                        "valu_active_frac_of_wave_cycles": d["SQ_ACTIVE_INST_VALU"]["mean_per_launch"] / d["SQ_WAVE_CYCLES"]["mean_per_launch"],
                        "wait_any_frac_of_wave_cycles": d["SQ_WAIT_ANY"]["mean_per_launch"] / d["SQ_WAVE_CYCLES"]["mean_per_launch"],
                        "waves_per_simd": waves / 1024.0}
                # THE BOUND THAT BINDS: VALU issue of the waves resident on a SIMD.  A lone wave issues at most one VALU instruction
                # per 4 cycles (MI355X_MICROARCH.md: issue cadence of a single wave), a SIMD with >= 2 resident waves one per 2.
                # Batches of up to 4 x #SIMDs robots run the one-wave-per-SIMD variant of the kernel (every wave alone on its SIMD:
                # the lone-wave ceiling applies), larger ones the two-waves-per-SIMD variant (orr_step; ORR_STEP_WAVES_PER_EU overrides).
                ipw = d["SQ_INSTS_VALU"]["mean_per_launch"] / waves
                simds = 4 * torch.cuda.get_device_properties(dev).multi_processor_count
                forced = os.environ.get("ORR_STEP_WAVES_PER_EU", "")
                resident = int(forced) if forced in ("1", "2") else (2 if waves > simds else 1)
                per_simd = max(1.0, waves / float(simds))            # waves a SIMD runs per launch (together or one after the other)
                kcyc = kern_ms * 1e-3 * clock_hz
                issue = {"source": "profiles/" + name + " (instruction counts) x this run's kernel_ms",
                         "insts_per_wave": ipw, "waves_per_simd_per_launch": per_simd, "waves_resident_per_simd": resident,
                         "cycles_per_inst_lone_wave": 4, "cycles_per_inst_simd": 2,
                         "shader_clock_ghz": clock_hz / 1e9, "kernel_cycles": kcyc,
                         "frac_of_lone_wave_ceiling": (ipw * per_simd * 4.0 / kcyc) if resident == 1 else None,
                         "frac_of_simd_peak": ipw * per_simd * 2.0 / kcyc,
                         "tail_frac": None}
                for rnd2 in ("r06", "r05", "r04", "r03", "r02"):
                    tl = os.path.join(ROOT, "profiles", "%s_wave_timeline.txt" % rnd2)
                    if args.config == "laikago4096" and not args.no_randomizer and os.path.exists(tl):
                        rows = {m.group(1).strip(): float(m.group(2)) for m in (re.match(r"^(.*\S)\s+([0-9.]+)$", ln.rstrip()) for ln in open(tl)) if m}
                        end, mean = rows.get("latest wave end = launch length (us)"), rows.get("mean wave duration (us)")
                        if end and mean:
                            issue["tail_frac"] = 1.0 - mean / end
                            issue["tail_source"] = ("profiles/%s_wave_timeline.txt (1 - mean wave duration / launch length); most of it is waves that end "
                                                    "EARLY - wave-level skips of contact work none of their robots needs - not late ones: "
                                                    "profiles/r03_wave_spread.txt, DESIGN.md section 6") % rnd2
                            break
                break
            except Exception as e:       # noqa: BLE001
                traffic, valu, issue = None, {"error": repr(e)}, None
        # which build of the step kernel ran (orr_step: more waves than SIMDs -> the two-waves-per-SIMD variant; name as in the rocprofv3 trace)
        forced_wpe = os.environ.get("ORR_STEP_WAVES_PER_EU", "")
        kernel_wpe = int(forced_wpe) if forced_wpe in ("1", "2") else (2 if (n + 3) // 4 > 4 * torch.cuda.get_device_properties(dev).multi_processor_count else 1)
        out = {
            "metric": "env steps/sec at N parallel robots", "value": world * n * args.steps / elapsed, "unit": "env steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "warmup_internal": WARMUP_FLOOR,
            "ms_per_step": 1e3 * elapsed / args.steps, "repeats": len(regions),
            "repeat_values": [world * n * args.steps / r["elapsed"] for r in regions],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "dtype_ref": "f64", "data": "synthetic",
            "config": {"workload": workload % n, "name": args.config,
                       "robots_per_gpu": n, "total_robots": world * n, "substeps_per_step": 33, "solver_iters": 9,
                       "randomizer": not args.no_randomizer, "control_latency_s": "U(0, 0.04) per episode" if not args.no_randomizer else 0.002,
                       "auto_reset": True, "actions": "reference pose + N(0,0.125^2), on device (one elementwise launch per step: orr_stress_actions; timed)",
                       "launch": "eager", "collective": "all_gather of episode returns every %d steps and at the end" % ROLLOUT,
                       "episodes_gathered": n_eps},
            "timed_breakdown": {"kernel_ms_total": kern_total_ms, "gather_ms": 1e3 * gather_s,
                                "other_ms": 1e3 * elapsed - kern_total_ms - 1e3 * gather_s,
                                "kernel_launches_timed": n_ev,
                                "note": "rank 0; kernel = median HIP-event duration of the bracketed orr_step_kernel launches x steps; gather = host time "
                                        "from the last queued kernel's end to the end of each rollout-boundary gather; other = action "
                                        "launches, launch gaps, barriers"},
            "dist": dist_info,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         # the waste ratios, hoisted: counter traffic over the algorithmic bytes of the built layout / of SURVEY 8(d)'s figure
                         # (> 1 = re-reads and spills), and the share of wave cycles spent waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES)
                         "traffic_over_alg": (traffic / (b_alg * n)) if traffic else None,
                         "traffic_over_alg_survey": (traffic / (b_survey * n)) if traffic else None,
                         "wait_any_frac": valu.get("wait_any_frac_of_wave_cycles") if valu else None,
                         # the same with SURVEY.md section 8(d)'s per-unit figure instead of the built layout's
                         "alg_bytes_survey": b_survey, "achieved_survey": achieved_survey, "frac_survey": achieved_survey / HBM_PEAK_GBS,
                         # THE BOUND THAT BINDS (valu_issue below, hoisted): share of a lone wave's VALU issue ceiling (one instruction per
                         # 4 cycles; one-wave kernel only) and of the SIMD's peak (one per 2 cycles) that the launch sustains
                         "frac_of_lone_wave_ceiling": issue["frac_of_lone_wave_ceiling"] if issue else None,
                         "frac_of_simd_peak": issue["frac_of_simd_peak"] if issue else None,
                         "pmc_stale": pmc_stale, "pmc_source_hash": lib_hash, "pmc_summaries_seen": pmc_seen,
                        "kernel": "orr_step_kernel<0, %d>" % kernel_wpe, "kernel_ms": kern_ms, "kernel_ms_mean": mid["kern_ms_mean"],
                         "kernel_ms_min": mid["kern_ms_min"], "kernel_ms_max": mid["kern_ms_max"], "kernel_ms_back_to_back": kern_b2b_ms,
                         "alg_bytes_per_robot_step": b_alg, "alg_bytes_per_launch": b_alg * n,
                         "pmc": valu, "valu_issue": issue,
                         "note": "instruction-issue-bound serial chain of %s (33 x (leg dynamics + rows + 9 PGS sweeps)); HBM fraction is reported "
                                 "because the north star asks for it, see DESIGN.md section 6"
                                 % ("a lone wave per SIMD" if kernel_wpe == 1 else "two co-resident waves per SIMD")},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(env)
        print(json.dumps(out))
    env.close()
    if torch.distributed.is_available() and torch.distributed.is_initialized():     # world > 1, or ORR_FORCE_DIST=1 (one-rank RCCL rehearsal)
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
