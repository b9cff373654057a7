#!/usr/bin/env python3
"""The float32 noise floor's OWN spread on the drift test's exact sample (runs on the GPU box).

tests/test_gpu_drift.py holds the HIP path's error against the float64 oracle to a multiple of the float32 oracle's error.  Round 3
set the multiples by hand (2 / 3 / 10) and loosened them twice after red runs.  This tool MEASURES them: on the test's sample (1024
robots, env seed 31, action seed 1, start states from the HIP reset) it runs
   run 0        the float32 parity build (-O2 -ffp-contract=off) from the test's start state: what the test compares the HIP path with
   runs 1..K    the same build from start states whose rigid-state components differ from it by ONE float32 ulp (random direction)
   run K+1      the -O3 -march=native float32 build from the unperturbed start
and takes, per field / horizon / quantile, the largest ratio between any two of these runs: what float32 noise alone does to that
quantile.  factor = max(1.5, 1.25 x that ratio).  For the task-level fields (reference pose, target observation, reward) the max is
replaced by a COUNT of discrete-event outliers (tests/drift.py).  The HIP path's own numbers are printed beside the floor's.

Writes tests/golden/drift_factors.json (what the test asserts against) and gpurun_out/drift_floor_spread_<robot>.txt (copied to
profiles/r05_drift_floor_spread_<robot>.txt).

usage: python tools/drift_floor_spread.py [--runs 12] [--out-dir gpurun_out]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=12, help="perturbed float32 runs (K)")
    ap.add_argument("--out-dir", default=os.path.join(ROOT, "gpurun_out"))
    ap.add_argument("--json", default=os.path.join(ROOT, "tests", "golden", "drift_factors.json"))
    args = ap.parse_args()
    from openroborl_amd import _abi, _lib
    from tests import drift, oracle_lib as ol
    from tests.test_gpu_drift import _three
    os.makedirs(args.out_dir, exist_ok=True)
    result = {"_about": "derived by tools/drift_floor_spread.py on the GPU box; see its docstring and tests/drift.py", "runs_perturbed": args.runs,
              "margin": drift.MARGIN, "factor_min": drift.FACTOR_MIN, "outlier_x": drift.OUTLIER_X, "source_hash": _lib.library_hash(),
              "oracle_hash": __import__("hashlib").sha256(open(os.path.join(ROOT, "oracle", "orr_oracle.c"), "rb").read()).hexdigest()[:32]}
    for robot in ("laikago", "mini_cheetah"):
        n = 1024
        env, o64, o32, dev_step, dev_state = _three(robot, n, seed=31)
        start32 = o32.state.copy()
        rng = np.random.RandomState(777)
        floors = [o32]
        for _ in range(args.runs):
            f = ol.OracleEnv(env.cfg, env.models, env.clips, n, robot_type=env.robot_type, clip_id=env.clip_id, threads=16, f32="parity")
            f.reset()
            f.state[:] = drift.perturb_one_ulp(start32, env.layout, rng)
            f.obs[:] = o32.obs
            floors.append(f)
        fn = ol.OracleEnv(env.cfg, env.models, env.clips, n, robot_type=env.robot_type, clip_id=env.clip_id, threads=16, f32=True)
        fn.reset()
        fn.state[:] = start32
        fn.obs[:] = o32.obs
        floors.append(fn)
        out, alive = drift.run_family(o64, floors, env.layout, env.models, env.robot_type, steps=25, seed=1, dev_step=dev_step, dev_state64=dev_state)
        tab = drift.spread_table(out, alive, n_floor=len(floors))
        text = drift.format_spread(tab, "%s, %d robots, randomiser on, actions = reference pose + N(0, 0.125^2); float32 runs: parity build, %d x "
                                        "one-ulp-perturbed starts, -O3 -march=native build; factor = max(%.1f, %.2f x max/min)"
                                   % (robot, n, args.runs, drift.FACTOR_MIN, 1.0 + drift.MARGIN))
        print(text, flush=True)
        open(os.path.join(args.out_dir, "drift_floor_spread_%s.txt" % robot), "w").write(text + "\n")
        result[robot] = {"factors": {str(h): {k: v for k, v in tab[h].items() if k != "alive"} for h in tab}}
        for f in floors[1:]:
            f.close()
        env.close(); o64.close(); o32.close()

    # the latency-ring scenario of test_latency_ring_multi_step_without_resync (64 robots, env seed 11, three steps of uniform actions)
    n = 64
    env, o64, o32, dev_step, dev_state = _three("laikago", n, seed=11)
    start32 = o32.state.copy()
    rng = np.random.RandomState(778)
    floors = [o32]
    for _ in range(args.runs):
        f = ol.OracleEnv(env.cfg, env.models, env.clips, n, robot_type=env.robot_type, clip_id=env.clip_id, threads=16, f32="parity")
        f.reset()
        f.state[:] = drift.perturb_one_ulp(start32, env.layout, rng)
        f.obs[:] = o32.obs
        floors.append(f)
    arng = np.random.RandomState(3)
    acts = [arng.uniform(-0.15, 0.15, (n, 12)).astype(np.float32) for _ in range(3)]
    out, alive = drift.run_family(o64, floors, env.layout, env.models, env.robot_type, steps=3, actions=acts, dev_step=dev_step, dev_state64=dev_state)
    m = alive[3]
    per_run = [drift.ring_errors(f.state.astype(np.float64), o64.state, env.layout, m, _abi.RING_DEPTH, _abi.RING_ENTRY) for f in floors]
    ring = drift.ring_spread(per_run)
    dev = drift.ring_errors(dev_state(), o64.state, env.layout, m, _abi.RING_DEPTH, _abi.RING_ENTRY)
    lines = ["latency ring after 3 un-synced env steps, 64 robots (latencies 0-40 ms), %d float32 runs" % len(floors)]
    for name, _ in drift.RING_GROUPS:
        for q, pct in (("median", 50.0), ("max", 100.0)):
            r = ring[name][q]
            r["device"] = float(np.percentile(dev[name], pct))
            lines.append("%-12s %-6s | f32 run 0 %.2e  min %.2e  max %.2e | max/min %6.2f  factor %6.2f | HIP %.2e (%.2f x run 0)"
                         % (name, q, r["q0"], r["min"], r["max"], r["max_ratio_between_two_runs"], r["factor"], r["device"], r["device"] / max(r["q0"], 1e-30)))
    print("\n".join(lines))
    open(os.path.join(args.out_dir, "drift_floor_spread_ring.txt"), "w").write("\n".join(lines) + "\n")
    result["ring"] = ring
    with open(args.json, "w") as f:
        json.dump(result, f, indent=1, sort_keys=True)
    # gpurun only brings gpurun_out/ back
    with open(os.path.join(args.out_dir, "drift_factors.json"), "w") as f:
        json.dump(result, f, indent=1, sort_keys=True)
    print("written", args.json)


if __name__ == "__main__":
    main()
