"""Float32 drift calibration (test infrastructure): how far does a float32 evaluation of the env step drift from the float64 oracle
over 1 .. 25 env steps of contact dynamics, and is the HIP path's drift of that size?

Three runs from ONE float32-representable start state, with the same actions:
  f64  the parity oracle (oracle/liborr_oracle.so)
  f32  the same C source compiled with `real = float`, -O2 -ffp-contract=off, no -march=native / fast-math (make -C oracle f32):
       the NOISE FLOOR of a float32 implementation of this algorithm
  dev  the device under test (the HIP path through the C-ABI; or, on a box without a GPU, anything with the same interface)
Errors are taken against f64 per robot (max over the components of a field), then summarised as quantiles over the robots
that are still in their first episode on all three sides.
"""
import numpy as np

FIELDS = ("POS", "QUAT", "Q", "LINVEL", "ANGVEL", "QD", "REF_POSE")
OBS_GROUPS = (("obs_imu", slice(0, 12)), ("obs_lastact_motorang", slice(12, 84)), ("obs_target", slice(84, 160)))
HORIZONS = (1, 3, 10, 25)
QUANTS = (("median", 50.0), ("p99", 99.0), ("max", 100.0))


def reference_tracking_actions(obs, models, robot_type, rng, noise=0.125):
    """bench.py's synthetic policy: reference joint pose one control step ahead -> motor space - init + N(0, noise^2)."""
    n = obs.shape[0]
    jom = np.stack([models[t]["joint_of_motor"] for t in robot_type])
    off = np.stack([models[t]["motor_offset"] for t in robot_type])
    mdir = np.stack([models[t]["motor_dir"] for t in robot_type])
    init = np.stack([models[t]["init_motor_angles"] for t in robot_type])
    tar = np.take_along_axis(obs[:, 84 + 7:84 + 19], jom, axis=1)
    return np.clip((tar - off) * mdir - init + rng.randn(n, 12) * noise, -2 * np.pi, 2 * np.pi).astype(np.float32)


def run_three_way(dev_step, dev_state64, orc64, orc32, lay, models, robot_type, steps, seed=0, noise=0.125, actions=None):
    """Advance the three sides `steps` env steps with identical actions (driven by the float64 oracle's observation unless a list
    of action arrays is given).  Returns {horizon: {name: (err_dev[n], err_f32[n])}}, alive mask per horizon."""
    n = orc64.n
    rng = np.random.RandomState(seed)
    obs64 = orc64.obs.copy()
    alive = np.ones(n, dtype=bool)
    out, alive_at = {}, {}
    for k in range(steps):
        a = actions[k] if actions is not None else reference_tracking_actions(obs64, models, robot_type, rng, noise)
        od, rd, dd = dev_step(a)
        o32, r32, d32 = orc32.step(a)
        obs64, r64, d64 = orc64.step(a.astype(np.float64))
        h = k + 1
        if h in HORIZONS or h == steps:
            sd = dev_state64()
            rec = {}
            for name in FIELDS:
                sl = lay.sl(name)
                rec[name] = (np.abs(sd[:, sl] - orc64.state[:, sl]).max(axis=1),
                             np.abs(orc32.state[:, sl].astype(np.float64) - orc64.state[:, sl]).max(axis=1))
            for name, sl in OBS_GROUPS:
                rec[name] = (np.abs(od[:, sl] - obs64[:, sl]).max(axis=1), np.abs(o32[:, sl].astype(np.float64) - obs64[:, sl]).max(axis=1))
            rec["reward"] = (np.abs(rd - r64), np.abs(r32.astype(np.float64) - r64))
            out[h] = rec
            alive_at[h] = alive.copy()       # robots that had not terminated BEFORE this step on any side
        alive &= ~(np.asarray(dd, dtype=bool) | np.asarray(d32, dtype=bool) | np.asarray(d64, dtype=bool))
    return out, alive_at


def quantile_table(out, alive_at):
    """{horizon: {name: {"dev": {q: v}, "f32": {q: v}}}} over the alive robots."""
    tab = {}
    for h, rec in out.items():
        m = alive_at[h]
        tab[h] = {"alive": int(m.sum())}
        for name, (ed, e32) in rec.items():
            tab[h][name] = {"dev": {q: float(np.percentile(ed[m], p)) for q, p in QUANTS},
                            "f32": {q: float(np.percentile(e32[m], p)) for q, p in QUANTS}}
    return tab


def format_table(tab, title=""):
    lines = ["%s" % title, "%-22s %3s %6s | %-32s | %-32s | %s" % ("field", "h", "alive", "HIP - f64 (median / p99 / max)", "f32 oracle - f64", "ratio")]
    for h in sorted(tab):
        for name in list(FIELDS) + [g for g, _ in OBS_GROUPS] + ["reward"]:
            d, f = tab[h][name]["dev"], tab[h][name]["f32"]
            lines.append("%-22s %3d %6d | %9.2e %9.2e %9.2e   | %9.2e %9.2e %9.2e   | %5.2f %5.2f %5.2f" % (
                name, h, tab[h]["alive"], d["median"], d["p99"], d["max"], f["median"], f["p99"], f["max"],
                d["median"] / max(f["median"], 1e-30), d["p99"] / max(f["p99"], 1e-30), d["max"] / max(f["max"], 1e-30)))
    return "\n".join(lines)


# HIP quantile <= FACTOR x float32-oracle quantile + FLOOR (a few float32 ulps, for fields where the float32 oracle happens to be exact).
# Two float32 builds of the SAME source (-O2 vs -O3 -march=native) differ from each other by 0.4-1.4x on the median / p99 and
# 0.2-4.4x on the max over 1024 robots (heavy tail), measured on the CPU; HIP vs the -O2 build on the GPU box: 0.5-1.0 / 0.3-2.1 / 0.1-2.1.
# The tail quantiles are a handful of robots in chaotic contact states: an unrelated change of the kernel's rounding (one 6x6 solve
# re-associated) moved a p99 ratio from 1.08 to 2.05, so only the MEDIAN is held to the factor of 2; p99 gets 3.  The MAX over 1024
# robots is asserted (factor 10) only for the rigid state: the task-level fields (reference pose, target observation, reward) contain
# discrete events - the cycle sync re-anchors the reference origin in the step where the phase wraps, quaternions are standardised to
# w >= 0 - which ONE robot of a thousand takes a step earlier or later than the float64 oracle (seen: target-observation max 1.0 in the
# HIP run and 0.13 in the float32 oracle after a recompile that changed nothing but instruction scheduling); they are reported only.
FACTOR = {"median": 2.0, "p99": 3.0, "max": 10.0}
MAX_ASSERTED = ("POS", "QUAT", "Q", "LINVEL", "ANGVEL", "QD")
FLOOR = 2e-6


SMALL_SAMPLE = (("median", 50.0, 2.0), ("p90", 90.0, 3.0))     # tests with 64-256 robots: the max of a heavy-tailed sample is one robot's chaos


def assert_within_float32_floor(err_dev, err_f32, what, quantiles=SMALL_SAMPLE, floor=FLOOR):
    """err_*: per-robot errors against the float64 oracle (device under test / float32 oracle); quantiles: (name, percentile, factor)."""
    for q, pct, fac in quantiles:
        d, f = float(np.percentile(err_dev, pct)), float(np.percentile(err_f32, pct))
        assert d <= fac * f + floor, "%s %s: device error %.3g > %.1f x float32-oracle error %.3g (+ %.0e)" % (what, q, d, fac, f, floor)
