"""CPU-side checks of the measurement tools whose results are quoted in DESIGN.md: the mini-cheetah table in robots.py IS the identified
candidate of tools/mc_identify.py (profiles/r03_mc_identify.json), and the drift statistics behave on synthetic data."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_round3_minicheetah_table_is_its_records_candidate():
    """ROUND 3's table (robots.MINI_CHEETAH_R03, shipped in rounds 3-5, superseded in round 6) is the candidate of profiles/r03_mc_identify.json."""
    import mc_identify as mi
    from openroborl_amd import robots
    rec = json.load(open(os.path.join(ROOT, "profiles", "r03_mc_identify.json")))
    assert rec["verdict"] == "accepted" and rec["best"]["detail"]["F"] >= 0.9 and rec["best"]["robustness"]["mean_F"] >= 0.8
    best = rec["best"]["params"]
    lo = {k: v[1] for k, v in mi.PARAMS.items()}
    hi = {k: v[2] for k, v in mi.PARAMS.items()}
    assert all(lo[k] - 1e-12 <= best[k] <= hi[k] + 1e-12 for k in mi.NAMES)          # inside the stated plausible intervals
    # robots.py carries the candidate rounded to three digits, with toe radius and friction left at their round-2 values (both within
    # the +-10 % cloud the candidate was checked against; the sensitivity sweep shows neither matters)
    shipped = dict(toe_m=0.214, lo_m=0.091, lo_com_z=-0.073, toe_r=0.0175, hip_z=0.011, foot_mu=1.0, shank_r=0.0094, shank_at=0.0196,
                   up_com_z=-0.023, limits=0.0)
    for k in mi.NAMES:
        tol = 0.1 * abs(best[k]) + 1e-3
        assert abs(shipped[k] - best[k]) <= tol, (k, shipped[k], best[k])
    m_tool = mi.build_model(np.array([shipped[k] for k in mi.NAMES]))
    m_ship = robots.mini_cheetah(**robots.MINI_CHEETAH_R03)
    for key, val in m_ship.items():
        if isinstance(val, str):
            continue
        np.testing.assert_allclose(np.asarray(m_tool[key], dtype=float), np.asarray(val, dtype=float), atol=2e-6, err_msg=key)
    # what was NOT varied is what the reference fixes: control constants, link lengths, base / hip / thigh masses
    old = mi.build_model(np.array([mi.PARAMS[k][0] for k in mi.NAMES]))
    for key in ("kp", "kd", "init_motor_angles", "motor_dir", "motor_offset", "joint_of_motor", "init_pos", "init_quat", "base_mass"):
        np.testing.assert_array_equal(np.asarray(old[key]), np.asarray(m_ship[key]), err_msg=key)
    np.testing.assert_allclose(old["link_mass"][[0, 1]], m_ship["link_mass"][[0, 1]])   # hip and thigh masses


def test_drift_statistics_on_synthetic_errors():
    from tests import drift
    rng = np.random.RandomState(0)
    n = 512
    e32 = np.abs(rng.standard_cauchy(n)) * 1e-4                 # heavy-tailed, like the real thing
    out = {h: {name: (e32 * 0.9, e32) for name in list(drift.FIELDS) + [g for g, _ in drift.OBS_GROUPS] + ["reward"]} for h in drift.HORIZONS}
    alive = {h: np.ones(n, dtype=bool) for h in drift.HORIZONS}
    tab = drift.quantile_table(out, alive)
    assert tab[1]["POS"]["dev"]["median"] == 0.9 * tab[1]["POS"]["f32"]["median"]
    assert "ratio" in drift.format_table(tab).splitlines()[1]
    drift.assert_within_float32_floor(e32 * 1.5, e32, "ok")
    try:
        drift.assert_within_float32_floor(e32 * 4.0, e32, "must fail")
    except AssertionError:
        pass
    else:
        raise AssertionError("a 4x larger median was accepted")


def test_round5_laikago_table_is_its_records_candidate_and_the_record_follows_its_protocol():
    """ROUND 5's table (robots.LAIKAGO_R05, superseded in round 6) IS the chosen candidate of tools/laikago_identify.py's recorded run
    (profiles/r05_laikago_identify.json), the run followed the protocol stated in the tool's docstring (fit on trot + spin, hold-out run
    once on the chosen candidate), and what the search did NOT vary is what the reference fixes."""
    import laikago_identify as li
    from openroborl_amd import robots
    rec = json.load(open(os.path.join(ROOT, "profiles", "r05_laikago_identify.json")))
    assert rec["fit"] == ["laikago_trot", "laikago_spin"] and rec["holdout"] == ["laikago_trot0", "laikago_pace"]
    assert "BEFORE the sweep was run" in li.__doc__ and rec["criterion"].split()[:8] == li.__doc__.split("PROTOCOL AND CRITERION")[1].split()[:8]
    ch = rec["chosen"]
    assert rec["verdict"] == "accepted" and all(ch["fit"][p]["F"] >= 0.8 for p in rec["fit"])            # rule 2
    assert ch["robustness"]["mean_score"] >= 0.6                                                         # rule 3: not a knife edge
    assert all(rec["shipped_table_fit"][p]["F"] <= 0.01 for p in rec["fit"])                             # where the search started: nobody walks
    assert rec["row_c_pinned_by_holdout"] == all(ch["holdout"][p]["F"] >= 0.5 for p in rec["holdout"])   # rule 5, whatever it came to
    for k, (v0, lo, hi) in rec["params"].items():
        assert lo - 1e-12 <= ch["theta"][k] <= hi + 1e-12, k                                             # inside the stated box
        assert (v0, lo, hi) == tuple(li.PARAMS[k][:3]), k                                                # the box in the tool is the box of the run
    # ONE entry differs, on purpose: the hip plane's height is pinned by the clips (stance toes on the ground: tools/diag/clip_toe_clearance.py),
    # not by a policy; the search's winner had put it 2.4 cm lower, the fit-set ablation shows the fit does not care, the calibrated value ships
    assert abs(ch["theta"]["hip_z"] - (-0.068136)) < 1e-5
    m_tool, m_ship = li.build_model(dict(ch["theta"], hip_z=-0.044)), robots.laikago(**robots.LAIKAGO_R05)
    for key, val in m_ship.items():
        if isinstance(val, str):
            continue
        b = np.asarray(m_tool[key], dtype=float)
        np.testing.assert_allclose(np.asarray(val, dtype=float), b, rtol=3e-5, atol=3e-5 * max(1.0, float(np.abs(b).max())), err_msg=key)   # 5 digits
    # the reference point of the search = round 4's table, reproduced by robots.LAIKAGO_R04
    r4, t4 = robots.laikago(**robots.LAIKAGO_R04), li.build_model(li.shipped_theta())
    for key, val in r4.items():
        if not isinstance(val, str):
            np.testing.assert_allclose(np.asarray(val, dtype=float), np.asarray(t4[key], dtype=float), atol=1e-9, err_msg=key)
    for key in ("kp", "kd", "init_motor_angles", "motor_dir", "motor_offset", "joint_of_motor", "init_pos", "init_quat", "joint_axis"):
        np.testing.assert_array_equal(np.asarray(m_ship[key]), np.asarray(r4[key]), err_msg=key)          # control constants, conventions
    # link lengths (trans2minicheetah.m:3-5): knee below the hip pitch axis, toe below the knee
    assert np.allclose(m_ship["joint_pos"][2], [0, 0, -0.25223]) and np.allclose(m_ship["toe_pos"][0], [0, 0, -0.251])
    # the solver constants that shipped in round 5 were the library defaults, not the candidate's: the candidate's table is accepted under them too
    ab = json.load(open(os.path.join(ROOT, "profiles", "r05_laikago_identify_ablation.json")))
    assert all(ab["table_with_shipped_config"]["fit"][p]["F"] >= 0.8 for p in rec["fit"])
    assert ab["single_reverted"]["soft"]["score"][0] < 0.5 and ab["single_reverted"]["foot_friction"]["score"][0] < 0.5    # what it hangs on
    assert min(ab["single_reverted"][k]["score"][0] for k in ("chassis", "hip_r", "knee_r")) >= 0.8                      # not the fall proxies
    assert ab["single_reverted"]["hip_z"]["score"][0] >= 0.85                                                          # nor the hip height (see above)


def test_laikago_identify_runs_end_to_end_on_the_oracle_backend(tmp_path):
    """The identification tool's whole flow - random stage, local stage, acceptance, closest-accepted choice with its cloud, the fit-alone
    and the once-only hold-out evaluation, the JSON record, and the fit-only ablation of that record - at toy size on the CPU oracle
    (2 robots, 12 steps: the numbers mean nothing, the plumbing is what is tested; the GPU run is profiles/r05_laikago_identify.json)."""
    import subprocess
    out = str(tmp_path / "li.json")
    tool = os.path.join(ROOT, "tools", "laikago_identify.py")
    r = subprocess.run([sys.executable, tool, "--backend", "oracle", "--robots", "2", "--steps", "12", "--minutes", "0.12", "--out", out],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.load(open(out))
    assert rec["fit"] == ["laikago_trot", "laikago_spin"] and rec["holdout"] == ["laikago_trot0", "laikago_pace"]
    assert rec["search"]["candidates"] >= 32 and set(rec["chosen"]["holdout"]) == set(rec["holdout"]) and set(rec["chosen"]["fit_alone"]) == set(rec["fit"])
    assert all(set(c["fit"]) == set(rec["fit"]) for c in rec["search"]["top_by_score"])          # no candidate of the search saw a hold-out policy
    ab = str(tmp_path / "ab.json")
    r = subprocess.run([sys.executable, tool, "--backend", "oracle", "--robots", "2", "--steps", "12", "--ablate", out, "--out", ab],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    a = json.load(open(ab))
    assert a["fit"] == rec["fit"] and "table_with_shipped_config" in a and "greedy" in a


def test_solver_constants_follow_the_recorded_cross_robot_rule():
    """config.make_config's solver constants are what tools/identify_r6.py's rule P5 decided (profiles/r06_constants_rule.json): the set
    PyBullet is remembered to run with, adopted because it is preferred on each robot's policies and costs the other robot's nothing."""
    from openroborl_amd import config
    rec = json.load(open(os.path.join(ROOT, "profiles", "r06_constants_rule.json")))
    assert rec["sets"]["PYB"] == config.PYBULLET_REMEMBERED and rec["sets"]["LIB"] == config.BULLET_LIBRARY_DEFAULTS
    a, b = rec["direction_A"], rec["direction_B"]
    assert rec["adopt_PYB"] == (a["preferred_on_laikago"] and a["validates_on_mini_cheetah"] and b["preferred_on_mini_cheetah"] and b["validates_on_laikago"])
    # the rule, recomputed from the recorded cells
    lk = ["laikago_pace", "laikago_spin", "laikago_trot", "laikago_trot0"]
    cell = rec["cells"]
    assert a["preferred_on_laikago"] == (np.mean([cell["PYB/" + p]["J"] for p in lk]) >= np.mean([cell["LIB/" + p]["J"] for p in lk]))
    assert b["validates_on_laikago"] == all(cell["LIB/" + p][k] - cell["PYB/" + p][k] <= 0.02 for p in lk for k in ("F", "J"))
    assert a["validates_on_mini_cheetah"] == all(cell["LIB/minicheetah_trot"][k] - cell["PYB/minicheetah_trot"][k] <= 0.02 for k in ("F", "J"))
    c = config.make_config(4)
    want = config.PYBULLET_REMEMBERED if rec["adopt_PYB"] else config.BULLET_LIBRARY_DEFAULTS
    for k, v in want.items():
        assert abs(getattr(c, k) - v) < 1e-7, k


def test_identify_r6_box_freezes_what_the_protocol_says_it_freezes():
    """tools/identify_r6.py P1 / P8: whatever the candidate, the clip-pinned hip height and toe radius, the termination-only proxies, the
    Laikago's shank sphere and every constant the reference states are those of the reference-point table; the reference point of the box
    IS round 4's Laikago table / round 2's mini-cheetah table with the clip-calibrated hip height."""
    import identify_r6 as ir
    from openroborl_amd import robots
    rng = np.random.RandomState(3)
    for robot, ref in (("laikago", robots.laikago(**robots.LAIKAGO_R04)), ("mini_cheetah", robots.mini_cheetah(**dict(robots.MINI_CHEETAH_R02, hip_z=0.011)))):
        spec = ir.SPECS[robot]
        m0 = ir.build_model(robot, ir.reference_theta(spec))
        for key, val in ref.items():
            if not isinstance(val, str) and key not in ("link_inertia",):          # mini-cheetah shank inertia: the slender-rod rule of round 3
                np.testing.assert_allclose(np.asarray(m0[key], dtype=float), np.asarray(val, dtype=float), atol=1e-12, err_msg=key)
        assert ir.distance(spec, ir.reference_theta(spec)) == 0.0
        for i in range(24):
            th = ir.random_theta(spec, rng, i % 2)
            m = ir.build_model(robot, th)
            for k, (v0, lo, hi) in spec["params"].items():
                assert lo <= th[k] <= hi and lo <= v0 <= hi
            frozen = ["kp", "kd", "init_motor_angles", "motor_dir", "motor_offset", "joint_of_motor", "init_pos", "init_quat", "joint_axis", "toe_radius",
                      "fall_radius", "fall_body", "toe_pos"] + (["shank_radius", "shank_pos"] if robot == "laikago" else [])
            for key in frozen:
                np.testing.assert_array_equal(np.asarray(m[key]), np.asarray(ref[key]), err_msg=key)
            assert np.allclose(m["joint_pos"][0::3, 2], ref["joint_pos"][0::3, 2])                   # hip plane height (clip toe clearance)
            # chassis corners move only with the COM shift (they are attached to the hips' centre), never in size
            assert np.allclose(np.ptp(np.asarray(m["fall_pos"])[:8], axis=0), np.ptp(np.asarray(ref["fall_pos"])[:8], axis=0))
    assert ir.splits()[3] == (["laikago_spin", "laikago_trot"], ["laikago_pace", "laikago_trot0"]) and len(ir.splits()) == 6   # round 5's split is one of the six


def test_identify_r6_runs_end_to_end_on_the_oracle_backend(tmp_path):
    """One run of the round-6 protocol (search on min-J, shortlist re-evaluation, cloud, once-only hold-out) and the smallest-table pass, at
    toy size on the CPU oracle: the plumbing, not the numbers (2 robots, 12 steps)."""
    import subprocess
    tool = os.path.join(ROOT, "tools", "identify_r6.py")
    out = str(tmp_path / "run.json")
    r = subprocess.run([sys.executable, tool, "run", "--backend", "oracle", "--robot", "laikago", "--fit", "laikago_trot", "laikago_spin", "--holdout",
                        "laikago_trot0", "laikago_pace", "--robots", "2", "--steps", "12", "--minutes", "0.1", "--out", out,
                        "--dump-all", str(tmp_path / "all.jsonl.gz")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.load(open(out))
    assert rec["fit"] == ["laikago_trot", "laikago_spin"] and rec["holdout"] == ["laikago_trot0", "laikago_pace"] and "P4. CROSS-VALIDATION" in rec["protocol"]
    ch = rec["chosen"]
    assert set(ch["fit_final"]["seed_1"]) == set(rec["fit"]) and set(ch["holdout"]["seed_1"]) == set(rec["holdout"])
    assert all(set(c["fit"]) == set(rec["fit"]) for c in rec["search"]["top_by_J"])                  # no candidate of the search saw a hold-out policy
    cell = ch["fit_final"]["seed_1"]["laikago_trot"]
    assert set(cell) >= {"F", "len", "J", "R", "terms", "dvx", "advx"} and set(cell["terms"]) == {"pose", "velocity", "end_effector", "root_pose", "root_velocity"}
    assert cell["J"] <= cell["R"] + 1e-9                                                                # J forfeits the reward after a failure
    mn = str(tmp_path / "min.json")
    r = subprocess.run([sys.executable, tool, "minimal", "--backend", "oracle", "--robots", "2", "--steps", "12", "--record", out, "--out", mn],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    m = json.load(open(mn))
    assert m["fit"] == rec["fit"] and set(m["minimal"]["still_moved"]) | {p["reverted"] for p in m["minimal"]["path"]} == set(m["single_reverted"])


def _check_all4_and_minimal(ir, robots, all4_name, min_name, freeze, moved):
    """P3 / P6 / P7 recomputed from one all-four run and its smallest-table pass; -> (model built from the record, the minimal record)."""
    spec = ir.spec_of("laikago", freeze)
    all4 = json.load(open(os.path.join(ROOT, "profiles", all4_name)))
    mn = json.load(open(os.path.join(ROOT, "profiles", min_name)))
    assert sorted(all4["fit"]) == ir.LAIKAGO_POLICIES and all4["holdout"] == [] and all4["verdict"] == "accepted" and bool(all4.get("freeze_geometry")) == freeze
    assert "P6. WHAT SHIPS" in all4["protocol"] and all4["protocol"].split()[:12] == ir.__doc__.split("==== PROTOCOL")[1].split("usage:")[0].split()[:12]
    assert ("P9. REVISION" in all4["protocol"]) == freeze                                      # P9 was written down before its runs, and after P6's
    assert set(all4["params"]) == set(spec["params"])
    for k, (v0, lo, hi) in all4["params"].items():
        assert (v0, lo, hi) == tuple(spec["params"][k]) and lo - 1e-12 <= all4["chosen"]["theta"][k] <= hi + 1e-12, k       # the tool's box is the run's box
    ch = all4["chosen"]
    assert ch["recheck"]["F"] >= 0.8 and ch["robustness"]["mean_min_F"] >= 0.8                                                # P3
    assert all(c["F"] >= 0.8 for c in ch["fit_final"]["seed_1"].values())
    # P7, recomputed: every step of the path kept acceptance and min-J within 0.01 of the chosen candidate's; nothing more could be put back
    assert mn["of"].endswith(all4_name) and mn["tolerance_J"] == 0.01
    assert {k: v for k, v in mn["minimal"]["theta"].items() if k not in mn["minimal"]["still_moved"] and k in spec["params"]} == \
           {k: spec["params"][k][0] for k in spec["params"] if k not in mn["minimal"]["still_moved"]}
    for step in mn["minimal"]["path"]:
        assert step["min_F"] >= 0.8 and step["min_J"] >= mn["chosen"]["min_J"] - 0.01
    for k, e in mn["minimal"]["effect_of_each_moved_entry"].items():
        assert e["min_F_if_put_back"] < 0.8 or e["min_J_if_put_back"] < mn["chosen"]["min_J"] - 0.01, k                      # each remaining entry costs something
    assert set(k for k in moved if k in spec["params"] or k in spec["switches"]) == set(mn["minimal"]["still_moved"])
    m_tool = ir.build_model("laikago", mn["minimal"]["theta"])
    m_ship = robots.laikago(**dict(robots.LAIKAGO_R04, **robots.laikago_theta_kwargs(moved)))
    for key, val in m_ship.items():
        if not isinstance(val, str):
            b = np.asarray(m_tool[key], dtype=float)
            np.testing.assert_allclose(np.asarray(val, dtype=float), b, rtol=3e-5, atol=3e-5 * max(1.0, float(np.abs(b).max())), err_msg=key)
    return m_ship, mn


def _check_cv(ir, name, freeze, seeds):
    cv = json.load(open(os.path.join(ROOT, "profiles", name)))
    assert [(r["fit"], r["holdout"]) for r in cv["splits"]] == [(f, h) for f, h in ir.splits()] and [r["seed"] for r in cv["splits"]] == seeds
    for r, row in zip(cv["splits"], cv["table"]):
        assert bool(r.get("freeze_geometry")) == freeze and ("hip_x" in r["chosen"]["theta"]) == (not freeze)
        assert set(r["chosen"]["holdout"]["seed_1"]) == set(r["holdout"]) and set(r["chosen"]["fit_final"]["seed_1"]) == set(r["fit"])
        assert r["transfers"] == (r["verdict"] == "accepted" and all(r["chosen"]["holdout"][s][p]["F"] >= 0.5 for s in r["chosen"]["holdout"] for p in r["holdout"]))
        assert row["transfers"] == r["transfers"] and len(row["cells"]) == 4
    held = [(row["split"], p, c["F"]) for row in cv["table"] for p, c in row["cells"].items() if c["role"] == "held out"]
    assert len(held) == 12
    return cv, held


def test_shipped_laikago_table_is_round6s_record_and_the_records_follow_the_protocol():
    """robots.laikago() IS the end of the smallest-table path (P7) of the all-four run of tools/identify_r6.py in its revision P9
    (profiles/r06_laikago_all4_p9.json, r06_laikago_minimal_p9.json), which ships by the rule P9 states; the first table of the round (P6 +
    P7, hip positions still in the box) is recorded the same way; both cross-validation records hold all six splits with their hold-out
    policies evaluated on the chosen candidate only; the rules are recomputed from the records."""
    import identify_r6 as ir
    from openroborl_amd import robots
    _, mn6 = _check_all4_and_minimal(ir, robots, "r06_laikago_all4.json", "r06_laikago_minimal.json", False, robots.LAIKAGO_R06_P6_MOVED)
    m9, mn9 = _check_all4_and_minimal(ir, robots, "r06_laikago_all4_p9.json", "r06_laikago_minimal_p9.json", True, robots.LAIKAGO_R06_MOVED)
    # P9's shipping rule: accepted on all four and min-J not more than 0.02 below the P6 / P7 table's
    assert all(c["F"] >= 0.8 for c in mn9["minimal"]["fit"].values()) and mn9["minimal"]["min_J"] >= mn6["minimal"]["min_J"] - 0.02
    m_ship = robots.laikago()
    for key, val in m_ship.items():
        if not isinstance(val, str):
            np.testing.assert_array_equal(np.asarray(val), np.asarray(m9[key]), err_msg=key)                # what ships = the P9 record
    r4 = robots.laikago(**robots.LAIKAGO_R04)
    for key in ("kp", "kd", "init_motor_angles", "motor_dir", "motor_offset", "joint_of_motor", "init_pos", "init_quat", "joint_axis", "toe_radius",
                "fall_radius", "fall_body", "shank_radius", "shank_pos", "joint_lo", "joint_hi"):
        np.testing.assert_array_equal(np.asarray(m_ship[key]), np.asarray(r4[key]), err_msg=key)          # the reference's constants + what P1 froze
    assert np.allclose(m_ship["joint_pos"][0::3, 2], -0.044) and np.allclose(m_ship["joint_pos"][2], [0, 0, -0.25223]) and np.allclose(m_ship["toe_pos"][0], [0, 0, -0.251])
    # hip_x / hip_y: laikago.py:54-59 (minus the coxa), behind the COM shift
    assert np.allclose(np.abs(m_ship["joint_pos"][0::3, 1]), 0.1157 - 0.032875) and np.allclose(m_ship["joint_pos"][0::3, 0] + 0.06, [0.21, 0.21, -0.21, -0.21])
    # P4, as it came out: pace walks on every table; spin is never predicted by a table that was not fitted on it
    cv, held = _check_cv(ir, "r06_laikago_cv.json", False, [100 + i for i in range(6)])
    assert all(f >= 0.99 for _, p, f in held if p == "laikago_pace") and all(f <= 0.01 for _, p, f in held if p == "laikago_spin")
    assert sum(r["transfers"] for r in cv["splits"]) == 2 and sum(f >= 0.5 for _, _, f in held) == 7
    # P9's matrix (weaker evidence: its design knew P4's outcomes): with the wheelbase frozen at the clip-pinned value spin is predicted by two
    # of the three tables that never saw it
    cv9, held9 = _check_cv(ir, "r06_laikago_cv_p9.json", True, [700 + i for i in range(6)])
    assert all(f >= 0.99 for _, p, f in held9 if p == "laikago_pace") and sorted(round(f, 2) for _, p, f in held9 if p == "laikago_spin") == [0.27, 0.80, 0.84]
    assert sum(r["transfers"] for r in cv9["splits"]) == 4 and sum(f >= 0.5 for _, _, f in held9) == 9


def test_shipped_minicheetah_table_is_round6s_record():
    """robots.mini_cheetah() IS the end of the smallest-table path of tools/identify_r6.py's mini-cheetah run (P8 + P7;
    profiles/r06_mc_identify.json, r06_mc_minimal.json).  IN SAMPLE: one policy exists, the record has no hold-out."""
    import identify_r6 as ir
    from openroborl_amd import robots
    spec = ir.SPECS["mini_cheetah"]
    rec = json.load(open(os.path.join(ROOT, "profiles", "r06_mc_identify.json")))
    mn = json.load(open(os.path.join(ROOT, "profiles", "r06_mc_minimal.json")))
    assert rec["fit"] == ["minicheetah_trot"] and rec["holdout"] == [] and rec["verdict"] == "accepted" and rec["accept_F"] == 0.9 and "holdout" not in rec["chosen"]
    assert rec["chosen"]["recheck"]["F"] >= 0.9 and rec["chosen"]["robustness"]["mean_min_F"] >= 0.8
    for k, (v0, lo, hi) in rec["params"].items():
        assert (v0, lo, hi) == tuple(spec["params"][k]) and lo - 1e-12 <= rec["chosen"]["theta"][k] <= hi + 1e-12, k
    for step in mn["minimal"]["path"]:
        assert step["min_F"] >= 0.9 and step["min_J"] >= mn["chosen"]["min_J"] - 0.01
    for k, e in mn["minimal"]["effect_of_each_moved_entry"].items():
        assert e["min_F_if_put_back"] < 0.9 or e["min_J_if_put_back"] < mn["chosen"]["min_J"] - 0.01, k
    assert set(robots.MINI_CHEETAH_R06_MOVED) == set(mn["minimal"]["still_moved"])
    m_tool, m_ship = ir.build_model("mini_cheetah", mn["minimal"]["theta"]), robots.mini_cheetah()
    for key, val in m_ship.items():
        if not isinstance(val, str):
            b = np.asarray(m_tool[key], dtype=float)
            np.testing.assert_allclose(np.asarray(val, dtype=float), b, rtol=3e-5, atol=3e-5 * max(1.0, float(np.abs(b).max())), err_msg=key)
    r2 = robots.mini_cheetah(**robots.MINI_CHEETAH_R02)
    for key in ("kp", "kd", "init_motor_angles", "motor_dir", "motor_offset", "joint_of_motor", "init_pos", "init_quat", "joint_axis", "toe_radius",
                "fall_radius", "fall_body", "base_mass", "shank_radius", "shank_pos", "joint_lo", "joint_hi"):
        np.testing.assert_array_equal(np.asarray(m_ship[key]), np.asarray(r2[key]), err_msg=key)
    assert np.allclose(m_ship["joint_pos"][0::3, 2], 0.011)                      # hip plane: the clip's lowest toe on the ground


def test_cross_validation_replicates_with_other_search_seeds():
    """profiles/r06_laikago_cv_replication.json: the six splits once more with search seeds 500 + i (the recorded run: 100 + i), after the
    tables had shipped.  The pattern of the 6 x 4 matrix is the same: which splits transfer, pace always, spin never."""
    cv = json.load(open(os.path.join(ROOT, "profiles", "r06_laikago_cv.json")))
    rep = json.load(open(os.path.join(ROOT, "profiles", "r06_laikago_cv_replication.json")))
    assert [r["seed"] for r in cv["splits"]] == [100 + i for i in range(6)] and [r["seed"] for r in rep["splits"]] == [500 + i for i in range(6)]
    assert [r["transfers"] for r in rep["splits"]] == [r["transfers"] for r in cv["splits"]] == [False, False, False, True, True, False]
    for a, b in zip(cv["table"], rep["table"]):
        assert a["fit"] == b["fit"] and a["verdict"] == b["verdict"] == "accepted"
        for p in a["cells"]:
            ca, cb = a["cells"][p], b["cells"][p]
            assert ca["role"] == cb["role"]
            if ca["role"] == "held out":
                assert (ca["F"] >= 0.5) == (cb["F"] >= 0.5) or p == "laikago_trot0", (a["split"], p, ca["F"], cb["F"])     # split 0's trot0: 0.04 / 0.43, both below
                if p in ("laikago_pace", "laikago_spin"):
                    assert abs(ca["F"] - cb["F"]) <= 0.01
