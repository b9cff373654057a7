"""Why `laikago_pace` runs ~0.3 m/s behind its clip (tools/lag_diagnosis.py): the clip's own stance toes
SKATE forward at ~0.35 m/s when it is replayed through the reference's leg kinematics (trans2minicheetah.m:3-9 link lengths; a dog's mocap
retargeted to rigid hips), the simulated stance toes STICK (millimetres of slip per stance, friction rows rarely at their bound), so a robot that
follows the clip's joint angles advances at v_clip - skate.  CPU only: float64 oracle + sub-step trace."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def diag():
    import lag_diagnosis as L
    from openroborl_amd import robots
    base = robots.laikago()
    kin = L.clip_kinematics("laikago_pace", base, 6.0)
    sim = L.simulate("laikago_pace", "laikago_pace", base, 4, 160, 60)
    return L, kin, sim


def test_clip_stance_toes_skate_whatever_the_table_or_threshold(diag):
    L, kin, _ = diag
    from openroborl_amd import robots
    assert abs(kin["v_clip"] - 1.086) < 0.005
    assert 0.30 <= kin["skate_mean"] <= 0.42                          # 0.344 m/s of 1.086
    for table in (robots.laikago(), robots.laikago(**robots.LAIKAGO_R04)):   # not a property of the identified entries
        for mm in (3.0, 10.0, 15.0):
            k = L.clip_kinematics("laikago_pace", table, mm)
            assert 0.30 <= k["skate_mean"] <= 0.45, (mm, k["skate_mean"])


def test_simulated_stance_toes_stick_and_the_lag_is_the_clip_skate(diag):
    _, kin, sim = diag
    assert sim["finished_window"] == 1.0
    lag = sim["v_ref"] - sim["v_sim"]
    assert 0.22 <= lag <= 0.40, lag                                   # 0.31 m/s (HIP path, 1024 robots: 0.29-0.32)
    assert abs(lag - kin["skate_mean"]) < 0.10                        # the lag IS the kinematic skate: 0.310 vs 0.344
    assert abs(sim["v_sim"] - kin["v_noslip_mean"]) < 0.10            # 0.774 vs 0.757
    for leg in sim["legs"]:
        assert abs(leg["slip_fwd_mm_per_stance"]) < 15.0, leg         # -4 .. +1 mm per stance; the clip's toes skate 16 .. 157 mm per stance
        assert abs(leg["toe_v_fwd_in_stance"]) < 0.06, leg            # toes at rest on the ground
        assert leg["friction_at_bound"] < 0.35, leg                   # 0.06 .. 0.15: the cone is not what limits the push
    assert 0.97 <= sim["mean_normal_force_over_weight"] <= 1.03       # the trace's impulses carry the robot


def test_more_friction_does_not_buy_the_lag_back(diag):
    L, kin, sim = diag
    from openroborl_amd import robots
    hi = L.simulate("laikago_pace", "laikago_pace", robots.laikago(foot_friction=1.0), 4, 160, 60)
    assert (hi["v_ref"] - hi["v_sim"]) > 0.2                          # mu 1.0: lag 0.28 (mu 0.5: 0.31)
    assert abs(hi["v_sim"] - sim["v_sim"]) < 0.08
