"""The GPU side of the measurement tools stays runnable: tools/laikago_identify.py's HipProbe (four candidates in the four robot-type
slots of one launch) on the two tables whose results are quoted in DESIGN.md section 7.2."""
import json
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_identification_probe_separates_the_round_4_table_from_the_identified_one():
    """One launch group = [round-4 table, the recorded chosen candidate, round-4 table, chosen candidate] under the fit policy laikago_trot,
    200 steps, 64 robots each: the slots do not leak into each other (the same table in two slots gives the same level - not the same bits:
    the robots of two slots have different global indices, i.e. other start phases) and the chosen candidate keeps walking where round 4's
    table has long fallen."""
    import laikago_identify as li
    rec = json.load(open(os.path.join(ROOT, "profiles", "r05_laikago_identify.json")))
    r4, ch = li.shipped_theta(), rec["chosen"]["theta"]
    probe = li.HipProbe(64)
    out = probe.run_group("laikago_trot", "laikago_trot", [r4, ch, dict(r4), dict(ch)], li.config_overrides(r4), steps=200)
    assert abs(out[0]["len"] - out[2]["len"]) < 45 and abs(out[1]["len"] - out[3]["len"]) < 20        # same table, another slot: same level
    assert out[2]["F"] <= 0.5 and out[3]["F"] >= 0.8
    assert out[0]["len"] < 160 and out[0]["F"] <= 0.5                  # round 4's table: mean survival ~130 steps
    assert out[1]["len"] > 170 and out[1]["F"] >= 0.8                  # the identified table: measured 182 of 200 steps, 0.9 still up (the warm-up starts are the weaker ones)
