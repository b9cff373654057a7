"""Row C (pybullet.stepSimulation) against REAL PyBullet output -- when somebody has produced it.

tools/pybullet_ref.py --dump-substeps writes tests/golden/pybullet_substep_<robot>.npz on a machine that has pybullet +
pybullet_data (this image and the GPU boxes do not: SURVEY.md section 8c).  Until that fixture exists these tests skip and
the physics stays "parity unpinned" (DESIGN.md section 7).  What runs everywhere: the harness degrades cleanly without
pybullet and the seeded inputs it would feed to PyBullet are the ones the HIP-vs-oracle parity test uses."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import oracle_lib as ol
from tests.parity_inputs import substep_parity_inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_harness_reports_unavailable_without_pybullet():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pybullet_ref
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pybullet_ref.py"), "--time"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["pybullet"] == ("available" if pybullet_ref.available() else "unavailable")


def test_parity_inputs_are_deterministic_and_float32_exact():
    for robot in ("laikago", "mini_cheetah"):
        _, _, _, st, tau = substep_parity_inputs(robot)
        _, _, _, st2, tau2 = substep_parity_inputs(robot)
        np.testing.assert_array_equal(st, st2)
        np.testing.assert_array_equal(tau, tau2)
        assert st.shape[0] == 64 and tau.shape == (64, 12)
        np.testing.assert_array_equal(tau.astype(np.float32).astype(np.float64), tau)
        z = st[:, 2]
        assert z.min() < 0.45 and z.max() > 0.3          # airborne as well as penetrating configurations


@pytest.mark.parametrize("robot", ["laikago", "mini_cheetah"])
def test_oracle_substeps_against_real_pybullet(robot):
    path = os.path.join(ol.GOLDEN, "pybullet_substep_%s.npz" % robot)
    if not os.path.exists(path):
        pytest.skip("no PyBullet fixture (run tools/pybullet_ref.py --dump-substeps where pybullet is installed): physics parity unpinned")
    g = np.load(path)
    cfg, models, clips, st, tau = substep_parity_inputs(robot)
    np.testing.assert_allclose(g["state_in"], st[:, 0:37], atol=0)          # the fixture was made from these very inputs
    orc = ol.OracleEnv(cfg, models, clips, st.shape[0], robot_type=[k for k, m in enumerate(models) if m is not None][0], clip_id=0)
    orc.state[:] = st
    for nsub, key, ptol, vtol in ((1, "state_1", 2e-4, 5e-2), (8, "state_8", 2e-3, 2e-1)):
        for i in range(st.shape[0]):
            for _ in range(nsub if nsub == 1 else 7):
                orc.L.orc_physics_substep(orc.h, ol.P(orc.state[i]), ol.P(np.ascontiguousarray(tau[i])))
        got, want = orc.state[:, 0:37], g[key]
        # tolerances to be tightened once the URDF inertial tables replace the hand-authored ones (tools/pybullet_ref.py --dump-urdf)
        np.testing.assert_allclose(got[:, 0:7], want[:, 0:7], atol=ptol, err_msg="base pose after %d sub-steps" % nsub)
        np.testing.assert_allclose(got[:, 13:25], want[:, 13:25], atol=10 * ptol, err_msg="joint angles after %d sub-steps" % nsub)
        assert np.median(np.abs(got[:, 7:13] - want[:, 7:13])) < vtol
    orc.close()
