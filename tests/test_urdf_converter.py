"""URDF -> model-table converter (openroborl_amd/urdf.py).  The reference's URDFs live in pybullet_data and
are not available here, so the converter is pinned by a round trip: a table is written out as URDF text with RANDOMLY ROTATED link
frames, inertial frames and a shifted base origin, and must come back unchanged through the XML parser and the frame algebra."""
import numpy as np
import pytest

from openroborl_amd import robots, urdf


@pytest.mark.parametrize("name,names", [("laikago", urdf.LAIKAGO_JOINTS), ("mini_cheetah", urdf.MINI_CHEETAH_JOINTS)])
def test_round_trip_through_rotated_urdf(name, names):
    model = robots.ROBOTS[name]()
    for seed in (None, 0, 1):
        rng = None if seed is None else np.random.RandomState(seed)
        text = urdf.model_to_urdf(model, names, rng)
        blank = {k: (np.zeros_like(v) if k in ("link_mass", "link_com", "link_inertia", "link_inertia_pa", "joint_pos", "joint_axis",
                                                "joint_lo", "joint_hi", "toe_pos", "lower_com", "base_inertia") else v)
                 for k, v in model.items()}
        blank["base_mass"] = 0.0
        got = urdf.model_from_urdf(text, blank, names)
        assert got["base_mass"] == pytest.approx(model["base_mass"], rel=1e-12)
        np.testing.assert_allclose(got["base_inertia"], model["base_inertia"], atol=1e-12)
        for k in ("link_mass", "link_com", "joint_pos", "joint_axis", "toe_pos"):
            np.testing.assert_allclose(got[k], model[k], atol=1e-12, err_msg=k)
        # the toe is merged into the lower leg: the writer cannot un-merge it, so the sum of the two inertia parts is what survives
        np.testing.assert_allclose(got["link_inertia"] + got["link_inertia_pa"], model["link_inertia"] + model["link_inertia_pa"], atol=1e-12)
        fin = np.abs(model["joint_lo"]) < 1e8
        np.testing.assert_allclose(got["joint_lo"][fin], model["joint_lo"][fin], atol=1e-12)
        np.testing.assert_allclose(got["joint_hi"][fin], model["joint_hi"][fin], atol=1e-12)
        assert np.all(got["joint_lo"][~fin] < -1e8) and np.all(got["joint_hi"][~fin] > 1e8)
        # the toe link's <contact> block and collision sphere (Bullet's per-link contact properties)
        assert got["toe_radius"] == pytest.approx(model["toe_radius"]) and got["foot_friction"] == pytest.approx(model["foot_friction"])
        # what the kernel requires of a model (orr_set_model): joints about +-x (hip) / +-y (upper, lower leg) of the kinematic frame
        ax = got["joint_axis"].reshape(4, 3, 3)
        assert np.allclose(np.abs(ax[:, 0]), [1, 0, 0], atol=1e-12) and np.allclose(np.abs(ax[:, 1:]), [0, 1, 0], atol=1e-12)


def test_parser_reads_plain_urdf_elements():
    text = """<robot name="r"><link name="a"><inertial><origin xyz="0.1 0 -0.2" rpy="0 0 1.5707963267948966"/><mass value="2.5"/>
              <inertia ixx="1" ixy="0" ixz="0" iyy="2" iyz="0" izz="3"/></inertial></link><link name="b"/>
              <joint name="j" type="revolute"><parent link="a"/><child link="b"/><origin xyz="0 0.5 0" rpy="0.1 0.2 0.3"/>
              <axis xyz="0 0 2"/><limit lower="-1" upper="2" effort="1" velocity="1"/></joint></robot>"""
    links, joints = urdf.parse_urdf(text)
    assert links["a"]["mass"] == 2.5 and np.allclose(links["a"]["com"], [0.1, 0, -0.2]) and links["b"]["mass"] == 0.0
    R = links["a"]["R_inertial"]
    assert np.allclose(R @ np.diag([1.0, 2.0, 3.0]) @ R.T, np.diag([2.0, 1.0, 3.0]), atol=1e-12)     # a quarter turn about z swaps xx / yy
    j = joints["j"]
    assert j["parent"] == "a" and j["child"] == "b" and j["lower"] == -1 and j["upper"] == 2 and np.allclose(j["xyz"], [0, 0.5, 0])
    assert np.allclose(urdf.mat_to_rpy(j["R"]), [0.1, 0.2, 0.3], atol=1e-12)


def test_toe_contact_block_fills_the_soft_contact_entries():
    """URDF <contact><stiffness/><damping/><lateral_friction/> on the toe links -> contact_stiffness / contact_damping / foot_friction
    of the table (Bullet's BT_CONTACT_FLAG_CONTACT_STIFFNESS_DAMPING)."""
    model = robots.laikago(contact_stiffness=0.0, contact_damping=0.0)       # rigid toes (the shipped table's are soft since round 5)
    soft = dict(model, contact_stiffness=30000.0, contact_damping=1000.0, foot_friction=3.0, toe_radius=0.03)
    text = urdf.model_to_urdf(soft, urdf.LAIKAGO_JOINTS, np.random.RandomState(2))
    assert "<stiffness" in text and text.count("<contact>") == 4
    got = urdf.model_from_urdf(text, robots.laikago(), urdf.LAIKAGO_JOINTS)
    assert (got["contact_stiffness"], got["contact_damping"], got["foot_friction"], got["toe_radius"]) == (30000.0, 1000.0, 3.0, 0.03)
    rigid = urdf.model_from_urdf(urdf.model_to_urdf(model, urdf.LAIKAGO_JOINTS), robots.laikago(contact_stiffness=0.0, contact_damping=0.0),
                                 urdf.LAIKAGO_JOINTS)
    assert rigid["contact_stiffness"] == 0.0 and "<stiffness" not in urdf.model_to_urdf(model, urdf.LAIKAGO_JOINTS)
