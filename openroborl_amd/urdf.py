"""URDF -> robot model table ("optional URDF -> model-table converter").

The reference loads its robots from `pybullet_data` URDFs that are neither under /root/reference nor in this image
(robots/laikago.py:23, robots/mini_cheetah.py:23), so the inertial tables of `robots.py` are hand-authored.  This module turns
such a URDF into the same table WITHOUT PyBullet (plain XML + frame algebra), for whoever has the files:

    from openroborl_amd import robots, urdf
    m = urdf.model_from_urdf(open("laikago_toes_limits.urdf").read(), robots.laikago(), urdf.LAIKAGO_JOINTS)

Frames (robots.py docstring): the table keeps every link-frame quantity in axes PARALLEL TO THE KINEMATIC BODY FRAME (x forward,
y left, z up) at zero motor angles, i.e. at q_urdf = motor_offset; the kinematic frame is the base link frame turned by the robot's
INIT_QUAT (the orientation in which the reference spawns the base: robots/laikago.py:49); the base frame's origin is the base COM.
`model_to_urdf` writes a table back out as URDF text with arbitrarily rotated link frames (used by the round-trip test, and a way
to look at a table in any URDF viewer).
"""
import xml.etree.ElementTree as ET

import numpy as np

# joint names of the two reference robots in table order (URDF leg order; hip, upper, lower + the fixed toe joint)
LAIKAGO_JOINTS = {
    "legs": [["%s_hip_motor_2_chassis_joint" % p, "%s_upper_leg_2_hip_motor_joint" % p, "%s_lower_leg_2_upper_leg_joint" % p]
             for p in ("FR", "FL", "RR", "RL")],                                   # robots/laikago.py:31-44
    "toes": ["jtoe%s" % p for p in ("FR", "FL", "RR", "RL")],                       # robots/minitaur.py:842-844 name pattern
}
MINI_CHEETAH_JOINTS = {
    "legs": [["torso_to_abduct_%s_j" % p, "abduct_%s_to_thigh_%s_j" % (p, p), "thigh_%s_to_knee_%s_j" % (p, p)]
             for p in ("fr", "fl", "hr", "hl")],                                   # robots/mini_cheetah.py:31-44
    "toes": ["toe_%s_joint" % p for p in ("fr", "fl", "hr", "hl")],
}


def rpy_to_mat(rpy):
    """URDF fixed-axis roll-pitch-yaw: R = Rz(yaw) Ry(pitch) Rx(roll)."""
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def mat_to_rpy(R):
    p = -np.arcsin(np.clip(R[2, 0], -1.0, 1.0))
    if abs(np.cos(p)) > 1e-9:
        return np.array([np.arctan2(R[2, 1], R[2, 2]), p, np.arctan2(R[1, 0], R[0, 0])])
    return np.array([0.0, p, np.arctan2(-R[0, 1], R[1, 1])])


def axis_angle_mat(axis, ang):
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1.0 - np.cos(ang)) * (K @ K)


def quat_to_mat(q):
    x, y, z, w = np.asarray(q, dtype=np.float64) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _sym6(M):
    return np.array([M[0, 0], M[1, 1], M[2, 2], M[0, 1], M[0, 2], M[1, 2]])


def _sym3(s):
    return np.array([[s[0], s[3], s[4]], [s[3], s[1], s[5]], [s[4], s[5], s[2]]])


def _pa(m, d):
    x, y, z = d
    return m * np.array([y * y + z * z, x * x + z * z, x * x + y * y, -x * y, -x * z, -y * z])


def _vec(s, n=3):
    v = [float(x) for x in (s or "").split()]
    return np.array(v if len(v) == n else [0.0] * n)


def parse_urdf(text):
    """-> (links, joints): links[name] = dict(mass, com, R_inertial, inertia 3x3 in the inertial frame);
    joints[name] = dict(type, parent, child, xyz, R, axis, lower, upper)."""
    root = ET.fromstring(text)
    links, joints = {}, {}
    for ln in root.findall("link"):
        ine = ln.find("inertial")
        d = dict(mass=0.0, com=np.zeros(3), R_inertial=np.eye(3), inertia=np.zeros((3, 3)))
        if ine is not None:
            o = ine.find("origin")
            if o is not None:
                d["com"] = _vec(o.get("xyz"))
                d["R_inertial"] = rpy_to_mat(_vec(o.get("rpy")))
            if ine.find("mass") is not None:
                d["mass"] = float(ine.find("mass").get("value"))
            i = ine.find("inertia")
            if i is not None:
                g = lambda k: float(i.get(k, "0"))
                d["inertia"] = np.array([[g("ixx"), g("ixy"), g("ixz")], [g("ixy"), g("iyy"), g("iyz")], [g("ixz"), g("iyz"), g("izz")]])
        # Bullet's per-link contact properties (URDF <contact> block: lateral_friction, stiffness + damping) and a spherical collision
        # shape: what the toe links of the reference's robots carry (the table's foot_friction, contact_stiffness / contact_damping,
        # toe_radius)
        ct = ln.find("contact")
        if ct is not None:
            for tag, key in (("lateral_friction", "lateral_friction"), ("stiffness", "contact_stiffness"), ("damping", "contact_damping")):
                e = ct.find(tag)
                if e is not None:
                    d[key] = float(e.get("value"))
        sp = ln.find("collision/geometry/sphere")
        if sp is not None:
            d["sphere_radius"] = float(sp.get("radius"))
        links[ln.get("name")] = d
    for jn in root.findall("joint"):
        o, ax, lim = jn.find("origin"), jn.find("axis"), jn.find("limit")
        joints[jn.get("name")] = dict(
            type=jn.get("type"), parent=jn.find("parent").get("link"), child=jn.find("child").get("link"),
            xyz=_vec(o.get("xyz")) if o is not None else np.zeros(3), R=rpy_to_mat(_vec(o.get("rpy"))) if o is not None else np.eye(3),
            axis=_vec(ax.get("xyz")) if ax is not None else np.array([1.0, 0.0, 0.0]),
            lower=float(lim.get("lower", "-1e9")) if lim is not None and jn.get("type") == "revolute" else -1e9,
            upper=float(lim.get("upper", "1e9")) if lim is not None and jn.get("type") == "revolute" else 1e9)
    return links, joints


def model_from_urdf(text, template, names):
    """Inertial / geometric fields of `template` (a robots.py model dict: it supplies the control constants, INIT_QUAT, the motor
    directions / offsets and the hand-authored collision proxies) replaced by what the URDF says.  names: LAIKAGO_JOINTS-style dict."""
    links, joints = parse_urdf(text)
    m = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in template.items()}
    children = {j["child"] for j in joints.values()}
    base = next(n for n in links if n not in children)
    K = quat_to_mat(template["init_quat"])            # kinematic frame <- base link frame
    motor_of_joint = np.argsort(template["joint_of_motor"])
    bl = links[base]
    m["base_mass"] = bl["mass"]
    Rb = K @ bl["R_inertial"]
    m["base_inertia"] = _sym6(Rb @ bl["inertia"] @ Rb.T)
    for leg in range(4):
        W = K                                       # kinematic <- current parent link frame, at q_urdf = motor_offset
        origin_shift = bl["com"]                     # the table's base frame sits at the base COM
        for k in range(3):
            j = 3 * leg + k
            jd = joints[names["legs"][leg][k]]
            mot = motor_of_joint[j]
            off, direc = float(template["motor_offset"][mot]), float(template["motor_dir"][mot])
            m["joint_pos"][j] = W @ (jd["xyz"] - origin_shift)
            Wc = W @ jd["R"] @ axis_angle_mat(jd["axis"], off)      # child link frame at the offset pose
            m["joint_axis"][j] = Wc @ (jd["axis"] / np.linalg.norm(jd["axis"]))
            lo, hi = jd["lower"] - off, jd["upper"] - off
            m["joint_lo"][j], m["joint_hi"][j] = (lo, hi) if direc > 0 else (-hi, -lo)
            ld = links[jd["child"]]
            Rl = Wc @ ld["R_inertial"]
            I = Rl @ ld["inertia"] @ Rl.T
            c = Wc @ ld["com"]
            if k < 2:
                m["link_mass"][j], m["link_com"][j], m["link_inertia"][j] = ld["mass"], c, _sym6(I)
                m["link_inertia_pa"][j] = 0.0
            else:      # lower leg + fixed toe merged for the dynamics (robots.py _build)
                td = joints.get(names["toes"][leg])
                if td is None:    # other naming: the fixed joint behind the lower leg (robots/laikago.py:45 `jtoe\d*`, mini_cheetah.py:45 `toe_`)
                    td = next(jj for jj in joints.values() if jj["parent"] == jd["child"] and jj["type"] == "fixed")
                tl = links[td["child"]]
                Wt = Wc @ td["R"]
                c_t = Wc @ td["xyz"] + Wt @ tl["com"]
                Rt = Wt @ tl["R_inertial"]
                mm = ld["mass"] + tl["mass"]
                cc = (ld["mass"] * c + tl["mass"] * c_t) / mm
                m["link_mass"][j], m["link_com"][j] = mm, cc
                m["link_inertia"][j] = _sym6(I) + _sym6(Rt @ tl["inertia"] @ Rt.T)
                m["link_inertia_pa"][j] = _pa(ld["mass"], c - cc) + _pa(tl["mass"], c_t - cc)
                m["toe_pos"][leg] = Wc @ td["xyz"]
                m["lower_com"][leg] = c
                # contact properties of the toe link, when the URDF states them (the four toes of the reference's robots are alike)
                if "lateral_friction" in tl:
                    m["foot_friction"] = tl["lateral_friction"]
                if "contact_stiffness" in tl:
                    m["contact_stiffness"], m["contact_damping"] = tl["contact_stiffness"], tl.get("contact_damping", 0.0)
                if "sphere_radius" in tl:
                    m["toe_radius"] = tl["sphere_radius"]
            W, origin_shift = Wc, np.zeros(3)
    return m


def model_to_urdf(model, names, rng=None):
    """A table as URDF text.  rng (numpy RandomState): turn every link frame by a random rotation (the same robot, differently
    framed -- what the round-trip test feeds to model_from_urdf); None: link frames parallel to the kinematic frame."""
    K = quat_to_mat(model["init_quat"])
    motor_of_joint = np.argsort(model["joint_of_motor"])

    def frame():
        if rng is None:
            return np.eye(3)
        return axis_angle_mat(rng.randn(3), rng.uniform(-np.pi, np.pi))

    def inertial(mass, com, I6, W):
        """link-frame <inertial> of a body whose table COM / inertia (kinematic-parallel axes) are com / I6; W = kinematic <- link."""
        Ri = frame()
        Il = Ri.T @ W.T @ _sym3(I6) @ W @ Ri
        c, rpy = W.T @ com, mat_to_rpy(Ri)
        return ('<inertial><origin xyz="%.17g %.17g %.17g" rpy="%.17g %.17g %.17g"/><mass value="%.17g"/>'
                '<inertia ixx="%.17g" ixy="%.17g" ixz="%.17g" iyy="%.17g" iyz="%.17g" izz="%.17g"/></inertial>'
                % (c[0], c[1], c[2], rpy[0], rpy[1], rpy[2], mass, Il[0, 0], Il[0, 1], Il[0, 2], Il[1, 1], Il[1, 2], Il[2, 2]))

    out = ['<?xml version="1.0"?>', '<robot name="%s">' % model["name"]]
    Wb = K                                            # kinematic <- base link frame is fixed by INIT_QUAT
    base_com_link = np.zeros(3) if rng is None else rng.uniform(-0.02, 0.02, 3)   # the base link frame need not sit at the COM
    out.append('<link name="base">%s</link>' % inertial(model["base_mass"], Wb @ base_com_link, model["base_inertia"], Wb))
    for leg in range(4):
        W, parent, shift = Wb, "base", base_com_link
        for k in range(3):
            j = 3 * leg + k
            mot = motor_of_joint[j]
            off, direc = float(model["motor_offset"][mot]), float(model["motor_dir"][mot])
            Wc = K @ frame() if rng is not None else np.eye(3)          # child link frame at the offset pose, in kinematic axes
            axis_c = Wc.T @ model["joint_axis"][j]
            Ro = W.T @ Wc @ axis_angle_mat(axis_c, off).T               # joint origin rotation (pose at q_urdf = 0)
            xyz, rpy = W.T @ model["joint_pos"][j] + shift, mat_to_rpy(Ro)
            lo, hi = model["joint_lo"][j], model["joint_hi"][j]
            ulo, uhi = ((lo, hi) if direc > 0 else (-hi, -lo))
            child = "leg%d_link%d" % (leg, k)
            cont = abs(lo) > 1e8
            out.append('<joint name="%s" type="%s"><parent link="%s"/><child link="%s"/><origin xyz="%.17g %.17g %.17g" '
                       'rpy="%.17g %.17g %.17g"/><axis xyz="%.17g %.17g %.17g"/>%s</joint>'
                       % (names["legs"][leg][k], "continuous" if cont else "revolute", parent, child, xyz[0], xyz[1], xyz[2],
                          rpy[0], rpy[1], rpy[2], axis_c[0], axis_c[1], axis_c[2],
                          "" if cont else '<limit lower="%.17g" upper="%.17g" effort="100" velocity="100"/>' % (ulo + off, uhi + off)))
            if k < 2:
                out.append('<link name="%s">%s</link>' % (child, inertial(model["link_mass"][j], model["link_com"][j], model["link_inertia"][j], Wc)))
            else:
                # un-merge lower leg and toe: the toe is a point-like sphere at toe_pos (its own inertia is not recoverable from the
                # merged table, so the writer puts the merged COM-inertia on the lower leg and a massless toe link behind the fixed joint)
                mm, cc = model["link_mass"][j], model["link_com"][j]
                I6 = model["link_inertia"][j] + model["link_inertia_pa"][j]
                out.append('<link name="%s">%s</link>' % (child, inertial(mm, cc, I6, Wc)))
                Wt = K @ frame() if rng is not None else np.eye(3)
                rpy_t, xyz_t = mat_to_rpy(Wc.T @ Wt), Wc.T @ model["toe_pos"][leg]
                out.append('<joint name="%s" type="fixed"><parent link="%s"/><child link="toe%d"/><origin xyz="%.17g %.17g %.17g" '
                           'rpy="%.17g %.17g %.17g"/></joint><link name="toe%d"><contact><lateral_friction value="%.17g"/>%s</contact>'
                           '<collision><geometry><sphere radius="%.17g"/></geometry></collision></link>'
                           % (names["toes"][leg], child, leg, xyz_t[0], xyz_t[1], xyz_t[2], rpy_t[0], rpy_t[1], rpy_t[2], leg,
                              model["foot_friction"],
                              '<stiffness value="%.17g"/><damping value="%.17g"/>' % (model["contact_stiffness"], model["contact_damping"])
                              if model.get("contact_stiffness", 0.0) > 0 else "", model["toe_radius"]))
            W, parent, shift = Wc, child, np.zeros(3)
    out.append("</robot>")
    return "\n".join(out)


if __name__ == "__main__":   # python -m openroborl_amd.urdf <file.urdf> laikago|mini_cheetah : the URDF's table next to the hand-authored one
    import sys
    from . import robots
    path, name = sys.argv[1], sys.argv[2]
    hand = robots.ROBOTS[name]()
    got = model_from_urdf(open(path).read(), hand, LAIKAGO_JOINTS if name == "laikago" else MINI_CHEETAH_JOINTS)
    np.set_printoptions(precision=6, suppress=True, linewidth=160)
    for k in ("base_mass", "base_inertia", "link_mass", "link_com", "link_inertia", "link_inertia_pa", "joint_pos", "joint_axis", "joint_lo",
              "joint_hi", "toe_pos", "lower_com", "toe_radius", "foot_friction", "contact_stiffness", "contact_damping"):
        print("== %s\nURDF:\n%s\nrobots.py:\n%s" % (k, np.asarray(got[k]), np.asarray(hand[k])))
