"""Independent numpy rigid-body reference used to check the oracle's ABA (NOT a copy of it):
builds the joint-space mass matrix from link Jacobians, M = sum_b J_b^T diag(I_b_world, m_b) J_b,
in the generalised velocity u = [omega_world, v_com_base_world, kinematic joint rates]."""
import numpy as np


def quat_to_mat(q):
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def qmul(a, b):
    x1, y1, z1, w1 = a
    x0, y0, z0, w0 = b
    return np.array([x1 * w0 + y1 * z0 - z1 * y0 + w1 * x0, -x1 * z0 + y1 * w0 + z1 * x0 + w1 * y0,
                     x1 * y0 - y1 * x0 + z1 * w0 + w1 * z0, -x1 * x0 - y1 * y0 - z1 * z0 + w1 * w0])


def qconj(q):
    return np.array([-q[0], -q[1], -q[2], q[3]])


def rodrigues(axis, ang):
    a = np.asarray(axis, dtype=float)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)


def sym6(v):
    xx, yy, zz, xy, xz, yz = v
    return np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]])


def joint_maps(model):
    dirj = np.zeros(12); offj = np.zeros(12); motor_of_joint = np.zeros(12, dtype=int)
    for m in range(12):
        j = int(model["joint_of_motor"][m])
        dirj[j] = model["motor_dir"][m]; offj[j] = model["motor_offset"][m]; motor_of_joint[j] = m
    return dirj, offj, motor_of_joint


def kinematics(model, pos, quat, q_urdf, mass_ratio=(1, 1), inertia_ratio=(1, 1)):
    """Returns per body: R (3x3 world), origin, com (world), mass, inertia (world), plus joint axes (world)."""
    dirj, offj, _ = joint_maps(model)
    a = dirj * (np.asarray(q_urdf) - offj)
    Rb = quat_to_mat(qmul(np.asarray(quat, float), qconj(np.asarray(model["init_quat"], float))))
    bodies = [dict(R=Rb, o=np.asarray(pos, float), m=model["base_mass"] * mass_ratio[0],
                   I=sym6(model["base_inertia"]) * inertia_ratio[0], c=np.zeros(3), parent=-1)]
    axes_w = []
    for j in range(12):
        par = 0 if j % 3 == 0 else j  # body index of parent (body = j+1, parent body = j)
        P = bodies[par]
        o = P["o"] + P["R"] @ model["joint_pos"][j]
        R = P["R"] @ rodrigues(model["joint_axis"][j], a[j])
        g = int(model["link_group"][j])
        I = sym6(model["link_inertia"][j]) * inertia_ratio[g] + sym6(model["link_inertia_pa"][j]) * mass_ratio[g]
        bodies.append(dict(R=R, o=o, m=model["link_mass"][j] * mass_ratio[g], I=I, c=model["link_com"][j], parent=par))
        axes_w.append(R @ model["joint_axis"][j])
    for b in bodies:
        b["cw"] = b["o"] + b["R"] @ b["c"]
        b["Iw"] = b["R"] @ b["I"] @ b["R"].T
    return bodies, np.array(axes_w)


def body_jacobians(bodies, axes_w):
    """J_b (6x18): [omega_b; v_com_b] = J_b u."""
    Js = []
    p0 = bodies[0]["o"]
    for bi, b in enumerate(bodies):
        J = np.zeros((6, 18))
        J[0:3, 0:3] = np.eye(3)
        r = b["cw"] - p0
        J[3:6, 0:3] = -np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
        J[3:6, 3:6] = np.eye(3)
        k = bi
        while k >= 1:
            j = k - 1
            ax = axes_w[j]
            J[0:3, 6 + j] = ax
            J[3:6, 6 + j] = np.cross(ax, b["cw"] - bodies[k]["o"])
            k = bodies[k]["parent"]
        Js.append(J)
    return Js


def mass_matrix(bodies, Js):
    M = np.zeros((18, 18))
    for b, J in zip(bodies, Js):
        M += J[0:3].T @ b["Iw"] @ J[0:3] + b["m"] * (J[3:6].T @ J[3:6])
    return M


def gravity_force(bodies, Js, gz):
    Q = np.zeros(18)
    for b, J in zip(bodies, Js):
        Q += J[3:6].T @ np.array([0, 0, b["m"] * gz])
    return Q


def energy_momentum(model, pos, quat, q_urdf, angvel, linvel, qd_urdf, gz):
    dirj, offj, _ = joint_maps(model)
    bodies, axes = kinematics(model, pos, quat, q_urdf)
    Js = body_jacobians(bodies, axes)
    u = np.concatenate([angvel, linvel, dirj * np.asarray(qd_urdf)])
    M = mass_matrix(bodies, Js)
    ke = 0.5 * u @ M @ u
    pe = sum(-b["m"] * gz * b["cw"][2] for b in bodies)
    mtot = sum(b["m"] for b in bodies)
    com = sum(b["m"] * b["cw"] for b in bodies) / mtot
    P = np.zeros(3); Lc = np.zeros(3)
    for b, J in zip(bodies, Js):
        w = J[0:3] @ u; v = J[3:6] @ u
        P += b["m"] * v
        Lc += b["Iw"] @ w + b["m"] * np.cross(b["cw"] - com, v)
    return ke, pe, P, Lc, mtot
