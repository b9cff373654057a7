"""Laikago -> mini-cheetah motion-clip retargeting ("motion-clip ... retargeter").

Restates what the reference's offline script does (task/motions/trans2minicheetah.m:1-75 with the leg kinematics of :137-168; the
Python scratch version is trans_data.py:38-69): per frame, the Laikago toe positions (hip frame, leg kinematics with the Laikago link
lengths) are scaled by the ratio of the leg lengths, shifted by the difference of the hip offsets, and solved for the mini-cheetah
joint angles; the base position is scaled by the same ratio and lowered by 45 mm; the base orientation is re-expressed in the
mini-cheetah's body axes.  The reference ships one clip made this way, `minicheetah_trot.txt` = retarget(`laikago_trot.txt`):
tests/test_retarget.py reproduces it to the 5 decimals it was printed with.

    python -m openroborl_amd.retarget laikago_pace > minicheetah_pace.txt
"""
import json
import os

import numpy as np

LAIKAGO_LEG = (0.032875, 0.25223, 0.251)          # coxa, femur, tibia (trans2minicheetah.m:3-5)
MINI_CHEETAH_LEG = (0.062, 0.209, 0.18)           # (:28-30)
LAIKAGO_OFFSET = np.array([0.0, 0.6, -0.66])      # clip angle -> kinematic angle (:8-12)
LAIKAGO_SIGN = np.array([-1, 1, 1, 1, 1, 1, -1, 1, 1, 1, 1, 1], dtype=np.float64)
MINI_SIGN = np.array([1, -1, -1] * 4, dtype=np.float64)                              # kinematic angle -> mini-cheetah clip angle (:32)
SIDE = np.array([-1.0, 1.0, -1.0, 1.0])           # legs FR, FL, RR, RL: right = -1, left = +1 (:13-17)
BASE_DROP = 0.045                                 # (:45)
AXES = np.array([[0.0, 1.0, 0.0], [0.0, 0.0, 1.0], [1.0, 0.0, 0.0]])                 # body-axis permutation between the two URDFs (:34-36)


def leg_fk(angle, coxa, femur, tibia, side):
    """Toe position in the hip frame for kinematic angles (abduction, hip, knee) (:154-168)."""
    s1, s2, s3 = np.sin(angle)
    c1, c2, c3 = np.cos(angle)
    c23, s23 = c2 * c3 - s2 * s3, s2 * c3 + c2 * s3
    return np.array([-(tibia * s23 + femur * s2),
                     coxa * side * c1 + (tibia * c23 + femur * c2) * s1,
                     coxa * side * s1 - (tibia * c23 + femur * c2) * c1])


def leg_ik(p, coxa, femur, tibia, side):
    """Inverse of leg_fk (knee bent backwards) (:137-152)."""
    x, y, z = p
    r2 = y * y + z * z - coxa * coxa
    D = np.clip((r2 + x * x - femur * femur - tibia * tibia) / (2.0 * tibia * femur), -0.99999999999999, 0.99999999999999)
    gamma = np.arctan2(-np.sqrt(1.0 - D * D), D)
    theta = -np.arctan2(z, y) - np.arctan2(np.sqrt(r2), side * coxa)
    theta = theta - 2.0 * np.pi if theta > np.pi else (theta + 2.0 * np.pi if theta < -np.pi else theta)
    alpha = np.arctan2(-x, np.sqrt(r2)) - np.arctan2(tibia * np.sin(gamma), femur + tibia * np.cos(gamma))
    return np.array([-theta, alpha, gamma])


def _wxyz_to_mat(q):
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _mat_to_wxyz(R):
    """Quaternion with a non-negative scalar part (what rotm2quat returns for these near-identity rotations)."""
    w = 0.5 * np.sqrt(max(1.0 + R[0, 0] + R[1, 1] + R[2, 2], 0.0))
    if w > 1e-6:
        return np.array([w, (R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w)])
    i = int(np.argmax(np.diag(R)))
    j, k = (i + 1) % 3, (i + 2) % 3
    s = np.sqrt(max(1.0 + R[i, i] - R[j, j] - R[k, k], 0.0)) * 2.0
    q = np.zeros(4)
    q[0], q[1 + i], q[1 + j], q[1 + k] = (R[k, j] - R[j, k]) / s, 0.25 * s, (R[j, i] + R[i, j]) / s, (R[k, i] + R[i, k]) / s
    return q if q[0] >= 0 else -q


def retarget_frames(frames):
    """[F, 19] Laikago frames (position 3, orientation 4, joints 12) -> [F, 19] mini-cheetah frames."""
    frames = np.asarray(frames, dtype=np.float64)
    c1, f1, t1 = LAIKAGO_LEG
    c2, f2, t2 = MINI_CHEETAH_LEG
    k = (f2 + t2) / (f1 + t1)
    out = frames.copy()
    out[:, 0:3] *= k
    out[:, 2] -= BASE_DROP
    for i, fr in enumerate(frames):
        # the script feeds the stored (x, y, z, w) to a (w, x, y, z) routine and stores the result the same way round (:37-43,47-49);
        # reproduced as is -- for the Laikago clips, whose base quaternion is close to (0.5, 0.5, 0.5, 0.5), the two readings agree
        q = _mat_to_wxyz(AXES @ _wxyz_to_mat(fr[3:7]))
        out[i, 3:6], out[i, 6] = q[1:4], q[0]
        for leg in range(4):
            a = (fr[7 + 3 * leg:10 + 3 * leg] + LAIKAGO_OFFSET) * LAIKAGO_SIGN[3 * leg:3 * leg + 3]
            p = leg_fk(a, c1, f1, t1, SIDE[leg])
            p[1] += SIDE[leg] * (c2 - c1)          # relative to the hip, not to the hip joint (:50-52)
            out[i, 7 + 3 * leg:10 + 3 * leg] = leg_ik(p * k, c2, f2, t2, SIDE[leg]) * MINI_SIGN[3 * leg:3 * leg + 3]
    return out


def retarget_clip(clip):
    """Clip dictionary (the reference's JSON layout, task/motion_data.py:72-112) -> retargeted clip dictionary."""
    out = dict(clip)
    out["Frames"] = [[round(float(x), 5) for x in row] for row in retarget_frames(clip["Frames"])]
    return out


if __name__ == "__main__":
    import sys
    name = sys.argv[1]
    path = name if os.path.exists(name) else os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "motions", name + ".txt")
    print(json.dumps(retarget_clip(json.load(open(path))), indent=1))
