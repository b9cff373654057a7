"""Restatement of the handful of `pybullet_utils.transformations` helpers the reference calls.

pybullet (and its pure-Python `pybullet_utils.transformations`, an [x, y, z, w]-ordered variant
of C. Gohlke's public `transformations.py`) is NOT installed in this image and is not vendored
under /root/reference, so the reference's `pose3d.py` / `motion_data.py` cannot be imported
without it.  This file restates the published semantics of exactly the functions those two
modules call (call sites: pose3d.py:228-230,355; motion_data.py:442,499,585-586,611,630,661).

CAVEAT (recorded in DESIGN.md): golden vectors produced through this shim pin the reference's
*own* code (frame lookup, blending, finite differences, heading, axis-angle ...) but the
quaternion *conventions* of the shim itself are pinned only by self-consistency with pose3d.py
(which hard-codes xyzw: pose3d.py:31,132-136,168,184,299) and by the motion data (w is the
last component of every root quaternion).

Test infrastructure only.  Never imported by the product package.
"""
import math

import numpy as np

_EPS = np.finfo(float).eps * 4.0


def quaternion_multiply(quaternion1, quaternion0):
    """Hamilton product q1 * q0, both [x, y, z, w]."""
    x0, y0, z0, w0 = quaternion0
    x1, y1, z1, w1 = quaternion1
    return np.array((
        x1 * w0 + y1 * z0 - z1 * y0 + w1 * x0,
        -x1 * z0 + y1 * w0 + z1 * x0 + w1 * y0,
        x1 * y0 - y1 * x0 + z1 * w0 + w1 * z0,
        -x1 * x0 - y1 * y0 - z1 * z0 + w1 * w0), dtype=np.float64)


def quaternion_conjugate(quaternion):
    q = np.array(quaternion, dtype=np.float64, copy=True)
    q[0:3] = -q[0:3]
    return q


def quaternion_inverse(quaternion):
    q = quaternion_conjugate(quaternion)
    return q / np.dot(q, q)


def quaternion_about_axis(angle, axis):
    q = np.array([axis[0], axis[1], axis[2], 0.0], dtype=np.float64)
    qlen = math.sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2])
    if qlen > _EPS:
        q *= math.sin(angle / 2.0) / qlen
    q[3] = math.cos(angle / 2.0)
    return q


def quaternion_from_euler(ai, aj, ak, axes='sxyz'):
    if axes != 'sxyz':
        raise NotImplementedError(axes)
    ai, aj, ak = ai / 2.0, aj / 2.0, ak / 2.0
    ci, si = math.cos(ai), math.sin(ai)
    cj, sj = math.cos(aj), math.sin(aj)
    ck, sk = math.cos(ak), math.sin(ak)
    cc, cs, sc, ss = ci * ck, ci * sk, si * ck, si * sk
    return np.array((cj * sc - sj * cs, cj * ss + sj * cc, cj * cs - sj * sc,
                     cj * cc + sj * ss), dtype=np.float64)


def _unit(v):
    v = np.array(v, dtype=np.float64, copy=True)
    return v / math.sqrt(np.dot(v, v))


def quaternion_slerp(quat0, quat1, fraction, spin=0, shortestpath=True):
    q0 = _unit(quat0[:4])
    q1 = _unit(quat1[:4])
    if fraction == 0.0:
        return q0
    elif fraction == 1.0:
        return q1
    d = np.dot(q0, q1)
    if abs(abs(d) - 1.0) < _EPS:
        return q0
    if shortestpath and d < 0.0:
        d = -d
        q1 *= -1.0
    angle = math.acos(d) + spin * math.pi
    if abs(angle) < _EPS:
        return q0
    isin = 1.0 / math.sin(angle)
    q0 *= math.sin((1.0 - fraction) * angle) * isin
    q1 *= math.sin(fraction * angle) * isin
    q0 += q1
    return q0
