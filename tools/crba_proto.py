"""numpy prototype of the kernel's forward dynamics (composite-rigid-body mass matrix + recursive Newton-Euler bias,
leg-wise block elimination), checked against the oracle's articulated-body algorithm (orc_dynamics_probe).

Everything is expressed in world-aligned axes with the origin O at the base COM (an inertial frame that coincides
with it at this instant).  Spatial vectors are (angular; linear), u = [omega_w, v_com_w, joint rates].

  M = [[ Ic_tot , F ],      F_j = Ic_j S_j  (6-vector per joint),  H block-diagonal (one 3x3 per leg)
       [ F^T    , H ]]
  T_L = F_L H_L^-1;   A0 = Ic_tot - sum_L T_L F_L^T;   a0 = -A0^-1 (p_tot + sum_L T_L (tau_L - C_L))
  qdd_L = H_L^-1 (tau_L - C_L - F_L^T a0)
  impulse response of a row (Jb, jl on leg L):  da0 = A0^-1 (Jb - T_L jl);  dqdd_L = H_L^-1 jl - T_L^T da0;
                                                 dqdd_K = -T_K^T da0

usage: python tools/crba_proto.py        (CPU only; prints the max deviations from the oracle)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cross(a, b):
    return np.cross(a, b)


def sp_inertia_mul(I, h, m, w, v):
    """[I, h x; -h x, m] (w; v) = momentum about O."""
    return I @ w + cross(h, v), m * v - cross(h, w)


def forward_dynamics(bodies, axes_w, angvel, linvel, qd, tau, gz, want_minv=False):
    p0 = bodies[0]["o"]
    # per body: spatial inertia about O
    IO, hh, mm = [], [], []
    for b in bodies:
        c = b["cw"] - p0
        IO.append(b["Iw"] + b["m"] * (c @ c * np.eye(3) - np.outer(c, c)))
        hh.append(b["m"] * c)
        mm.append(b["m"])
    # motion subspaces
    S = []
    for j in range(12):
        d = bodies[j + 1]["o"] - p0
        S.append((axes_w[j], cross(d, axes_w[j])))
    # velocities, acceleration bias, link forces
    V = [(np.asarray(angvel, float), np.asarray(linvel, float))]
    A = [(np.zeros(3), np.zeros(3))]
    for j in range(12):
        par = 0 if j % 3 == 0 else j
        sa, sl = S[j][0] * qd[j], S[j][1] * qd[j]
        w, v = V[par][0] + sa, V[par][1] + sl
        V.append((w, v))
        # A_k = A_parent + V_k x (S qd)   (motion cross product)
        A.append((A[par][0] + cross(w, sa), A[par][1] + cross(w, sl) + cross(v, sa)))
    f = []
    for b in range(13):
        w, v = V[b]
        Pa, Pl = sp_inertia_mul(IO[b], hh[b], mm[b], w, v)
        Fa, Fl = sp_inertia_mul(IO[b], hh[b], mm[b], A[b][0], A[b][1])
        # V x* P = (w x Pa + v x Pl ; w x Pl)
        f.append((Fa + cross(w, Pa) + cross(v, Pl), Fl + cross(w, Pl)))
    # composite inertias (suffix sums along each leg), F, H, C
    Ic_tot = [IO[0].copy(), hh[0].copy(), mm[0]]
    p_tot = [f[0][0].copy(), f[0][1].copy()]
    legs = []
    for L in range(4):
        Ic = [None] * 3
        fs = [None] * 3
        accI, acch, accm = np.zeros((3, 3)), np.zeros(3), 0.0
        fa, fl = np.zeros(3), np.zeros(3)
        for k in (2, 1, 0):
            b = 1 + 3 * L + k
            accI = accI + IO[b]; acch = acch + hh[b]; accm += mm[b]
            fa = fa + f[b][0]; fl = fl + f[b][1]
            Ic[k] = (accI.copy(), acch.copy(), accm)
            fs[k] = (fa.copy(), fl.copy())
        F = np.zeros((6, 3)); C = np.zeros(3)
        for k in range(3):
            j = 3 * L + k
            Fa, Fl = sp_inertia_mul(Ic[k][0], Ic[k][1], Ic[k][2], S[j][0], S[j][1])
            F[0:3, k] = Fa; F[3:6, k] = Fl
            C[k] = S[j][0] @ fs[k][0] + S[j][1] @ fs[k][1]
        H = np.zeros((3, 3))
        for i in range(3):
            for k in range(i, 3):
                H[i, k] = H[k, i] = S[3 * L + i][0] @ F[0:3, k] + S[3 * L + i][1] @ F[3:6, k]
        Hinv = np.linalg.inv(H)
        T = F @ Hinv
        legs.append(dict(F=F, H=H, Hinv=Hinv, T=T, C=C))
        Ic_tot[0] += Ic[0][0]; Ic_tot[1] += Ic[0][1]; Ic_tot[2] += Ic[0][2]
        p_tot[0] += fs[0][0]; p_tot[1] += fs[0][1]

    def skew(h):
        return np.array([[0, -h[2], h[1]], [h[2], 0, -h[0]], [-h[1], h[0], 0]])
    I6 = np.zeros((6, 6))
    I6[0:3, 0:3] = Ic_tot[0]; I6[0:3, 3:6] = skew(Ic_tot[1]); I6[3:6, 0:3] = -skew(Ic_tot[1]); I6[3:6, 3:6] = Ic_tot[2] * np.eye(3)
    A0 = I6.copy()
    rhs = -np.concatenate(p_tot)
    for L in range(4):
        lg = legs[L]
        A0 -= lg["T"] @ lg["F"].T
        rhs -= lg["T"] @ (tau[3 * L:3 * L + 3] - lg["C"])
    A0inv = np.linalg.inv(A0)
    a0 = A0inv @ rhs
    acc = np.zeros(18)
    for L in range(4):
        lg = legs[L]
        acc[6 + 3 * L:9 + 3 * L] = lg["Hinv"] @ (tau[3 * L:3 * L + 3] - lg["C"] - lg["F"].T @ a0)
    acc[0:3] = a0[0:3]
    acc[3:6] = a0[3:6] + cross(angvel, linvel) + np.array([0, 0, gz])   # spatial -> classical, uniform gravity field
    Minv = None
    if want_minv:
        Minv = np.zeros((18, 18))
        for k in range(18):
            Jb = np.zeros(6); jl = np.zeros(12)
            if k < 6:
                Jb[k] = 1
            else:
                jl[k - 6] = 1
            fb = Jb.copy()
            for L in range(4):
                fb -= legs[L]["T"] @ jl[3 * L:3 * L + 3]
            da0 = A0inv @ fb
            Minv[0:6, k] = da0
            for L in range(4):
                Minv[6 + 3 * L:9 + 3 * L, k] = legs[L]["Hinv"] @ jl[3 * L:3 * L + 3] - legs[L]["T"].T @ da0
    return acc, Minv


def main():
    from tests import phys_ref as pr
    from tests.oracle_lib import P
    from tests.test_oracle_physics import make_env, random_state
    worst_a, worst_m = 0.0, 0.0
    for robot in ("laikago", "mini_cheetah"):
        env, model = make_env(robot)
        lay = env.lay
        rng = np.random.RandomState(3)
        dirj, offj, motor_of_joint = pr.joint_maps(model)
        for trial in range(6):
            random_state(env, model, rng)
            s = env.state[0]
            tau_m = rng.randn(12) * 5
            acc = np.zeros(18); Minv = np.zeros((18, 18))
            env.L.orc_dynamics_probe(env.h, P(s), P(tau_m), P(acc), P(Minv))
            bodies, axes = pr.kinematics(model, s[lay.sl("POS")], s[lay.sl("QUAT")], s[lay.sl("Q")])
            qd = dirj * s[lay.sl("QD")]
            a2, M2 = forward_dynamics(bodies, axes, s[lay.sl("ANGVEL")], s[lay.sl("LINVEL")], qd, tau_m[motor_of_joint], -10.0, True)
            worst_a = max(worst_a, np.abs(a2 - acc).max() / max(1.0, np.abs(acc).max()))
            worst_m = max(worst_m, np.abs(M2 - Minv).max())
        env.close()
    print("max relative deviation of the accelerations: %.3e" % worst_a)
    print("max deviation of M^-1:                      %.3e" % worst_m)
    assert worst_a < 1e-9 and worst_m < 1e-9


if __name__ == "__main__":
    main()
