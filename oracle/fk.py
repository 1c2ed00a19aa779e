"""Oracle cheetah forward kinematics (numpy fp64).  Test infrastructure.

Restates src/all_optimizations.py:66-190: the reference's rot_x/rot_y/rot_z (transposes
of the usual active rotations), the 14-frame body-from-inertial chain and the 20 marker
offsets.  The Jacobian is obtained by carrying d(RI_k)/d(angle) matrices through the same
chain (product rule) - deliberately a different method from the HIP kernels' geometric
(axis x lever-arm) Jacobian so the two check each other; both are pinned to the sympy
Jacobian of the reference's own expressions (tests/golden/cheetah_fk.npz).
"""
import numpy as np

N_STATES = 45
MARKERS = ["l_eye", "r_eye", "nose", "neck_base", "spine", "tail_base", "tail_mid", "tail_tip",
           "l_shoulder", "l_front_knee", "l_front_ankle", "r_shoulder", "r_front_knee",
           "r_front_ankle", "l_hip", "l_back_knee", "l_back_ankle", "r_hip", "r_back_knee",
           "r_back_ankle"]
# state layout [x, y, z, phi_0..13, theta_0..13, psi_0..13] (all_optimizations.py:182-185)
PHI, THETA, PSI = 3, 17, 31
ACTIVE = np.array([0, 1, 2, PHI + 0, PHI + 1, PHI + 3] + [THETA + i for i in range(14)] +
                  [PSI + 0, PSI + 1, PSI + 3, PSI + 4, PSI + 5])
# model variances Q (all_optimizations.py:245-252); zero -> state unused
Q_SIGMA = np.array([4, 7, 5,
                    13, 32, 0, 10, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                    9, 18, 43, 53, 90, 118, 247, 186, 194, 164, 295, 243, 334, 149,
                    26, 12, 0, 34, 43, 51, 0, 0, 0, 0, 0, 0, 0, 0], dtype=np.float64)


def bounds45():
    """Box bounds of src/all_optimizations.py:403-483, 0-based state index -> (lo, hi)."""
    lo = np.full(N_STATES, -np.inf)
    hi = np.full(N_STATES, np.inf)
    for p in (4, 18, 5, 19, 33, 20, 21, 7, 35):       # 1-based p, |x| <= pi/6
        lo[p - 1], hi[p - 1] = -np.pi / 6, np.pi / 6
    for p in (22, 36, 23, 37):                         # |x| <= pi/1.5
        lo[p - 1], hi[p - 1] = -np.pi / 1.5, np.pi / 1.5
    for p in (24, 26, 28, 30):                         # |x| <= pi/2
        lo[p - 1], hi[p - 1] = -np.pi / 2, np.pi / 2
    for p in (25, 27):                                 # |x + pi/2| <= pi/2
        lo[p - 1], hi[p - 1] = -np.pi, 0.0
    for p in (29, 31):                                 # |x - pi/2| <= pi/2
        lo[p - 1], hi[p - 1] = 0.0, np.pi
    return lo, hi


def _rot(axis, ang):
    """Reference rot_x / rot_y / rot_z (all_optimizations.py:66-91) and d/d(ang), batched."""
    c, s = np.cos(ang), np.sin(ang)
    z, o = np.zeros_like(c), np.ones_like(c)
    if axis == "x":
        R = [[o, z, z], [z, c, s], [z, -s, c]]
        dR = [[z, z, z], [z, -s, c], [z, -c, -s]]
    elif axis == "y":
        R = [[c, z, -s], [z, o, z], [s, z, c]]
        dR = [[-s, z, -c], [z, z, z], [c, z, -s]]
    else:
        R = [[c, s, z], [-s, c, z], [z, z, o]]
        dR = [[-s, c, z], [-c, -s, z], [z, z, z]]
    R = np.stack([np.stack(r, -1) for r in R], -2)
    dR = np.stack([np.stack(r, -1) for r in dR], -2)
    return R, dR


# (frame k, parent frame or None, [(axis, state index)] applied right-to-left i.e. last entry first)
_CHAIN = [
    (0, None, [("z", PSI + 0), ("x", PHI + 0), ("y", THETA + 0)]),
    (1, 0, [("z", PSI + 1), ("x", PHI + 1), ("y", THETA + 1)]),
    (2, 1, [("y", THETA + 2)]),
    (3, 2, [("z", PSI + 3), ("x", PHI + 3), ("y", THETA + 3)]),
    (4, 3, [("z", PSI + 4), ("y", THETA + 4)]),
    (5, 4, [("z", PSI + 5), ("y", THETA + 5)]),
    (6, 2, [("y", THETA + 6)]), (7, 6, [("y", THETA + 7)]),
    (8, 2, [("y", THETA + 8)]), (9, 8, [("y", THETA + 9)]),
    (10, 3, [("y", THETA + 10)]), (11, 10, [("y", THETA + 11)]),
    (12, 3, [("y", THETA + 12)]), (13, 12, [("y", THETA + 13)]),
]
# marker, parent marker (None = head origin), frame, offset  (all_optimizations.py:138-165)
_LINKS = [
    ("l_eye", None, 0, (0, 0.03, 0)), ("r_eye", None, 0, (0, -0.03, 0)), ("nose", None, 0, (0.055, 0, -0.055)),
    ("neck_base", None, 1, (-0.28, 0, 0)), ("spine", "neck_base", 2, (-0.37, 0, 0)),
    ("tail_base", "spine", 3, (-0.37, 0, 0)), ("tail_mid", "tail_base", 4, (-0.28, 0, 0)),
    ("tail_tip", "tail_mid", 5, (-0.36, 0, 0)),
    ("l_shoulder", "neck_base", 2, (-0.04, 0.08, -0.10)), ("l_front_knee", "l_shoulder", 6, (0, 0, -0.24)),
    ("l_front_ankle", "l_front_knee", 7, (0, 0, -0.28)),
    ("r_shoulder", "neck_base", 2, (-0.04, -0.08, -0.10)), ("r_front_knee", "r_shoulder", 8, (0, 0, -0.24)),
    ("r_front_ankle", "r_front_knee", 9, (0, 0, -0.28)),
    ("l_hip", "tail_base", 3, (0.12, 0.08, -0.06)), ("l_back_knee", "l_hip", 10, (0, 0, -0.32)),
    ("l_back_ankle", "l_back_knee", 11, (0, 0, -0.25)),
    ("r_hip", "tail_base", 3, (0.12, -0.08, -0.06)), ("r_back_knee", "r_hip", 12, (0, 0, -0.32)),
    ("r_back_ankle", "r_back_knee", 13, (0, 0, -0.25)),
]


def cheetah_fk(q, with_jac=False):
    """q[..., 45] -> positions[..., 20, 3] (row order = MARKERS) and optionally
    d(positions)/dq [..., 20, 3, 45]."""
    q = np.asarray(q, dtype=np.float64)
    batch = q.shape[:-1]
    q2 = q.reshape(-1, N_STATES)
    B = q2.shape[0]
    RI, dRI = {}, {}
    for k, parent, rots in _CHAIN:
        P = RI[parent] if parent is not None else np.tile(np.eye(3), (B, 1, 1))
        dP = dict(dRI[parent]) if (parent is not None and with_jac) else {}
        mats = [_rot(ax, q2[:, idx]) for ax, idx in rots]
        E = mats[0][0]
        for m in mats[1:]:
            E = E @ m[0]
        RI[k] = E @ P
        if with_jac:
            d = {a: E @ dm for a, dm in dP.items()}
            for j, (ax, idx) in enumerate(rots):
                dE = None
                for jj, m in enumerate(mats):
                    f = m[1] if jj == j else m[0]
                    dE = f if dE is None else dE @ f
                d[idx] = dE @ P
            dRI[k] = d
    pos, dpos = {}, {}
    head = q2[:, 0:3]
    out = np.empty((B, 20, 3))
    J = np.zeros((B, 20, 3, N_STATES)) if with_jac else None
    for li, (name, parent, k, off) in enumerate(_LINKS):
        off = np.asarray(off, dtype=np.float64)
        base = head if parent is None else pos[parent]
        pos[name] = base + np.einsum("bji,j->bi", RI[k], off)   # RI_k^T @ off
        out[:, li] = pos[name]
        if with_jac:
            if parent is None:
                dj = np.zeros((B, 3, N_STATES))
                dj[:, 0, 0] = dj[:, 1, 1] = dj[:, 2, 2] = 1.0
            else:
                dj = dpos[parent].copy()
            for idx, dR in dRI[k].items():
                dj[:, :, idx] += np.einsum("bji,j->bi", dR, off)
            dpos[name] = dj
            J[:, li] = dj
    out = out.reshape(batch + (20, 3))
    if with_jac:
        return out, J.reshape(batch + (20, 3, N_STATES))
    return out
