"""Oracle generic-skeleton forward kinematics (numpy).  Test infrastructure - see oracle/__init__.py.

Restates the pose construction of ``build_model`` (src/build.py:28-80): a skeleton dictionary
``{links, dofs, positions, markers}`` -> per part ``pose = pose[parent] + M_parent(q) @ (rest offset)``.
The reference builds this symbolically with sympy; its bookkeeping has two consequences that are kept here
because the stored results (KAT-3: data/results/traj_results.pickle, data/old_results/run1.pickle) depend on them:

  * every marker part gets all three rotational dofs (:36-37), whatever the pickle's ``dofs`` says;
  * only the ``<part>_i`` entries of ``rot_dict`` enter the poses (:77), and ``rot_dict[child + "_i"]`` is transposed
    once per link in which the part is the child (:76), never composed with its parent - so the offset of a link is
    rotated by its parent's OWN three angles only: by R_loc(parent)^T if the parent has been a child an even
    number of times so far, by R_loc(parent) otherwise, with R_loc = Rz(psi) Rx(phi) Ry(theta) (:54-60) in the
    reference's rot_x/rot_y/rot_z convention (build.py:418-454 = all_optimizations.py:66-91);
  * a part that is the child of two links (a kinematic loop in the skeleton graph) takes the LAST definition, but
    keeps its first position in the output order (dict semantics).

State layout (:68-69): ``[x, y, z, phi_0..phi_{L-1}, theta_0.., psi_0..]``, angle index = order of ``dofs``.
"""
import numpy as np

from .fk import _rot


def skeleton_fk(skel, q):
    """q[N, 3 + 3L] -> positions[N, n_pose, 3] in the reference's pose_dict order (build.py:79-83)."""
    q = np.atleast_2d(np.asarray(q, dtype=np.float64))
    links, positions = skel["links"], skel["positions"]
    dofs = {k: list(v) for k, v in skel["dofs"].items()}
    for joint in skel["markers"]:
        dofs[joint] = [1, 1, 1]
    parts = list(dofs.keys())
    L = len(positions)
    assert q.shape[1] == 3 + 3 * L, "state must be [x y z | phi | theta | psi]"
    N = q.shape[0]
    eye = np.broadcast_to(np.eye(3), (N, 3, 3))
    rot_i = {}
    for i, part in enumerate(parts):
        R = eye
        if dofs[part][1]:
            R = _rot("y", q[:, 3 + L + i])[0] @ R
        if dofs[part][0]:
            R = _rot("x", q[:, 3 + i])[0] @ R
        if dofs[part][2]:
            R = _rot("z", q[:, 3 + 2 * L + i])[0] @ R
        rot_i[part] = np.swapaxes(R, 1, 2)
    pose = {}
    for link in links:
        if len(link) == 1:
            pose[link[0]] = q[:, :3]
            continue
        a, b = link
        if a not in pose:
            pose[a] = q[:, :3]
        off = np.asarray(positions[b], dtype=np.float64) - np.asarray(positions[a], dtype=np.float64)
        rot_i[b] = np.swapaxes(rot_i[b], 1, 2)
        pose[b] = pose[a] + rot_i[a] @ off
    return np.stack([pose[k] for k in pose], axis=1), list(pose.keys())
