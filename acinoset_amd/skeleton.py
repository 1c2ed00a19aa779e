"""Generic-skeleton forward kinematics on the GPU.

Reference: ``build_model`` in src/build.py:28-95 turns a skeleton dictionary ``{links, dofs, positions, markers}``
(skeletons/*.pickle, loaded by ``load_skeleton`` :18-26) into sympy expressions for every part's position and
``lambdify``s them (``pose_to_3d``, :85-86).  Here the same construction is compiled ONCE, on the host, into a short
program of link operations ``pose[child] = pose[parent] + M(q) @ offset`` that a HIP kernel evaluates for all
frames.  The compiler keeps the reference's bookkeeping exactly (see oracle/skeleton_fk.py for the list): marker
parts get all three dofs, a link's offset is rotated by its parent's own angles only, transposed or not depending
on how often the parent has been a child so far, a part defined twice keeps its last definition and its first place
in the output order.  Pinned by the reference's stored runs (KAT-3) to 4e-15.
"""
import ctypes as C
import pickle

import numpy as np
import torch

from . import _lib, calib
from ._lib import SkelOp, check, lib, ptr, stream_ptr

MAX_OPS = 64


def load_skeleton(skel_file):
    """build.py:18-26."""
    with open(skel_file, "rb") as handle:
        return pickle.load(handle)


def compile_skeleton(skel):
    """-> dict(names, n_angles, ops): ``names`` = output part order (pose_dict order, build.py:79-83); ``ops`` = list
    of (child_slot, parent_slot, angle_index, dof_mask, use_untransposed, offset[3]) in evaluation order."""
    links, positions = skel["links"], skel["positions"]
    dofs = {k: list(v) for k, v in skel["dofs"].items()}
    for joint in skel["markers"]:                                    # :36-37
        dofs[joint] = [1, 1, 1]
    parts = list(dofs.keys())
    inv_is_transposed = {p: True for p in parts}                     # rot_dict[p + "_i"] = R_loc(p)^T   (:62)
    names, ops = [], []

    def slot(name):
        if name not in names:
            names.append(name)
        return names.index(name)

    for link in links:
        if len(link) == 1:                                           # :71-72
            slot(link[0])
            continue
        a, b = link
        if a not in parts or b not in parts:
            raise KeyError(f"link {link} names a part without dofs")
        sa = slot(a)
        off = np.asarray(positions[b], dtype=np.float64) - np.asarray(positions[a], dtype=np.float64)
        inv_is_transposed[b] = not inv_is_transposed[b]              # :76
        sb = slot(b)
        mask = (1 if dofs[a][0] else 0) | (2 if dofs[a][1] else 0) | (4 if dofs[a][2] else 0)
        ops.append((sb, sa, parts.index(a), mask, not inv_is_transposed[a], off))
    if len(ops) > MAX_OPS:
        raise ValueError(f"skeleton has {len(ops)} links; the kernel program holds {MAX_OPS}")
    return dict(names=names, n_angles=len(positions), ops=ops)


def skeleton_fk(skel, q):
    """``pose_to_3d(*q_n)`` for every frame: q[N, 3 + 3L] -> positions[N, n_parts, 3] (numpy in -> numpy out, CUDA
    tensor in -> CUDA tensor out).  ``skel`` is a skeleton dictionary or the result of ``compile_skeleton``."""
    prog = skel if "ops" in skel else compile_skeleton(skel)
    _lib.require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device())
    x = calib._to_dev(q, dev)
    if x.dim() == 1:
        x = x.unsqueeze(0)
    L = prog["n_angles"]
    if x.dim() != 2 or x.shape[1] != 3 + 3 * L:
        raise ValueError(f"state must be [N, {3 + 3 * L}] = [x y z | phi | theta | psi] for this skeleton")
    n_pose = len(prog["names"])
    ops = (SkelOp * max(len(prog["ops"]), 1))()
    for i, (child, parent, angle, mask, untransposed, off) in enumerate(prog["ops"]):
        ops[i].child, ops[i].parent, ops[i].angle = child, parent, angle
        ops[i].flags = mask | (8 if untransposed else 0)
        ops[i].off[0], ops[i].off[1], ops[i].off[2] = off
    pos = torch.empty((x.shape[0], n_pose, 3), dtype=torch.float64, device=dev)
    check(lib().acino_skeleton_fk(ptr(x), x.shape[0], L, n_pose, ops, len(prog["ops"]), ptr(pos), stream_ptr()))
    return calib._ret(pos, q)
