"""Generic-skeleton FK: oracle vs the reference's stored runs (KAT-3), the host-side
skeleton compiler vs the oracle (CPU), and the HIP kernel vs both (GPU)."""
import json
import os

import numpy as np
import pytest

from oracle import skeleton_fk as osk
from oracle.fk import _rot


def _skels(golden_dir):
    g = np.load(os.path.join(golden_dir, "kat34_build_runs.npz"))
    return g, {tag: json.loads(str(g[f"{tag}_skeleton_json"])) for tag in ("traj", "run1", "cheetah")}


def _run_program(prog, q):
    """numpy interpreter of the compiled link program (what the kernel executes)."""
    L, N = prog["n_angles"], q.shape[0]
    pos = np.repeat(q[:, None, :3], len(prog["names"]), axis=1).copy()
    for child, parent, angle, mask, untransposed, off in prog["ops"]:
        R = np.broadcast_to(np.eye(3), (N, 3, 3))
        if mask & 2:
            R = _rot("y", q[:, 3 + L + angle])[0] @ R
        if mask & 1:
            R = _rot("x", q[:, 3 + angle])[0] @ R
        if mask & 4:
            R = _rot("z", q[:, 3 + 2 * L + angle])[0] @ R
        M = R if untransposed else np.swapaxes(R, 1, 2)
        pos[:, child] = pos[:, parent] + M @ off
    return pos


def test_oracle_reproduces_stored_runs(golden_dir):
    """KAT-3: positions stored by the reference (build.py:335-343) = pose_to_3d(x) of the stored states."""
    g, sk = _skels(golden_dir)
    for tag in ("traj", "run1"):
        pos, names = osk.skeleton_fk(sk[tag], g[f"{tag}_x"])
        assert pos.shape == g[f"{tag}_positions"].shape and names[0] == "chin"
        assert np.abs(pos - g[f"{tag}_positions"]).max() < 5e-15


def test_compiled_program_equals_oracle(golden_dir):
    from acinoset_amd import skeleton
    g, sk = _skels(golden_dir)
    rng = np.random.default_rng(11)
    for tag, s in sk.items():
        prog = skeleton.compile_skeleton(s)
        L = prog["n_angles"]
        q = rng.uniform(-2.5, 2.5, (40, 3 + 3 * L))
        want, names = osk.skeleton_fk(s, q)
        assert names == prog["names"]
        assert np.abs(_run_program(prog, q) - want).max() < 1e-14
    one = dict(links=[["a"], ["a", "b"]], dofs=dict(a=[0, 1, 0], b=[0, 0, 0]), positions=dict(a=[0, 0, 0], b=[1, 0, 0]),
               markers=[])
    prog = skeleton.compile_skeleton(one)
    assert prog["names"] == ["a", "b"] and prog["ops"][0][3] == 2          # single-part link, theta-only parent
    q = np.array([[1.0, 2.0, 3.0, 0, 0, 0.3, 0.7, 0, 0]])
    assert np.abs(_run_program(prog, q) - osk.skeleton_fk(one, q)[0]).max() < 1e-15


@pytest.mark.gpu
def test_hip_skeleton_fk(gpu_lib, golden_dir):
    import torch
    from acinoset_amd import skeleton
    g, sk = _skels(golden_dir)
    for tag in ("traj", "run1"):
        assert np.abs(skeleton.skeleton_fk(sk[tag], g[f"{tag}_x"]) - g[f"{tag}_positions"]).max() < 1e-14
    rng = np.random.default_rng(12)
    for tag, s in sk.items():
        L = len(s["positions"])
        q = rng.uniform(-3, 3, (1000, 3 + 3 * L))
        assert np.abs(skeleton.skeleton_fk(s, q) - osk.skeleton_fk(s, q)[0]).max() < 1e-13
    out = skeleton.skeleton_fk(skeleton.compile_skeleton(sk["traj"]), torch.tensor(g["traj_x"][:3], device="cuda"))
    assert out.is_cuda and out.shape == (3, 15, 3)
    assert skeleton.skeleton_fk(sk["traj"], np.zeros((0, 48))).shape == (0, 15, 3)
    with pytest.raises(ValueError):
        skeleton.skeleton_fk(sk["traj"], np.zeros((2, 45)))
