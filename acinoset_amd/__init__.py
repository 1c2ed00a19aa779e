"""acinoset_amd - MI355X-native (gfx950, fp64 HIP) triangulation + Full Trajectory Estimation hot path
of AcinoSet.  Python host layer over the C ABI of libacinoset_hip.so (include/acinoset_hip.h).

    from acinoset_amd import calib, fte
    pts3d = calib.triangulate_points_fisheye(p1, p2, k1, d1, r1, t1, k2, d2, r2, t2)
    results, info = fte.fte_solve(meas, likelihood, K, D, R, t, Ts)
"""
from . import _lib  # noqa: F401

__all__ = ["calib", "fte", "synth", "dist", "io"]
