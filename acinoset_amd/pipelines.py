"""Array-level forms of the reference's per-clip reconstruction pipelines around the accelerated path.

    tri(...)        src/all_optimizations.py:909-940   adjacent-pair triangulation of every (frame, marker)
    sba_points(...) src/all_optimizations.py:868-895   the same points refined by ``bundle_adjust_points_only``
                                                       (the reference reaches it through ``app.sba_points_fisheye`` of
                                                       its un-shipped ``lib`` package; calib.py:327-341 is the solver)
    fte / ekf       -> acinoset_amd.fte.fte_solve, acinoset_amd.ekf.ekf

Both return ``positions[N, 20, 3]`` with NaN where a marker has no estimate - the layout the reference writes to
``tri.pickle`` / ``sba.pickle`` (:895-903, :932-938).
"""
import numpy as np
import torch

from . import calib, sba


def tri(det, k_arr, d_arr, r_arr, t_arr, dlc_thresh):
    """det[N, C, L, 3] (x, y, likelihood) -> positions[N, L, 3] (NaN where no adjacent camera pair sees the marker)."""
    out = calib.triangulate_pairs_dense(det, dlc_thresh, k_arr, d_arr, r_arr, t_arr, return_masks=False)
    return out if isinstance(out, np.ndarray) else out.cpu().numpy()


def sba_points(det, k_arr, d_arr, r_arr, t_arr, dlc_thresh, f_scale=50):
    """Triangulate, then refine every 3-D point against ALL cameras that see it (Cauchy loss, cameras fixed).
    Returns (positions[N, L, 3], residuals dict(before=, after=)) - points seen by fewer than two adjacent cameras
    stay NaN, exactly the points the reference's pairwise triangulation yields (calib.py:394-423)."""
    det = det.detach().cpu().numpy() if isinstance(det, torch.Tensor) else np.asarray(det, dtype=np.float64)
    N, C, L, _ = det.shape
    p3 = tri(det, k_arr, d_arr, r_arr, t_arr, dlc_thresh)
    have = np.isfinite(p3).all(-1)                                   # [N, L]
    valid = (det[..., 2] > dlc_thresh) & np.isfinite(det[..., 0]) & np.isfinite(det[..., 1])   # [N, C, L]
    idx = -np.ones((N, L), dtype=np.int64)
    idx[have] = np.arange(int(have.sum()))
    n_idx, c_idx, l_idx = np.nonzero(valid & have[:, None, :])
    points_2d = det[n_idx, c_idx, l_idx, :2]
    positions = np.full((N, L, 3), np.nan)
    if len(n_idx) == 0:
        return positions, dict(before=np.zeros(0), after=np.zeros(0))
    pts, residuals = sba.bundle_adjust_points_only(points_2d, p3[have], idx[n_idx, l_idx], c_idx, k_arr, d_arr, r_arr,
                                                   t_arr, calib.project_points_fisheye, f_scale=f_scale)
    positions[have] = pts
    return positions, residuals
