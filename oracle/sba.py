"""Oracle sparse bundle adjustment (numpy + scipy).  Test infrastructure - see oracle/__init__.py.

Restates src/calib/calib.py:196-390: data preparation for checkerboard SBA (:210-263), the residual
functions (:312-316, :355-359) and the two scipy.optimize.least_squares calls (:335, :381-385; TRF, Cauchy
loss, x_scale='jac').  The reference differentiates by finite differences through a sparsity mask and calls
cv2.fisheye.projectPoints once per observation; here the residual is vectorised (same arithmetic) and the
same least_squares settings are used, so KAT-2 (src/calib_with_gui.ipynb cell 29: final costs 5.3361e+01 /
2.2845e+01) pins it.
"""
import numpy as np
from scipy.optimize import least_squares
from scipy.sparse import lil_matrix

from . import camera


def rodrigues_to_vec(R):
    """cv2.Rodrigues(matrix) -> rvec."""
    R = np.asarray(R, dtype=np.float64)
    u, _s, vt = np.linalg.svd(R)          # cv2.Rodrigues projects onto SO(3) first (R = U V^T)
    R = u @ vt
    c = (np.trace(R) - 1.0) / 2.0
    c = min(1.0, max(-1.0, c))
    th = np.arccos(c)
    if th < 1e-12:
        return np.zeros(3)
    ax = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / (2 * np.sin(th))
    return ax * th


def prepare_calib_board_data(img_pts_arr, fnames_arr, board_shape, k_arr, d_arr, r_arr, t_arr, triangulate_func):
    """calib.py:210-263 (dict order = first-seen order is replaced by sorted names: the reference iterates a
    set-built dict, whose order is arbitrary; costs do not depend on it)."""
    n_cam = len(img_pts_arr)
    names = {}
    for fnames in fnames_arr:
        for f in fnames:
            names[f] = names.get(f, 0) + 1
    keep = sorted(f for f, v in names.items() if v >= 2)
    per_img = board_shape[0] * board_shape[1]
    points_3d, point_3d_indices, points_2d, camera_indices = [], [], [], []
    counter = 0
    for fname in keep:
        tri_pts, tri_cams = [], []
        for cam in range(n_cam):
            if fname in fnames_arr[cam]:
                f_idx = list(fnames_arr[cam]).index(fname)
                tri_pts.append(f_idx)
                tri_cams.append(cam)
                points_2d.extend(np.array(img_pts_arr[cam][f_idx]).reshape(per_img, 2))
                point_3d_indices.extend(range(counter, counter + per_img))
                camera_indices.extend([cam] * per_img)
        a, b = tri_cams[0], tri_cams[1]
        est = triangulate_func(img_pts_arr[a][tri_pts[0]], img_pts_arr[b][tri_pts[1]],
                               k_arr[a], d_arr[a], r_arr[a], t_arr[a], k_arr[b], d_arr[b], r_arr[b], t_arr[b])
        points_3d.extend(est)
        counter += per_img
    return (np.array(points_2d, dtype=np.float32), np.array(points_3d, dtype=np.float32),
            np.array(point_3d_indices, dtype=int), np.array(camera_indices, dtype=int))


def residuals(obj_pts, r_mats, t_arr, k_arr, d_arr, point_3d_indices, camera_indices, points_2d, project_func=None):
    """(reprojected - points_2d).ravel(), vectorised per camera (calib.py:357-359); project_func = the injected
    projection (default: the fisheye model, app.py:220-223)."""
    project_func = project_func or camera.project_points_fisheye
    out = np.empty((len(points_2d), 2))
    for c in range(len(k_arr)):
        sel = camera_indices == c
        if sel.any():
            out[sel] = project_func(obj_pts[point_3d_indices[sel]], k_arr[c], d_arr[c], r_mats[c], t_arr[c])
    return (out - points_2d).ravel()


def sparsity(n_cameras, n_params_per_camera, camera_indices, n_points, point_indices):
    """calib.py:196-207."""
    m = camera_indices.size * 2
    n = n_cameras * n_params_per_camera + n_points * 3
    A = lil_matrix((m, n), dtype=int)
    i = np.arange(camera_indices.size)
    for s in range(n_params_per_camera):
        A[2 * i, camera_indices * n_params_per_camera + s] = 1
        A[2 * i + 1, camera_indices * n_params_per_camera + s] = 1
    for s in range(3):
        A[2 * i, n_cameras * n_params_per_camera + point_indices * 3 + s] = 1
        A[2 * i + 1, n_cameras * n_params_per_camera + point_indices * 3 + s] = 1
    return A


def cauchy_cost(res, f_scale=1.0):
    """scipy's cost for loss='cauchy': 0.5 * f_scale^2 * sum ln(1 + (r/f_scale)^2)."""
    return 0.5 * f_scale ** 2 * float(np.sum(np.log1p((np.asarray(res) / f_scale) ** 2)))


def bundle_adjust_points_and_extrinsics(points_2d, points_3d, point_3d_indices, camera_indices, k_arr, d_arr, r_arr,
                                        t_arr, max_nfev=1000, ftol=1e-10, verbose=0, project_func=None):
    """calib.py:369-390 with the reference's least_squares settings."""
    n_points, n_cameras = len(points_3d), len(k_arr)
    r_vecs = np.array([rodrigues_to_vec(r) for r in r_arr]).flatten()
    x0 = np.concatenate([r_vecs, np.asarray(t_arr, dtype=np.float64).flatten(), np.asarray(points_3d, dtype=np.float64).flatten()])

    def unpack(params):
        r_end = n_cameras * 3
        t_end = r_end + n_cameras * 3
        rm = np.array([camera.rodrigues(r) for r in params[:r_end].reshape(n_cameras, 3)])
        return params[t_end:].reshape(n_points, 3), rm, params[r_end:t_end].reshape(n_cameras, 3, 1)

    def fun(params):
        pts, rm, tt = unpack(params)
        return residuals(pts, rm, tt, k_arr, d_arr, point_3d_indices, camera_indices, points_2d, project_func)

    f0 = fun(x0)
    A = sparsity(n_cameras, 6, camera_indices, n_points, point_3d_indices)
    res = least_squares(fun, x0, jac_sparsity=A, verbose=verbose, x_scale="jac", ftol=ftol, method="trf", loss="cauchy",
                        max_nfev=max_nfev)
    pts, rm, tt = unpack(res.x)
    return pts, rm, tt, dict(before=f0, after=res.fun), res


def bundle_adjust_points_only(points_2d, points_3d, point_3d_indices, camera_indices, k_arr, d_arr, r_arr, t_arr,
                              f_scale=50, max_nfev=500, ftol=1e-15, verbose=0):
    """calib.py:327-341."""
    n_points = len(points_3d)
    x0 = np.asarray(points_3d, dtype=np.float64).flatten()

    def fun(params):
        return residuals(params.reshape(n_points, 3), r_arr, t_arr, k_arr, d_arr, point_3d_indices, camera_indices,
                         points_2d)

    f0 = fun(x0)
    A = sparsity(len(k_arr), 0, camera_indices, n_points, point_3d_indices)
    res = least_squares(fun, x0, jac_sparsity=A, verbose=verbose, x_scale="jac", ftol=ftol, method="trf", loss="cauchy",
                        f_scale=f_scale, max_nfev=max_nfev)
    return res.x.reshape(n_points, 3), dict(before=f0, after=res.fun), res
