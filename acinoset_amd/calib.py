"""Drop-in replacements for the camera-model entry points of AcinoSet's ``src/calib/calib.py``.

Same names, argument order and array conventions as the reference:

    triangulate_points_fisheye(img_pts_1, img_pts_2, k1, d1, r1, t1, k2, d2, r2, t2) -> (M,3)   calib.py:121-130
    triangulate_points(...)                                                          -> (M,3)   calib.py:52-61
    project_points_fisheye(obj_pts, k, d, r, t)                                      -> (M,2)   calib.py:132-136
    project_points(obj_pts, k, d, r, t)                                              -> (M,2)   calib.py:64-66
    get_pairwise_3d_points_from_df(points_2d_df, k_arr, d_arr, r_arr, t_arr, triangulate_func)  calib.py:394-423

Inputs may be numpy arrays (returned as numpy, like cv2 does) or torch CUDA tensors (returned as CUDA
tensors, no host round trip).  All arithmetic runs in hand-written HIP kernels through the C ABI of
libacinoset_hip.so; there is no CPU path - a missing library or GPU raises RuntimeError.
"""
import numpy as np
import torch

from . import _lib
from ._lib import CAM_STRIDE, PINHOLE_STRIDE, check, lib, ptr, stream_ptr


def _dev():
    _lib.require_gpu()
    return torch.device("cuda", torch.cuda.current_device())


def _to_dev(a, dev):
    if isinstance(a, torch.Tensor):
        return a.to(device=dev, dtype=torch.float64).contiguous()
    return torch.as_tensor(np.ascontiguousarray(np.asarray(a, dtype=np.float64)), device=dev)


def _ret(t, like):
    return t if isinstance(like, torch.Tensor) else t.cpu().numpy()


def _host(a):
    if isinstance(a, torch.Tensor):
        return a.detach().cpu().numpy().astype(np.float64)
    return np.asarray(a, dtype=np.float64)


def _rodrigues(rvec):
    """cv2.Rodrigues for an rvec handed to project_points (calib.py:65 passes `r` straight through)."""
    rvec = rvec.reshape(3)
    th = np.linalg.norm(rvec)
    if th < np.finfo(float).eps:
        return np.eye(3)
    kx, ky, kz = rvec / th
    K = np.array([[0, -kz, ky], [kz, 0, -kx], [-ky, kx, 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def fisheye_record(k, d, r, t):
    """One 24-double camera record [fx fy cx cy | k1..k4 | R | t | alpha 0 0 0] (include/acinoset_hip.h)."""
    k, d, r, t = _host(k), _host(d).reshape(-1), _host(r), _host(t).reshape(-1)
    if k.shape != (3, 3) or d.size != 4 or t.size != 3:
        raise ValueError("fisheye camera needs k (3,3), d (4,) or (4,1), t (3,) or (3,1)")
    if r.size == 3:
        r = _rodrigues(r)
    if r.shape != (3, 3):
        raise ValueError("r must be a 3x3 rotation matrix (or an rvec)")
    rec = np.zeros(CAM_STRIDE)
    rec[0:4] = k[0, 0], k[1, 1], k[0, 2], k[1, 2]
    rec[4:8] = d
    rec[8:17] = r.reshape(-1)
    rec[17:20] = t
    rec[20] = k[0, 1] / k[0, 0]
    return rec


def fisheye_records(k_arr, d_arr, r_arr, t_arr):
    k_arr, r_arr = _host(k_arr), _host(r_arr)
    d_arr = _host(d_arr).reshape(len(k_arr), -1)
    t_arr = _host(t_arr).reshape(len(k_arr), -1)
    return np.stack([fisheye_record(k_arr[i], d_arr[i], r_arr[i], t_arr[i]) for i in range(len(k_arr))])


def pinhole_record(k, d, r, t):
    k, d, r, t = _host(k), _host(d).reshape(-1), _host(r), _host(t).reshape(-1)
    if d.size not in (4, 5, 8, 12, 14):
        raise ValueError("OpenCV pinhole distortion vectors have 4, 5, 8, 12 or 14 entries")
    if d.size == 14 and (d[12] != 0 or d[13] != 0):
        raise ValueError("tilted sensor model (tau_x, tau_y) is not supported")
    if r.size == 3:
        r = _rodrigues(r)
    rec = np.zeros(PINHOLE_STRIDE)
    rec[0:4] = k[0, 0], k[1, 1], k[0, 2], k[1, 2]
    rec[4:4 + d.size] = d
    rec[18:27] = r.reshape(-1)
    rec[27:30] = t
    return rec


def _triangulate(fn_name, rec_fn, img_pts_1, img_pts_2, cam_a, cam_b):
    dev = _dev()
    p1 = _to_dev(img_pts_1, dev).reshape(-1, 2)
    p2 = _to_dev(img_pts_2, dev).reshape(-1, 2)
    if p1.shape != p2.shape:
        raise ValueError("img_pts_1 and img_pts_2 must hold the same number of points")
    ca = torch.as_tensor(rec_fn(*cam_a), device=dev)
    cb = torch.as_tensor(rec_fn(*cam_b), device=dev)
    out = torch.empty((p1.shape[0], 3), dtype=torch.float64, device=dev)
    check(getattr(lib(), fn_name)(ptr(p1), ptr(p2), p1.shape[0], ptr(ca), ptr(cb), ptr(out), stream_ptr()))
    return _ret(out, img_pts_1)


def triangulate_points_fisheye(img_pts_1, img_pts_2, k1, d1, r1, t1, k2, d2, r2, t2):
    """Two-view fisheye triangulation (calib.py:121-130): undistort both views, 4x4 DLT, dehomogenise."""
    return _triangulate("acino_triangulate_fisheye", fisheye_record, img_pts_1, img_pts_2, (k1, d1, r1, t1),
                        (k2, d2, r2, t2))


def triangulate_points(img_pts_1, img_pts_2, k1, d1, r1, t1, k2, d2, r2, t2):
    """Two-view pinhole triangulation (calib.py:52-61)."""
    return _triangulate("acino_triangulate_pinhole", pinhole_record, img_pts_1, img_pts_2, (k1, d1, r1, t1),
                        (k2, d2, r2, t2))


def _project(fn_name, rec_fn, obj_pts, k, d, r, t):
    dev = _dev()
    X = _to_dev(obj_pts, dev).reshape(-1, 3)
    cam = torch.as_tensor(rec_fn(k, d, r, t), device=dev)
    out = torch.empty((X.shape[0], 2), dtype=torch.float64, device=dev)
    check(getattr(lib(), fn_name)(ptr(X), X.shape[0], ptr(cam), ptr(out), stream_ptr()))
    return _ret(out, obj_pts)


def project_points_fisheye(obj_pts, k, d, r, t):
    """Fisheye projection (calib.py:132-136)."""
    return _project("acino_project_fisheye", fisheye_record, obj_pts, k, d, r, t)


def project_points(obj_pts, k, d, r, t):
    """Pinhole / rational-model projection (calib.py:64-66); `r` may be a matrix or an rvec."""
    return _project("acino_project_pinhole", pinhole_record, obj_pts, k, d, r, t)


def undistort_points_fisheye(pts, k, d, max_iter=10, eps=1e-8):
    """cv2.fisheye.undistortPoints(pts, k, d) -> normalised coordinates (calib.py:124-125)."""
    dev = _dev()
    p = _to_dev(pts, dev).reshape(-1, 2)
    cam = torch.as_tensor(fisheye_record(k, d, np.eye(3), np.zeros(3)), device=dev)
    out = torch.empty_like(p)
    check(lib().acino_undistort_fisheye(ptr(p), p.shape[0], ptr(cam), ptr(out), int(max_iter), float(eps),
                                        stream_ptr()))
    return _ret(out, pts)


# --------------------------------------------------------------------------- dense index path
def pinhole_records(k_arr, d_arr, r_arr, t_arr):
    k_arr, r_arr = _host(k_arr), _host(r_arr)
    d_arr = _host(d_arr).reshape(len(k_arr), -1)
    t_arr = _host(t_arr).reshape(len(k_arr), -1)
    return np.stack([pinhole_record(k_arr[i], d_arr[i], r_arr[i], t_arr[i]) for i in range(len(k_arr))])


def triangulate_pairs_dense(det, thresh, k_arr, d_arr, r_arr, t_arr, return_masks=True, model="fisheye"):
    """det[N,C,L,3] (x, y, likelihood) -> tri[N,L,3] (NaN where no adjacent pair), npairs[N,L] u8,
    pairmask[N,L] u8.  The dense form of get_pairwise_3d_points_from_df (adjacent pairs, mean).
    ``model``: "fisheye" (triangulate_points_fisheye, calib.py:121-130) or "pinhole" (triangulate_points,
    calib.py:52-61) - the two functions the reference injects through ``triangulate_func`` (app.py:215-223)."""
    dev = _dev()
    d = _to_dev(det, dev)
    if d.dim() != 4 or d.shape[-1] != 3:
        raise ValueError("det must be [N, C, L, 3]")
    if model not in ("fisheye", "pinhole"):
        raise ValueError("model must be 'fisheye' or 'pinhole'")
    N, Cn, L, _ = d.shape
    recs = fisheye_records if model == "fisheye" else pinhole_records
    cams = torch.as_tensor(recs(k_arr, d_arr, r_arr, t_arr), device=dev)
    if cams.shape[0] != Cn:
        raise ValueError("camera count mismatch")
    tri = torch.empty((N, L, 3), dtype=torch.float64, device=dev)
    npairs = torch.empty((N, L), dtype=torch.uint8, device=dev)
    mask = torch.empty((N, L), dtype=torch.uint8, device=dev)
    fn = lib().acino_triangulate_pairs if model == "fisheye" else lib().acino_triangulate_pairs_pinhole
    check(fn(ptr(d), N, Cn, L, float(thresh), ptr(cams), ptr(tri), ptr(npairs), ptr(mask), stream_ptr()))
    if not return_masks:
        return _ret(tri, det)
    return _ret(tri, det), _ret(npairs, det), _ret(mask, det)


def reproject_residuals(pts3, det, thresh, k_arr, d_arr, r_arr, t_arr):
    """pts3[N,L,3], det[N,C,L,3] -> residuals[N,C,L,2] (NaN where invalid) and
    sums = (count, sum r, sum r^2, 0.5*sum log1p(r^2))."""
    dev = _dev()
    p = _to_dev(pts3, dev)
    d = _to_dev(det, dev)
    N, Cn, L, _ = d.shape
    cams = torch.as_tensor(fisheye_records(k_arr, d_arr, r_arr, t_arr), device=dev)
    res = torch.empty((N, Cn, L, 2), dtype=torch.float64, device=dev)
    sums = torch.zeros(4, dtype=torch.float64, device=dev)
    check(lib().acino_reproject_residuals(ptr(p), ptr(d), N, Cn, L, float(thresh), ptr(cams), ptr(res), ptr(sums),
                                          stream_ptr()))
    return _ret(res, det), _ret(sums, det)


def triangulate_reproject_dense(det, thresh, k_arr, d_arr, r_arr, t_arr):
    """``triangulate_pairs_dense`` followed by ``reproject_residuals`` of the triangulated points, in ONE pass over
    the detections (BASELINE config 2): returns tri[N,L,3], npairs, pairmask, residuals[N,C,L,2], sums[4]."""
    dev = _dev()
    d = _to_dev(det, dev)
    if d.dim() != 4 or d.shape[-1] != 3:
        raise ValueError("det must be [N, C, L, 3]")
    N, Cn, L, _ = d.shape
    cams = torch.as_tensor(fisheye_records(k_arr, d_arr, r_arr, t_arr), device=dev)
    if cams.shape[0] != Cn:
        raise ValueError("camera count mismatch")
    tri = torch.empty((N, L, 3), dtype=torch.float64, device=dev)
    npairs = torch.empty((N, L), dtype=torch.uint8, device=dev)
    mask = torch.empty((N, L), dtype=torch.uint8, device=dev)
    res = torch.empty((N, Cn, L, 2), dtype=torch.float64, device=dev)
    sums = torch.zeros(4, dtype=torch.float64, device=dev)
    check(lib().acino_triangulate_reproject(ptr(d), N, Cn, L, float(thresh), ptr(cams), ptr(tri), ptr(npairs), ptr(mask),
                                            ptr(res), ptr(sums), stream_ptr()))
    return _ret(tri, det), _ret(npairs, det), _ret(mask, det), _ret(res, det), _ret(sums, det)


def dataframe_to_dense(points_2d_df, n_cameras):
    """Long DataFrame [frame, camera, marker, x, y, (likelihood)] -> dense det[N,C,L,3] plus the sorted
    frame and marker keys.  Rows absent from the frame get likelihood -inf (never valid)."""
    df = points_2d_df
    frames = np.sort(df["frame"].unique())
    markers = np.array(sorted(df["marker"].unique()), dtype=object)
    fi = np.searchsorted(frames, df["frame"].to_numpy())
    mi = np.searchsorted(markers.astype(str), df["marker"].to_numpy().astype(str))
    ci = df["camera"].to_numpy().astype(np.int64)
    if ci.size and (ci.min() < 0 or ci.max() >= n_cameras):
        raise ValueError("camera index outside the rig")
    det = np.zeros((len(frames), n_cameras, len(markers), 3))
    det[..., 2] = -np.inf
    det[fi, ci, mi, 0] = df["x"].to_numpy(dtype=np.float64)
    det[fi, ci, mi, 1] = df["y"].to_numpy(dtype=np.float64)
    det[fi, ci, mi, 2] = df["likelihood"].to_numpy(dtype=np.float64) if "likelihood" in df else 1.0
    return det, frames, markers


def get_pairwise_3d_points_from_df(points_2d_df, k_arr, d_arr, r_arr, t_arr, triangulate_func=None):
    """Adjacent-pair triangulation of a long detections DataFrame (calib.py:394-423).

    Output rows are sorted by (frame, marker) with ``frame`` as float64, exactly as the reference's
    groupby().mean().reset_index().  The caller pre-filters by likelihood (all_optimizations.py:262-263),
    so every row present is valid.  ``triangulate_func`` is the reference's injection seam (calib.py:394,
    412-413; app.py:215-223 injects either pair function): ``triangulate_points_fisheye`` (default) and
    ``triangulate_points`` select the fused dense HIP kernel of that camera model; any other callable raises
    (there is no CPU path here to run foreign Python per pair).
    Like the reference, raises KeyError when no adjacent pair exists at all.
    """
    import pandas as pd
    if triangulate_func is None or triangulate_func is triangulate_points_fisheye:
        model = "fisheye"
    elif triangulate_func is triangulate_points:
        model = "pinhole"
    else:
        raise NotImplementedError("triangulate_func must be acinoset_amd.calib.triangulate_points_fisheye or "
                                  "acinoset_amd.calib.triangulate_points (the dense HIP index path has no CPU fallback)")
    n_cam = len(k_arr)
    det, frames, markers = dataframe_to_dense(points_2d_df, n_cam)
    det[..., 2] = np.where(np.isfinite(det[..., 2]), np.inf, -np.inf)   # presence == valid
    tri, cnt, _ = triangulate_pairs_dense(det, 0.0, k_arr, d_arr, r_arr, t_arr, model=model)
    has = cnt > 0
    if not has.any():
        raise KeyError("['frame', 'marker'] not in index")
    n_i, l_i = np.nonzero(has)
    return pd.DataFrame({"frame": np.asarray(frames, dtype=np.float64)[n_i], "marker": markers[l_i],
                         "x": tri[n_i, l_i, 0], "y": tri[n_i, l_i, 1], "z": tri[n_i, l_i, 2]})
