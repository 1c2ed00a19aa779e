"""Sparse bundle adjustment on the GPU - drop-ins for the SBA entry points of ``src/calib/calib.py``.

    prepare_calib_board_data_for_bundle_adjustment(...)            calib.py:210-263
    prepare_manual_points_for_bundle_adjustment(...)               calib.py:266-304
    bundle_adjust_points_only(..., project_func, f_scale=50)       calib.py:327-341
    bundle_adjust_board_points_only(...)                           calib.py:319-324
    bundle_adjust_points_and_extrinsics(..., project_func)         calib.py:369-390
    bundle_adjust_board_points_and_extrinsics(...)                 calib.py:362-366

Same argument order and return values (``obj_pts[, r_arr, t_arr], residuals`` with ``residuals = dict(before=,
after=)`` flat ``(reprojected - points_2d).ravel()`` vectors).  ``project_func`` selects the camera model exactly as
the reference's two call sites do (app.py:215-223): ``project_points_fisheye`` -> cv2.fisheye, ``project_points`` ->
cv2.projectPoints (rational / tangential / thin-prism pinhole); the arithmetic is the analytic-Jacobian Levenberg-Marquardt solver in csrc/sba.hip, which
minimises the SAME robust cost as scipy's ``least_squares(loss='cauchy', f_scale=...)``.  The last solve's
summary (costs, iterations, status) is kept in ``last_info``.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, calib
from ._lib import SbaInfo, SbaParams, check, lib, ptr, stream_ptr

last_info = None
# acino_sba_params::precision: fp64 throughout, or BASELINE config 5's "bf16 residuals with fp32 accumulate"
PRECISIONS = {"f64": 0, "bf16": 1}


def _camera_model(project_func):
    """0 = cv2.fisheye model (the reference's sba_board_points_fisheye, app.py:220-223), 1 = cv2.projectPoints pinhole
    model (sba_board_points, app.py:215-218); decided by the injected projection function's name."""
    name = getattr(project_func, "__name__", "")
    if project_func is None or "fisheye" in name:
        return 0
    if name == "project_points":
        return 1
    raise NotImplementedError(f"GPU bundle adjustment knows the reference's two camera models (got project_func={name})")


def _csr_by_point(point_3d_indices, n_points):
    idx = np.asarray(point_3d_indices, dtype=np.int64)
    if idx.size and (idx.min() < 0 or idx.max() >= n_points):
        raise ValueError("point_3d_indices out of range")
    order = np.argsort(idx, kind="stable").astype(np.int32)
    start = np.zeros(n_points + 1, dtype=np.int32)
    np.cumsum(np.bincount(idx, minlength=n_points), out=start[1:])
    return start, order


class ReduceHook:
    """The callback of ``acino_sba_solve_sharded``: combines ``n`` doubles at an address inside the workspace tensor
    ``ws`` over all ranks, in place (op 0 sum, 1 max).  ``comm`` offers ``all_reduce(tensor, op)`` - by default
    torch.distributed (backend "nccl" = RCCL on the GPU node; with "gloo" device memory is staged through the host,
    which is how several ranks can share one GPU in the tests)."""

    def __init__(self, ws, group=None):
        self.ws, self.group, self.calls, self.error = ws, group, 0, None
        self.sizes = []                      # doubles per reduction, in call order
        self.fn = _lib.REDUCE_FN(self._call)

    def all_reduce(self, t, op):
        import torch.distributed as dist
        rop = dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM
        if t.is_cuda and dist.get_backend(self.group) == "gloo":
            h = t.cpu()
            dist.all_reduce(h, op=rop, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=rop, group=self.group)
        if t.is_cuda:
            torch.cuda.current_stream().synchronize()

    def _call(self, _user, buf, n, op, _stream):
        try:
            off = int(buf) - self.ws.data_ptr()
            if off < 0 or off % 8 or off + 8 * n > self.ws.numel():
                raise ValueError("reduction buffer outside the workspace")
            self.all_reduce(self.ws[off:off + 8 * n].view(torch.float64), op)
            self.calls += 1
            self.sizes.append(int(n))
            return 0
        except Exception as e:      # an exception must not unwind through the C frames
            self.error = e
            return 1


def _solve(points_2d, points_3d, point_3d_indices, camera_indices, k_arr, d_arr, r_arr, t_arr, optimize_cameras,
           f_scale, max_iter, ftol, gtol, lam0=1e-3, model=0, group=None, sharded=False, precision="f64", host_checks=True):
    global last_info
    _lib.require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device())
    n_cams = len(k_arr)
    pts0 = np.ascontiguousarray(np.asarray(points_3d, dtype=np.float64).reshape(-1, 3))
    uv = np.ascontiguousarray(np.asarray(points_2d, dtype=np.float64).reshape(-1, 2))
    cam_idx = np.ascontiguousarray(np.asarray(camera_indices, dtype=np.int32).reshape(-1))
    n_points, n_obs = pts0.shape[0], uv.shape[0]
    if cam_idx.size != n_obs or len(point_3d_indices) != n_obs:
        raise ValueError("points_2d, point_3d_indices and camera_indices must have one entry per observation")
    # (host_checks=False leaves both tests below to the library, which makes them itself on the device - the C ABI's own
    #  guard, exercised by the tests - and answers ACINO_ERR_INVALID_ARG = ValueError)
    if host_checks and n_obs and (cam_idx.min() < 0 or cam_idx.max() >= n_cams):
        raise ValueError("camera_indices out of range")
    if host_checks and n_obs:
        # one GPU lane per (point, camera) slot (csrc/sba.hip): one observation per pair
        key = np.asarray(point_3d_indices, dtype=np.int64).reshape(-1) * n_cams + cam_idx
        if np.unique(key).size != key.size:
            raise ValueError("a point is observed twice by the same camera: merge the duplicate observations first")
    intr = np.zeros((n_cams, 16))
    Rt = np.zeros((n_cams, 12))
    for c in range(n_cams):
        k = np.asarray(k_arr[c], dtype=np.float64)
        dist = np.asarray(d_arr[c], dtype=np.float64).reshape(-1)
        if model == 0:
            if abs(k[0, 1]) > 1e-12 * abs(k[0, 0]):
                raise NotImplementedError("skewed fisheye intrinsics are not supported by the GPU bundle adjustment "
                                          "(calib.py:78 calibrates with CALIB_FIX_SKEW)")
            if dist.size != 4:
                raise ValueError("fisheye cameras have 4 distortion coefficients")
        elif dist.size not in (4, 5, 8, 12):
            raise ValueError("pinhole distortion vector must have 4, 5, 8 or 12 entries (cv2.projectPoints)")
        intr[c, :4] = [k[0, 0], k[1, 1], k[0, 2], k[1, 2]]
        intr[c, 4:4 + dist.size] = dist
        r = np.asarray(r_arr[c], dtype=np.float64)
        r = calib._rodrigues(r) if r.size == 3 else r
        if optimize_cameras:                      # calib.py:373 passes every rotation through cv2.Rodrigues, which
            u, _s, vt = np.linalg.svd(r)          # projects it onto SO(3) (scene files carry ~1e-8 of round-off)
            r = u @ vt
        Rt[c, :9] = r.reshape(-1)
        Rt[c, 9:] = np.asarray(t_arr[c], dtype=np.float64).reshape(-1)
    start, order = _csr_by_point(point_3d_indices, n_points)

    prm = SbaParams(n_cams=n_cams, optimize_cameras=int(bool(optimize_cameras)), n_points=n_points, n_obs=n_obs,
                    f_scale=float(f_scale), lam0=float(lam0), ftol=float(ftol), gtol=float(gtol), max_iter=int(max_iter),
                    camera_model=int(model), precision=PRECISIONS[precision])
    nbytes = lib().acino_sba_workspace_bytes(n_cams, n_points, n_obs)
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
    ws_ptr = (ws.data_ptr() + 255) // 256 * 256
    d = lambda a: torch.as_tensor(a, device=dev)
    d_intr, d_Rt, d_pts, d_uv = d(intr), d(Rt), d(pts0), d(uv)
    d_cam, d_start, d_order = d(cam_idx), d(start), d(order)
    res_b = torch.empty((n_obs, 2), dtype=torch.float64, device=dev)
    res_a = torch.empty((n_obs, 2), dtype=torch.float64, device=dev)
    info = SbaInfo()
    if sharded:
        hook = ReduceHook(ws, group)
        status = lib().acino_sba_solve_sharded(C.byref(prm), ptr(d_intr), ptr(d_Rt), ptr(d_pts), ptr(d_uv), ptr(d_cam),
                                               ptr(d_start), ptr(d_order), C.c_void_p(ws_ptr), nbytes, ptr(res_b),
                                               ptr(res_a), C.byref(info), hook.fn, None, stream_ptr())
        if hook.error is not None:
            raise hook.error
        check(status)
    else:
        check(lib().acino_sba_solve(C.byref(prm), ptr(d_intr), ptr(d_Rt), ptr(d_pts), ptr(d_uv), ptr(d_cam),
                                    ptr(d_start), ptr(d_order), C.c_void_p(ws_ptr), nbytes, ptr(res_b), ptr(res_a),
                                    C.byref(info), stream_ptr()))
    torch.cuda.current_stream().synchronize()
    last_info = info.as_dict()
    Rt_o = d_Rt.cpu().numpy()
    return (d_pts.cpu().numpy(), Rt_o[:, :9].reshape(n_cams, 3, 3).copy(), Rt_o[:, 9:].reshape(n_cams, 3, 1).copy(),
            dict(before=res_b.cpu().numpy().ravel(), after=res_a.cpu().numpy().ravel()))


def bundle_adjust_points_only(points_2d, points_3d, point_3d_indices, camera_indices, k_arr, d_arr, r_arr, t_arr,
                              project_func=None, f_scale=50, max_iter=200, ftol=1e-15, gtol=1e-10):
    """calib.py:327-341: refine the 3-D points, cameras fixed; Cauchy loss with scale ``f_scale`` px."""
    pts, _r, _t, residuals = _solve(points_2d, points_3d, point_3d_indices, camera_indices, k_arr, d_arr, r_arr, t_arr,
                                    False, f_scale, max_iter, ftol, gtol, model=_camera_model(project_func))
    return pts, residuals


def bundle_adjust_points_and_extrinsics(points_2d, points_3d, point_3d_indices, camera_indices, k_arr, d_arr, r_arr,
                                        t_arr, project_func=None, max_iter=300, ftol=1e-10, gtol=1e-10, precision="f64"):
    """calib.py:369-390: refine the 3-D points and every camera's rotation + translation (Cauchy loss, scale 1).
    ``precision="bf16"``: BASELINE config 5's mixed mode (residual / Jacobian rows in bf16, blocks accumulated in fp32)."""
    return _solve(points_2d, points_3d, point_3d_indices, camera_indices, k_arr, d_arr, r_arr, t_arr, True, 1.0,
                  max_iter, ftol, gtol, model=_camera_model(project_func), precision=precision)


def bundle_adjust_points_and_extrinsics_sharded(points_2d, points_3d, point_3d_indices, camera_indices, k_arr, d_arr,
                                                r_arr, t_arr, project_func=None, max_iter=300, ftol=1e-10, gtol=1e-10,
                                                group=None, precision="f64"):
    """The same refinement with the POINTS spread over the ranks of a torch.distributed group (one process per GPU)
    and the cameras shared: every rank passes its own points / observations (``point_3d_indices`` local, 0-based) and
    the same initial poses; the reduced camera system is summed over the ranks each iteration (a 6C x 6C block + 6C
    vector - BASELINE config 5's extrinsic refinement over all sequences).  Returns this rank's refined points, the
    common poses and this rank's residuals; ``last_info`` carries the GLOBAL costs."""
    return _solve(points_2d, points_3d, point_3d_indices, camera_indices, k_arr, d_arr, r_arr, t_arr, True, 1.0,
                  max_iter, ftol, gtol, model=_camera_model(project_func), group=group, sharded=True, precision=precision)


def prepare_calib_board_data_for_bundle_adjustment(img_pts_arr, fnames_arr, board_shape, k_arr, d_arr, r_arr, t_arr,
                                                   triangulate_func=None):
    """calib.py:210-263.  Boards seen by >= 2 cameras, in sorted file-name order (the reference iterates a
    set-built dict, i.e. an arbitrary order; the optimum does not depend on it).  Initial points from the first two
    cameras that see each board, all boards triangulated in one batched launch per camera pair."""
    triangulate_func = triangulate_func or calib.triangulate_points_fisheye
    n_cam = len(img_pts_arr)
    fnames_arr = [list(f) for f in fnames_arr]
    count = {}
    for fnames in fnames_arr:
        for f in set(fnames):
            count[f] = count.get(f, 0) + 1
    keep = sorted(f for f, v in count.items() if v >= 2)
    per_img = int(board_shape[0] * board_shape[1])
    lookup = [{f: i for i, f in reversed(list(enumerate(fnames)))} for fnames in fnames_arr]   # .index() = first hit
    points_2d, point_3d_indices, camera_indices = [], [], []
    pair_jobs = {}
    for b, fname in enumerate(keep):
        seen = [(cam, lookup[cam][fname]) for cam in range(n_cam) if fname in lookup[cam]]
        for cam, f_idx in seen:
            points_2d.append(np.asarray(img_pts_arr[cam][f_idx], dtype=np.float64).reshape(per_img, 2))
            point_3d_indices.append(np.arange(b * per_img, (b + 1) * per_img))
            camera_indices.append(np.full(per_img, cam))
        (ca, fa), (cb, fb) = seen[0], seen[1]
        pair_jobs.setdefault((ca, cb), []).append((b, fa, fb))
    points_3d = np.zeros((len(keep) * per_img, 3))
    for (ca, cb), jobs in pair_jobs.items():
        pa = np.concatenate([np.asarray(img_pts_arr[ca][fa], dtype=np.float64).reshape(per_img, 2) for _, fa, _ in jobs])
        pb = np.concatenate([np.asarray(img_pts_arr[cb][fb], dtype=np.float64).reshape(per_img, 2) for _, _, fb in jobs])
        est = np.asarray(triangulate_func(pa, pb, k_arr[ca], d_arr[ca], r_arr[ca], t_arr[ca],
                                          k_arr[cb], d_arr[cb], r_arr[cb], t_arr[cb])).reshape(-1, 3)
        for j, (b, _, _) in enumerate(jobs):
            points_3d[b * per_img:(b + 1) * per_img] = est[j * per_img:(j + 1) * per_img]
    if not keep:
        return (np.zeros((0, 2), np.float32), np.zeros((0, 3), np.float32), np.zeros(0, int), np.zeros(0, int))
    return (np.concatenate(points_2d).astype(np.float32), points_3d.astype(np.float32),
            np.concatenate(point_3d_indices).astype(int), np.concatenate(camera_indices).astype(int))


def prepare_manual_points_for_bundle_adjustment(img_pts_arr, k_arr, d_arr, r_arr, t_arr, triangulate_func=None):
    """calib.py:266-304: img_pts_arr[n_points, n_cameras, 2] with NaN where a camera does not see the point."""
    triangulate_func = triangulate_func or calib.triangulate_points_fisheye
    pts = np.asarray(img_pts_arr, dtype=np.float64).swapaxes(0, 1)
    n_cam, n_pts = pts.shape[0], pts.shape[1]
    points_2d, point_3d_indices, camera_indices, first_two = [], [], [], []
    p = 0
    for i in range(n_pts):
        cams = [c for c in range(n_cam) if not np.isnan(pts[c, i]).any()]
        if len(cams) > 1:
            points_2d.extend(pts[c, i] for c in cams)
            camera_indices.extend(cams)
            point_3d_indices.extend([p] * len(cams))
            first_two.append((cams[0], cams[1], i))
            p += 1
    points_3d = np.zeros((p, 1, 3))
    groups = {}
    for j, (a, b, i) in enumerate(first_two):
        groups.setdefault((a, b), []).append((j, i))
    for (a, b), items in groups.items():
        ia = np.array([i for _, i in items])
        est = np.asarray(triangulate_func(pts[a, ia], pts[b, ia], k_arr[a], d_arr[a], r_arr[a], t_arr[a],
                                          k_arr[b], d_arr[b], r_arr[b], t_arr[b])).reshape(-1, 3)
        points_3d[[j for j, _ in items], 0] = est
    return (np.array(points_2d, dtype=np.float32).reshape(-1, 2), points_3d.astype(np.float32),
            np.array(point_3d_indices, dtype=int), np.array(camera_indices, dtype=int))


def bundle_adjust_board_points_only(img_pts_arr, fnames_arr, board_shape, k_arr, d_arr, r_arr, t_arr,
                                    triangulate_func=None, project_func=None):
    """calib.py:319-324."""
    data = prepare_calib_board_data_for_bundle_adjustment(img_pts_arr, fnames_arr, board_shape, k_arr, d_arr, r_arr,
                                                          t_arr, triangulate_func)
    return bundle_adjust_points_only(*data, k_arr, d_arr, r_arr, t_arr, project_func)


def bundle_adjust_board_points_and_extrinsics(img_pts_arr, fnames_arr, board_shape, k_arr, d_arr, r_arr, t_arr,
                                              triangulate_func=None, project_func=None):
    """calib.py:362-366."""
    data = prepare_calib_board_data_for_bundle_adjustment(img_pts_arr, fnames_arr, board_shape, k_arr, d_arr, r_arr,
                                                          t_arr, triangulate_func)
    return bundle_adjust_points_and_extrinsics(*data, k_arr, d_arr, r_arr, t_arr, project_func)


# ---------------------------------------------------------------------------------------------------------------
# BASELINE config 5: extrinsic refinement over the marker trajectories of many sequences (app.py:201-223 feeds board
# corners or hand-picked points into calib.py:345-390; here the points are the FTE marker positions of every clip and
# the observations their above-threshold detections, the six extrinsics shared by all sequences)
# ---------------------------------------------------------------------------------------------------------------
def dense_observations(det, dlc_thresh, min_views=2):
    """Observation lists of dense detections det[N, C, L, 3] = (x, y, likelihood), built on the device: every (frame,
    marker) seen by >= min_views cameras above the threshold (calib.py:281 keeps points with > 1 view) becomes a point;
    observations are point-major, cameras ascending.  Returns (keep[N, L] bool, uv[M, 2], cam_idx[M] int32,
    pt_start[P + 1] int32, pt_obs[M] int32) as device tensors."""
    lik = det[..., 2].permute(0, 2, 1)                    # [N, L, C]
    seen = lik > dlc_thresh
    keep = seen.sum(-1) >= min_views                      # [N, L]
    sel = seen & keep[..., None]
    idx = torch.nonzero(sel)                              # rows sorted by (frame, marker, camera)
    uv = det.permute(0, 2, 1, 3)[idx[:, 0], idx[:, 1], idx[:, 2], :2].contiguous()
    cam_idx = idx[:, 2].to(torch.int32).contiguous()
    counts = sel.sum(-1)[keep]                            # observations per kept point, in point order
    pt_start = torch.zeros(counts.numel() + 1, dtype=torch.int32, device=det.device)
    pt_start[1:] = torch.cumsum(counts, 0).to(torch.int32)
    pt_obs = torch.arange(uv.shape[0], dtype=torch.int32, device=det.device)
    return keep, uv, cam_idx, pt_start, pt_obs


def bundle_adjust_dense_points_and_extrinsics(det, points_3d, k_arr, d_arr, r_arr, t_arr, dlc_thresh=0.5, precision="f64",
                                              max_iter=100, ftol=1e-10, gtol=1e-10, f_scale=1.0, lam0=1e-3, min_views=2,
                                              group=None):
    """calib.py:369-390 on DENSE data, everything resident on the device: det[N, C, 20, 3] detections and
    points_3d[N, 20, 3] initial points (e.g. ``positions`` of an FTE solve, clips concatenated along N); fisheye model.
    Returns (points[N, 20, 3] - refined where a point had >= min_views views, input value elsewhere -, r_arr[C, 3, 3],
    t_arr[C, 3, 1], info) with info = the solver summary plus ``n_points``, ``n_obs`` and the rms residuals (px)
    before / after.  ``group``: a torch.distributed group whose ranks each hold their own sequences (points sharded,
    cameras replicated: the reduced camera system - (6C)^2 + 6C doubles - is all-reduced every iteration)."""
    global last_info
    _lib.require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device())
    det = calib._to_dev(det, dev)
    pts_all = calib._to_dev(points_3d, dev).to(torch.float64).contiguous().clone()
    n_cams = len(k_arr)
    if det.dim() != 4 or det.shape[1] != n_cams or tuple(pts_all.shape) != (det.shape[0], det.shape[2], 3):
        raise ValueError("det must be [N, C, L, 3] and points_3d [N, L, 3] for the C cameras of the rig")
    keep, uv, cam_idx, pt_start, pt_obs = dense_observations(det, float(dlc_thresh), min_views)
    n_points, n_obs = int(pt_start.numel() - 1), int(uv.shape[0])
    if n_points < 1:
        raise ValueError("no point is seen by enough cameras")
    intr = np.zeros((n_cams, 16))
    Rt = np.zeros((n_cams, 12))
    for c in range(n_cams):
        k = np.asarray(k_arr[c], dtype=np.float64)
        intr[c, :4] = [k[0, 0], k[1, 1], k[0, 2], k[1, 2]]
        intr[c, 4:8] = np.asarray(d_arr[c], dtype=np.float64).reshape(-1)[:4]
        r = np.asarray(r_arr[c], dtype=np.float64)
        u, _s, vt = np.linalg.svd(calib._rodrigues(r) if r.size == 3 else r)     # (as _solve: cv2.Rodrigues' SO(3) projection)
        Rt[c, :9] = (u @ vt).reshape(-1)
        Rt[c, 9:] = np.asarray(t_arr[c], dtype=np.float64).reshape(-1)
    d_intr, d_Rt = torch.as_tensor(intr, device=dev), torch.as_tensor(Rt, device=dev)
    d_pts = pts_all[keep].contiguous()
    prm = SbaParams(n_cams=n_cams, optimize_cameras=1, n_points=n_points, n_obs=n_obs, f_scale=float(f_scale),
                    lam0=float(lam0), ftol=float(ftol), gtol=float(gtol), max_iter=int(max_iter), camera_model=0,
                    precision=PRECISIONS[precision])
    nbytes = lib().acino_sba_workspace_bytes(n_cams, n_points, n_obs)
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
    ws_ptr = (ws.data_ptr() + 255) // 256 * 256
    res_b = torch.empty((n_obs, 2), dtype=torch.float64, device=dev)
    res_a = torch.empty((n_obs, 2), dtype=torch.float64, device=dev)
    info = SbaInfo()
    hook = ReduceHook(ws, group) if group is not None else None
    status = lib().acino_sba_solve_sharded(C.byref(prm), ptr(d_intr), ptr(d_Rt), ptr(d_pts), ptr(uv), ptr(cam_idx),
                                           ptr(pt_start), ptr(pt_obs), C.c_void_p(ws_ptr), nbytes, ptr(res_b), ptr(res_a),
                                           C.byref(info), hook.fn if hook else _lib.REDUCE_FN(0), None, stream_ptr())
    if hook is not None and hook.error is not None:
        raise hook.error
    check(status)
    torch.cuda.current_stream().synchronize()
    last_info = info.as_dict()
    out = dict(last_info, n_points=n_points, n_obs=n_obs, precision=precision,
               rms_before=float(res_b.pow(2).mean().sqrt()), rms_after=float(res_a.pow(2).mean().sqrt()),
               reduce_calls=hook.calls if hook else 0, reduce_sizes=sorted(set(hook.sizes)) if hook else [])
    pts_all[keep] = d_pts
    Rt_o = d_Rt.cpu().numpy()
    return pts_all, Rt_o[:, :9].reshape(n_cams, 3, 3).copy(), Rt_o[:, 9:].reshape(n_cams, 3, 1).copy(), out


def refine_extrinsics_from_clips(dets, k_arr, d_arr, r_arr, t_arr, Ts, dlc_thresh=0.5, precision="bf16", fte_iter=60,
                                 sba_iter=60, fte_kw=None, sba_kw=None):
    """BASELINE config 5 end to end on one GPU: the clips' trajectories are estimated with the current rig (fte_solve_clips:
    all clips as one chain, ``precision`` = "bf16": bf16 residual / Jacobian rows, fp32 accumulation), then the marker
    positions of ALL clips and their above-threshold detections go through one bundle adjustment of points + the shared
    extrinsics in the same precision mode.  Returns (r_arr, t_arr, info) with info = dict(fte=..., sba=...)."""
    from . import fte
    outs = fte.fte_solve_clips(dets, k_arr, d_arr, r_arr, t_arr, Ts, dlc_thresh=dlc_thresh, max_iter=fte_iter,
                               return_numpy=False, precision=precision, **(fte_kw or {}))
    dev = torch.device("cuda", torch.cuda.current_device())
    pos = torch.cat([o[0]["positions"] for o in outs], 0)
    det_all = torch.cat([calib._to_dev(d, dev) for d in dets], 0)
    _pts, r_new, t_new, info = bundle_adjust_dense_points_and_extrinsics(det_all, pos, k_arr, d_arr, r_arr, t_arr, dlc_thresh,
                                                                         precision=precision, max_iter=sba_iter,
                                                                         **(sba_kw or {}))
    return r_new, t_new, dict(fte=outs[0][1], sba=info)
