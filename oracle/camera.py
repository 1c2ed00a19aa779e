"""Oracle camera models (numpy fp64).  Test infrastructure - see oracle/__init__.py.

Restates the OpenCV routines the reference calls through src/calib/calib.py
(cv2 is not vendored; algorithms follow OpenCV 4.x fisheye.cpp / triangulate.cpp /
calibration.cpp / undistort.dispatch.cpp) and the reference's own closed-form
projection ``pt3d_to_2d`` (src/all_optimizations.py:193-209, src/build.py:457-473).
"""
import numpy as np


# --------------------------------------------------------------------------- fisheye
def undistort_points_fisheye(pts, k, d, max_iter=10, eps=1e-8):
    """cv2.fisheye.undistortPoints(pts, k, d) with R=P=None -> normalised coordinates.

    Reference call sites: src/calib/calib.py:124-125.  Kannala-Brandt inverse by Newton
    on theta*(1+k1 th^2+k2 th^4+k3 th^6+k4 th^8) = theta_d, OpenCV default criteria
    (COUNT+EPS, 10, 1e-8); non-converged / sign-flipped points become -1e6 as in OpenCV >= 4.5.
    """
    pts = np.asarray(pts, dtype=np.float64).reshape(-1, 2)
    k = np.asarray(k, dtype=np.float64)
    d = np.asarray(d, dtype=np.float64).reshape(-1)
    fx, fy, cx, cy = k[0, 0], k[1, 1], k[0, 2], k[1, 2]
    alpha = k[0, 1] / fx
    pw_y = (pts[:, 1] - cy) / fy
    pw_x = (pts[:, 0] - cx) / fx - alpha * pw_y
    theta_d = np.sqrt(pw_x * pw_x + pw_y * pw_y)
    theta_d = np.minimum(np.maximum(-np.pi / 2.0, theta_d), np.pi / 2.0)
    theta = theta_d.copy()
    converged = np.abs(theta_d) <= eps
    active = ~converged
    for _ in range(max_iter):
        t2 = theta * theta
        t4 = t2 * t2
        t6 = t4 * t2
        t8 = t4 * t4
        k0t2, k1t4, k2t6, k3t8 = d[0] * t2, d[1] * t4, d[2] * t6, d[3] * t8
        fix = (theta * (1 + k0t2 + k1t4 + k2t6 + k3t8) - theta_d) / \
              (1 + 3 * k0t2 + 5 * k1t4 + 7 * k2t6 + 9 * k3t8)
        theta = np.where(active, theta - fix, theta)
        newly = active & (np.abs(fix) < eps)
        converged |= newly
        active &= ~newly
        if not active.any():
            break
    with np.errstate(divide="ignore", invalid="ignore"):
        scale = np.where(np.abs(theta_d) > eps, np.tan(theta) / theta_d, 0.0)
    flipped = ((theta_d < 0) & (theta > 0)) | ((theta_d > 0) & (theta < 0))
    ok = converged & ~flipped
    out = np.stack([pw_x * scale, pw_y * scale], axis=1)
    out[~ok] = -1000000.0
    return out


def triangulate_dlt(p1, p2, x1, x2):
    """cv2.triangulatePoints(P1, P2, x1, x2): per point the 4x4 A, SVD, last right-singular vector.

    Returns homogeneous (4, M) like OpenCV.  Reference call site: src/calib/calib.py:128.
    """
    x1 = np.asarray(x1, dtype=np.float64).reshape(-1, 2)
    x2 = np.asarray(x2, dtype=np.float64).reshape(-1, 2)
    p1 = np.asarray(p1, dtype=np.float64)
    p2 = np.asarray(p2, dtype=np.float64)
    A = np.empty((x1.shape[0], 4, 4))
    A[:, 0, :] = x1[:, 0:1] * p1[2] - p1[0]
    A[:, 1, :] = x1[:, 1:2] * p1[2] - p1[1]
    A[:, 2, :] = x2[:, 0:1] * p2[2] - p2[0]
    A[:, 3, :] = x2[:, 1:2] * p2[2] - p2[1]
    _, _, vt = np.linalg.svd(A)
    return vt[:, 3, :].T


def triangulate_points_fisheye(img_pts_1, img_pts_2, k1, d1, r1, t1, k2, d2, r2, t2):
    """Restatement of src/calib/calib.py:121-130 (two-view fisheye triangulation)."""
    pts_1 = undistort_points_fisheye(np.asarray(img_pts_1).reshape(-1, 2), k1, d1)
    pts_2 = undistort_points_fisheye(np.asarray(img_pts_2).reshape(-1, 2), k2, d2)
    p1 = np.hstack((np.asarray(r1, dtype=np.float64), np.asarray(t1, dtype=np.float64).reshape(3, 1)))
    p2 = np.hstack((np.asarray(r2, dtype=np.float64), np.asarray(t2, dtype=np.float64).reshape(3, 1)))
    pts_4d = triangulate_dlt(p1, p2, pts_1, pts_2)
    return (pts_4d[:3] / pts_4d[3]).T


def project_points_fisheye(obj_pts, k, d, r, t):
    """Restatement of src/calib/calib.py:132-136 (cv2.fisheye.projectPoints; r is a 3x3 matrix)."""
    X = np.asarray(obj_pts, dtype=np.float64).reshape(-1, 3)
    k = np.asarray(k, dtype=np.float64)
    d = np.asarray(d, dtype=np.float64).reshape(-1)
    r = np.asarray(r, dtype=np.float64)
    if r.size == 3:
        r = rodrigues(r.reshape(3))
    Y = X @ r.T + np.asarray(t, dtype=np.float64).reshape(1, 3)
    a = Y[:, 0] / Y[:, 2]
    b = Y[:, 1] / Y[:, 2]
    rr = np.sqrt(a * a + b * b)
    th = np.arctan(rr)
    th2 = th * th
    th_d = th * (1 + d[0] * th2 + d[1] * th2 ** 2 + d[2] * th2 ** 3 + d[3] * th2 ** 4)
    with np.errstate(divide="ignore", invalid="ignore"):
        cdist = np.where(rr > 1e-8, th_d / rr, 1.0)
    alpha = k[0, 1] / k[0, 0]
    xd = a * cdist
    yd = b * cdist
    return np.stack([(xd + alpha * yd) * k[0, 0] + k[0, 2], yd * k[1, 1] + k[1, 2]], axis=1)


def pt3d_to_2d(X, K, D, R, t, with_jac=False):
    """The reference's closed-form projection inside the NLP (src/all_optimizations.py:193-209).

    Identical to fisheye projection except r = sqrt(a^2+b^2+1e-12) and no r->0 branch.
    X: (..., 3).  Returns uv (..., 2) and optionally d(uv)/dX (..., 2, 3) and camera-frame z.
    """
    X = np.asarray(X, dtype=np.float64)
    K = np.asarray(K, dtype=np.float64)
    D = np.asarray(D, dtype=np.float64).reshape(-1)
    R = np.asarray(R, dtype=np.float64)
    t = np.asarray(t, dtype=np.float64).reshape(-1)
    xc = X[..., 0] * R[0, 0] + X[..., 1] * R[0, 1] + X[..., 2] * R[0, 2] + t[0]
    yc = X[..., 0] * R[1, 0] + X[..., 1] * R[1, 1] + X[..., 2] * R[1, 2] + t[1]
    zc = X[..., 0] * R[2, 0] + X[..., 1] * R[2, 1] + X[..., 2] * R[2, 2] + t[2]
    a = xc / zc
    b = yc / zc
    r = (a ** 2 + b ** 2 + 1e-12) ** 0.5
    th = np.arctan(r)
    th2 = th * th
    poly = 1 + D[0] * th2 + D[1] * th2 ** 2 + D[2] * th2 ** 3 + D[3] * th2 ** 4
    th_D = th * poly
    m = th_D / r
    u = K[0, 0] * a * m + K[0, 2]
    v = K[1, 1] * b * m + K[1, 2]
    uv = np.stack([u, v], axis=-1)
    if not with_jac:
        return uv
    dthD = 1 + 3 * D[0] * th2 + 5 * D[1] * th2 ** 2 + 7 * D[2] * th2 ** 3 + 9 * D[3] * th2 ** 4
    dm_dr = (dthD / (1 + r * r) * r - th_D) / (r * r)
    dm_da = dm_dr * a / r
    dm_db = dm_dr * b / r
    du_da = K[0, 0] * (m + a * dm_da)
    du_db = K[0, 0] * a * dm_db
    dv_da = K[1, 1] * b * dm_da
    dv_db = K[1, 1] * (m + b * dm_db)
    iz = 1.0 / zc
    # d(a,b)/dXc = [[iz, 0, -a iz], [0, iz, -b iz]]
    du_dXc = np.stack([du_da * iz, du_db * iz, -(du_da * a + du_db * b) * iz], axis=-1)
    dv_dXc = np.stack([dv_da * iz, dv_db * iz, -(dv_da * a + dv_db * b) * iz], axis=-1)
    J = np.stack([du_dXc @ R, dv_dXc @ R], axis=-2)
    return uv, J, zc


# --------------------------------------------------------------------------- pinhole (parity unpinned)
def _dist14(d):
    d = np.asarray(d, dtype=np.float64).reshape(-1)
    assert d.size in (4, 5, 8, 12, 14), "OpenCV distortion vectors have 4, 5, 8, 12 or 14 entries"
    k = np.zeros(14)
    k[:d.size] = d
    return k


def rodrigues(rvec):
    """cv2.Rodrigues(rvec) -> 3x3 (used when callers hand project_points an rvec, calib.py:65)."""
    rvec = np.asarray(rvec, dtype=np.float64).reshape(3)
    th = np.linalg.norm(rvec)
    if th < np.finfo(float).eps:
        return np.eye(3)
    kx, ky, kz = rvec / th
    Kx = np.array([[0, -kz, ky], [kz, 0, -kx], [-ky, kx, 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)


def project_points(obj_pts, k, d, r, t):
    """Restatement of src/calib/calib.py:64-66 (cv2.projectPoints; rational model k1,k2,p1,p2,k3,k4,k5,k6,s1..s4)."""
    X = np.asarray(obj_pts, dtype=np.float64).reshape(-1, 3)
    k = np.asarray(k, dtype=np.float64)
    kk = _dist14(d)
    r = np.asarray(r, dtype=np.float64)
    R = rodrigues(r) if r.size == 3 else r
    Y = X @ R.T + np.asarray(t, dtype=np.float64).reshape(1, 3)
    x = Y[:, 0] / Y[:, 2]
    y = Y[:, 1] / Y[:, 2]
    r2 = x * x + y * y
    r4 = r2 * r2
    r6 = r4 * r2
    a1 = 2 * x * y
    a2 = r2 + 2 * x * x
    a3 = r2 + 2 * y * y
    cdist = 1 + kk[0] * r2 + kk[1] * r4 + kk[4] * r6
    icdist2 = 1.0 / (1 + kk[5] * r2 + kk[6] * r4 + kk[7] * r6)
    xd = x * cdist * icdist2 + kk[2] * a1 + kk[3] * a2 + kk[8] * r2 + kk[9] * r4
    yd = y * cdist * icdist2 + kk[2] * a3 + kk[3] * a1 + kk[10] * r2 + kk[11] * r4
    return np.stack([xd * k[0, 0] + k[0, 2], yd * k[1, 1] + k[1, 2]], axis=1)


def undistort_points(pts, k, d, iters=5):
    """cv2.undistortPoints(pts, k, d) (no R/P): 5 fixed-point iterations (OpenCV default criteria)."""
    pts = np.asarray(pts, dtype=np.float64).reshape(-1, 2)
    k = np.asarray(k, dtype=np.float64)
    kk = _dist14(d)
    x0 = (pts[:, 0] - k[0, 2]) / k[0, 0]
    y0 = (pts[:, 1] - k[1, 2]) / k[1, 1]
    x, y = x0.copy(), y0.copy()
    alive = np.ones(x.shape, dtype=bool)
    for _ in range(iters):
        r2 = x * x + y * y
        icdist = (1 + ((kk[7] * r2 + kk[6]) * r2 + kk[5]) * r2) / (1 + ((kk[4] * r2 + kk[1]) * r2 + kk[0]) * r2)
        bad = alive & (icdist < 0)
        x = np.where(bad, x0, x)
        y = np.where(bad, y0, y)
        alive &= ~bad
        dx = 2 * kk[2] * x * y + kk[3] * (r2 + 2 * x * x) + kk[8] * r2 + kk[9] * r2 * r2
        dy = kk[2] * (r2 + 2 * y * y) + 2 * kk[3] * x * y + kk[10] * r2 + kk[11] * r2 * r2
        x = np.where(alive, (x0 - dx) * icdist, x)
        y = np.where(alive, (y0 - dy) * icdist, y)
    return np.stack([x, y], axis=1)


def triangulate_points(img_pts_1, img_pts_2, k1, d1, r1, t1, k2, d2, r2, t2):
    """Restatement of src/calib/calib.py:52-61 (two-view pinhole triangulation)."""
    pts_1 = undistort_points(np.asarray(img_pts_1).reshape(-1, 2), k1, d1)
    pts_2 = undistort_points(np.asarray(img_pts_2).reshape(-1, 2), k2, d2)
    p1 = np.hstack((np.asarray(r1, dtype=np.float64), np.asarray(t1, dtype=np.float64).reshape(3, 1)))
    p2 = np.hstack((np.asarray(r2, dtype=np.float64), np.asarray(t2, dtype=np.float64).reshape(3, 1)))
    pts_4d = triangulate_dlt(p1, p2, pts_1, pts_2)
    return (pts_4d[:3] / pts_4d[3]).T
