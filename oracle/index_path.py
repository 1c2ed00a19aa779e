"""Oracle for the adjacent-pair triangulation index path.  Test infrastructure.

Restates src/calib/calib.py:394-423 (get_pairwise_3d_points_from_df): camera pairs
(i, i+1) only, inner merge on (frame, marker), concat in pair order, groupby-mean.
pandas' group mean is a Kahan-compensated sum in row (= pair) order divided by the
count; the dense form below reproduces exactly that order so means are bit-exact.
"""
import numpy as np


def pairwise_dense(det, thresh, k_arr, d_arr, r_arr, t_arr, triangulate_func):
    """det[N,C,L,3] (x, y, likelihood) -> tri[N,L,3] (NaN where no adjacent pair),
    npairs[N,L] uint8, pair_mask[N,L] uint8 (bit i = pair (i,i+1) contributed)."""
    det = np.asarray(det, dtype=np.float64)
    N, C, L, _ = det.shape
    valid = det[..., 2] > thresh
    ssum = np.zeros((N, L, 3))
    comp = np.zeros((N, L, 3))
    cnt = np.zeros((N, L), dtype=np.uint8)
    mask = np.zeros((N, L), dtype=np.uint8)
    for i in range(C - 1):
        both = valid[:, i, :] & valid[:, i + 1, :]
        if not both.any():
            continue
        n_idx, l_idx = np.nonzero(both)
        p3 = triangulate_func(det[n_idx, i, l_idx, :2].reshape(-1, 1, 2),
                              det[n_idx, i + 1, l_idx, :2].reshape(-1, 1, 2),
                              k_arr[i], d_arr[i], r_arr[i], t_arr[i],
                              k_arr[i + 1], d_arr[i + 1], r_arr[i + 1], t_arr[i + 1])
        # Kahan step, as pandas' group_mean does per row
        y = p3 - comp[n_idx, l_idx]
        tt = ssum[n_idx, l_idx] + y
        comp[n_idx, l_idx] = (tt - ssum[n_idx, l_idx]) - y
        ssum[n_idx, l_idx] = tt
        cnt[n_idx, l_idx] += 1
        mask[n_idx, l_idx] |= np.uint8(1 << i)
    tri = np.full((N, L, 3), np.nan)
    has = cnt > 0
    tri[has] = ssum[has] / cnt[has][:, None]
    return tri, cnt, mask


def get_pairwise_3d_points_from_df(points_2d_df, k_arr, d_arr, r_arr, t_arr, triangulate_func):
    """DataFrame form with the reference's output conventions: rows sorted by
    (frame, marker), ``frame`` float64, KeyError when no adjacent pair exists at all."""
    import pandas as pd
    df = points_2d_df
    frames = np.sort(df["frame"].unique())
    markers = sorted(df["marker"].unique())
    n_cam = len(k_arr)
    f_idx = {f: i for i, f in enumerate(frames)}
    m_idx = {m: i for i, m in enumerate(markers)}
    det = np.zeros((len(frames), n_cam, len(markers), 3))
    det[..., 2] = -np.inf
    fi = df["frame"].map(f_idx).to_numpy()
    mi = df["marker"].map(m_idx).to_numpy()
    ci = df["camera"].to_numpy().astype(int)
    det[fi, ci, mi, 0] = df["x"].to_numpy(dtype=np.float64)
    det[fi, ci, mi, 1] = df["y"].to_numpy(dtype=np.float64)
    det[fi, ci, mi, 2] = np.inf  # caller pre-filtered by likelihood: presence == valid
    tri, cnt, _ = pairwise_dense(det, 0.0, k_arr, d_arr, r_arr, t_arr, triangulate_func)
    if not (cnt > 0).any():
        raise KeyError("['frame', 'marker'] not in index")
    n_i, l_i = np.nonzero(cnt > 0)
    return pd.DataFrame({"frame": np.asarray(frames, dtype=np.float64)[n_i],
                         "marker": np.asarray(markers, dtype=object)[l_i],
                         "x": tri[n_i, l_i, 0], "y": tri[n_i, l_i, 1], "z": tri[n_i, l_i, 2]})
