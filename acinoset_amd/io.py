"""Input / output formats either side of the hot path: the scene JSON, DLC
detection tables, the dense detections array of the C ABI, and the result files.

Reference: src/calib/utils.py:65-120 (save/load_scene, load_points, create_dlc_points_2d_file),
src/all_optimizations.py:548-559 (fte.pickle layout), :932-937 (tri positions scatter).  Pure host code."""
import json
import os
import pickle
from datetime import datetime

import numpy as np


def load_scene(fpath):
    """utils.load_scene (utils.py:84-101): -> k_arr[C,3,3], d_arr[C,4,1], r_arr[C,3,3], t_arr[C,3,1], camera_resolution."""
    with open(fpath, "r") as f:
        data = json.load(f)
    camera_resolution = tuple(data["camera_resolution"])
    k_arr = np.array([c["k"] for c in data["cameras"]], dtype=np.float64)
    d_arr = np.array([c["d"] for c in data["cameras"]], dtype=np.float64)
    r_arr = np.array([c["r"] for c in data["cameras"]], dtype=np.float64)
    t_arr = np.array([c["t"] for c in data["cameras"]], dtype=np.float64)
    return k_arr, d_arr, r_arr, t_arr, camera_resolution


def save_scene(out_fpath, k_arr, d_arr, r_arr, t_arr, camera_resolution):
    """utils.save_scene (utils.py:65-81)."""
    cameras = [dict(k=np.asarray(k).tolist(), d=np.asarray(d).tolist(), r=np.asarray(r).tolist(), t=np.asarray(t).tolist())
               for k, d, r, t in zip(k_arr, d_arr, r_arr, t_arr)]
    data = {"created_timestamp": str(datetime.now()), "camera_resolution": list(camera_resolution), "cameras": cameras}
    with open(out_fpath, "w") as f:
        json.dump(data, f)


def load_points(fpath):
    """utils.load_points (utils.py:32-40).  The shipped files name the square size ``board_square_len`` while
    the loader reads ``board_edge_len`` (key drift noted in SURVEY.md f-3): both are accepted."""
    with open(fpath, "r") as f:
        data = json.load(f)
    fnames = list(data["points"].keys())
    points = np.array(list(data["points"].values()), dtype=np.float32)
    board_shape = tuple(data["board_shape"])
    edge = data.get("board_edge_len", data.get("board_square_len"))
    return points, fnames, board_shape, edge, tuple(data["camera_resolution"])


def dlc_wide_to_long(wide_dfs):
    """The reshaping of utils.create_dlc_points_2d_file (utils.py:105-120) on already-loaded DLC tables
    (one per camera; MultiIndex columns scorer / bodyparts / coords): long DataFrame
    [frame, camera, marker, x, y, likelihood], rows ordered by camera, frame, marker (alphabetical)."""
    import pandas as pd
    out = []
    for cam, df in enumerate(wide_dfs):
        d = df.droplevel(0, axis=1) if getattr(df.columns, "nlevels", 1) == 3 else df
        parts = sorted(set(d.columns.get_level_values(0)))
        frames = np.asarray(d.index)
        n, m = len(frames), len(parts)
        block = {"frame": np.repeat(frames, m), "camera": np.full(n * m, cam, dtype=object),
                 "marker": np.tile(np.array(parts, dtype=object), n)}
        for coord in ("x", "y", "likelihood"):
            block[coord] = np.stack([d[(p, coord)].to_numpy(dtype=np.float64) for p in parts], axis=1).reshape(-1)
        out.append(pd.DataFrame(block))
    long_df = pd.concat(out, ignore_index=True)
    return long_df[["frame", "camera", "marker", "x", "y", "likelihood"]]


def read_dlc_table(path):
    """One DeepLabCut table -> (bodyparts, values[N, K, 3] = x, y, likelihood per body part, frame index[N]).

    With pytables installed this is ``pandas.read_hdf`` (what the reference calls, utils.py:108).  Without it the two
    things needed are taken from the file directly, which works for the uncompressed fixed-format tables DeepLabCut writes
    (the shipped data/*.h5): the column MultiIndex is stored as pickle text (``V<name>`` tokens, body parts in column order,
    each followed by x / y / likelihood), the table itself as contiguous records of one int64 row index and 3 K float64."""
    import re
    try:
        import pandas as pd
        df = pd.read_hdf(path)
        d = df.droplevel(0, axis=1) if getattr(df.columns, "nlevels", 1) == 3 else df
        parts = list(dict.fromkeys(d.columns.get_level_values(0)))
        vals = np.stack([np.stack([d[(q, c)].to_numpy(dtype=np.float64) for c in ("x", "y", "likelihood")], 1) for q in parts], 1)
        return parts, vals, np.asarray(d.index, dtype=np.int64)
    except ImportError:
        pass
    with open(path, "rb") as f:
        raw = f.read()
    toks = [t.decode() for t in re.findall(rb"V([A-Za-z0-9_\-]+)\n", raw[:1 << 16])]
    parts = []
    if "x" in toks and toks.index("x") >= 1:        # ... scorer, <first body part>, x, y, likelihood, <next body part>, ... names
        for name in toks[toks.index("x") - 1:]:
            if name == "names":
                break
            if name not in ("x", "y", "likelihood") and name not in parts:
                parts.append(name)
    if not parts:
        raise ValueError(f"{path}: no DeepLabCut column index found (is this a pandas fixed-format table?)")
    rec = 8 + 24 * len(parts)
    dt = np.dtype([("i", "<i8"), ("v", "<f8", (3 * len(parts),))])
    for off in range(0, min(len(raw) - 3 * rec, 1 << 16), 8):
        head = np.frombuffer(raw, dtype="<i8", count=1, offset=off)[0]
        if head != 0:
            continue
        n = (len(raw) - off) // rec
        idx = np.frombuffer(raw, dtype=dt, count=min(n, 64), offset=off)["i"]
        if len(idx) >= 3 and np.array_equal(idx, np.arange(len(idx))):
            tab = np.frombuffer(raw, dtype=dt, count=n, offset=off)
            good = int(np.argmax(tab["i"] != np.arange(n))) if (tab["i"] != np.arange(n)).any() else n
            tab = tab[:good]
            return parts, tab["v"].reshape(good, len(parts), 3).copy(), tab["i"].copy()
    raise ValueError(f"{path}: no uncompressed table of {len(parts)} body parts found")


def create_dlc_points_2d_file(dlc_df_fpaths):
    """utils.create_dlc_points_2d_file (utils.py:105-120): the per-camera DLC tables as ONE long table
    [frame, camera, marker, x, y, likelihood], camera = position of the file in the list."""
    import pandas as pd
    out = []
    for cam, path in enumerate(dlc_df_fpaths):
        parts, vals, frames = read_dlc_table(path)
        order = sorted(range(len(parts)), key=lambda k: parts[k])      # (rows ordered by frame, marker - as the reference's unstack)
        n, m = len(frames), len(parts)
        block = {"frame": np.repeat(frames, m), "camera": np.full(n * m, cam, dtype=object),
                 "marker": np.tile(np.array([parts[k] for k in order], dtype=object), n)}
        for ci, coord in enumerate(("x", "y", "likelihood")):
            block[coord] = vals[:, order, ci].reshape(-1)
        out.append(pd.DataFrame(block))
    long_df = pd.concat(out, ignore_index=True)
    return long_df[["frame", "camera", "marker", "x", "y", "likelihood"]]


def dense_detections(points_2d_df, n_cameras, markers, start_frame=None, end_frame=None):
    """Long detections table -> the boundary's dense det[N, C, L, 3] = (x, y, likelihood) in the given marker
    order, frames start_frame..end_frame-1 (default: the table's range).  Missing rows get likelihood 0."""
    df = points_2d_df
    f = df["frame"].to_numpy().astype(np.int64)
    lo = int(f.min()) if start_frame is None else int(start_frame)
    hi = int(f.max()) + 1 if end_frame is None else int(end_frame)
    idx = {m: i for i, m in enumerate(markers)}
    keep = (f >= lo) & (f < hi) & df["marker"].isin(idx).to_numpy()
    det = np.zeros((hi - lo, n_cameras, len(markers), 3))
    fi = f[keep] - lo
    ci = df["camera"].to_numpy()[keep].astype(np.int64)
    mi = df["marker"][keep].map(idx).to_numpy().astype(np.int64)
    if ci.size and (ci.min() < 0 or ci.max() >= n_cameras):
        raise ValueError("camera index outside the rig")
    for k, col in enumerate(("x", "y", "likelihood")):
        det[fi, ci, mi, k] = df[col].to_numpy(dtype=np.float64)[keep]
    return det, lo


def positions_from_points_3d_df(points_3d_df, markers, start_frame, n_frames):
    """The scatter of all_optimizations.py:932-937: positions[N, len(markers), 3], NaN where not triangulated."""
    positions = np.full((n_frames, len(markers), 3), np.nan)
    idx = {m: i for i, m in enumerate(markers)}
    fr = points_3d_df["frame"].to_numpy().astype(np.int64) - int(start_frame)
    mk = points_3d_df["marker"].map(idx)
    ok = mk.notna().to_numpy() & (fr >= 0) & (fr < n_frames)
    positions[fr[ok], mk[ok].to_numpy().astype(np.int64)] = points_3d_df[["x", "y", "z"]].to_numpy(dtype=np.float64)[ok]
    return positions


def save_fte(results, out_fpath):
    """fte.pickle as written by all_optimizations.py:548-559: dict(positions, x, dx, ddx, start_frame) with
    x/dx/ddx as lists of per-frame lists (what convert_m builds) and positions as a list of (20,3) arrays."""
    data = dict(positions=[np.asarray(p) for p in results["positions"]],
                x=np.asarray(results["x"]).tolist(), dx=np.asarray(results["dx"]).tolist(),
                ddx=np.asarray(results["ddx"]).tolist(), start_frame=int(results.get("start_frame", 0)))
    os.makedirs(os.path.dirname(os.path.abspath(out_fpath)), exist_ok=True)
    with open(out_fpath, "wb") as f:
        pickle.dump(data, f)
    return out_fpath


def load_fte(fpath):
    with open(fpath, "rb") as f:
        d = pickle.load(f)
    return dict(positions=np.asarray(d["positions"]), x=np.asarray(d["x"]), dx=np.asarray(d["dx"]),
                ddx=np.asarray(d["ddx"]), start_frame=d.get("start_frame", 0))
