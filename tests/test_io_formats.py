"""CPU: input/output formats of the path (scene JSON, DLC tables, dense detections, result pickle) against the
reference's own utils.py output (golden) and its shipped data files (copied as fixtures)."""
import json
import os

import numpy as np
import pandas as pd

from acinoset_amd import io as aio


def test_scene_roundtrip_and_reference_file(golden_dir, tmp_path):
    K, D, R, t, res = aio.load_scene(os.path.join(golden_dir, "dummy_scene.json"))
    assert K.shape == (6, 3, 3) and D.shape == (6, 4, 1) and R.shape == (6, 3, 3) and t.shape == (6, 3, 1)
    assert res == (2704, 1520) and abs(K[0, 0, 0] - 1239.734301643185) < 1e-12
    out = tmp_path / "scene.json"
    aio.save_scene(out, K, D, R, t, res)
    K2, D2, R2, t2, res2 = aio.load_scene(out)
    assert np.array_equal(K, K2) and np.array_equal(D, D2) and np.array_equal(R, R2) and np.array_equal(t, t2)
    assert set(json.load(open(out))) == {"created_timestamp", "camera_resolution", "cameras"}   # utils.py:76-80


def test_dlc_wide_to_long_matches_reference(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "dlc_tables.json")))
    parts = g["parts"]
    cols = pd.MultiIndex.from_product([["DLC_resnet50_test"], parts, ["x", "y", "likelihood"]],
                                      names=["scorer", "bodyparts", "coords"])
    wide = [pd.DataFrame(np.array(w), columns=cols) for w in g["wide"]]
    long_df = aio.dlc_wide_to_long(wide)
    assert list(long_df.columns) == g["long_columns"]
    assert [int(v) for v in long_df["frame"]] == g["long_frame"]
    assert [int(v) for v in long_df["camera"]] == g["long_camera"]
    assert list(long_df["marker"]) == g["long_marker"]
    assert np.array_equal(long_df[["x", "y", "likelihood"]].to_numpy(dtype=float), np.array(g["long_xyl"]))
    det, lo = aio.dense_detections(long_df, 3, parts)
    assert det.shape == (6, 3, 5, 3) and lo == 0
    w1 = np.array(g["wide"][1]).reshape(6, 5, 3)
    assert np.array_equal(det[:, 1], w1)
    det2, lo2 = aio.dense_detections(long_df[long_df["likelihood"] > 0.5], 3, parts, start_frame=2, end_frame=5)
    assert det2.shape == (3, 3, 5, 3) and lo2 == 2 and ((det2[..., 2] == 0) | (det2[..., 2] > 0.5)).all()


def test_tri_scatter_and_fte_pickle(tmp_path):
    df = pd.DataFrame({"frame": [10.0, 10.0, 12.0], "marker": ["nose", "spine", "nose"],
                       "x": [1.0, 2.0, 3.0], "y": [4.0, 5.0, 6.0], "z": [7.0, 8.0, 9.0]})
    pos = aio.positions_from_points_3d_df(df, ["l_eye", "nose", "spine"], start_frame=10, n_frames=3)
    assert pos.shape == (3, 3, 3) and pos[0, 1].tolist() == [1.0, 4.0, 7.0] and pos[2, 1].tolist() == [3.0, 6.0, 9.0]
    assert np.isnan(pos[1]).all() and np.isnan(pos[0, 0]).all()
    res = dict(positions=np.zeros((4, 20, 3)), x=np.ones((4, 25)), dx=np.zeros((4, 25)), ddx=np.zeros((4, 25)), start_frame=7)
    path = aio.save_fte(res, str(tmp_path / "fte" / "fte.pickle"))
    import pickle
    raw = pickle.load(open(path, "rb"))
    assert set(raw) == {"positions", "x", "dx", "ddx", "start_frame"} and isinstance(raw["x"], list) and len(raw["x"][0]) == 25
    back = aio.load_fte(path)
    assert back["x"].shape == (4, 25) and back["positions"].shape == (4, 20, 3) and back["start_frame"] == 7


def test_video_windows_cover_every_frame_once_deep_enough():
    """build.video_windows (host logic of build.solve_video): windows of the reference's length, consecutive ones overlapping,
    the last one pulled back to end on the last frame; every frame lies in a window, and - away from the two ends of the
    video - at least overlap / 2 frames deep in one of them (the stitching takes a frame from the window in which it lies
    deepest)."""
    from acinoset_amd import build
    for f0, f1, w, ov in ((0, 6239, 100, 20), (60, 459, 100, 20), (0, 99, 100, 20), (5, 250, 100, 0), (0, 1000, 64, 63)):
        st = build.video_windows(f0, f1, w, ov)
        assert st[0] == f0 and st[-1] + w - 1 == f1 and all(b > a for a, b in zip(st, st[1:]))
        assert all(b - a <= w - ov for a, b in zip(st, st[1:]))
        depth = np.full(f1 - f0 + 1, -1)
        for s in st:
            d = np.minimum(np.arange(w), w - 1 - np.arange(w))
            sl = slice(s - f0, s - f0 + w)
            depth[sl] = np.maximum(depth[sl], d)
        assert (depth >= 0).all()
        inner = depth[w // 2: len(depth) - w // 2]
        assert inner.size == 0 or inner.min() >= ov // 2 - 1
    assert build.video_windows(0, 6239, 100, 20)[:3] == [0, 80, 160] and len(build.video_windows(0, 6239, 100, 20)) == 78
    for bad in ((0, 50, 100, 20), (0, 500, 100, 100)):
        try:
            build.video_windows(*bad)
        except ValueError:
            continue
        raise AssertionError(bad)
