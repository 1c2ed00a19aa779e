"""CPU: the C-ABI library loads and exports every symbol include/acinoset_hip.h declares; host-side
argument handling; the product path fails loudly without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import __graft_entry__ as ge
from acinoset_amd import _lib, calib, fte

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    ge.build()
    return _lib.lib()


def test_every_declared_symbol_is_exported_and_bound(lib):
    hdr = open(os.path.join(ROOT, "include", "acinoset_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(acino_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 40
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert set(_lib.SIGNATURES) <= declared
    assert lib.acino_abi_version() == _lib.ABI_VERSION == 3     # (3: acino_skel_fte_* - include/acinoset_hip.h)
    assert lib.acino_sizeof_fte_params() == C.sizeof(_lib.FteParams)
    assert lib.acino_sizeof_fte_state() == C.sizeof(_lib.FteState)


def test_argument_validation_happens_before_any_device_call(lib):
    null = C.c_void_p(0)
    rc = lib.acino_triangulate_pairs(null, 10, 9, 20, 0.5, null, null, null, null, null)
    assert rc == -1 and b"n_cams" in lib.acino_last_error_string()
    rc = lib.acino_project_fisheye(null, -1, null, null, null)
    assert rc == -1
    assert lib.acino_project_fisheye(null, 0, C.c_void_p(8), null, null) == 0      # empty input is a no-op
    assert lib.acino_selftest_mfma(null, null, 6, null, null) == -1
    p = fte.make_params(100, 6, 1 / 120)
    nbytes = lib.acino_fte_workspace_bytes(C.byref(p))
    assert nbytes > 34 * 80 * 80 * 8                           # chunked solver: ONE 80x80 matrix per super-block (G_k) ...
    p_bcr = fte.make_params(100, 6, 1 / 120, chunk_nodes=-1)
    assert lib.acino_fte_workspace_bytes(C.byref(p_bcr)) > 34 * 3 * 80 * 80 * 8 and lib.acino_fte_workspace_bytes(C.byref(p_bcr)) > nbytes   # ... whole-chain reduction: five
    p_bad = fte.make_params(100, 6, 1 / 120, pin_left=True, n_global=200, n_offset=50)   # offset not a multiple of 3
    h = C.c_void_p()
    assert lib.acino_fte_create(C.byref(h), C.byref(p_bad), C.c_void_p(256), C.c_void_p(256), C.c_void_p(256), nbytes, null) == -1
    with pytest.raises(ValueError):
        _lib.check(-1)
    with pytest.raises(RuntimeError):
        _lib.check(-2)


def test_params_mirror_reference_constants():
    p = fte.make_params(10, 6, 1 / 120)
    lo, hi = fte.bounds45()
    assert np.isfinite(lo).sum() == 21
    assert abs(p.inv_r_meas - 0.2) < 1e-15 and (p.redesc_a, p.redesc_b, p.redesc_c) == (3.0, 10.0, 20.0)
    assert abs(p.q_w[0] - (1 / 16) * 120 ** 4) < 1e-3                                  # x: sigma 4 -> 1/16 / Ts^4
    assert [p.lo[i] for i in range(3)] == [-np.inf] * 3 and p.hi[20] == np.inf        # x, y, z and psi_0 are free
    assert abs(p.lo[13] + np.pi) < 1e-15 and p.hi[13] == 0.0                          # theta_7 in [-pi, 0]
    with pytest.raises(ValueError):
        fte.make_params(10, 6, 1 / 120, Q=np.ones(45))
    assert len(fte.MARKERS) == 20 and fte.ACTIVE.tolist()[:6] == [0, 1, 2, 3, 4, 6]


def test_camera_records_and_shapes():
    k = np.array([[1000.0, 2.0, 640], [0, 1001.0, 360], [0, 0, 1]])
    rec = calib.fisheye_record(k, np.array([[0.1], [0.2], [0.3], [0.4]]), np.eye(3), np.array([[1.0], [2.0], [3.0]]))
    assert rec.shape == (24,) and rec[20] == 2.0 / 1000.0 and rec[4:8].tolist() == [0.1, 0.2, 0.3, 0.4]
    rec2 = calib.fisheye_record(k, np.zeros(4), np.array([0.0, 0.0, np.pi / 2]), np.zeros(3))     # rvec accepted
    assert np.allclose(rec2[8:17].reshape(3, 3), [[0, -1, 0], [1, 0, 0], [0, 0, 1]], atol=1e-15)
    with pytest.raises(ValueError):
        calib.fisheye_record(k, np.zeros(5), np.eye(3), np.zeros(3))
    pin = calib.pinhole_record(k, np.arange(8) * 0.01, np.eye(3), np.zeros(3))
    assert pin.shape == (32,) and pin[4:12].tolist() == (np.arange(8) * 0.01).tolist()
    with pytest.raises(ValueError):
        calib.pinhole_record(k, np.zeros(6), np.eye(3), np.zeros(3))


def test_dataframe_to_dense_layout():
    import pandas as pd
    rows = [dict(frame=f, camera=c, marker=m, x=f + c, y=10 * f, likelihood=0.9)
            for f in (3, 5) for c in (0, 2) for m in ("nose", "l_eye")]
    det, frames, markers = calib.dataframe_to_dense(pd.DataFrame(rows), 3)
    assert det.shape == (2, 3, 2, 3) and frames.tolist() == [3, 5] and markers.tolist() == ["l_eye", "nose"]
    assert det[1, 2, 1].tolist() == [7.0, 50.0, 0.9]
    assert np.isneginf(det[:, 1, :, 2]).all()                  # camera 1 never saw anything


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the behaviour WITHOUT a GPU")
def test_product_path_fails_loudly_without_gpu():
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        calib.project_points_fisheye(np.zeros((1, 3)), np.eye(3), np.zeros(4), np.eye(3), np.zeros(3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        fte.cheetah_fk(np.zeros((1, 45)))


def test_no_product_module_imports_the_oracle():
    pkg = os.path.join(ROOT, "acinoset_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{fn} imports the oracle"


def test_window_plan_and_params_host_logic():
    """Host-side logic of round 2 that needs no GPU: the overlapping-window plan, the new acino_fte_params fields."""
    from acinoset_amd import dist as adist
    from acinoset_amd import fte
    plan, halo = adist.window_plan(10000, 4, 190)
    assert halo == 192 and [p[2:] for p in plan] == adist.shard_plan(10000, 4)
    assert plan[0][:2] == (0, 2499 + 192) and plan[3][:2] == (7500 - 192, 10000)
    for (w0, w1, n0, n1) in plan:
        assert w0 % 3 == 0 and w0 <= n0 < n1 <= w1 and (w1 == 10000 or (w1 - w0) % 3 == 0)
        assert (n0 - w0 in (0, halo)) and (w1 - n1 in (0, halo))
    with pytest.raises(ValueError):
        adist.window_plan(600, 4, 192)                     # shards shorter than the halo
    assert adist.window_plan(500, 1, 96)[0] == [(0, 500, 0, 500)]
    p = fte.make_params(100, 6, 1 / 120, precision="bf16", bcr_levels=3, trunc_tol=1e-9, own_first=12, own_count=60)
    assert (p.precision, p.bcr_levels, p.trunc_tol, p.own_first, p.own_count) == (1, 3, 1e-9, 12, 60)
    assert fte.make_params(100, 6, 1 / 120).precision == 0 and fte.PRECISIONS["bf16_residuals"] == 2
    with pytest.raises(ValueError):
        fte.make_params(100, 6, 1 / 120, precision="fp8")
    # the solver layout (acino_fte_plan is host logic: no device call) and the automatic truncation depth derived from it
    mk = lambda n, **kw: fte.make_params(n, 6, 1 / 120, **kw)
    assert fte.solver_plan(mk(10000)) == dict(m=14, n_chunks=239, n_sep=238, levels=8)
    assert fte.solver_plan(mk(10000, chunk_nodes=-1)) == dict(m=0, n_chunks=0, n_sep=0, levels=12)
    assert fte.solver_plan(mk(9999, pin_right=True, n_global=20000)) == dict(m=14, n_chunks=238, n_sep=238, levels=8)   # sharded contexts: chunked too (round 4); the pin joins the separators and, 3 333 = 238 x 14 + 1, the last run takes 15 nodes
    assert fte.solver_plan(mk(9, chunk_nodes=5)) == dict(m=5, n_chunks=1, n_sep=0, levels=0)
    assert fte.auto_bcr_levels(mk(10000), 160) == 2 and fte.auto_bcr_levels(mk(10000), 384) == 4      # 42 * 2^K frames
    assert fte.auto_bcr_levels(mk(10000, chunk_nodes=-1), 384) == 7 and fte.auto_bcr_levels(mk(700, chunk_nodes=-1), 384) == 0
    assert fte.auto_bcr_levels(mk(300), 160) == 0                                        # chain too short to truncate
    assert fte.auto_bcr_levels(mk(10000), fte.FTEContext.TRUNC_DISTANCE) == 1 and fte.FTEContext.REFINE_SWEEPS == 7
    # escalation after a refused step: one level more, or as many as the refused bound calls for (couplings square per level)
    import types
    nxt = lambda levels, bound=None, tol=1e-12: fte.FTEContext._next_levels(
        types.SimpleNamespace(params=mk(10000, bcr_levels=levels, trunc_tol=tol, refine_sweeps=7)), bound)
    assert nxt(1) == 2 and nxt(1, 1.0) == 2 and nxt(1, 0.0) == 2            # no size to extrapolate from
    assert nxt(1, 1.3e-8) == 2                                              # 1.3e-8 squared is under 1e-13
    assert nxt(1, 3.3e-6) == 3 and nxt(2, 2e-7) == 3                        # needs the fourth power / the square
    assert nxt(1, 0.05) == 5 and nxt(1, 0.3) == 6 and nxt(6) == 7 and nxt(7) == 0 and nxt(4, 0.3) == 0 and nxt(0, 1e-3) == 0    # (the chain has 8 levels: beyond 7, the complete reduction)


def test_build_id_matches_sources_and_cpu_baseline_worker(tmp_path):
    """The shared object carries the hash of the sources it was built from (build() and the PMC summaries key on it);
    the CPU-baseline worker of bench.py runs as a plain numpy process."""
    import subprocess
    import sys
    import numpy as np
    from acinoset_amd import _lib
    from oracle import fk as ofk
    from oracle import synth as osynth
    assert _lib.built_id() == _lib.source_hash() == _lib.lib().acino_build_id().decode()
    seq = osynth.make_sequence(12, "sprint")
    path = str(tmp_path / "s.npz")
    np.savez(path, det=seq["det"], K=seq["K"], D=seq["D"], R=seq["R"], t=seq["t"], Ts=seq["Ts"], xa=seq["q_true"][:, ofk.ACTIVE])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "oracle.cpu_baseline", path, "0", "12", "1"], cwd=root, capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == 0 and float(out.stdout.strip()) > 0.0
    import bench
    probe = bench.probe_reference_cpu_path()
    assert set(probe["probe"]) == {"cv2", "pyomo", "ipopt"} and ("unavailable" in probe["note"]) == (not probe["available"])


def test_bench_baseline_bookkeeping():
    """bench.py's CPU-baseline hygiene (host only): the fingerprint names the oracle sources / sample / host it was measured
    with and is stable; the reference-path probe reports what is missing here and would time the repo's own Pyomo
    formulation where Pyomo + IPOPT exist."""
    import bench
    sample = np.arange(4 * 6 * 20 * 3, dtype=np.float64).reshape(4, 6, 20, 3)
    f1, f2 = bench._baseline_fingerprint(sample), bench._baseline_fingerprint(sample.copy())
    assert f1 == f2 and len(f1["oracle_sources_sha256"]) == 16 and f1["nproc"] == os.cpu_count()
    assert bench._baseline_fingerprint(sample + 1.0)["sample_sha256"] != f1["sample_sha256"]
    probe = bench.probe_reference_cpu_path()
    assert set(probe["probe"]) == {"cv2", "pyomo", "ipopt"}
    if not (probe["probe"]["pyomo"] and probe["probe"]["ipopt"]):
        assert probe["available"] is False and "oracle/pyomo_model.py" in probe["note"]


def test_chunk_plan_of_sharded_ranks():
    """Pinned (sharded) contexts take the chunked solver since round 4: the pins join the separator chain, and a right pin is
    never a run of its own (acino_fte_plan is host logic: no GPU needed)."""
    for n, pl, pr in ((5000, True, True), (1251, True, False), (1248, False, True), (30, True, True), (6, True, True), (39, False, True)):
        if pr:
            n -= n % 3
        p = fte.make_params(n, 6, 1 / 120, n_global=3 * n + 300, n_offset=(n // 3) * 3 if pl else 0, pin_left=pl, pin_right=pr)
        plan = fte.solver_plan(p)
        nodes = (n + 2) // 3                         # swept nodes (the right pin is the last of them)
        assert plan["m"] >= 2 and plan["n_sep"] == plan["n_chunks"] - 1 + int(pl) + int(pr), plan
        last = nodes - (plan["n_chunks"] - 1) * plan["m"]
        assert 1 <= last <= plan["m"] + 1 and (not pr or last >= 2 or plan["n_chunks"] == 1), (n, plan, last)
    whole = fte.solver_plan(fte.make_params(999, 6, 1 / 120, n_global=5000, n_offset=999, pin_left=True, pin_right=True, chunk_nodes=-1))
    assert whole["n_chunks"] == 0


def test_bench_launches_its_own_ranks_and_prints_one_json_line():
    """`python bench.py --gpus N` started WITHOUT torch.distributed.run (how the driver starts N = 1) must launch its own N ranks and
    leave exactly one JSON line on stdout - Gloo / c10d banners and progress belong to stderr.  --dry-run stops before the GPU work:
    launch, rendezvous on 127.0.0.1, one all-reduce, one all-gather of the ranks' devices, the line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    pr = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--dry-run"],
                        capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert pr.returncode == 0, pr.stderr[-2000:]
    lines = [ln for ln in pr.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, pr.stdout
    obj = json.loads(lines[0])
    assert obj["n_gpus"] == 2 and obj["world"] == 2 and obj["self_launched"] and obj["all_reduce_check"]
    assert obj["devices_seen"] == [0, 1]                   # every rank reports the device its LOCAL_RANK names
    assert "self-launch" in pr.stderr
    # a launcher's ranks with the wrong --gpus are refused with a message, not silently run
    env2 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    pr2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                         timeout=120, env=env2, cwd=root)
    assert pr2.returncode != 0 and "WORLD_SIZE" in pr2.stderr


def test_bench_json_line_filter():
    import bench
    line, rest = bench._json_line_of('[Gloo] Rank 0 is connected\n{"not": "it"}\n{"metric": "m", "value": 1}\ntrailing')
    assert line == '{"metric": "m", "value": 1}' and rest == ['[Gloo] Rank 0 is connected', '{"not": "it"}', 'trailing']
    assert bench._json_line_of("nothing here")[0] is None


def test_level_split_tables_cover_every_tile_once():
    """Fused narrow levels of the separator reduction (csrc/seplevel.hip): for every T the T workgroups of a node together
    compute each of the 55 output tiles (15 lower tiles of P_l, 15 of P_r, 25 of X) exactly once, every workgroup computes the
    strips of W its tiles read, every strip of W_l / W_r has exactly one workgroup that stores it, and the shares are balanced."""
    import ctypes as C
    want = {5 * a + b for a in range(5) for b in range(a + 1)}
    want |= {25 + t for t in want} | {50 + t for t in range(25)}
    for T in range(1, 17):
        buf = (C.c_int32 * (T * 64))()
        assert _lib.lib().acino_debug_level_split(T, buf) == 0
        rows = np.frombuffer(buf, dtype=np.int32).reshape(T, 64)
        seen, stored, counts = [], 0, []
        for g in range(T):
            strips, stores, nt = int(rows[g, 0]), int(rows[g, 1]), int(rows[g, 2])
            tiles = [int(c) for c in rows[g, 3:3 + nt]]
            assert all(c == -1 for c in rows[g, 3 + nt:]) and tiles == sorted(tiles)
            need = 0
            for c in tiles:
                kind, a, b = c // 25, (c % 25) // 5, c % 5
                need |= ((1 << a) | (1 << b)) if kind == 0 else (((32 << a) | (32 << b)) if kind == 1 else ((32 << a) | (1 << b)))
            assert strips == need and stores & ~strips == 0 and stored & stores == 0, (T, g)
            stored |= stores
            seen += tiles
            counts.append(nt)
        assert sorted(seen) == sorted(want), T
        assert stored == 0x3FF, T
        assert max(counts) <= -(-55 // T) + 8, (T, counts)          # no workgroup left with most of the node
    assert _lib.lib().acino_debug_level_split(17, (C.c_int32 * 64)()) != 0
