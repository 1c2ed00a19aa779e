#!/usr/bin/env python3
"""Benchmark of the FTE hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Started under torch.distributed.run (RANK / WORLD_SIZE in the environment) the ranks
are taken as given; started as a plain `python bench.py --gpus N` the script launches its own N ranks (self_launch()) and
forwards rank 0's JSON line.  Either way stdout carries exactly ONE JSON line; everything else (progress, Gloo / c10d /
RCCL banners) goes to stderr.

A "step" is ONE Levenberg-Marquardt iteration over the whole synthetic sequence: damped block system ->
block-cyclic-reduction solve -> trial iterate -> reprojection residuals + analytic Jacobians + normal-
equation assembly at the trial -> accept/reject.  Workload = BASELINE.json's metric configuration:
6 cameras x 20 markers x 10 000 frames, fp64, detections resident in HBM before the timed region.
With N GPUs the SAME 10 000-frame sequence is sharded by frames (strong scaling) with one RCCL all-reduce
on the separator ("temporal-coupling") rows per step.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_FRAMES = 10000
N_CAMS = 6
# Algorithmic flops per frame per LM iteration (dense accounting), by PHASE of the iteration.
ALG_FLOPS = {"assemble": 218.0e3,   # FK 1.1k + projection 7.8k + Jacobians 14.4k + chain 16.7k + weights 10k + J^T W J 156k + J^T r 12k
             "elim": 52.0e3,        # block Cholesky 25^3/3 + three 25x25 triangular solves
             "update": 187.5e3,     # six 2*25^3 trailing GEMM updates of the banded factorisation
             "factor": 239.5e3,     # elim + update: what the chunk sweep does for its interior nodes in one kernel
             "backsub": 10.0e3}     # forward/backward substitution
# every phase of the factorisation is carried by several kernels (the chunk sweep for the interior nodes of the runs; the
# block cyclic reduction of the separator chain with its wide / narrow level kernels); a kernel's algorithmic work = the
# chain nodes it processed x 3 frames x the phase's flops per frame
PHASE_OF = {"elim": "elim", "elim_deep": "elim", "update0": "update", "update": "update", "update_deep": "update",
            "backsub0": "backsub", "backsub": "backsub", "backsub_tail": "backsub", "assemble": "assemble",
            "chunk_sweep": "factor", "chunk_backsub": "backsub", "refine": "backsub"}
ALG_FLOPS_STEP = 4.7e5              # SURVEY 8d total
ALG_BYTES_STEP = 3600.0             # compulsory bytes / frame / iteration (detections 2880 + x in/out 720)
FP64_PEAK_TFLOPS = 78.6             # MI355X FP64 vector = matrix peak (AMD datasheet)
HBM_PEAK_GBS = 8000.0
PROFILE_DIR = "round6"        # profiles/<dir>/pmc_*.json: quoted only when their build_id matches the loaded library


def _log(msg):
    """Progress marker on stderr (stdout carries exactly one JSON line)."""
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def probe_reference_cpu_path(det=None, rig=None, Ts=None, x0_full=None):
    """SURVEY 8(d) "CPU baseline timing" (ii): is the reference's own CPU path (Pyomo + IPOPT, cv2) on this host?  Where
    Pyomo and an ipopt binary exist, the repo's own Pyomo formulation of the FTE NLP (oracle/pyomo_model.py - validated
    against the reference's model text through float stand-ins, tests/test_oracle_golden.py) is built and solved with the
    reference's solver options at N = 100 and N = 1000 frames and the timings reported."""
    import importlib.util
    import shutil
    have = {m: importlib.util.find_spec(m) is not None for m in ("cv2", "pyomo")}
    have["ipopt"] = shutil.which("ipopt") is not None
    missing = [k for k, v in have.items() if not v]
    if not (have["pyomo"] and have["ipopt"]) or det is None:
        return dict(available=False, probe=have,
                    note="reference CPU path unavailable on this host (missing: " + ", ".join(missing) +
                         "): the Pyomo/IPOPT FTE solve cannot be timed here (oracle/pyomo_model.py holds the formulation "
                         "and runs where they exist); the baseline reported is this repo's numpy/scipy oracle of the same "
                         "LM iteration (kind = port)")
    out = dict(available=True, probe=have, runs=[])
    from oracle import pyomo_model
    for n in (100, 1000):
        try:
            out["runs"].append(pyomo_model.time_reference_cpu_path(det, rig, Ts, x0_full, n_frames=n))
        except Exception as exc:                       # pragma: no cover
            out["runs"].append(dict(frames=n, error=f"{type(exc).__name__}: {exc}"))
            break
    return out


def _baseline_fingerprint(det_sample):
    """What the CPU figure was measured ON and WITH, so that a change between rounds is attributable: sha256 of the
    oracle sources in the timed loop, of the sample's bytes, and the host's CPU / numpy / scipy."""
    import hashlib
    import platform
    import scipy
    h = hashlib.sha256()
    for f in ("cpu_baseline.py", "fte.py", "fk.py", "camera.py", "loss.py"):
        h.update(open(os.path.join(ROOT, "oracle", f), "rb").read())
    cpu = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:                                    # pragma: no cover
        pass
    return dict(oracle_sources_sha256=h.hexdigest()[:16],
                sample_sha256=hashlib.sha256(np.ascontiguousarray(det_sample).tobytes()).hexdigest()[:16],
                host_cpu=cpu or platform.processor(), nproc=os.cpu_count(), numpy=np.__version__, scipy=scipy.__version__,
                python=platform.python_version())


def cpu_baseline(det, rig, Ts, x0_full, sample_frames=10000, iters=3):
    """The numpy/scipy oracle's LM iteration timed on the host on a bounded sample: 1 thread on the first
    `sample_frames` frames, then all cores (one single-threaded oracle process per core, each on its own contiguous
    block of the same frames - the frame-sharded form of the same iteration), plus BASELINE config 1 (one frame,
    6-camera adjacent-pair triangulation of 20 keypoints through the oracle's numpy path)."""
    from oracle import camera as ocam
    from oracle import fk as ofk
    from oracle import index_path as oidx
    from oracle.cpu_baseline import lm_iterations as _oracle_lm_iterations
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=1)
    except Exception:                                  # pragma: no cover
        limiter = None
    n = min(sample_frames, det.shape[0])
    xa = x0_full[:n, ofk.ACTIVE]
    hist = []
    dt = _oracle_lm_iterations(det[:n], rig, Ts, xa, iters, history=hist)
    per_iter = dt / (iters + 0.5)      # the initial evaluation is ~half an iteration of work
    # config 1: N = 1
    t0 = time.perf_counter()
    reps = 200
    for _ in range(reps):
        oidx.pairwise_dense(det[:1], 0.5, *rig, ocam.triangulate_points_fisheye)
    c1 = (time.perf_counter() - t0) / reps
    if limiter is not None and hasattr(limiter, "restore_original_limits"):
        limiter.restore_original_limits()
    single = dict(value=n / per_iter, unit="frames/s", cores=1,
                  sample=f"{iters} LM iterations (+ initial evaluation) of the numpy/scipy oracle on the first {n} "
                         f"frames of the same sequence, 1 thread, {dt:.1f} s")
    out = dict(value=single["value"], unit="frames/s", cores=1, kind="port", sample=single["sample"],
               single_thread=single, fingerprint=_baseline_fingerprint(det[:n]),
               reference_cpu_path=probe_reference_cpu_path(det, rig, Ts, x0_full),
               config1_single_frame_triangulation=dict(seconds_per_frame=c1, frames_per_s=1.0 / c1, cores=1,
                                                       what="oracle.index_path.pairwise_dense on 1 frame x 6 cameras x "
                                                            "20 keypoints (5 adjacent pairs, numpy SVD DLT), mean of "
                                                            f"{reps} calls"))
    # the oracle iterations just timed double as a full-size parity check: the HIP solve, from the same start on the same
    # frames, must produce the same trial costs (untimed; the oracle is the checker here, never the thing measured)
    try:
        from acinoset_amd import fte as _fte
        ctx = _fte.FTEContext(det[:n], *rig, Ts, ftol=0.0, xtol=0.0, gtol=0.0)
        ctx.set_x(xa)
        gpu_costs = []
        for _ in hist:
            ctx.step()
            gpu_costs.append(ctx.state()["cost_trial"])
        ctx.close()
        out["parity_at_this_size"] = dict(
            frames=n, iterations=len(hist), trial_cost_oracle=[h["Ft"] for h in hist], trial_cost_gpu=gpu_costs,
            max_rel_diff=max(abs(a - h["Ft"]) / abs(h["Ft"]) for a, h in zip(gpu_costs, hist)),
            what="trial cost of LM iteration 1..k from the same start: HIP solve vs oracle.fte.lm_solve")
    except Exception as exc:                           # pragma: no cover
        out["parity_at_this_size"] = dict(error=f"{type(exc).__name__}: {exc}")
    ncpu = os.cpu_count() or 1
    # How many of the host's hardware threads: 32 single-threaded processes, NOT every core.  On this sample (bounded to seconds
    # of CPU work) more blocks are slower, not faster - 128 processes measured 39.5 k frames/s against 51 k with 32: every block
    # pays the interpreter start, the FTEProblem set-up and the sparse symbolic factorisation, and blocks under ~100 frames are
    # dominated by them.  `cores` says what was used, `nproc` what the host has; the figure is a reported baseline, never credit.
    procs = max(1, min(ncpu, 32, n // 96))
    try:
        import subprocess
        import tempfile
        K, D, R, t = rig
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "sample.npz")
            np.savez(path, det=det[:n], K=K, D=D, R=R, t=t, Ts=Ts, xa=xa)
            bounds = np.linspace(0, n, procs + 1).astype(int)
            env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1", PYTHONPATH=ROOT)
            passes = []
            for _pass in range(3):          # three passes, the MEDIAN is reported (a ~1 s measurement on a busy host wanders by +-15 %)
                t1 = time.perf_counter()
                ps = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_baseline", path, str(a), str(b), str(iters)],
                                       stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, cwd=ROOT, text=True)
                      for a, b in zip(bounds[:-1], bounds[1:])]
                each = []
                for pr in ps:   # plain numpy processes (no torch, no GPU); bounded: a stuck host never hangs the bench
                    o, _ = pr.communicate(timeout=max(60.0, 6.0 * dt))
                    each.append(float(o.strip().splitlines()[-1]))
                passes.append((max(each), time.perf_counter() - t1, each))
        passes.sort(key=lambda v: v[0])
        slow, wall, each = passes[len(passes) // 2]
        out["all_cores"] = dict(value=n / (slow / (iters + 0.5)), unit="frames/s", cores=procs, nproc=ncpu, each_seconds=each,
                                slowest_block_seconds_of_the_three_passes=[v[0] for v in passes],
                                sample=f"the same {n} frames cut into {procs} contiguous blocks, one single-threaded "
                                       f"oracle process per block (python -m oracle.cpu_baseline), {iters} LM iterations "
                                       f"each, run together, three passes: median slowest block {slow:.2f} s (wall incl. interpreter "
                                       f"start {wall:.1f} s); blocks are solved independently (no coupling across block "
                                       "boundaries), i.e. an upper bound for a frame-sharded CPU port")
    except Exception as exc:                           # pragma: no cover
        for pr in locals().get("ps", []):
            if pr.poll() is None:
                pr.kill()
        out["all_cores"] = dict(value=None, error=repr(exc), nproc=ncpu)
    # the reported baseline is the BETTER of the two figures: 32 processes (of `nproc` hardware threads - see above) or one thread
    if out["all_cores"].get("value") and out["all_cores"]["value"] > out["value"]:
        out.update(value=out["all_cores"]["value"], cores=out["all_cores"]["cores"], sample=out["all_cores"]["sample"])
    return out


def _rod(v):
    """Rodrigues vector -> rotation matrix (host helper for the synthetic rig perturbation)."""
    th = float(np.linalg.norm(v))
    if th < 1e-12:
        return np.eye(3)
    k = np.asarray(v) / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


def secondary_metrics(det, rig, Ts):
    """SURVEY 8(d)'s other two quantities, single GPU: config 2 (triangulation + reprojection residual over the
    whole 10 000-frame detection tensor, HBM-resident) and the end-to-end solve wall-clock to a fixed tolerance
    (config 3: 1 000 frames from the reference's nose-line initialisation; and the benchmark sequence itself)."""
    from acinoset_amd import calib, fte, synth
    from acinoset_amd._lib import check, lib, ptr, stream_ptr
    out = {}
    _log("secondary: config 2")
    dev = torch.device("cuda", torch.cuda.current_device())
    d = torch.as_tensor(det, device=dev)
    N, Cn, L, _ = d.shape
    cams = torch.as_tensor(calib.fisheye_records(*rig), device=dev)
    tri = torch.empty((N, L, 3), dtype=torch.float64, device=dev)
    npairs = torch.empty((N, L), dtype=torch.uint8, device=dev)
    mask = torch.empty((N, L), dtype=torch.uint8, device=dev)
    res = torch.empty((N, Cn, L, 2), dtype=torch.float64, device=dev)
    sums = torch.zeros(4, dtype=torch.float64, device=dev)

    def once():     # triangulation + reprojection residual of the triangulated points, one pass over the detections
        check(lib().acino_triangulate_reproject(ptr(d), N, Cn, L, 0.5, ptr(cams), ptr(tri), ptr(npairs), ptr(mask), ptr(res),
                                                ptr(sums), stream_ptr()))
    for _ in range(3):
        once()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        once()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    bytes_per_frame = Cn * L * 3 * 8 + L * 3 * 8 + 2 * L + Cn * L * 2 * 8      # detections in; points, pair counts/masks, residuals out
    out["config2_triangulate_reproject"] = dict(frames=N, ms=ms, frames_per_s=N / (ms * 1e-3),
                                                hbm_gbs=N * bytes_per_frame / (ms * 1e-3) / 1e9,
                                                frac_hbm=N * bytes_per_frame / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                algorithmic_bytes_per_frame=bytes_per_frame)
    # config 3: 1 000-frame straight run through the rig ("trot"), nose-line init (all_optimizations.py:268-277), solve to the default tolerances (ftol = xtol = 1e-10)
    _log("secondary: config 3")
    seq = synth.make_sequence(1000, "trot")
    r3 = (seq["K"], seq["D"], seq["R"], seq["t"])
    det3 = torch.as_tensor(seq["det"], device=dev)
    for tag in ("warm", "timed"):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _res, info = fte.fte_solve(det3[..., :2], det3[..., 2], *r3, Ts=seq["Ts"], max_iter=200, return_numpy=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    out["config3_solve_1k_frames"] = dict(seconds=dt, iterations=info["iter"], status=info["status_name"], cost=info["cost"],
                                          init="nose line", includes="init triangulation, workspace setup, LM loop, outputs")
    # config 5's FTE half on one GPU: 8 clips x 1 000 frames, one HIP stream each (the runtime multiplexes them onto its 4 hardware queues), 20 LM steps
    _log("secondary: config 5 (8 streams)")
    x3 = fte.nose_line_init(det3, *r3, 0.5)[:, fte.ACTIVE]
    streams = [torch.cuda.Stream() for _ in range(8)]
    ctxs = []
    for b in range(8):
        with torch.cuda.stream(streams[b % 8]):
            c = fte.FTEContext(det3, *r3, seq["Ts"], ftol=0.0, xtol=0.0, gtol=0.0, clamp_lambda=True, shared_gpu=True)
            c.enable_graph(True)
            c.set_x(x3)
            for _ in range(3):
                c.step()
            ctxs.append(c)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        for b, c in enumerate(ctxs):
            with torch.cuda.stream(streams[b % 8]):
                c.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for c in ctxs:
        c.close()
    out["config5_batched_fte_8x1k"] = dict(sequences=8, frames_each=1000, steps=20, streams=8, ms_per_round=1e3 * dt / 20,
                                           frames_per_s=8 * 1000 * 20 / dt, precision="f64")
    # ... and at full width: 64 clips x 1 000 frames laid end to end as ONE chain (prior cut at the clip boundaries)
    det64 = det3.repeat(64, 1, 1, 1)
    side = torch.cuda.Stream()
    for prec, key in (("f64", "config5_clips_one_chain_64x1k"), ("bf16", "config5_bf16")):
        _log(f"secondary: config 5, 64 clips as one chain, {prec}")
        c = fte.FTEContext(det64, *r3, seq["Ts"], ftol=0.0, xtol=0.0, gtol=0.0, clamp_lambda=True, clip_len=1000, precision=prec)
        with torch.cuda.stream(side):
            c.enable_graph(True)
            c.set_x(x3.repeat(64, 1) if isinstance(x3, torch.Tensor) else np.tile(x3, (64, 1)))
            for _ in range(3):
                c.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                c.step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            c.profile_begin()
            for _ in range(5):
                c.step()
            prof = c.profile_end()
        c.close()
        out[key] = dict(sequences=64, frames_each=1000, steps=20, ms_per_step=1e3 * dt / 20,
                        frames_per_s=64 * 1000 * 20 / dt,
                        precision="f64" if prec == "f64" else "bf16 residual / Jacobian rows, fp32 accumulation of M_l and v_l, "
                                                              "fp64 from the 6x6 spatial blocks on (acino_fte_params::precision = 1)",
                        assemble_ms_per_step=prof["assemble"]["ms"] / 5,
                        lm="one controller over the sum of the clips' costs")
    # ... and config 5's other half: the six shared extrinsics refined over the marker positions of all 64 clips
    # (sba.bundle_adjust_dense_points_and_extrinsics: points = FTE positions, observations = above-threshold detections)
    try:
        from acinoset_amd import sba as _sba
        _log("secondary: config 5, extrinsic refinement over 64 x 1000 frames")
        rng = np.random.default_rng(7)
        pos64 = torch.as_tensor(seq["pos_true"], device=dev).repeat(64, 1, 1)
        pos64 = pos64 + 0.005 * torch.randn(pos64.shape, dtype=torch.float64, device=dev)
        Rp = np.array([_rod(rng.normal(0, 1, 3) / np.sqrt(3) * np.radians(0.5)) @ r3[2][c] for c in range(len(r3[2]))])
        tp = np.asarray(r3[3], dtype=np.float64).reshape(-1, 3, 1) + rng.normal(0, 1, (len(r3[2]), 3, 1)) / np.sqrt(3) * 1e-2
        c5 = {}
        for prec in ("f64", "bf16"):
            for _rep in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _pts, _r, _t, info = _sba.bundle_adjust_dense_points_and_extrinsics(det64, pos64, r3[0], r3[1], Rp, tp, 0.5, precision=prec,
                                                                                    max_iter=10, ftol=0.0, gtol=0.0)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            it = max(info["iterations"], 1)
            c5[prec] = dict(seconds=dt, iterations=info["iterations"], ms_per_outer_iteration=1e3 * dt / it, n_points=info["n_points"],
                            n_obs=info["n_obs"], observations_per_s=info["n_obs"] * it / dt, rms_before_px=info["rms_before"],
                            rms_after_px=info["rms_after"], cost_initial=info["cost_initial"], cost_final=info["cost_final"],
                            # the fused kernels are VALU-issue bound (profiles/: SIMD issue 64 % busy), not HBM bound: priced on the
                            # fp64 peak with the flops they execute per observation - 48 fp64 matrix instructions (2 048 flop each) and
                            # ~930 vector instructions per batch of ten points (60 observation slots): 1.64 + ~1.0 kflop per observation
                            fp64_flop_fraction_per_iteration=2.6e3 * info["n_obs"] * it / dt / 78.6e12,
                            hbm_fraction_on_bytes_moved=112.0 * info["n_obs"] * it / dt / 8.0e12,
                            includes="observation lists built on the device, workspace allocation, slot table, residuals before / after, "
                                     "10 LM iterations (one 64-byte read-back each)")
        c5["bytes_per_observation"] = dict(algorithmic_survey=200, as_implemented=112, compulsory=40,
                                           note="fused path, per LM iteration: slot table + detections + points read by both passes, V / V^-1 / g_p written "
                                                "once and read once, trial points written and copied (728 MB for 6.5 M observations); the 6 x 3 coupling "
                                                "blocks never leave the chip (the table form of round 4a moved 456 B per observation)")
        c5["all_reduce_doubles_per_iteration_when_sharded"] = (6 * len(r3[2])) ** 2 + 6 * len(r3[2])
        out["config5_sba_extrinsics"] = c5
        del pos64
    except Exception as exc:                           # pragma: no cover
        out["config5_sba_extrinsics"] = dict(error=f"{type(exc).__name__}: {exc}")
    del det64
    # the benchmark sequence under the linear-solver variants: the round-1/2 solver (block cyclic reduction over the whole
    # chain), the chunked sweep with a COMPLETE reduction of the separator chain, and the default (separator chain reduced
    # until the remaining nodes are TRUNC_DISTANCE frames apart, dropped couplings re-introduced by refinement sweeps whose
    # measured contraction bounds the error: state.trunc_eps)
    # the skeleton-driven FTE of src/build.py on REAL detections: 400 frames of the shipped DeepLabCut tables (rows kept under
    # tests/golden/), shipped human skeleton and 2-camera scene, every pose fed by the part of its own name
    try:
        from acinoset_amd import build as _build
        gd = os.path.join(ROOT, "tests", "golden")
        gsk = np.load(os.path.join(gd, "skel_fte_model.npz"))
        sk = json.loads(str(gsk["skeleton_json"]))
        dets = np.load(os.path.join(gd, "human_dlc_slice.npz"))["det"].astype(np.float64)
        tabs = [(list(gsk["parts"]), dets[:, c]) for c in range(dets.shape[1])]
        _log("secondary: skeleton FTE (human, shipped detections)")
        for _rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model, _p3 = _build.build_model(sk, scene=(gsk["K"], gsk["D"], gsk["R"], gsk["t"]), dlc_tables=tabs, n_frames=400,
                                            start_frame=60, pairing="name")
            t1 = time.perf_counter()
            _res, sinfo = _build.solve_model(model, max_iter=300)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        n_w = int((model.weights > 0).sum()) * 2
        out["skeleton_fte_human_real_detections"] = dict(
            frames=400, cams=int(dets.shape[1]), poses=len(model.names), states_per_frame=int(model.P), active_states=int(len(model.active)),
            seconds_build=t1 - t0, seconds_solve=t2 - t1, iterations=sinfo["iterations"], status=sinfo["status_name"],
            cost_initial=sinfo["cost_initial"], cost_final=sinfo["cost_final"],
            mean_abs_residual_px=sinfo["cost_final"] * _build.R_MEAS / max(n_w, 1), ms_per_iteration=1e3 * (t2 - t1) / max(sinfo["iterations"], 1),
            data="rows 60..459 of the reference's data/Ex1Cam{3,4}...h5, skeletons/new_human.pickle, data/4_cam_scene_static_sba.json",
            note="L1 measurement loss, constant model weight 0.002 (src/build.py:186-191, 299); csrc/skel_fte.hip")
        # ... the same 400-frame clip with every compute unit: alternating Schwarz over half-overlapping windows (build.solve_model_parallel)
        _log("secondary: skeleton FTE, one clip on every CU (alternating Schwarz over windows)")
        for _rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _pres, pinfo = _build.solve_model_parallel(model)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
        out["skeleton_fte_human_real_detections"]["alternating_schwarz_over_windows"] = dict(
            seconds_solve=t1 - t0, outer_iterations=pinfo["outer_iterations"], windows=pinfo["windows"], status=pinfo["status_name"],
            cost_final=pinfo["cost_final"], gnorm_inf=pinfo["gnorm_inf"], cost_final_single_workgroup=sinfo["cost_final"],
            note="whole-clip cost and projected gradient of the outer iteration's end point; the single-workgroup solve above is the exact "
                 "banded factorisation of the same clip")
        # ... and the WHOLE shipped video (6 240 frames): 78 windows of the reference's 100 frames (build.py:131-133), overlapping
        # by 20, as ONE batched solve - one workgroup and one device-side controller per window (acino_skel_fte_solve_batch)
        full = np.load(os.path.join(gd, "human_dlc_full.npz"))
        ftabs = [(list(full["parts"]), full[f"det{c}"].astype(np.float64)) for c in range(2)]
        _log("secondary: skeleton FTE, the whole shipped video as a batch of windows")
        for _rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _vres, vinfos, vstarts = _build.solve_video(sk, scene=(gsk["K"], gsk["D"], gsk["R"], gsk["t"]), dlc_tables=ftabs, first_frame=0,
                                                        last_frame=6239, window=100, overlap=20, pairing="name", max_iter=1500)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
        vits = [i["iterations"] for i in vinfos]
        out["skeleton_fte_whole_shipped_video"] = dict(
            frames=6240, windows=len(vstarts), window_frames=100, overlap=20, seconds_incl_model_building=t1 - t0,
            frames_per_s=6240 / (t1 - t0), iterations_min=min(vits), iterations_max=max(vits), iterations_sum=int(sum(vits)),
            windows_not_converged=int(sum(i["status_name"] not in ("ftol", "xtol", "gtol") for i in vinfos)),
            data="every row of the reference's data/Ex1Cam{3,4}...h5 (tests/golden/human_dlc_full.npz)",
            note="every kernel but the factorisation runs over the frames of all windows; the factorisation is one workgroup per window "
                 "(34 us per frame and iteration); finished windows are skipped")
    except Exception as exc:                           # pragma: no cover
        out["skeleton_fte_human_real_detections"] = dict(error=f"{type(exc).__name__}: {exc}")
    x10 = fte.triangulation_init(d, *rig, 0.5)[:, fte.ACTIVE]
    inc = {}
    for tag, kw in (("whole_chain_bcr_complete", dict(chunk_nodes=-1, bcr_levels=0)), ("chunked_complete", dict(bcr_levels=0)),
                    ("chunked_truncated_refined_default", {})):
        _log(f"secondary: solver variant {tag}")
        c = fte.FTEContext(d, *rig, Ts, ftol=0.0, xtol=0.0, gtol=0.0, clamp_lambda=True, **kw)
        with torch.cuda.stream(side):
            c.enable_graph(True)
            c.set_x(x10)
            for _ in range(3):
                c.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                c.step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            stK = c.state()
        inc[tag] = dict(ms_per_step=1e3 * dt / 20, frames_per_s=N * 20 / dt, cost_after_23_steps=stK["cost"], accepted=stK["accepted"],
                        trunc_eps=stK["trunc_eps"], status=stK["status_name"], plan=fte.solver_plan(c.params),
                        bcr_levels=int(c.params.bcr_levels), refine_sweeps=int(c.params.refine_sweeps))
        c.close()
    out["solver_variants_10k"] = inc
    _log("secondary: end-to-end solve of the 10 000-frame sequence")
    for key, kw in (("solve_10k_frames", {}), ("solve_10k_frames_complete_reduction", {"bcr_levels": 0}),
                    ("solve_10k_frames_whole_chain_bcr", {"bcr_levels": 0, "chunk_nodes": -1})):
        for _rep in range(2):                  # (second run: graph capture, allocator and first-touch costs paid)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _res, info = fte.fte_solve(d[..., :2], d[..., 2], *rig, Ts=Ts, max_iter=200, init="triangulation", return_numpy=False,
                                       reuse_context=True, **kw)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        out[key] = dict(seconds=dt, iterations=info["iter"], status=info["status_name"], cost=info["cost"],
                        init="per-frame triangulation", frames_per_s_end_to_end=N / dt, trunc_eps=info.get("trunc_eps", 0.0),
                        bcr_levels=info.get("bcr_levels"),
                        includes="init triangulation, detections copied into the kept context (reuse_context: workspace, constants and "
                                 "captured graph stay from the first solve of this shape), LM loop to ftol = xtol = 1e-10, outputs")
    fte.clear_context_cache()
    return out


def _shard_model_figures(world, halo, frames=N_FRAMES):
    """What profiles/<PROFILE_DIR>/shard_model.txt (scripts/shard_model.py: real pinned / windowed contexts of every rank stepped
    in lock step on ONE GPU, collectives emulated and not timed) predicts for this rank count and sequence length: the per-rank
    ms of both drivers before collectives, to hold the measured line against."""
    import re
    out = dict(source=f"profiles/{PROFILE_DIR}/shard_model.txt", frames=int(frames),
               note="per-rank ms per iteration on one GPU, collectives not included")
    pre = rf"frames {int(frames)} \| "
    try:
        for line in open(os.path.join(ROOT, "profiles", PROFILE_DIR, "shard_model.txt")):
            m = re.match(pre + r"world 1: .* single-GPU step ([0-9.]+) ms", line)
            if m:
                out["single_gpu_ms"] = float(m.group(1))
            m = re.match(pre + rf"world {world}: .*chunked sweep.* = ([0-9.]+) ms per iteration", line)
            if m:
                out["separators_ms"] = float(m.group(1))
            m = re.match(pre + rf"windows: world {world}, halo (\d+): .*default.*: step ([0-9.]+) ms", line)
            if m and (int(m.group(1)) == int(halo) or "windows_ms" not in out):
                out["windows_ms"], out["windows_halo"] = float(m.group(2)), int(m.group(1))
    except OSError as exc:
        out["error"] = repr(exc)
    return out


_JSON_OUT = None


def _claim_stdout():
    """stdout carries exactly one JSON line: keep a private handle on the real stdout and point fd 1 at stderr, so that
    whatever else writes to fd 1 - Python prints, the C++ banners of Gloo / c10d / RCCL - cannot land in front of it."""
    global _JSON_OUT
    if _JSON_OUT is None:
        sys.stdout.flush()
        _JSON_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    return _JSON_OUT


def _emit(obj):
    out = _claim_stdout()
    out.write(json.dumps(obj) + "\n")
    out.flush()


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _json_line_of(text):
    """The last line of `text` that parses as a JSON object with a "metric" key (None if there is none); the other lines."""
    found, rest = None, []
    for line in text.splitlines():
        obj = None
        if line.startswith("{"):
            try:
                obj = json.loads(line)
            except ValueError:
                obj = None
        if isinstance(obj, dict) and "metric" in obj:
            if found is not None:
                rest.append(found)
            found = line
        else:
            rest.append(line)
    return found, rest


def self_launch(n_ranks, argv, extra_env=None):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script through torch.distributed.run on
    127.0.0.1 (a free port), pass rank 0's JSON line through to stdout, everything else to stderr.  If the sharded run
    dies (it has never met a multi-GPU node: a failing collective must not leave the caller without a number) the
    collective-free placement is run instead - N replicas, weak scaling - and the line says so (`fallback`)."""
    import subprocess
    me = os.path.abspath(__file__)

    def run(more):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), me] + list(argv) + more
        _log("self-launch: " + " ".join(cmd))
        env = dict(os.environ, ACINO_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "8")
        env.update(extra_env or {})
        pr = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, cwd=ROOT, text=True)
        line, rest = _json_line_of(pr.stdout)
        for other in rest:
            print(other, file=sys.stderr)
        return pr.returncode, line

    rc, line = run([])
    if line is None and "--shard" not in argv and "--dry-run" not in argv:
        _log(f"self-launch: the sharded run ended with rc {rc} and no JSON line; running {n_ranks} replicas instead")
        rc2, line = run(["--shard", "replicas"])
        if line is not None:
            obj = json.loads(line)
            obj["fallback"] = f"the frame-sharded run ended with rc {rc} before printing; this line is --shard replicas"
            line, rc = json.dumps(obj), rc2
    if line is None:
        raise SystemExit(rc or 1)
    _emit(json.loads(line))
    return rc


def _dry_run(rank, world, backend, dev_seen):
    """--dry-run: everything around the measurement - launch, rendezvous, one all-reduce, one all-gather, the single JSON
    line - without touching a GPU (what the CPU test of the launch path runs)."""
    import torch.distributed as dist
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t)
        seen = [None] * world
        dist.all_gather_object(seen, dev_seen)
    else:
        seen = [dev_seen]
    if rank == 0:
        _emit({"metric": "FTE frames/sec (residual+Jac+LM step), 6-cam x 20-joint", "value": None, "unit": "frames/s",
               "n_gpus": world, "dry_run": True, "world": world, "backend": backend, "devices_seen": seen,
               "all_reduce_check": float(t.item()) == world * (world + 1) / 2,
               "self_launched": os.environ.get("ACINO_BENCH_SELF_LAUNCHED") == "1"})
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=N_FRAMES)
    ap.add_argument("--repeats", type=int, default=5, help="timed blocks of --steps steps; the headline is their median")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the config-2 / solve-to-tolerance extras")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly in the timed region")
    ap.add_argument("--dry-run", action="store_true", help="launch, rendezvous, one collective and the JSON line only (no GPU work)")
    ap.add_argument("--shard", choices=("both", "windows", "separators", "replicas"), default="both",
                    help="N > 1: 'both' (default) = the headline is the separator system - the north star's RCCL all-reduce on the "
                         "temporal-coupling rows, exact step - and the overlapping-window driver is timed in the same run "
                         "(drivers.windows); or one of: overlapping windows (two all-gathers per step, step exact to the decay over --halo frames) or "
                         "exact separator system (one all-reduce + two all-gathers); 'replicas' = every rank solves its own copy of the "
                         "whole sequence (weak scaling, no data-path collective)")
    ap.add_argument("--halo", type=int, default=192, help="--shard windows: frames of overlap on either side")
    ap.add_argument("--bcr-levels", type=int, default=None,
                    help="reduction levels before the remaining nodes are solved on their own + refined (default: the library's "
                         "automatic choice, FTEContext.TRUNC_DISTANCE; 0 = complete reduction; single GPU only)")
    ap.add_argument("--refine-sweeps", type=int, default=None, help="block-Jacobi sweeps over the dropped couplings")
    ap.add_argument("--chunk-nodes", type=int, default=0, help="nodes per run of the chunked solver (0 auto, -1 whole-chain BCR)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N` (how the driver starts N = 1): be the launcher
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE = {world}: start one rank per GPU "
                         "(torch.distributed.run --nproc-per-node N), or run `python bench.py --gpus N` and let it launch them")
    _claim_stdout()
    # ACINO_DIST_BACKEND=gloo ACINO_FORCE_DEVICE=0 lets several ranks share one GPU (functional check of the
    # multi-process path on a single-GPU box); the driver's runs use RCCL, one rank per GPU.
    backend = os.environ.get("ACINO_DIST_BACKEND", "nccl")
    if args.dry_run and not torch.cuda.is_available():
        backend = "gloo"
    dev_index = int(os.environ.get("ACINO_FORCE_DEVICE", local_rank))
    import torch.distributed as dist
    if args.dry_run:
        if world > 1:
            dist.init_process_group(backend="gloo")
        return _dry_run(rank, world, backend, dev_index)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            assert dev_index < torch.cuda.device_count(), f"rank {rank}: device {dev_index} of {torch.cuda.device_count()}"
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=backend)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    from acinoset_amd import dist as adist
    from acinoset_amd import fte, synth

    # ---- synthetic workload (every rank builds the same sequence, then keeps its shard) ----
    if rank == 0:
        _log(f"building the {args.frames}-frame sequence")
    seq = synth.make_sequence(args.frames, "loop")
    det = seq["det"]
    rig = (seq["K"], seq["D"], seq["R"], seq["t"])
    x0_full = fte.triangulation_init(det, *rig, 0.5)
    windows = world > 1 and args.shard == "windows"

    def time_other_driver(kind):
        """K timed steps of the driver that is NOT the headline of this run (same sequence, same start, same barrier /
        synchronise bracket, max over ranks), with HIP-event times of its collectives."""
        cm = dict(ftol=0.0, xtol=0.0, gtol=0.0, clamp_lambda=True, shared_gpu="ACINO_FORCE_DEVICE" in os.environ)
        if kind == "windows":
            sv, (w0_, w1_, a0, a1) = adist.make_windowed(torch.as_tensor(det), *rig, seq["Ts"], rank, world, halo=args.halo, **cm)
            xl = torch.as_tensor(x0_full[w0_:w1_][:, fte.ACTIVE])
        else:
            sv, (a0, a1) = adist.make_sharded(torch.as_tensor(det), *rig, seq["Ts"], rank, world, **cm)
            xl = torch.as_tensor(x0_full[a0:a1][:, fte.ACTIVE])
        sd = torch.cuda.Stream()
        sd.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(sd):
            if not args.no_graph:
                (sv if kind == "windows" else sv.b.ctx).enable_graph(True)
            sv.set_x(xl)
            for _ in range(args.warmup):
                sv.step()
            dist.barrier()
            torch.cuda.synchronize()
            t0_ = time.perf_counter()
            for _ in range(args.steps):
                sv.step()
            dist.barrier()
            torch.cuda.synchronize()
            dt_ = time.perf_counter() - t0_
            sv.collect_timing(True)
            for _ in range(args.steps):
                sv.step()
            torch.cuda.synchronize()
            coll_ = sv.timing_summary()
            sv.collect_timing(False)
            st_ = (sv.ctx if kind == "windows" else sv.b.ctx).state()
        tm = torch.tensor([dt_], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dt_ = float(tm.item())
        (sv.ctx if kind == "windows" else sv.b.ctx).close()
        return dict(ms_per_step=1e3 * dt_ / args.steps, value=args.frames * args.steps / dt_, unit="frames/s", collectives_per_step=coll_,
                    lm_state={k: st_[k] for k in ("cost", "iter", "accepted", "status_name", "trunc_eps")},
                    what=("overlapping windows (halo %d frames): 2 all-gathers / step, step exact to the decay over the halo" % args.halo)
                    if kind == "windows" else "separator system: 1 all-reduce on the separator rows + 2 all-gathers / step, exact step")

    def time_replicas():
        """Weak scaling over sequences (config 5's placement): every rank runs the SINGLE-GPU solver on its own copy of the
        whole sequence, no collective in the data path; same barrier / synchronise bracket, max over ranks."""
        cx = fte.FTEContext(torch.as_tensor(det), *rig, seq["Ts"], ftol=0.0, xtol=0.0, gtol=0.0, clamp_lambda=True)
        sd = torch.cuda.Stream()
        sd.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(sd):
            if not args.no_graph:
                cx.enable_graph(True)
            cx.set_x(x0_full[:, fte.ACTIVE])
            for _ in range(args.warmup):
                cx.step()
            dist.barrier()
            torch.cuda.synchronize()
            t0_ = time.perf_counter()
            for _ in range(args.steps):
                cx.step()
            dist.barrier()
            torch.cuda.synchronize()
            dt_ = time.perf_counter() - t0_
        tm = torch.tensor([dt_], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dt_ = float(tm.item())
        cx.close()
        return dict(ms_per_step=1e3 * dt_ / args.steps, value=world * args.frames * args.steps / dt_, unit="frames/s", scaling="weak",
                    what=f"{world} independent {args.frames}-frame sequences, one per GPU, single-GPU solver, no data-path collective")

    # N > 1, --shard both: BOTH sharded drivers are timed the same way first; the faster one becomes the headline (and is the
    # one profiled below), the other one, the replicas (weak scaling) and the one-GPU model stand beside it
    pre = {}
    replicas = None
    if world > 1 and args.shard == "both":
        for kind in ("separators", "windows"):
            if rank == 0:
                _log(f"driver {kind}")
            pre[kind] = time_other_driver(kind)
        windows = pre["windows"]["value"] > pre["separators"]["value"]
    if world > 1 and args.shard in ("both", "replicas"):
        if rank == 0:
            _log("replicas")
        replicas = time_replicas()
    dev_seen = dict(rank=rank, device=dev_index, name=torch.cuda.get_device_name(dev_index),
                    uuid=str(getattr(torch.cuda.get_device_properties(dev_index), "uuid", "")))
    devices_seen = [dev_seen]
    if world > 1:
        devices_seen = [None] * world
        dist.all_gather_object(devices_seen, dev_seen)
    if world > 1 and args.shard == "replicas":
        if rank == 0:
            _emit({"metric": "FTE frames/sec (residual+Jac+LM step), 6-cam x 20-joint", "value": replicas["value"], "unit": "frames/s",
                   "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": replicas["ms_per_step"],
                   "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                   "config": {"workload": f"FTE LM iteration, {N_CAMS} cam x 20 markers x {args.frames} frames PER GPU (replicas), loop "
                                          "trajectory, seed 20210313", "frames": args.frames, "cams": N_CAMS, "markers": 20, "states": 25,
                              "parallelism": f"{world} replicas, no data-path collective"},
                   "world": world, "backend": backend, "devices_seen": devices_seen,
                   "every_rank_its_own_device": len({d["device"] for d in devices_seen}) == world,
                   "drivers": {"headline": "replicas", "replicas": replicas}, "roofline": None, "cpu_baseline": None})
        dist.barrier()
        dist.destroy_process_group()
        return
    common = dict(ftol=0.0, xtol=0.0, gtol=0.0, clamp_lambda=True,                           # never stops: every step is full work
                  shared_gpu="ACINO_FORCE_DEVICE" in os.environ)                             # (ranks sharing one GPU: functional runs only)
    solver_kw = {}
    if world == 1:
        solver_kw["chunk_nodes"] = args.chunk_nodes
        if args.bcr_levels is not None:
            solver_kw["bcr_levels"] = args.bcr_levels
        if args.refine_sweeps is not None:
            solver_kw["refine_sweeps"] = args.refine_sweeps
    if windows:
        solver, (w0, w1, n0, n1) = adist.make_windowed(torch.as_tensor(det), *rig, seq["Ts"], rank, world, halo=args.halo, **common)
        x0_local = torch.as_tensor(x0_full[w0:w1][:, fte.ACTIVE])
        ctx = solver.ctx
    else:
        solver, (n0, n1) = adist.make_sharded(torch.as_tensor(det), *rig, seq["Ts"], rank, world, **common,
                                              **solver_kw)
        x0_local = torch.as_tensor(x0_full[n0:n1][:, fte.ACTIVE])
        ctx = solver.b.ctx

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    use_graph = world == 1 and not args.no_graph
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        if not args.no_graph:
            # no host sync inside a step: capture once, replay (N > 1: the phases between the collectives)
            (solver if windows else ctx).enable_graph(True)
        solver.set_x(x0_local)
        for _ in range(args.warmup):
            solver.step()
        sync()
        if rank == 0:
            _log("timed region")
        # ---- timed region: blocks of exactly K steps, each bracketed by barrier + synchronise; the headline is the MEDIAN
        #      block (a block is 11 ms at K = 20: a single one wanders by +-2 % from box to box and run to run) -----------
        block_dt = []
        for _rep in range(args.repeats):
            t0 = time.perf_counter()
            for _ in range(args.steps):
                solver.step()
            sync()
            block_dt.append(time.perf_counter() - t0)
        # ---- the same K steps again, launched eagerly with HIP events around every kernel (events cannot be
        #      recorded inside a captured graph); kernel durations do not depend on how they were launched ----
        solver.set_x(x0_local)            # same start, same LM trajectory as the timed region
        for _ in range(args.warmup):
            solver.step()
        sync()
        if windows:
            solver.enable_graph(False)    # (events cannot be recorded inside the driver's captured phases either)
        ctx.profile_begin()
        solver.collect_timing(True)       # N > 1: HIP events around the collectives of every step
        t1 = time.perf_counter()
        for _ in range(args.steps):
            solver.step()
        sync()
        dt_eager = time.perf_counter() - t1
        prof = ctx.profile_end()
        coll = solver.timing_summary()
        solver.collect_timing(False)
    tmax = torch.tensor(block_dt, dtype=torch.float64, device="cuda" if backend == "nccl" or world == 1 else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)          # every block: the slowest rank's time
    block_dt = [float(v) for v in tmax.tolist()]
    dt = float(np.median(block_dt))
    st = ctx.state()
    assert st["status"] != 7, "the timed steps were refused by the truncation check: not a valid measurement"

    if rank == 0:
        n_loc = n1 - n0
        ms_step = 1e3 * dt / args.steps
        value = args.frames * args.steps / dt
        dom = max(PHASE_OF, key=lambda k: prof[k]["ms"])            # the kernel with the largest summed launch time
        launches = max(prof[dom]["launches"], 1)
        avg_ms = prof[dom]["ms"] / launches
        phase = PHASE_OF[dom]
        total_units = (n_loc if phase == "assemble" else (n_loc + 2) // 3) * args.steps
        share = prof[dom]["units"] / max(total_units, 1)             # of the chain's nodes (frames for assemble)
        flops_per_launch = ALG_FLOPS[phase] * n_loc * args.steps * share / launches
        achieved = flops_per_launch / (max(avg_ms, 1e-9) * 1e-3) / 1e12
        gpu_ms_step = sum(v["ms"] for v in prof.values()) / args.steps
        kname = fte.FTEContext.PROF_KERNELS[dom]
        # HBM bytes / launch and matrix-core counters of the dominant kernel: from the committed PMC passes under
        # profiles/ (counters need their own rocprofv3 runs) - quoted ONLY when that summary was measured on this very
        # binary (build_id = hash over all HIP sources, embedded in the .so); otherwise null, never a stale number
        traffic, mfma, pmc_note = None, None, None
        per_kernel = None
        try:
            from acinoset_amd import _lib
            pdir = os.path.join(ROOT, "profiles", PROFILE_DIR)
            tj = json.load(open(os.path.join(pdir, "pmc_traffic.json")))
            mj = json.load(open(os.path.join(pdir, "pmc_mfma_lds.json")))
            bid = _lib.built_id()
            if tj.get("build_id") != bid or mj.get("build_id") != bid:
                pmc_note = (f"profiles/{PROFILE_DIR} was measured on build {tj.get('build_id')}, this library is {bid}: "
                            "traffic / pmc not quoted")
            elif args.frames == N_FRAMES and world == 1:
                traffic = tj["kernels"]["acino::" + kname]["bytes_per_launch"]
                mk = mj["kernels"]["acino::" + kname]
                mfma = dict(mfma_f64_flops_executed_per_launch=mk["mfma_f64_flops_per_launch"],
                            mfma_util_percent=mk["mfma_util_percent"], lds_bank_conflict_rate=mk["lds_bank_conflict_rate"])
                # per kernel: executed / algorithmic flops and measured / algorithmic bytes (regressions show up here)
                per_kernel = {}
                for cls, kn in fte.FTEContext.PROF_KERNELS.items():
                    full = "acino::" + kn
                    if full not in tj["kernels"] or prof[cls]["launches"] == 0:
                        continue
                    lps = prof[cls]["launches"] / args.steps
                    ent = dict(launches_per_step=lps, hbm_bytes_per_step=tj["kernels"][full]["bytes_per_launch"] * lps)
                    if full in mj["kernels"]:
                        ent["mfma_flops_executed_per_step"] = mj["kernels"][full]["mfma_f64_flops_per_launch"] * lps
                        ent["mfma_util_percent"] = mj["kernels"][full]["mfma_util_percent"]
                    per_kernel[cls] = ent
                hbm_step = sum(v["hbm_bytes_per_step"] for v in per_kernel.values())
                exe_step = sum(v.get("mfma_flops_executed_per_step", 0.0) for v in per_kernel.values())
                per_kernel["_step"] = dict(hbm_bytes=hbm_step, hbm_over_algorithmic=hbm_step / (ALG_BYTES_STEP * args.frames),
                                           mfma_flops_executed=exe_step,
                                           mfma_executed_over_algorithmic_solve=exe_step / ((ALG_FLOPS["elim"] + ALG_FLOPS["update"] +
                                                                                             ALG_FLOPS["backsub"]) * args.frames))
        except Exception as exc:
            pmc_note = f"no PMC summary quoted ({exc!r})"
        out = {
            "metric": "FTE frames/sec (residual+Jac+LM step), 6-cam x 20-joint",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "timing": {"what": f"median of {len(block_dt)} blocks of {args.steps} steps, each bracketed by barrier + synchronise",
                       "ms_per_step_blocks": [1e3 * v / args.steps for v in block_dt],
                       "ms_per_step_min": 1e3 * min(block_dt) / args.steps, "ms_per_step_max": 1e3 * max(block_dt) / args.steps},
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"FTE LM iteration, {N_CAMS} cam x 20 markers x {args.frames} frames (BASELINE configs[3] shape"
                                   f"{'' if world > 1 else ' on one GPU'}), loop trajectory, seed 20210313",
                       "linear_solver": (dict(plan=fte.solver_plan(ctx.params), bcr_levels=int(ctx.params.bcr_levels),
                                              refine_sweeps=int(ctx.params.refine_sweeps), trunc_tol=float(ctx.params.trunc_tol),
                                              note="chunked sweep; separator chain reduced bcr_levels levels, the remaining nodes solved on "
                                                   "their own and corrected by refine_sweeps block-Jacobi sweeps over the dropped "
                                                   "couplings; the error bound measured from the sweeps' contraction (lm_state.trunc_eps) "
                                                   "is checked against trunc_tol INSIDE every timed step (status 7 otherwise); "
                                                   "bcr_levels = 0: complete reduction") if world == 1 else None),
                       "frames": args.frames, "cams": N_CAMS, "markers": 20, "states": 25,
                       "parallelism": (f"frames sharded x{world}, " + (f"overlapping windows (halo {args.halo} frames), 2 all-gathers / step"
                                                                        if windows else "separator system, 1 all-reduce + 2 all-gathers / step"))
                       if world > 1 else "single GPU"},
            "roofline": {"bound": "mfma", "kernel": kname,
                         "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic,
                         "launches_per_step": launches / args.steps, "avg_launch_ms": avg_ms,
                         "algorithmic_flops_per_launch": flops_per_launch,
                         "share_of_phase": {"phase": phase, "nodes_or_frames": share},
                         "pmc": mfma, "pmc_note": pmc_note, "per_kernel": per_kernel,
                         "step": {"achieved_tflops": ALG_FLOPS_STEP * n_loc / (ms_step * 1e-3) / 1e12,
                                  "frac_fp64": ALG_FLOPS_STEP * n_loc / (ms_step * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                                  "achieved_hbm_gbs": ALG_BYTES_STEP * n_loc / (ms_step * 1e-3) / 1e9,
                                  "frac_hbm": ALG_BYTES_STEP * n_loc / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "gpu_kernel_ms_per_step": gpu_ms_step,
                                  "launch": "eager" if args.no_graph else ("hipGraph replay" if world == 1 else ("2 graph phases + 2 all-gathers" if windows else "4 hipGraph phases + 3 collectives")),
                                  "ms_per_step_eager_with_events": 1e3 * dt_eager / args.steps},
                         "kernel_ms_per_step": {k: v["ms"] / args.steps for k, v in prof.items()}},
            "lm_state": {k: st[k] for k in ("cost", "iter", "accepted", "lam", "status_name", "trunc_eps")},
        }
        if world > 1:
            out["world"], out["backend"], out["devices_seen"] = world, backend, devices_seen
            out["every_rank_its_own_device"] = len({d["device"] for d in devices_seen}) == world
            out["drivers"] = {"headline": "windows" if windows else "separators",
                              "chosen": "the faster of the two sharded drivers in their own timed runs" if pre else f"--shard {args.shard}",
                              "separators": pre.get("separators"), "windows": pre.get("windows"),
                              "replicas": replicas, "modelled_on_one_gpu": _shard_model_figures(world, args.halo, args.frames)}
            out["collectives"] = {"backend": "RCCL (torch.distributed nccl)" if backend == "nccl" else backend,
                                  "per_step": coll,
                                  "payload_bytes": ({"all_gather_edge_slabs": world * 2 * (args.halo + 3) * 25 * 8, "all_gather_scalars": world * 8 * 8}
                                                    if windows else
                                                    {"all_reduce_separators": (world - 1) * (2 * 80 * 80 + 80) * 8,
                                                     "all_gather_edges": world * 6 * 25 * 8, "all_gather_scalars": world * 8 * 8}),
                                  "note": "rank 0's HIP-event time around each collective on the launch stream (includes "
                                          "waiting for the slowest rank); kernel_ms_per_step above is rank 0's shard"}
        if world == 1 and not args.no_secondary:
            out["secondary"] = secondary_metrics(det, rig, seq["Ts"])
            # time to solution, first class beside the per-iteration headline (what an inexact step must answer to)
            out["end_to_end"] = {k: out["secondary"][k] for k in ("solve_10k_frames", "config3_solve_1k_frames",
                                                                   "skeleton_fte_human_real_detections") if k in out["secondary"]}
        if not args.no_cpu_baseline and world == 1:
            _log("cpu baseline (oracle on the host cores)")
            out["cpu_baseline"] = cpu_baseline(det, rig, seq["Ts"], x0_full)
            _log("done")
        elif world > 1:
            out["cpu_baseline"] = None
        if world == 1:
            out["world"], out["backend"], out["devices_seen"] = 1, None, devices_seen
        _emit(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
