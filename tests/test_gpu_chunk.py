"""GPU (-m gpu): the chunked substructuring solver (csrc/chunk.hip) against the block cyclic reduction over the whole
chain (chunk_nodes = -1, the round-1/2 solver) and against the oracle's banded Cholesky.  Both solve the same damped
Gauss-Newton system exactly, so every LM step must produce the same trial iterate (to rounding) whatever the run length,
including runs of one interior node, chains shorter than a run, partial last nodes, clips and windows."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fk as ofk
from oracle import fte as ofte


@pytest.fixture(scope="module")
def mods(gpu_lib):
    from acinoset_amd import calib, fte, synth
    return calib, fte, synth


def _start(fte, seq, n, seed, sigma=0.02):
    x0 = np.zeros((n, 45))
    x0[:, fte.ACTIVE] = seq["q_true"][:, fte.ACTIVE] + np.random.default_rng(seed).normal(0, sigma, (n, 25))
    lo, hi = fte.bounds45()
    return np.clip(x0, lo, hi)[:, fte.ACTIVE]


def _trial(ctx):
    """The trial iterate of the last step (frames x 25), through the C ABI."""
    from acinoset_amd._lib import check, lib, ptr, stream_ptr
    buf = torch.empty((ctx.N, 25), dtype=torch.float64, device=ctx.device)
    st = ctx.state()
    # after an accepted step the trial became the current iterate
    which = 0 if st["last_accept"] else 1
    check(lib().acino_fte_copy_frames(ctx._h, which, 0, 0, ctx.N, ptr(buf), stream_ptr()))
    return buf.cpu().numpy()


def _walk(fte, det, rig, Ts, xa, steps, **kw):
    ctx = fte.FTEContext(det, *rig, Ts, ftol=0.0, xtol=0.0, gtol=0.0, clamp_lambda=True, **kw)
    ctx.set_x(xa)
    out = []
    for _ in range(steps):
        ctx.step()
        st = ctx.state()
        out.append((st["cost_trial"], st["last_accept"], st["pred"], st["status"], _trial(ctx)))
    ctx.close()
    return out


@pytest.mark.parametrize("n", [3, 4, 7, 10, 24, 59, 100, 301])
@pytest.mark.parametrize("m", [0, 2, 3, 5])
def test_chunk_sweep_equals_block_cyclic_reduction(mods, n, m):
    calib, fte, synth = mods
    seq = synth.make_sequence(n, "sprint" if n < 200 else "loop")
    rig = (seq["K"], seq["D"], seq["R"], seq["t"])
    xa = _start(fte, seq, n, n)
    ref = _walk(fte, seq["det"], rig, seq["Ts"], xa, 5, chunk_nodes=-1)
    got = _walk(fte, seq["det"], rig, seq["Ts"], xa, 5, chunk_nodes=m)
    for it, (r, g) in enumerate(zip(ref, got)):
        assert g[3] == 0 and r[3] == 0, (it, g[3], r[3])
        assert g[1] == r[1], f"iteration {it}: accept decisions differ"
        # (clips of 3 .. 10 frames are badly conditioned - DESIGN section 5: differences of 1e-5 between two exact solvers
        #  after a few iterations at lambda ~ 1e-6 - the decisions stay identical)
        tol = 1e-9 if n >= 24 else 1e-6
        assert abs(g[0] - r[0]) <= 0.1 * tol * abs(r[0]), (it, g[0], r[0])
        assert abs(g[2] - r[2]) <= 10 * tol * abs(r[2]) + 1e-14, (it, g[2], r[2])
        assert np.abs(g[4] - r[4]).max() < tol, (it, float(np.abs(g[4] - r[4]).max()))


@pytest.mark.parametrize("n,m", [(30, 2), (30, 4), (95, 3), (95, 0)])
def test_chunk_sweep_first_step_equals_oracle_banded_solve(mods, n, m):
    """One LM step from a perturbed start: the trial iterate against the oracle's banded Cholesky (scipy) of the same
    damped system - the solver checked against something that is not a GPU kernel."""
    calib, fte, synth = mods
    seq = synth.make_sequence(n, "sprint")
    rig = (seq["K"], seq["D"], seq["R"], seq["t"])
    xa = _start(fte, seq, n, 7 * n)
    got = _walk(fte, seq["det"], rig, seq["Ts"], xa, 1, chunk_nodes=m, lam0=1e-3)[0]
    prob = ofte.FTEProblem(seq["det"][..., :2], seq["det"][..., 2], *rig, seq["Ts"])
    x = np.clip(xa, prob.lo, prob.hi)
    F, g, H, _nb = prob.evaluate(x)
    fixed = prob.active_set(x, g, H)
    delta, _diag = prob.solve_banded(H, g, 1e-3, fixed)
    xt = np.clip(x + delta, prob.lo, prob.hi)
    assert np.abs(got[4] - xt).max() < 1e-9 * max(1.0, float(np.abs(delta).max())), float(np.abs(got[4] - xt).max())
    assert abs(got[0] - prob.evaluate(xt)[0]) < 1e-9 * abs(F)


def test_chunk_sweep_with_clips_and_windows(mods):
    """Clip boundaries inside runs and inside nodes; a window with an owned range (sums over owned frames only)."""
    calib, fte, synth = mods
    seq = synth.make_sequence(40, "sprint")
    rig = (seq["K"], seq["D"], seq["R"], seq["t"])
    det = np.concatenate([seq["det"]] * 5, 0)
    xa = np.concatenate([_start(fte, seq, 40, s) for s in range(5)], 0)
    for m in (2, 3, 0):
        ref = _walk(fte, det, rig, seq["Ts"], xa, 4, chunk_nodes=-1, clip_len=40)
        got = _walk(fte, det, rig, seq["Ts"], xa, 4, chunk_nodes=m, clip_len=40)
        for r, g in zip(ref, got):
            assert g[1] == r[1] and abs(g[0] - r[0]) <= 1e-10 * abs(r[0]) and np.abs(g[4] - r[4]).max() < 1e-9
    seq = synth.make_sequence(200, "sprint")
    xa = _start(fte, seq, 200, 5)
    kw = dict(n_global=1000, n_offset=300, own_first=50, own_count=100)
    ref = _walk(fte, seq["det"], rig, seq["Ts"], xa, 4, chunk_nodes=-1, **kw)
    got = _walk(fte, seq["det"], rig, seq["Ts"], xa, 4, chunk_nodes=4, **kw)
    for r, g in zip(ref, got):
        assert abs(g[0] - r[0]) <= 1e-10 * abs(r[0]) and np.abs(g[4] - r[4]).max() < 1e-9


def test_chunk_sweep_at_bench_size(mods):
    """10 000 frames, automatic run length: same LM trajectory as the whole-chain reduction over 6 iterations."""
    calib, fte, synth = mods
    seq = synth.make_sequence(10000, "loop")
    rig = (seq["K"], seq["D"], seq["R"], seq["t"])
    xa = fte.triangulation_init(seq["det"], *rig, 0.5)[:, fte.ACTIVE]
    ref = _walk(fte, seq["det"], rig, seq["Ts"], xa, 6, chunk_nodes=-1)
    got = _walk(fte, seq["det"], rig, seq["Ts"], xa, 6, chunk_nodes=0)
    for it, (r, g) in enumerate(zip(ref, got)):
        assert g[3] == 0 and g[1] == r[1]
        assert abs(g[0] - r[0]) <= 1e-10 * abs(r[0]), (it, g[0], r[0])
        assert np.abs(g[4] - r[4]).max() < 1e-8, (it, float(np.abs(g[4] - r[4]).max()))


def test_chunk_sweep_random_chain_and_run_lengths(mods):
    """25 seeded (frames, run length) pairs - chains of 1 ... 133 nodes, runs of 2 ... 14 nodes, partial last nodes, last
    runs of every length: the step of ONE linear solve, node by node, against the whole-chain reduction; the same solve
    repeated three times must not change a bit (the sweep's waves synchronise through LDS counters: a missing fence would
    show up as run-to-run differences).  scripts/chunk_stress.py is the long form."""
    import ctypes as C
    calib, fte, synth = mods
    from acinoset_amd._lib import check, lib, ptr, stream_ptr
    rng = np.random.default_rng(2024)
    for _ in range(25):
        n = int(rng.integers(3, 400))
        m = int(rng.choice([0, 2, 3, 4, 5, 7, 9, 14]))
        seq = synth.make_sequence(n, "sprint" if n < 200 else "loop")
        rig = (seq["K"], seq["D"], seq["R"], seq["t"])
        xa = _start(fte, seq, n, n + m)
        T = (n + 2) // 3
        got = {}
        for tag, cn in (("bcr", -1), ("chunk", m)):
            ctx = fte.FTEContext(seq["det"], *rig, seq["Ts"], ftol=0.0, xtol=0.0, gtol=0.0, chunk_nodes=cn, bcr_levels=0)
            ctx.set_x(xa)
            runs = []
            for _rep in range(3 if tag == "chunk" else 1):
                check(lib().acino_fte_reduce_local(ctx._h, stream_ptr()))
                check(lib().acino_fte_backsub_local(ctx._h, C.c_void_p(0), 0, 1, stream_ptr()))
                buf = torch.zeros(T * 80, dtype=torch.float64, device="cuda")
                check(lib().acino_fte_debug_read(ctx._h, 0, ptr(buf), T * 80, stream_ptr()))
                torch.cuda.synchronize()
                runs.append(buf.cpu().numpy().reshape(T, 80)[:, :75])
            ctx.close()
            assert all(np.array_equal(r, runs[0]) for r in runs[1:]), (n, m, "not reproducible")
            got[tag] = runs[0]
        d = np.abs(got["bcr"] - got["chunk"]).max() / np.abs(got["bcr"]).max()
        assert d < 1e-8, (n, m, d)


@pytest.mark.parametrize("n,steps", [(10000, 2000), (3331, 2000), (1005, 2500), (999, 2500), (190, 3000)])
def test_soak_thousands_of_steps_bit_reproducible_under_foreign_load(mods, n, steps):
    """Soak (round-3 verdict, robustness): >= 2 000 LM steps per size as hipGraph replays - sweep with its LDS-counter
    sub-barriers, the one-launch separator tail with its cross-workgroup hand-offs, the back-substitution - while a
    FOREIGN stream keeps launching LDS-heavy filler kernels that take CUs away at random moments.  Sizes: the benchmark
    (239 runs of 14), a last run of 3 interior nodes (1 005 frames: 335 nodes in runs of 4 - the shape of the one
    placement-dependent crash recorded in NOTES_perf.md), a last run of one node, short chains.  The whole trajectory is run
    TWICE from the same start: state and iterate must agree to the last bit (a lost hand-off, a missed fence or a race
    shows up as a run-to-run difference long before it shows up as a wrong answer), every step must be verified
    (status 0, finite cost), and the first pass must not have raised a numeric / time-out flag."""
    from acinoset_amd._lib import check, lib
    calib, fte, synth = mods
    seq = synth.make_sequence(n, "loop")
    rig = (seq["K"], seq["D"], seq["R"], seq["t"])
    det = torch.as_tensor(seq["det"], device="cuda")
    x0 = fte.triangulation_init(det, *rig, 0.5)[:, fte.ACTIVE]
    main, foreign = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for rep in range(2):
        with torch.cuda.stream(main):
            ctx = fte.FTEContext(det, *rig, seq["Ts"], ftol=0.0, xtol=0.0, gtol=0.0, clamp_lambda=True)
            ctx.enable_graph(True)
            ctx.set_x(x0)
        for k in range(steps):
            with torch.cuda.stream(main):
                ctx.step()
            if k % 16 == 0:                              # foreign load: 48 workgroups x 64 KB of LDS, a few us each
                check(lib().acino_debug_poison_lds(48, 4, C_void(foreign.cuda_stream)))
            if k % 500 == 499:
                with torch.cuda.stream(main):
                    st = ctx.state()
                assert st["status"] == 0 and np.isfinite(st["cost"]), (n, k, st)
        with torch.cuda.stream(main):
            st = ctx.state()
            x = ctx.result()[0].cpu().numpy()
            ctx.close()
        torch.cuda.synchronize()
        outs.append((st, x))
    (s0, x0_), (s1, x1_) = outs
    assert s0["iter"] == steps and s0["status"] == 0
    assert s0["cost"] == s1["cost"] and s0["lam"] == s1["lam"] and s0["accepted"] == s1["accepted"], (s0, s1)
    assert np.array_equal(x0_, x1_), float(np.abs(x0_ - x1_).max())
    print(f"soak n={n}: {steps} steps x 2, cost {s0['cost']:.12f}, accepted {s0['accepted']}, bit-identical")


def C_void(v):
    import ctypes
    return ctypes.c_void_p(v)


def test_separator_tail_falls_back_to_the_per_level_kernels_on_a_device_that_cannot_hold_it(mods, monkeypatch):
    """Round-4 advisor: the one-launch back-substitution of the separator chain (k_sep_tail) needs every one of its workgroups
    resident, so a context is only given it where occupancy x compute units of the device hold them all; elsewhere (a CU mask, a
    partitioned GPU - here: ACINO_SEP_TAIL_CAPACITY pretends a device with room for one workgroup) the per-level kernels
    run.  Same LM path either way: decisions equal, trial iterates to rounding."""
    calib, fte, synth = mods
    n = 3331
    seq = synth.make_sequence(n, "loop")
    rig = (seq["K"], seq["D"], seq["R"], seq["t"])
    xa = _start(fte, seq, n, 11)
    ref = _walk(fte, seq["det"], rig, seq["Ts"], xa, 4)
    monkeypatch.setenv("ACINO_SEP_TAIL_CAPACITY", "1")
    got = _walk(fte, seq["det"], rig, seq["Ts"], xa, 4)
    # ... and it IS the other set of kernels: per-level back-substitution launches, one refinement launch per sweep instead of one
    ctx = fte.FTEContext(seq["det"], *rig, seq["Ts"], ftol=0.0, xtol=0.0, gtol=0.0, clamp_lambda=True)
    ctx.set_x(xa)
    ctx.step()
    ctx.profile_begin()
    ctx.step()
    launches = {k: v["launches"] for k, v in ctx.profile_end().items()}
    ctx.close()
    assert launches["backsub"] >= 1 and launches["refine"] == fte.FTEContext.REFINE_SWEEPS, launches
    for it, (r, g) in enumerate(zip(ref, got)):
        assert g[3] == 0 and r[3] == 0 and g[1] == r[1], (it, g[3], r[3])
        assert abs(g[0] - r[0]) <= 1e-11 * abs(r[0]), (it, g[0], r[0])
        assert np.abs(g[4] - r[4]).max() < 1e-10, (it, float(np.abs(g[4] - r[4]).max()))


@pytest.mark.parametrize("n,kw", [(999, {}), (999, dict(bcr_levels=0)), (3331, {}), (400, dict(shared_gpu=True)), (190, {}),
                                   (2500, dict(chunk_nodes=4))])
def test_fused_narrow_levels_equal_the_per_phase_kernels(mods, monkeypatch, n, kw):
    """Round 6: a narrow level of the separator reduction is ONE launch (csrc/seplevel.hip: elimination and Schur products, the
    products kept as per-side running sums, the isolated level factored inside k_sep_tail).  ACINO_NO_FUSED_LEVELS=1 selects the
    per-phase kernels of csrc/bcr.hip for the same chain: both are exact eliminations of the same system, so the LM walk - trial
    cost, accept / reject, predicted reduction, trial iterate - must agree to rounding.  Cases: truncated + refined (tail kernel),
    complete reduction, a shared GPU (per-level back-substitution: the sums are folded for it), a chain too short to truncate,
    many short runs (wide first level on the old kernels, narrow ones fused)."""
    calib, fte, synth = mods
    seq = synth.make_sequence(n, "loop")
    rig = (seq["K"], seq["D"], seq["R"], seq["t"])
    xa = _start(fte, seq, n, 7 * n)
    monkeypatch.setenv("ACINO_NO_FUSED_LEVELS", "1")
    ref = _walk(fte, seq["det"], rig, seq["Ts"], xa, 6, **kw)
    monkeypatch.delenv("ACINO_NO_FUSED_LEVELS")
    got = _walk(fte, seq["det"], rig, seq["Ts"], xa, 6, **kw)
    for it, (r, g) in enumerate(zip(ref, got)):
        assert r[3] == g[3] == 0, (it, r[3], g[3])
        assert r[1] == g[1], it
        assert abs(r[0] - g[0]) <= 1e-9 * abs(r[0]), (it, r[0], g[0])
        assert abs(r[2] - g[2]) <= 1e-7 * abs(r[2]) + 1e-12, (it, r[2], g[2])
        assert np.abs(r[4] - g[4]).max() < 1e-8, (it, np.abs(r[4] - g[4]).max())


@pytest.mark.parametrize("n,levels,sweeps", [(3331, 3, 7), (3331, 3, 12), (10000, 1, 7), (10000, 1, 12)])
def test_truncation_bound_with_many_sweeps_and_the_tagged_handoffs(mods, n, levels, sweeps):
    """Round 6, second half.  (i) The refinement sweeps of k_sep_tail trade their vectors as tagged 64-bit word pairs (no flag):
    whole solves with few levels and many sweeps walk the path of the complete reduction - same iterations, same accepted steps,
    cost to rounding.  (ii) The bound on what the truncation leaves is measured from the FIRST clean pair of sweeps: with >= 6
    sweeps the last two updates are rounding noise, their ratio (~1) used to refuse converged solves; now every step verifies
    and the reported bound sits at the rounding level, below the default tolerance."""
    calib, fte, synth = mods
    seq = synth.make_sequence(n, "loop")
    rig = (seq["K"], seq["D"], seq["R"], seq["t"])
    xa = _start(fte, seq, n, 3 * n)
    ref = fte.FTEContext(seq["det"], *rig, seq["Ts"], bcr_levels=0)
    ref.set_x(xa)
    iref = ref.solve(80)
    ref.close()
    ctx = fte.FTEContext(seq["det"], *rig, seq["Ts"], bcr_levels=levels, refine_sweeps=sweeps, trunc_tol=fte.FTEContext.TRUNC_TOL)
    ctx.set_x(xa)
    worst = 0.0
    for _ in range(80):
        ctx.step()
        st = ctx.state()
        worst = max(worst, st["trunc_eps"])
        if st["status"] != 0:
            break
    ctx.close()
    assert st["status_name"] == iref["status_name"] and st["status"] in (1, 2, 3), (st["status_name"], iref["status_name"], worst)
    assert st["iter"] == iref["iter"] and st["accepted"] == iref["accepted"]
    assert abs(st["cost"] - iref["cost"]) <= 1e-12 * abs(iref["cost"])
    assert 0.0 < worst <= fte.FTEContext.TRUNC_TOL, worst
