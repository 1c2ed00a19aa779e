"""CPU, world_size 2 (gloo): the sharded LM driver (acinoset_amd/dist.py) with the oracle backend reproduces
the single-process oracle LM - shard plan, separator all-reduce, halo all-gather, global accept/reject."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, n_steps, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from acinoset_amd import dist as adist
    from oracle import fk, synth
    from oracle_backend import OracleBackend
    seq = synth.make_sequence(n_frames, "sprint")
    plan = adist.shard_plan(n_frames, world)
    n0, n1 = plan[rank]
    be = OracleBackend(seq["det"][n0:n1], seq["K"], seq["D"], seq["R"], seq["t"], seq["Ts"], n_frames, n0, rank, world,
                       ftol=0.0, xtol=0.0, gtol=0.0)
    drv = adist.ShardedFTE(be, rank, world)
    rng = np.random.default_rng(5)
    x0 = seq["q_true"][:, fk.ACTIVE] + rng.normal(0, 0.03, (n_frames, 25))
    drv.set_x(torch.as_tensor(x0[n0:n1]))
    for _ in range(n_steps):
        drv.step()
    x = drv.gather_x(max(b - a for a, b in plan)).numpy()
    if rank == 0:
        np.savez(out_path, x=x, cost=be.state()["cost"], it=be.state()["iter"], acc=be.state()["accepted"])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_driver_matches_single_process_oracle(tmp_path, world):
    from oracle import fk, synth
    from oracle import fte as ofte
    n_frames, n_steps = 27, 6
    out = str(tmp_path / "x.npz")
    mp.spawn(_worker, args=(world, _free_port(), n_frames, n_steps, out), nprocs=world, join=True)
    got = np.load(out)
    seq = synth.make_sequence(n_frames, "sprint")
    prob = ofte.FTEProblem(seq["det"][..., :2], seq["det"][..., 2], seq["K"], seq["D"], seq["R"], seq["t"], seq["Ts"])
    rng = np.random.default_rng(5)
    x0 = seq["q_true"][:, fk.ACTIVE] + rng.normal(0, 0.03, (n_frames, 25))
    x_ref, info = ofte.lm_solve(prob, x0, max_iter=n_steps, ftol=0.0, xtol=0.0, gtol=0.0)
    assert int(got["it"]) == n_steps and int(got["acc"]) == info["accepted"]
    assert abs(float(got["cost"]) - info["cost"]) < 1e-8 * abs(info["cost"])
    assert np.abs(got["x"] - x_ref).max() < 1e-8


def test_shard_plan():
    sys.path.insert(0, ROOT)
    from acinoset_amd.dist import shard_plan
    for n, w in ((10000, 8), (10000, 1), (1001, 4), (27, 3), (12, 2)):
        plan = shard_plan(n, w)
        assert plan[0][0] == 0 and plan[-1][1] == n
        for (a, b), (c, d) in zip(plan, plan[1:]):
            assert b == c and b % 3 == 0
        assert all(b - a >= 6 for a, b in plan) or w == 1
    with pytest.raises(ValueError):
        shard_plan(9, 2)


def _hook_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from acinoset_amd import sba
    ws = torch.zeros(4096, dtype=torch.uint8)
    hook = sba.ReduceHook(ws)
    vals = ws[256:256 + 8 * 5].view(torch.float64)
    vals.copy_(torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0], dtype=torch.float64) * (rank + 1))
    base = ws.data_ptr()
    rc_sum = hook._call(None, base + 256, 3, 0, None)          # sum over ranks of the first three
    rc_max = hook._call(None, base + 256 + 32, 1, 1, None)     # max over ranks of the last one
    rc_bad = hook._call(None, base + 4090, 4, 0, None)         # beyond the workspace: refused, no collective issued
    # through the C function pointer the library would call
    cfn_rc = hook.fn(None, base + 256 + 24, 1, 0, None)
    np.save(out_path + f".{rank}.npy", np.array(list(vals.numpy()) + [rc_sum, rc_max, rc_bad, cfn_rc, hook.calls]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sba_reduce_hook_sums_workspace_slices_over_ranks(tmp_path, world):
    """The callback acino_sba_solve_sharded calls (acinoset_amd.sba.ReduceHook): in-place sum / max of a slice of
    the workspace tensor over the ranks, addressed by raw pointer."""
    out = str(tmp_path / "hook")
    mp.spawn(_hook_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    tri = world * (world + 1) / 2
    for r in range(world):
        got = np.load(out + f".{r}.npy")
        assert list(got[:3]) == [tri, 2 * tri, 3 * tri]          # summed
        assert got[3] == 4.0 * tri                                 # summed through the C function pointer
        assert got[4] == 5.0 * world                               # max
        assert list(got[5:]) == [0, 0, 1, 0, 3]                    # return codes; three collectives completed


def _window_worker(rank, world, port, n_frames, halo, n_steps, out_path):
    try:                                   # several oracle processes on a few cores: one BLAS thread each
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)
    except Exception:
        pass
    torch.set_num_threads(1)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from acinoset_amd import dist as adist
    from oracle import fk, synth
    from oracle_backend import OracleWindowBackend
    seq = synth.make_sequence(n_frames, "trot")
    plan, halo = adist.window_plan(n_frames, world, halo)
    w0, w1, n0, n1 = plan[rank]
    be = OracleWindowBackend(seq["det"][w0:w1], seq["K"], seq["D"], seq["R"], seq["t"], seq["Ts"], n_frames, w0, n0 - w0, n1 - n0,
                             ftol=0.0, xtol=0.0, gtol=0.0)
    drv = adist.WindowedFTE(be, rank, world, (n0 - w0, n1 - n0), halo)
    rng = np.random.default_rng(5)
    x0 = seq["q_true"][:, fk.ACTIVE] + rng.normal(0, 0.03, (n_frames, 25))
    drv.set_x(torch.as_tensor(x0[w0:w1]))
    cost0 = be.state()["cost"]
    for _ in range(n_steps):
        drv.step()
    st = be.state()
    np.savez(out_path + f".{rank}.npz", x=drv.result_x().numpy(), cost=st["cost"], cost0=cost0, it=st["iter"], acc=st["accepted"],
             lam=st["lam"])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,halo", [(2, 96)])        # (3 and 4 shards: GPU tests, lock step and multi-process)
def test_windowed_driver_matches_single_process_oracle(tmp_path, world, halo):
    """The overlapping-window driver (dist.WindowedFTE: slab all-gather, owned-range sums, replicated controller) under
    gloo with the oracle backend.  Cost bookkeeping is exact (initial cost == the single-process cost, identical on every
    rank); the step is inexact by the decay over the 96-frame halo (a few 1e-3 with lambda -> 0), so the trajectory is
    compared after convergence."""
    from oracle import fk, synth
    from oracle import fte as ofte
    n_frames, n_steps = 110 * world, 16
    out = str(tmp_path / "w")
    mp.spawn(_window_worker, args=(world, _free_port(), n_frames, halo, n_steps, out), nprocs=world, join=True)
    parts = [np.load(out + f".{r}.npz") for r in range(world)]
    seq = synth.make_sequence(n_frames, "trot")
    prob = ofte.FTEProblem(seq["det"][..., :2], seq["det"][..., 2], seq["K"], seq["D"], seq["R"], seq["t"], seq["Ts"])
    rng = np.random.default_rng(5)
    x0 = np.clip(seq["q_true"][:, fk.ACTIVE] + rng.normal(0, 0.03, (n_frames, 25)), prob.lo, prob.hi)
    F0 = prob.evaluate(x0, need_jac=False)[0]
    assert all(abs(float(p["cost0"]) - F0) < 1e-9 * abs(F0) for p in parts)            # owned ranges add up exactly
    assert len({(float(p["cost"]), int(p["acc"]), float(p["lam"])) for p in parts}) == 1   # one controller, replicated
    xo, info = ofte.lm_solve(prob, x0, max_iter=60, ftol=1e-13)
    x = np.concatenate([p["x"] for p in parts])
    assert x.shape == xo.shape and int(parts[0]["it"]) == n_steps
    assert abs(float(parts[0]["cost"]) - info["cost"]) < 1e-5 * abs(info["cost"]), (float(parts[0]["cost"]), info["cost"])
    pos = fk.cheetah_fk(prob.full_state(x))
    pos_o = fk.cheetah_fk(prob.full_state(xo))
    assert np.abs(pos - pos_o).max() < 1e-3                                                # north-star tolerance, metres


def test_combine_partials_ors_the_numeric_flag_mask():
    """Slot 5 of the per-rank sums is a bit mask (bit 0 pivot, 1 sync timeout, 2 truncation).  The host combination must be
    the device's (k_control_gathered ORs): with a max, rank A (bit 0) and rank B (bit 2) would see 4, derive statuses 5 and
    7, and part ways in the next collective (round-3 advisor finding)."""
    sys.path.insert(0, ROOT)
    from acinoset_amd import dist as adist
    g = torch.zeros(3, 8, dtype=torch.float64)
    g[:, 0] = torch.tensor([1.0, 2.0, 3.0])
    g[0, 5], g[1, 5], g[2, 5] = 1.0, 4.0, 0.0
    tot = adist.combine_partials(g)
    assert tot[5].item() == 5.0 and tot[0].item() == 6.0
