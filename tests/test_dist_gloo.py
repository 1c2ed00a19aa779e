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
