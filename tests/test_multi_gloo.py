"""Multi-GPU exchange step on CPU: world_size 2, gloo backend.

Points (with all their observations) are sharded across ranks exactly as bench.py does it (shard_points), cameras are
replicated.  Each rank computes its PARTIAL normal-equation blocks with the CPU oracle; the reduced camera system is
summed with torch.distributed all_reduce -- the same exchange bundler_sfm_amd's LM loop performs through its
all-reduce hook on RCCL -- and must equal the single-rank system: S = sum_r (S_r - mu_r I) + mu I, E = sum_r E_r
(U_j, ea_j and the Schur sums are additive over points; lib/sba-1.5/sba_levmar.c:919-1339)."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)

import oracle_util as O  # noqa: E402
from bench import shard_points  # noqa: E402


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _scene():
    import bundler_sfm_amd as B
    m, n = 10, 240
    s = B.synth_ba(m, n, 5)
    # ragged: drop a few observations so that shards are not uniform
    keep = np.ones(len(s["colidx"]), bool)
    keep[np.arange(0, len(keep), 7)] = False
    rows = np.repeat(np.arange(n), 5)
    cnt = np.bincount(rows[keep], minlength=n)
    rowptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    return dict(m=m, n=n, rowptr=rowptr, colidx=s["colidx"][keep].copy(), proj=s["proj"].reshape(-1, 2)[keep].ravel().copy(),
                cams=s["cams"], pts=s["pts"])


def _vm(n, m, rowptr, colidx):
    vm = np.zeros((n, m), np.uint8)
    vm[np.repeat(np.arange(n), np.diff(rowptr)), colidx] = 1
    return vm


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    sc = _scene()
    m, n = sc["m"], sc["n"]
    lo, hi = shard_points(sc["rowptr"], world, rank)
    rp = sc["rowptr"][lo:hi + 1] - sc["rowptr"][lo]
    k0, k1 = sc["rowptr"][lo], sc["rowptr"][hi]
    args = (hi - lo, m, _vm(hi - lo, m, rp, sc["colidx"][k0:k1]), sc["proj"][2 * k0:2 * k1], sc["cams"], sc["pts"][3 * lo:3 * hi])
    q = O.port_run_sfm(*args, itmax=1, jac_mode=1, want_dumps=True)
    # the damping term is job-wide: mu = tau * max diag over ALL ranks (max-reduction, as in the LM loop)
    # exactly as the LM loop does it: U (hence its diagonal) is SUMMED over ranks first, V's diagonal is rank-local
    local_maxdiag = float(q["mu"][0]) / 1e-3
    udiag = torch.from_numpy(np.einsum("jii->ji", q["U"]).ravel() - float(q["mu"][0]))
    dist.all_reduce(udiag, op=dist.ReduceOp.SUM)
    vmax = torch.tensor([float((np.einsum("nii->ni", q["V"]) - float(q["mu"][0])).max())], dtype=torch.float64)
    dist.all_reduce(vmax, op=dist.ReduceOp.MAX)
    maxdiag = torch.tensor([max(float(udiag.max()), float(vmax[0]))], dtype=torch.float64)
    O.port().oracle_set_tau.argtypes = [__import__("ctypes").c_double]
    O.port().oracle_set_tau(1e-3 * float(maxdiag[0]) / local_maxdiag)
    q = O.port_run_sfm(*args, itmax=1, jac_mode=1, want_dumps=True)
    O.port().oracle_set_tau(-1.0)
    mu_r = float(q["mu"][0])
    assert abs(mu_r - 1e-3 * float(maxdiag[0])) <= 1e-12 * mu_r
    S = torch.from_numpy(q["S"] - mu_r * np.eye(q["S"].shape[0]))
    E = torch.from_numpy(q["E"].copy())
    cost = torch.tensor([q["info"][0]], dtype=torch.float64)
    nobs = torch.tensor([float(k1 - k0)], dtype=torch.float64)
    for t in (S, E, cost, nobs):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.savez(os.path.join(out_dir, "reduced.npz"), S=S.numpy(), E=E.numpy(), cost=cost.numpy(), nobs=nobs.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_points_partitions_and_balances():
    rowptr = np.concatenate([[0], np.cumsum(np.random.default_rng(0).integers(2, 30, 1000))]).astype(np.int32)
    for world in (1, 2, 3, 8):
        spans = [shard_points(rowptr, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == 1000
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        d = np.diff(rowptr).astype(float) ** 2
        loads = [d[a:b].sum() for a, b in spans]
        assert max(loads) <= 1.15 * (sum(loads) / world) + d.max()


def test_two_rank_reduction_equals_single_rank_system(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    red = np.load(tmp_path / "reduced.npz")
    sc = _scene()
    g = O.port_run_sfm(sc["n"], sc["m"], _vm(sc["n"], sc["m"], sc["rowptr"], sc["colidx"]), sc["proj"], sc["cams"], sc["pts"],
                       itmax=1, jac_mode=1, want_dumps=True)
    mu = float(g["mu"][0])
    Sg = g["S"] - mu * np.eye(g["S"].shape[0])
    assert red["nobs"][0] == len(sc["colidx"])
    assert abs(red["cost"][0] - g["info"][0]) <= 1e-12 * g["info"][0]
    assert np.abs(red["S"] - Sg).max() <= 1e-11 * np.abs(Sg).max()
    assert np.abs(red["E"] - g["E"]).max() <= 1e-11 * np.abs(g["E"]).max()
    # and the reduced system gives the single-rank camera step
    da = np.linalg.solve(red["S"] + mu * np.eye(len(red["E"])), red["E"])
    assert np.abs(da - g["dp"][:len(da)]).max() <= 1e-8 * np.abs(da).max()
