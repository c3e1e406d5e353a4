"""Multi-GPU exchange step through the real library on ONE GPU: two processes share cuda:0, each owns a
point shard, and the all-reduce hook sums device buffers across the two processes (gloo on a host copy -- RCCL refuses
two ranks on the same device; bench.py uses RCCL with one rank per GPU, the hook contract is identical).
The packed block-union exchange (solver.hip:exchange_block_union / k_schur_pack / k_schur_unpack) must reproduce the
single-process LM trajectory up to summation order."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)

from bench import shard_points  # noqa: E402

# (process start-up + RCCL initialisation took 87 s on one box of the pool: the 180 s default of tests/conftest.py is too tight here)
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]
ITERS = 6


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _scene(B):
    m, n = 15, 700                                    # deg 5: cameras j0 + 3 d -> three groups (mod 3) that share no point
    s = B.synth_ba(m, n, 5)
    keep = np.ones(len(s["colidx"]), bool)
    keep[np.arange(0, len(keep), 9)] = False          # ragged shards
    rows = np.repeat(np.arange(n), 5)
    cnt = np.bincount(rows[keep], minlength=n)
    rowptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    return dict(m=m, n=n, rowptr=rowptr, colidx=s["colidx"][keep].copy(),
                proj=s["proj"].reshape(-1, 2)[keep].ravel().copy(), cams=s["cams"], pts=s["pts"])


def _solve(B, sc, lo, hi, world, rank, hook=None, jac=None):
    rp = (sc["rowptr"][lo:hi + 1] - sc["rowptr"][lo]).astype(np.int32)
    k0, k1 = int(sc["rowptr"][lo]), int(sc["rowptr"][hi])
    opt = B.default_options(jacobian=B.JAC_ANALYTIC if jac is None else jac, verbose=0, itmax=ITERS,
                            opts=[1e-3, 0.0, 0.0, 0.0, 0.0, -1.0])
    pb = B.Problem(hi - lo, sc["m"], rp, sc["colidx"][k0:k1], sc["proj"][2 * k0:2 * k1], sc["cams"], sc["pts"][3 * lo:3 * hi],
                   options=opt, world_size=world, rank=rank, nvis_global=int(sc["rowptr"][-1]),
                   nvars_global=sc["m"] * 9 + 3 * sc["n"])
    if hook is not None:
        pb.set_allreduce(hook)
    assert pb.lm_begin() == 0
    pb.lm_iterate(ITERS)
    rc, info = pb.lm_finish()
    p = pb.download(want_cams=False)[0]
    info = np.append(info, pb.phase_ms("groups"))           # number of independent camera groups solved separately (0 = dense)
    pb.close()
    return p, info


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    import bundler_sfm_amd as B
    sc = _scene(B)
    lo, hi = shard_points(sc["rowptr"], world, rank)
    calls = []

    def hook(dev_ptr, count, op, _ctx):
        class _Buf:
            __cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (dev_ptr, False), "version": 2}
        t = torch.as_tensor(_Buf(), device="cuda:0")
        c = t.cpu()
        dist.all_reduce(c, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX)
        t.copy_(c)
        torch.cuda.synchronize()
        calls.append(count)
        return 0

    p, info = _solve(B, sc, lo, hi, world, rank, hook)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), p=p, info=info, lo=lo, hi=hi, max_count=max(calls), ncalls=len(calls))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("solver", ["dense", "auto"])
def test_two_ranks_one_gpu_match_single_rank(tmp_path, monkeypatch, solver):
    """solver = auto: the scene's cameras fall into three groups (camera index mod 3), found on every rank from the job-wide block
    union after the first exchange and then solved group by group (compsolve.hip.h)."""
    import torch
    import torch.multiprocessing as mp
    import bundler_sfm_amd as B
    monkeypatch.setenv("BSFM_REDUCED_SOLVER", solver)          # read by bsfm_default_options, inherited by the workers
    sc = _scene(B)
    assert all(len(set(sc["colidx"][sc["rowptr"][i]:sc["rowptr"][i + 1]] % 3)) == 1 for i in range(sc["n"]))
    p1, info1 = _solve(B, sc, 0, sc["n"], 1, 0)
    assert info1[1] < 0.05 * info1[0]                      # the LM run actually converged somewhere
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r = [np.load(tmp_path / f"rank{k}.npz") for k in (0, 1)]
    m = sc["m"]
    for k in (0, 1):
        # every rank ends with the same cameras (replicated) and the global cost
        assert abs(r[k]["info"][1] - info1[1]) <= 1e-9 * info1[1]
        assert r[k]["info"][5] == info1[5]
        assert r[k]["info"][-1] == (3 if solver == "auto" else 0) and info1[-1] == r[k]["info"][-1]
        cam = r[k]["p"][:9 * m]
        assert np.abs(cam - p1[:9 * m]).max() <= 1e-8 * np.abs(p1[:9 * m]).max()
        lo, hi = int(r[k]["lo"]), int(r[k]["hi"])
        pts = r[k]["p"][9 * m:]
        assert np.abs(pts - p1[9 * m + 3 * lo:9 * m + 3 * hi]).max() <= 1e-8
        # the exchange is the packed block union (<= lower triangle: m(m+1)/2 blocks of 81), never the dense (9m)^2 matrix
        assert int(r[k]["max_count"]) <= max(81 * m * (m + 1) // 2, 81 * m)
    assert np.array_equal(r[0]["p"][:9 * m], r[1]["p"][:9 * m])      # bitwise identical replicas


# ---- the collective on the C/C++ side of the boundary (comm.hip) ------------------------------------------------------------
def test_run_sfm_over_two_in_process_ranks_matches_one_gpu(monkeypatch):
    """run_sfm itself on "2 GPUs": opt.num_gpus = 2 makes the library shard the points over two host threads inside the one
    process (boundary.hip:run_multi).  Both ranks sit on the single device of this box, so the communicator falls back from
    ncclCommInitAll to its loopback transport (BSFM_ALLOW_SHARED_DEVICE) -- the whole multi-rank control flow (sharding,
    U || ea exchange, block-union exchange, packed reduced system, scalar decisions, per-rank download) is the production one."""
    import bundler_sfm_amd as B
    monkeypatch.setenv("BSFM_ALLOW_SHARED_DEVICE", "1")
    m, n = 24, 1500
    s = B.synth_ba(m, n, 6, banded=True)
    vm = B.dense_vmask(n, m, s["rowptr"], s["colidx"])
    outs = []
    for g, solver in ((1, B.SOLVER_DENSE), (2, B.SOLVER_DENSE), (3, B.SOLVER_DENSE), (2, B.SOLVER_ENVELOPE), (3, B.SOLVER_AUTO)):
        # (the last two: the envelope solver on a multi-rank job -- every rank derives the same camera numbering from the job-wide
        #  union of the reduced-camera blocks; the exchange buffer stays in the natural numbering)
        cams = B.copy_cameras(s["cams"]); pts = s["pts"].copy()
        opt = B.default_options(jacobian=B.JAC_ANALYTIC, verbose=0, itmax=8, num_gpus=g, reduced_solver=solver)
        rc, info = B.run_sfm(n, m, 0, vm, s["proj"], 1, 0, 1, 1, cams, pts, eps2=1e-12, options=opt)
        assert rc >= 0
        outs.append((info, np.array([list(c.R) + list(c.t) + [c.f] + list(c.k) for c in cams]), pts))
    i1, c1, p1 = outs[0]
    assert i1[1] < 0.05 * i1[0]
    for info, cam, pts in outs[1:]:
        assert list(info[5:10]) == list(i1[5:10])                       # same iterations / stop code / counters
        assert abs(info[1] - i1[1]) <= 1e-9 * i1[1]
        assert np.abs(cam - c1).max() <= 1e-8 * np.abs(c1).max()
        assert np.abs(pts - p1).max() <= 1e-8


def _rccl_world1(out):
    import ctypes as C
    import bundler_sfm_amd as B
    c = B.lib.bsfm_comm_create_from_env()
    res = dict(ok=bool(c), transport="", sum=0.0, idfile=os.environ["BSFM_COMM_ID_FILE"])
    if c:
        res["transport"] = B.lib.bsfm_comm_transport(c).decode()
        v = (C.c_double * 3)(1.0, 2.0, 3.5)
        rc = B.lib.bsfm_comm_allreduce_host(c, v, 3, 0)
        res["sum"] = float(v[0] + v[1] + v[2]) if rc == 0 else -1.0
        res["exists_before_destroy"] = os.path.exists(res["idfile"])
        # a BA problem that BELIEVES it is one of two ranks but reduces over this one-rank communicator: the exchange path
        # (packed block union, ncclAllReduce on the compute stream) runs end to end through RCCL
        s = B.synth_ba(12, 300, 6)
        pb = B.Problem(300, 12, s["rowptr"], s["colidx"], s["proj"], s["cams"], s["pts"], world_size=2, rank=0,
                       options=B.default_options(jacobian=1, verbose=0, itmax=4))
        B.lib.bsfm_problem_set_comm(pb.h, c)
        rc2, info = pb.solve()
        pb2 = B.Problem(300, 12, s["rowptr"], s["colidx"], s["proj"], s["cams"], s["pts"], options=B.default_options(jacobian=1, verbose=0, itmax=4))
        rc3, info3 = pb2.solve()
        res["ba"] = [float(info[1]), float(info3[1]), int(rc2), int(rc3)]
        pb.close(); pb2.close()
        B.lib.bsfm_comm_destroy(c)
        res["exists_after_destroy"] = os.path.exists(res["idfile"])
    import json
    json.dump(res, open(out, "w"))


def test_rccl_communicator_from_env_world_size_one(tmp_path):
    """bsfm_comm_create_from_env with RCCL forced on for a world of one rank: dlopen of librccl.so, ncclUniqueId hand-over file,
    ncclCommInitRank, ncclAllReduce enqueued on the problem's stream.  (More ranks need more GPUs than this box has; the
    N-rank path differs only in the value of WORLD_SIZE.)"""
    import json
    import subprocess
    out = tmp_path / "res.json"
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_PORT="29876", BSFM_COMM_FORCE_RCCL="1",
               BSFM_COMM_ID_FILE=str(tmp_path / "nccl.id"), HSA_ENABLE_IPC_MODE_LEGACY="0")
    code = f"import sys; sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {HERE!r}); import test_multi_gpu as t; t._rccl_world1({str(out)!r})"
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.load(open(out))
    assert res["ok"] and res["transport"] == "rccl"
    assert res["sum"] == 6.5
    assert res["exists_before_destroy"] and not res["exists_after_destroy"]
    c_two, c_one, rc2, rc3 = res["ba"]
    assert rc2 == rc3 and abs(c_two - c_one) <= 1e-12 * c_one


# ---- two PROCESSES on one GPU through the library's own communicator (ipc transport of comm.hip) -----------------------------
def _ipc_worker(out):
    import ctypes as C
    import json
    import torch
    import bundler_sfm_amd as B
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    B.lib.bsfm_comm_create_from_env.restype = C.c_void_p
    B.lib.bsfm_comm_transport.restype = C.c_char_p
    c = B.lib.bsfm_comm_create_from_env()
    res = dict(ok=bool(c), rank=rank)
    if c:
        c = C.c_void_p(c)
        res["transport"] = B.lib.bsfm_comm_transport(c).decode()
        v = (C.c_double * 2)(rank + 1.0, 10.0 * (rank + 1))
        res["rc_host"] = B.lib.bsfm_comm_allreduce_host(c, v, 2, 0)
        res["host_sum"] = [v[0], v[1]]
        # a message larger than the exchange buffer (BSFM_COMM_IPC_MB = 1 -> 131 072 doubles): three pieces; max as the operation
        n = 300001
        t = (torch.arange(n, dtype=torch.float64, device="cuda:0") % 1000) * (1.0 if rank == 0 else -0.5)
        B.lib.bsfm_comm_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        res["rc_big"] = B.lib.bsfm_comm_allreduce(c, C.c_void_p(t.data_ptr()), n, 0, None)
        torch.cuda.synchronize()
        exp = (torch.arange(n, dtype=torch.float64) % 1000) * 0.5
        res["big_err"] = float((t.cpu() - exp).abs().max())
        # the BA problem sharded over the two processes, reduced through this communicator
        sc = _scene(B)
        lo, hi = shard_points(sc["rowptr"], world, rank)
        rp = (sc["rowptr"][lo:hi + 1] - sc["rowptr"][lo]).astype(np.int32)
        k0, k1 = int(sc["rowptr"][lo]), int(sc["rowptr"][hi])
        opt = B.default_options(jacobian=B.JAC_ANALYTIC, verbose=0, itmax=ITERS, opts=[1e-3, 0.0, 0.0, 0.0, 0.0, -1.0])
        pb = B.Problem(hi - lo, sc["m"], rp, sc["colidx"][k0:k1], sc["proj"][2 * k0:2 * k1], sc["cams"], sc["pts"][3 * lo:3 * hi],
                       options=opt, world_size=world, rank=rank, nvis_global=int(sc["rowptr"][-1]), nvars_global=sc["m"] * 9 + 3 * sc["n"])
        B.lib.bsfm_problem_set_comm(pb.h, c)
        assert pb.lm_begin() == 0
        pb.lm_iterate(ITERS)
        rc, info = pb.lm_finish()
        p = pb.download(want_cams=False)[0]
        res["dist_ranks"] = int(pb.phase_ms("flow_dist"))
        pb.close()
        np.save(out + f".p{rank}.npy", p)
        res["info"] = [float(x) for x in info]
        res["lo"], res["hi"] = int(lo), int(hi)
        B.lib.bsfm_comm_destroy(c)
    json.dump(res, open(out + f".{rank}.json", "w"))


@pytest.mark.parametrize("dist_chol", [0, 1], ids=["replicated-cholesky", "distributed-cholesky"])
def test_two_processes_one_gpu_through_the_native_ipc_transport(tmp_path, dist_chol):
    """bsfm_comm_create_from_env with WORLD_SIZE = 2 between two PROCESSES that share cuda:0 -- the id-file
    hand-over (csrc/idfile.h), the control block in shared memory, the exchange buffers opened through hipIpc, chunked all-reduces in
    rank order and the whole multi-rank LM control flow of solver.hip, without torch.distributed in the loop.  (RCCL refuses two
    ranks on one device; with one GPU per rank the same entry point takes the RCCL branch.)"""
    import json
    import subprocess
    import bundler_sfm_amd as B
    sc = _scene(B)
    p1, info1 = _solve(B, sc, 0, sc["n"], 1, 0)
    out = str(tmp_path / "ipc")
    port = str(_free_port())
    procs = []
    for rank in (1, 0):                     # the waiting rank starts first
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   BSFM_COMM_TRANSPORT="ipc", BSFM_COMM_IPC_MB="1", BSFM_COMM_ID_FILE=str(tmp_path / "job.id"),
                   BSFM_COMM_TIMEOUT_S="60", HSA_ENABLE_IPC_MODE_LEGACY="0",
                   # dist_chol: the two ranks factor the (two-tile) reduced camera system TOGETHER inside the LM loop -- same bits as replicated
                   BSFM_DIST_CHOL=str(dist_chol), BSFM_FLOW_WGS="256")
        code = f"import sys; sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {HERE!r}); import test_multi_gpu as t; t._ipc_worker({out!r})"
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    errs = [p.communicate(timeout=420) for p in procs]
    assert all(p.returncode == 0 for p in procs), [e[1][-1500:] for e in errs]
    m = sc["m"]
    res = [json.load(open(out + f".{k}.json")) for k in (0, 1)]
    ps = [np.load(out + f".p{k}.npy") for k in (0, 1)]
    for k in (0, 1):
        r = res[k]
        assert r["ok"] and r["transport"] == "ipc"
        assert r["rc_host"] == 0 and r["host_sum"] == [3.0, 30.0]
        assert r["rc_big"] == 0 and r["big_err"] == 0.0
        assert abs(r["info"][1] - info1[1]) <= 1e-9 * info1[1] and r["info"][5] == info1[5]
        assert np.abs(ps[k][:9 * m] - p1[:9 * m]).max() <= 1e-8 * np.abs(p1[:9 * m]).max()
        assert np.abs(ps[k][9 * m:] - p1[9 * m + 3 * r["lo"]:9 * m + 3 * r["hi"]]).max() <= 1e-8
    assert np.array_equal(ps[0][:9 * m], ps[1][:9 * m])               # bitwise identical replicas
    assert not os.path.exists(str(tmp_path / "job.id"))                # rank 0 removed the id file on destroy
    assert [r["dist_ranks"] for r in res] == [2 * dist_chol] * 2
    # the distributed factorisation is the same arithmetic: the run must not differ from the replicated one by a single bit
    keep = os.path.join(os.path.dirname(str(tmp_path)), "ipc_two_ranks_cameras.npy")
    if dist_chol == 0:
        np.save(keep, ps[0][:9 * m])
    elif os.path.exists(keep):
        assert np.array_equal(np.load(keep), ps[0][:9 * m])


def test_ipc_transport_times_out_when_a_rank_is_missing(tmp_path):
    """One rank of a world of two never shows up: the other gives up after BSFM_COMM_TIMEOUT_S with a message, no hang."""
    import subprocess
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               BSFM_COMM_TRANSPORT="ipc", BSFM_COMM_ID_FILE=str(tmp_path / "job.id"), BSFM_COMM_TIMEOUT_S="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); import bundler_sfm_amd as B; import ctypes as C; "
            "B.lib.bsfm_comm_create_from_env.restype = C.c_void_p; c = B.lib.bsfm_comm_create_from_env(); print('COMM', bool(c))")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0 and "COMM False" in r.stdout
    assert "only 1 of 2 ranks reached the ipc group" in r.stderr


def test_bench_runs_as_two_ranks_on_one_gpu(tmp_path):
    """bench.py itself on N > 1 ranks -- the command the driver launches for its SCALE record -- as two processes sharing cuda:0
    over the ipc transport (one GPU per rank takes the RCCL branch of the same entry point): one JSON line from rank 0, both ranks
    counted, the headline run plus the structure-aware legs (independent groups, connected scene dense / envelope) on a small scene,
    every cost at the fixed iteration index equal to the single-process run's."""
    import json
    import subprocess
    common = ["--steps", "3", "--warmup", "1", "--cams", "120", "--points", "12000", "--no-cpu-baseline", "--no-matcher", "--no-end-to-end"]
    env1 = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env1.pop(k, None)
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common, env=env1, capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stderr[-1500:]
    ref = json.loads(one.stdout.strip().splitlines()[-1])
    port = str(_free_port())
    procs = []
    for rank in (1, 0):
        env = dict(env1, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   BSFM_COMM_TRANSPORT="ipc", BSFM_COMM_ID_FILE=str(tmp_path / "bench.id"), BSFM_COMM_TIMEOUT_S="60",
                   BSFM_DIST_CHOL="1", BSFM_FLOW_WGS="256")      # the two ranks factor the reduced camera system together (round 6)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=280) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    assert outs[0][0].strip() == ""                                    # rank 1 prints nothing
    lines = [ln for ln in outs[1][0].strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                             # ONE JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["rccl_ranks_seen"] == 2 and "ipc" in d["config"]["collective"]
    assert d["config"]["distributed_cholesky_ranks"] == 2
    assert d["steps"] == 3 and d["value"] > 0 and d["scaling"] == "strong" and d["cpu_baseline"] is None
    close = lambda a, b: abs(a - b) <= 1e-9 * abs(b)
    assert close(d["cost_after_3_iterations"], ref["cost_after_3_iterations"])
    sa, cs = d["structure_aware"], d["connected_scene"]
    assert "error" not in sa and "error" not in cs and sa["n_gpus"] == 2 and cs["n_gpus"] == 2
    assert close(sa["cost_after_3_iterations"], ref["structure_aware"]["cost_after_3_iterations"])
    assert close(cs["cost_after_3_iterations"], ref["connected_scene"]["cost_after_3_iterations"])
    assert close(cs["envelope_solver"]["cost_after_3_iterations"], ref["connected_scene"]["envelope_solver"]["cost_after_3_iterations"])
    assert "matcher" not in d and "end_to_end_run_sfm" not in d


def test_bench_line_has_the_contract_fields(tmp_path):
    """One JSON line with the fields the driver parses (metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better /
    scaling / vs_baseline / dtype / data / config.workload) and the two objects of this tier: `roofline` (bound, achieved <= peak, unit,
    frac = achieved / peak, traffic) for the dominant kernel and `cpu_baseline` (value, unit, cores, kind, sample) -- on a small scene,
    both timed windows."""
    import json
    import subprocess
    env1 = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env1.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--cams", "150", "--points", "15000",
           "--cpu-baseline", "cached", "--cpu-sample", "20,1500", "--no-matcher", "--no-end-to-end", "--no-connected", "--no-structure-aware"]
    one = subprocess.run(cmd, env=env1, capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stderr[-1500:]
    lines = [ln for ln in one.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["metric"] == "BA LM iterations/sec" and d["unit"] == "iterations/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert d["scaling"] in ("strong", "weak") and d["vs_baseline"] is None and isinstance(d["config"]["workload"], str)
    assert abs(d["value"] * d["ms_per_step"] - 1e3) <= 1.0                           # iterations/s and ms per step describe the same run
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["kernel"] == "k_chol_flow" and 0 < r["achieved"] <= r["peak"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-3 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] == "reference" and c["cores"] == 1 and c["value"] > 0 and isinstance(c["sample"], str) and "unit" in c
    assert "run_sfm" in d["config"]["timed_window"] and d["continued_past_convergence"]["iterations_per_s"] > 0
    two = subprocess.run(cmd + ["--window", "continue"], env=env1, capture_output=True, text=True, timeout=300)
    assert two.returncode == 0, two.stderr[-1500:]
    d2 = json.loads([ln for ln in two.stdout.strip().splitlines() if ln.startswith("{")][0])
    assert d2["continued_past_convergence"] is None and "rounds 1-3" in d2["config"]["timed_window"]


# ---- the DISTRIBUTED reduced-camera factorisation (chol_flow.hip.h: FlowDist; VERDICT r5 "missing #1") between processes sharing cuda:0 ------------
DIST_SIZES = (1100, 3700, 9000)          # 9, 29 and 71 tile columns


def _dist_system(n, seed):
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((n, 64))
    A = G @ G.T
    A[np.diag_indices(n)] += np.exp(rng.uniform(np.log(0.5), np.log(50.0), n)) * 64.0
    d = 1.0 + np.arange(n) / n
    return A * d[:, None] * d[None, :], rng.standard_normal(n)


def _dist_worker(out):
    import ctypes as C
    import json
    import bundler_sfm_amd.sfm as S
    import bundler_sfm_amd as B
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    B.lib.bsfm_comm_create_from_env.restype = C.c_void_p
    c = C.c_void_p(B.lib.bsfm_comm_create_from_env())
    res = dict(ok=bool(c), rank=rank, rc=[], info=None, stalled=None)
    if c:
        for n in DIST_SIZES:
            A, b = _dist_system(n, n)
            rc, x = S.dense_chol_solve_dist(c, A, b)
            res["rc"].append(int(rc))
            np.save(out + f".x{n}.{rank}.npy", x)
        # dpotrf's info = the first failing leading minor, whichever rank owns its tile column (and a second failure in another rank's column)
        A, b = _dist_system(1100, 7)
        A[700, 700] = -1.0; A[300, 300] = -1.0
        res["info"] = int(S.dense_chol_solve_dist(c, A, b)[0])
        # band envelope
        A, b = _dist_system(1500, 8)
        i, j = np.indices(A.shape); A[np.abs(i - j) > 300] = 0.0
        A[np.diag_indices(1500)] = np.abs(A).sum(axis=1) + 1.0
        rc, x = S.dense_chol_solve_dist(c, A, b, backend=2)
        res["rc_env"] = int(rc); np.save(out + f".xenv.{rank}.npy", x)
        # a task of rank 1 never signals: every rank's waits expire, every rank repeats the solve on its own
        os.environ["BSFM_FLOW_TEST_STALL"] = "5"; os.environ["BSFM_FLOW_TEST_STALL_RANK"] = "1"; os.environ["BSFM_FLOW_SPIN_MS"] = "30"
        A, b = _dist_system(1100, 9)
        rc, x = S.dense_chol_solve_dist(c, A, b)
        res["stalled"] = int(rc); np.save(out + f".xstall.{rank}.npy", x)
        B.lib.bsfm_comm_destroy(c)
    json.dump(res, open(out + f".{rank}.json", "w"))


@pytest.mark.parametrize("world", [2, 4])
def test_distributed_cholesky_is_bit_identical_to_the_one_rank_solve(tmp_path, world):
    """The replicated factorisation of the multi-GPU path (every rank factoring the whole reduced camera system, sba_levmar.c:1368) split by tile
    columns over the ranks -- here 2 and 4 PROCESSES sharing cuda:0 through hipIpc (the transport a node's GPUs would use over xGMI peer access):
    panel tiles, inverse diagonal factors, y, the solution and the hand-off counters are read through peer-mapped windows.  Same static order, same
    arithmetic: every rank's solution equals bsfm_dense_chol_solve's to the last bit at 9, 29 and 71 tile columns and on a band envelope; dpotrf's
    info is the first failing minor whichever rank owns it; a starved rank makes EVERY rank fall back to its own stream-ordered solve."""
    import json
    import subprocess
    import scipy.linalg as sl
    import bundler_sfm_amd.sfm as S
    out = str(tmp_path / "dist")
    port = str(_free_port())
    procs = []
    for rank in reversed(range(world)):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   BSFM_COMM_TRANSPORT="ipc", BSFM_COMM_IPC_MB="1", BSFM_COMM_ID_FILE=str(tmp_path / "job.id"), BSFM_COMM_TIMEOUT_S="120",
                   BSFM_FLOW_WGS=str(512 // world), HSA_ENABLE_IPC_MODE_LEGACY="0")
        code = f"import sys; sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {HERE!r}); import test_multi_gpu as t; t._dist_worker({out!r})"
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    errs = [p.communicate(timeout=800) for p in procs]
    assert all(p.returncode == 0 for p in procs), [e[1][-2500:] for e in errs]
    res = [json.load(open(out + f".{k}.json")) for k in range(world)]
    for n_i, n in enumerate(DIST_SIZES):
        A, b = _dist_system(n, n)
        rc1, x1 = S.dense_chol_solve(A, b)
        assert rc1 == 0
        for k in range(world):
            assert res[k]["ok"] and res[k]["rc"][n_i] == 0
            assert np.load(out + f".x{n}.{k}.npy").tobytes() == x1.tobytes(), (n, k)
    for k in range(world):
        assert res[k]["info"] == 301                      # the smaller of the two planted minors (rows 300 and 700)
        assert res[k]["stalled"] == 0
    A, b = _dist_system(1500, 8)
    i, j = np.indices(A.shape); A[np.abs(i - j) > 300] = 0.0
    A[np.diag_indices(1500)] = np.abs(A).sum(axis=1) + 1.0
    rc1, x1 = S.dense_chol_solve(A, b, backend=2)
    A9, b9 = _dist_system(1100, 9)
    ref9 = sl.cho_solve(sl.cho_factor(A9, lower=True), b9)
    for k in range(world):
        assert res[k]["rc_env"] == 0 and np.load(out + f".xenv.{k}.npy").tobytes() == x1.tobytes()
        assert np.abs(np.load(out + f".xstall.{k}.npy") - ref9).max() <= 1e-11 * np.abs(ref9).max()
    assert any("timed out" in e[1] for e in errs)        # the fallback announced itself
