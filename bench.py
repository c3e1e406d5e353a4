#!/usr/bin/env python
"""bench.py -- BA LM iterations/sec on the north-star synthetic problem (BASELINE.json configs[2]):
1 000 cameras / 500 000 points / 5 000 000 observations, FP64, one MI355X (or N ranks of one node).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is ONE outer Levenberg-Marquardt iteration of the resident problem: Jacobian, U/V/ea/eb blocks, and
at least one {Schur complement, dense Cholesky solve, back-substitution, cost evaluation} attempt, with the
observations already resident in HBM.  The stop rules are disabled (eps1 = eps2 = eps3 = 0, eps5 < 0) so that
exactly W + K iterations run; nothing inside an iteration is skipped.  Multi-GPU shards POINTS (with all their
observations) across ranks, cameras are replicated, and the reduced camera system is summed with RCCL
(torch.distributed "nccl") -- a fixed total problem, i.e. strong scaling.

Prints ONE JSON line on rank 0 (metric/value/roofline/cpu_baseline ...).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X datasheet FP64 matrix (== vector) peak; SURVEY.md section 7
NB = 128                        # Cholesky tile (bundler_sfm_amd/csrc/potrf.hip.h)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cams", type=int, default=1000)
    ap.add_argument("--points", type=int, default=500000)
    ap.add_argument("--deg", type=int, default=10)
    ap.add_argument("--jacobian", choices=["fd", "analytic"], default="fd",
                    help="fd = the reference's forward differences (run_sfm default), analytic = closed form")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--reduced-solver", choices=["dense", "auto"], default="dense",
                    help="dense (default, the reference's algorithm: Cholesky of the whole reduced camera system) or auto "
                         "(independent camera groups solved separately when the scene has them)")
    ap.add_argument("--no-structure-aware", action="store_true", help="skip the extra (non-headline) run with the opt-in group-by-group reduced solve")
    ap.add_argument("--cpu-sample", default="200,50000", help="cams,points of the bounded CPU-reference sample")
    return ap.parse_args()


def shard_points(rowptr, world, rank):
    """Contiguous point ranges balanced by sum of d_i^2 (Schur work), SURVEY 8(e)."""
    d = np.diff(rowptr).astype(np.float64)
    w = np.cumsum(d * d)
    total = w[-1] if len(w) else 0.0
    bounds = [0]
    for r in range(1, world):
        bounds.append(int(np.searchsorted(w, total * r / world)))
    bounds.append(len(d))
    return bounds[rank], bounds[rank + 1]


def syrk_flops_per_launch(sdim):
    """Algorithmic FP64 flop of the timed trailing-update launches of one factorisation (mirrors the schedule in
    csrc/potrf.hip.h:potrf_solve).  One 128x128 tile x one 128-deep panel = 2*128^3 flop; step k launches k_syrk_update
    on the T(T-1)/2 - 1 lower tiles of the columns past k+1 (T = nblk-k-1; the tile S_{k+2,k+2} and the first trailing
    column belong to the chain/side streams and are not part of the timed launches)."""
    nblk = (sdim + NB - 1) // NB
    tiles = [t * (t - 1) // 2 - 1 for t in range(nblk - 1, 2, -1)]
    return 2.0 * NB ** 3 * sum(tiles) / max(len(tiles), 1), len(tiles)


def pmc_summary():
    """Newest committed PMC summary (counters need their own profiler passes, so they cannot be collected inside this run)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    try:
        return (json.load(open(files[-1])), os.path.basename(files[-1])) if files else (None, None)
    except Exception:
        return None, None


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from that summary; None when no summary is committed."""
    d, _ = pmc_summary()
    k = d["kernels"].get(kernel) if d else None
    return None if not k else round(k["traffic_bytes"])


def pmc_mfma(kernel):
    """MFMA-busy evidence of `kernel` from the same summary: SQ_VALU_MFMA_BUSY_CYCLES per launch (summed over SIMDs),
    its share of all SIMD cycles of the device (GRBM_GUI_ACTIVE is summed over the 8 XCDs) and of the busy CU cycles."""
    d, name = pmc_summary()
    c = (d["kernels"].get(kernel) or {}).get("counters") if d else None
    if not c or "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
        return None
    out = {"source": "profiles/" + name, "SQ_VALU_MFMA_BUSY_CYCLES": round(c["SQ_VALU_MFMA_BUSY_CYCLES"])}
    if c.get("GRBM_GUI_ACTIVE"):
        out["mfma_busy_of_all_simd_cycles"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4), 4)
    if c.get("SQ_BUSY_CU_CYCLES"):
        out["mfma_busy_of_busy_cu_cycles"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * c["SQ_BUSY_CU_CYCLES"]), 4)
        if c.get("GRBM_GUI_ACTIVE"):
            out["cu_busy_fraction"] = round(c["SQ_BUSY_CU_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 256), 4)
    return out


def cpu_baseline(sample):
    """Reference SBA (oracle/_ref = the reference's own C sources) on a bounded sample, 1 thread."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_util as O
    import bundler_sfm_amd as B
    if not O.have_ref():
        return None
    m, n = sample
    s = B.synth_ba(m, n, 10)
    vm = B.dense_vmask(n, m, s["rowptr"], s["colidx"])
    itmax = 3
    r = O.ref_sba(n, m, vm, s["proj"], s["cams"], s["pts"], itmax=itmax, jac_mode=0)
    its = max(int(r["info"][5]), 1)
    return {"value": its / r["secs"], "unit": "LM iterations/s", "cores": 1, "kind": "reference",
            "sample": f"reference sba_motstr_levmar (FD Jacobian, vendored CLAPACK, gcc -O3), {m} cams / {n} pts / "
                      f"{10 * n} obs, itmax={itmax}: {r['secs']:.1f} s wall = {1e3 * r['secs'] / its:.0f} ms/iter",
            "host_cpus": os.cpu_count()}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0 and world > 1:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)

    import bundler_sfm_amd as B
    if B.lib.bsfm_device_count() <= 0:
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    if world > 1:
        # the library resolves the current device through the HIP runtime torch already initialised
        pass

    m, n, deg = args.cams, args.points, args.deg
    s = B.synth_ba(m, n, deg)
    nvis_global = int(s["rowptr"][-1])
    cnp = 9
    lo, hi = shard_points(s["rowptr"], world, rank)
    rp = (s["rowptr"][lo:hi + 1] - s["rowptr"][lo]).astype(np.int32)
    k0, k1 = int(s["rowptr"][lo]), int(s["rowptr"][hi])
    ci = s["colidx"][k0:k1]
    pr = s["proj"][2 * k0:2 * k1]
    pts = s["pts"][3 * lo:3 * hi]
    opt = B.default_options(jacobian=B.JAC_FD if args.jacobian == "fd" else B.JAC_ANALYTIC, verbose=0,
                            itmax=args.warmup + args.steps + 1000,
                            opts=[1e-3, 0.0, 0.0, 0.0, 0.0, -1.0],
                            reduced_solver=B.SOLVER_AUTO if args.reduced_solver == "auto" else B.SOLVER_DENSE)
    t_create = time.time()
    pb = B.Problem(hi - lo, m, rp, ci, pr, s["cams"], pts, options=opt, world_size=world, rank=rank,
                   nvis_global=nvis_global, nvars_global=m * cnp + 3 * n)
    t_create = time.time() - t_create

    if world > 1:
        def hook(dev_ptr, count, op, _ctx):
            class _Buf:
                __cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (dev_ptr, False), "version": 2}
            t = torch.as_tensor(_Buf(), device=f"cuda:{local_rank}")
            dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX)
            torch.cuda.synchronize()
            return 0
        pb.set_allreduce(hook)

    def sync():
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
        else:
            B.lib.bsfm_device_synchronize()

    if pb.lm_begin() != 0:
        raise SystemExit("lm_begin failed")
    pb.lm_iterate(args.warmup)
    att0 = pb.attempts()
    sync()
    t0 = time.perf_counter()
    stop = pb.lm_iterate(args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    att = pb.attempts() - att0
    rc, info = pb.lm_finish()
    done = int(info[5]) - args.warmup
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if done != args.steps:
        print(f"[bench] WARNING: ran {done} of {args.steps} steps (stop={stop})", file=sys.stderr)

    if rank == 0:
        sdim = m * cnp
        flops_launch, nlaunch = syrk_flops_per_launch(sdim)
        syrk_ms = pb.phase_ms("syrk")
        phases = {ph: round(pb.phase_ms(ph), 4) for ph in
                  ("jacobian", "cam_blocks", "point_blocks", "point_invert", "schur", "solve", "backsub", "residual")}
        roof = None
        if syrk_ms and syrk_ms > 0:
            lib_gflop = pb.phase_ms("syrk_gflop")       # the library's own count of what it launched
            assert abs(lib_gflop * 1e9 - flops_launch) <= 1e-9 * flops_launch, (lib_gflop, flops_launch)
            ach = flops_launch / (syrk_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": "k_syrk_update", "achieved": round(ach, 3), "peak": FP64_MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(ach / FP64_MFMA_PEAK_TFLOPS, 4), "traffic": pmc_traffic("k_syrk_update"),
                    "traffic_note": "HBM bytes per launch from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                    "this command (scripts/profile_round.sh -> profiles/*_pmc_traffic.json; 2 x FETCH_SIZE "
                                    "per the gfx950 correction); algorithmic C traffic is 2 x 128 KB per tile",
                    "mfma_counters": pmc_mfma("k_syrk_update"), "launches_per_solve": nlaunch, "avg_launch_ms": round(syrk_ms, 4),
                    "alg_flop_per_launch": flops_launch}
        # HBM-bound streaming kernels against the 8 TB/s roof: algorithmic bytes / HIP-event time
        nv_loc, np_loc, js = float(k1 - k0), float(hi - lo), 2 * cnp + 6
        deg2 = float(np.sum(np.diff(rp).astype(np.float64) * (np.diff(rp) + 1) / 2))        # co-visibility triples
        alg = {"jacobian": nv_loc * (8 * js + 8 + 24), "cam_blocks": nv_loc * (16 * cnp + 16 + 4),
               "point_blocks": nv_loc * (48 + 16 + 4) + np_loc * 72, "backsub": nv_loc * (8 * js + 8) + np_loc * 120,
               "residual": nv_loc * 56, "schur": deg2 * (2 * 8 * js + 48) + nv_loc * 24}
        hbm = {}
        for ph, nbytes in alg.items():
            ms = phases.get(ph, 0.0)
            if ms and ms > 0:
                gbs = nbytes / (ms * 1e-3) / 1e9
                hbm[ph] = {"alg_GB": round(nbytes / 1e9, 3), "ms": ms, "GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / 8000.0, 3)}
        hbm["note"] = ("algorithmic bytes per phase over its HIP-event time; schur counts the record gathers of all co-visibility "
                       "triples (served by L2/Infinity Cache: its unique footprint is the 8*js bytes per observation)")
        out = {
            "metric": "BA LM iterations/sec", "value": round(done / elapsed, 4), "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / max(done, 1), 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"synthetic BA {m} cams / {n} pts / {nvis_global} obs (BASELINE.json configs[2]), "
                                   f"cnp=9, {args.jacobian} Jacobian, point-sharded x{world}",
                       "cameras": m, "points": n, "observations": nvis_global, "jacobian": args.jacobian,
                       "reduced_solver": args.reduced_solver,
                       "solve_attempts_per_step": round(att / max(done, 1), 3), "problem_create_s": round(t_create, 2)},
            "phases_ms": phases, "hbm_kernels": hbm, "final_cost": info[1], "initial_cost": info[0],
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(tuple(int(v) for v in args.cpu_sample.split(",")))
            except Exception as exc:   # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"error": repr(exc)}
        else:
            out["cpu_baseline"] = None
        if world == 1 and not args.no_structure_aware and args.reduced_solver == "dense":
            # NOT the headline: the same problem with the opt-in group-by-group reduced solve (compsolve.hip.h).  This
            # generator's cameras fall into m/deg groups that share no point, so S is block diagonal up to a permutation;
            # `value` above is measured with the reference's algorithm (dense Cholesky of the whole S).
            try:
                opt2 = B.default_options(jacobian=opt.jacobian, verbose=0, itmax=opt.itmax, opts=list(opt.opts),
                                         reduced_solver=B.SOLVER_AUTO)
                pb2 = B.Problem(hi - lo, m, rp, ci, pr, s["cams"], pts, options=opt2)
                pb2.lm_begin(); pb2.lm_iterate(args.warmup)
                B.lib.bsfm_device_synchronize()
                t1 = time.perf_counter(); pb2.lm_iterate(args.steps); B.lib.bsfm_device_synchronize()
                el2 = time.perf_counter() - t1
                _, info2 = pb2.lm_finish()
                d2 = int(info2[5]) - args.warmup
                out["structure_aware"] = {"reduced_solver": "auto (independent camera groups, one workgroup each)",
                                          "iterations_per_s": round(d2 / el2, 3), "ms_per_step": round(1e3 * el2 / max(d2, 1), 4),
                                          "solve_ms": round(pb2.phase_ms("solve"), 4), "schur_ms": round(pb2.phase_ms("schur"), 4),
                                          "final_cost": info2[1],
                                          "final_cost_rel_diff_vs_dense": abs(info2[1] - info[1]) / info[1]}
                pb2.close()
            except Exception as exc:
                out["structure_aware"] = {"error": repr(exc)}
        print(json.dumps(out))
    pb.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
