"""Wall time of the drop-in boundary on the problem sizes incremental Bundler actually calls run_sfm with (14 .. 400 cameras):
GPU run_sfm end to end (dense vmask in, cameras / points out: index construction, allocation, upload, LM, download) next to the
reference's own run_sfm on one core of this host, plus the split of the GPU call (create / iterate)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bundler_sfm_amd as B
import oracle_util as O


def run(tag, n, m, vm, proj, cams, pts, s):
    opt = B.default_options(verbose=0)
    c2 = B.copy_cameras(cams); p2 = pts.copy()
    B.run_sfm(n, m, 0, vm, proj, 1, 0, 1, 1, c2, p2, eps2=1e-12, options=opt)          # warm-up (module load, context, pools)
    ts = []
    for _ in range(5):
        c2 = B.copy_cameras(cams); p2 = pts.copy()
        t = time.perf_counter()
        rc, info = B.run_sfm(n, m, 0, vm, proj, 1, 0, 1, 1, c2, p2, eps2=1e-12, options=opt)
        ts.append(time.perf_counter() - t)
    ph = {k: round(B.lib.bsfm_run_sfm_last_ms(k.encode()), 2) for k in ("total", "crs", "create", "lm", "download")}
    t = time.perf_counter()
    pb = B.Problem(n, m, s["rowptr"], s["colidx"], proj, cams, pts, options=opt)
    t_create = time.perf_counter() - t
    t = time.perf_counter(); pb.solve(); t_solve = time.perf_counter() - t
    split = {k: round(pb.phase_ms("create_" + k), 2) for k in ("total", "upload", "index", "alloc")}
    pb.close()
    line = (f"{tag}: GPU run_sfm {1e3 * min(ts):8.2f} ms {ph}, {int(info[5])} iterations (resident API: create {1e3 * t_create:.2f} ms {split}, "
            f"LM {1e3 * t_solve:.2f} ms = {1e3 * t_solve / max(info[5], 1):.3f} ms/iter)")
    if O.have_ref() and not os.environ.get("SMALL_NO_REF"):
        t = time.perf_counter(); O.ref_run_sfm(n, m, vm, proj, cams, pts); tr = time.perf_counter() - t
        line += f" | reference run_sfm (1 core) {1e3 * tr:9.1f} ms"
    print(line, flush=True)


ONLY = [int(v) for v in os.environ.get("SMALL_ONLY", "").split(",") if v]        # e.g. SMALL_ONLY=50 under the profiler
for m, n in ((14, 1500), (50, 10000), (100, 20000), (200, 50000), (400, 100000)):
    if ONLY and m not in ONLY:
        continue
    s = B.synth_ba(m, n, 10)
    vm = B.dense_vmask(n, m, s["rowptr"], s["colidx"])
    run(f"{m:4d} cams / {n:6d} pts / {10 * n:7d} obs", n, m, vm, s["proj"], s["cams"], s["pts"], s)
