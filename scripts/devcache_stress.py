"""Block cache under an incremental-reconstruction pattern: run_sfm on growing problems (8 .. 59 cameras, every size twice).
Run with BSFM_DEVCACHE_MB=8 / 64 to exercise the eviction path."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, bundler_sfm_amd as B
opt = B.default_options(verbose=0)
# an incremental reconstruction in miniature: growing problems, every size twice; with a small cap so that eviction runs
t0 = time.perf_counter()
for m in list(range(8, 60, 3)) * 2:
    n = 40 * m
    s = B.synth_ba(m, n, 4)
    vm = B.dense_vmask(n, m, s["rowptr"], s["colidx"])
    cams = B.copy_cameras(s["cams"]); pts = s["pts"].copy()
    rc, info = B.run_sfm(n, m, 0, vm, s["proj"], 1, 0, 1, 1, cams, pts, eps2=1e-12, options=opt)
    assert rc >= 0, (m, rc)
print("ok %.2f s" % (time.perf_counter() - t0))
