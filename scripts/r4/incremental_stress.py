"""An incremental reconstruction in miniature through the drop-in boundary: run_sfm on growing problems (8 .. 420 cameras: one tile to 30
tile columns of the dataflow Cholesky, more task orders than the process-wide cache keeps, the block cache evicting with
BSFM_DEVCACHE_MB=256), the whole sequence twice: the second pass must reproduce the first one's results bit for bit."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, bundler_sfm_amd as B
opt = B.default_options(verbose=0)
sizes = list(range(8, 60, 7)) + list(range(60, 421, 24))
res = []
t0 = time.perf_counter()
for rep in range(2):
    out = []
    for m in sizes:
        n = 60 * m
        s = B.synth_ba(m, n, 6, banded=(m % 2 == 0))
        vm = B.dense_vmask(n, m, s["rowptr"], s["colidx"])
        cams = B.copy_cameras(s["cams"]); pts = s["pts"].copy()
        rc, info = B.run_sfm(n, m, 0, vm, s["proj"], 1, 0, 1, 1, cams, pts, eps2=1e-12, options=opt)
        assert rc >= 0 and info[1] < info[0], (m, rc, info)
        out.append((m, int(info[5]), int(info[6]), float(info[1]), pts.tobytes(), bytes(cams)))
    res.append(out)
for a, b in zip(*res):
    assert a == b, ("pass 2 differs from pass 1 at", a[0])
print("ok: %d problems x 2 passes, %.2f s; iterations %s" % (len(sizes), time.perf_counter() - t0, [r[1] for r in res[0]]))
