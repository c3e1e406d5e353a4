"""Phase split of one LM iteration at the sizes incremental Bundler spends its time at (tens to hundreds of cameras)."""
import os
os.environ.setdefault("BSFM_PHASE_TIMING", "1")      # (phase events are off below 2 M observations since round 6)
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import os
os.environ.setdefault("BSFM_PHASE_TIMING", "1")      # (phase events are off below 2 M observations since round 6)
import bundler_sfm_amd as B
for m, n in ((14, 3000), (50, 10000), (100, 20000), (200, 50000), (400, 100000)):
    s = B.synth_ba(m, n, 10 if m >= 50 else 7, banded=(m >= 100))
    opt = B.default_options(verbose=0, itmax=1000, opts=[1e-3, 0.0, 0.0, 0.0, 0.0, -1.0])
    pb = B.Problem(n, m, s["rowptr"], s["colidx"], s["proj"], s["cams"], s["pts"], options=opt)
    pb.lm_begin(); pb.lm_iterate(3); B.lib.bsfm_device_synchronize()
    t = time.perf_counter(); pb.lm_iterate(20); B.lib.bsfm_device_synchronize(); t = (time.perf_counter() - t) / 20
    ph = {k: round(pb.phase_ms(k), 3) for k in ("jacobian", "cam_blocks", "point_blocks", "schur", "solve", "backsub", "residual")}
    print(f"{m:4d} cams / {n:6d} pts: {1e3 * t:.3f} ms per iteration; phases {ph}")
    pb.close()
