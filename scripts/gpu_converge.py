"""BASELINE.json configs[2]: the 1 000-camera / 500 000-point / 5 000 000-observation problem run to convergence with the
options run_sfm passes (sfm.c:705-714, eps2 = 1e-12), dense and group-by-group reduced solver."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bundler_sfm_amd as B

s = B.synth_ba(1000, 500000, 10)
for name, mode in (("dense", B.SOLVER_DENSE), ("auto", B.SOLVER_AUTO)):
    opt = B.default_options(verbose=0, opts=[1e-3, 1e-10, 1e-12, 1e-12, 0.0, 4e-2], reduced_solver=mode)
    pb = B.Problem(500000, 1000, s["rowptr"], s["colidx"], s["proj"], s["cams"], s["pts"], options=opt)
    B.lib.bsfm_device_synchronize()
    t = time.perf_counter(); rc, info = pb.solve(); B.lib.bsfm_device_synchronize(); t = time.perf_counter() - t
    print(f"{name}: {int(info[5])} iterations, stop reason {int(info[6])}, {int(info[9])} linear systems, cost {info[0]:.6e} -> {info[1]:.6e}, "
          f"{1e3 * t:.1f} ms total = {1e3 * t / max(info[5], 1):.2f} ms/iteration")
    pb.close()
