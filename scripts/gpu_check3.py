import sys, time, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import bundler_sfm_amd as B
phases = ("jacobian", "cam_blocks", "point_blocks", "point_invert", "schur", "solve", "backsub", "residual")
def run(m, n, itmax, jac=B.JAC_ANALYTIC, tag=""):
    s = B.synth_ba(m, n, 10)
    opt = B.default_options(jacobian=jac, verbose=0, itmax=itmax)
    t0 = time.time()
    pb = B.Problem(n, m, s['rowptr'], s['colidx'], s['proj'], s['cams'], s['pts'], options=opt)
    t1 = time.time()
    rc, info = pb.solve(); t2 = time.time()
    print(f"{tag} m={m} n={n} create {t1-t0:.3f}s solve {t2-t1:.4f}s iters {info[5]:.0f} attempts {info[9]:.0f} stop {info[6]:.0f} cost {info[0]:.6e}->{info[1]:.6e}  ms/iter {1e3*(t2-t1)/max(info[5],1):.3f}")
    print("    " + "  ".join(f"{ph}={pb.phase_ms(ph):.3f}" for ph in phases))
    pb.close()
run(50, 10000, 150, tag="cfg2")
run(1000, 20000, 3, tag="chol9000")
if len(sys.argv) > 1 and sys.argv[1] == "big":
    run(1000, 500000, 10, tag="cfg3")
