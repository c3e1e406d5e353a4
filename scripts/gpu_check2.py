import sys, time, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import bundler_sfm_amd as B
import oracle_util as O
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "cfg2"):
    m, n = 50, 10000
    s = B.synth_ba(m, n, 10)
    vm = B.dense_vmask(n, m, s['rowptr'], s['colidx'])
    for jac in (B.JAC_ANALYTIC, B.JAC_FD):
        opt = B.default_options(jacobian=jac, verbose=0)
        pb = B.Problem(n, m, s['rowptr'], s['colidx'], s['proj'], s['cams'], s['pts'], options=opt)
        t = time.time(); rc, info = pb.solve(); t = time.time() - t
        p, _, _ = pb.download()
        q = O.port_run_sfm(n, m, vm, s['proj'], s['cams'], s['pts'], itmax=150, jac_mode=1 if jac == B.JAC_ANALYTIC else 0)
        print("cfg2 jac", jac, "gpu info", np.array2string(info, precision=8), "t %.4f" % t, "ms/iter %.3f" % (1e3 * t / max(info[5], 1)))
        print("      port info", np.array2string(q['info'], precision=8))
        print("      p rel diff", np.abs(p - q['p']).max() / np.abs(q['p']).max())
        for ph in ("jacobian", "cam_blocks", "point_blocks", "point_invert", "schur", "solve", "backsub", "residual"): print("      ", ph, "%.4f ms" % pb.phase_ms(ph))
        pb.close()
if which in ("all", "chol"):
    # Sdim = 9000 with few points: exercises the dense solve at the north-star size
    m, n = 1000, 20000
    s = B.synth_ba(m, n, 10)
    for backend in (0, 1):
        opt = B.default_options(jacobian=B.JAC_ANALYTIC, verbose=0, potrf_backend=backend, itmax=3)
        pb = B.Problem(n, m, s['rowptr'], s['colidx'], s['proj'], s['cams'], s['pts'], options=opt)
        t = time.time(); rc, info = pb.solve(); t = time.time() - t
        print("chol9000 backend", backend, "info", np.array2string(info, precision=8), "t %.4f" % t)
        for ph in ("jacobian", "cam_blocks", "point_blocks", "point_invert", "schur", "solve", "backsub", "residual"): print("      ", ph, "%.4f ms" % pb.phase_ms(ph))
        pb.close()
