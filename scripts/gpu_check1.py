import sys, time, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import bundler_sfm_amd as B
import oracle_util as O
print("devices", B.lib.bsfm_device_count())
m, n = 20, 2000
s = B.synth_ba(m, n, 10)
vm = B.dense_vmask(n, m, s['rowptr'], s['colidx'])
# --- dense chol
rng = np.random.default_rng(1)
for N in (100, 300):
    A = rng.standard_normal((N, N)); A = A @ A.T + N * np.eye(N); b = rng.standard_normal(N)
    rc, x = B.dense_chol_solve(A, b); print("chol", N, rc, np.abs(x - np.linalg.solve(A, b)).max())
    rc, x = B.dense_chol_solve(A, b, 1); print("chol rocsolver", N, rc, np.abs(x - np.linalg.solve(A, b)).max())
for jac in (B.JAC_ANALYTIC, B.JAC_FD):
    opt = B.default_options(jacobian=jac, verbose=0)
    pb = B.Problem(n, m, s['rowptr'], s['colidx'], s['proj'], s['cams'], s['pts'], options=opt)
    e, cost = pb.residuals()
    r0 = O.ref_sba(n, m, vm, s['proj'], s['cams'], s['pts'], itmax=1, jac_mode=1 if jac == B.JAC_ANALYTIC else 0, want_blocks=True)
    print("jac", jac, "cost gpu", cost, "ref init", r0['info'][0], "rel", abs(cost - r0['info'][0]) / cost)
    # blocks at the reference's post-iteration p
    cams1 = O.cams_from_packed(s['cams'], r0['p'], m, 9)
    pts1 = r0['p'][m * 9:]
    pb2 = B.Problem(n, m, s['rowptr'], s['colidx'], s['proj'], cams1, pts1, options=opt)
    ne = pb2.normal_equations(0.0)
    print("  U rel", np.abs(ne['U'] - r0['U']).max() / np.abs(r0['U']).max(), " V rel", np.abs(ne['V'] - r0['V']).max() / np.abs(r0['V']).max(),
          " S rel", np.abs(ne['S'] - r0['S']).max() / np.abs(r0['S']).max())
    # full LM
    t = time.time(); rc, info = pb.solve(); t = time.time() - t
    rr = O.ref_sba(n, m, vm, s['proj'], s['cams'], s['pts'], itmax=150, jac_mode=1 if jac == B.JAC_ANALYTIC else 0)
    p, _, _ = pb.download()
    print("  gpu info", np.array2string(info, precision=6), "t", t)
    print("  ref info", np.array2string(rr['info'], precision=6), "t", rr['secs'])
    print("  p rel diff", np.abs(p - rr['p']).max() / np.abs(rr['p']).max())
    for ph in ("jacobian", "cam_blocks", "point_blocks", "schur", "solve", "backsub", "residual"): print("   ", ph, pb.phase_ms(ph))
