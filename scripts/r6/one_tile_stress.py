"""Randomised stress of the one-tile solve (n <= 128: k_flow_solve_one) and of run-to-run bit-identity of a small LM run."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bundler_sfm_amd.sfm as S
import bundler_sfm_amd as B
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
bad, worst = 0, 0.0
for c in range(400):
    n = int(rng.integers(1, 129))
    G = rng.standard_normal((n, n + 3))
    A = G @ G.T + n * np.eye(n) * rng.random()
    A += 1e-6 * np.eye(n)
    b = rng.standard_normal(n)
    rc, x = S.dense_chol_solve(A, b)
    ref = np.linalg.solve(A, b)
    err = np.abs(x - ref).max() / max(np.abs(ref).max(), 1e-300)
    cond = np.linalg.cond(A)
    worst = max(worst, err / cond)
    if rc != 0 or not np.isfinite(err) or err > 1e-13 * cond:
        bad += 1; print(f"case {c}: n = {n}: rc {rc} err {err:.2e} cond {cond:.2e}  <-- BAD", flush=True)
# a failing pivot: info = k
for n, k in ((40, 17), (128, 128), (100, 1)):
    A = np.eye(n) * 4.0; A[k - 1, k - 1] = -1.0
    rc, x = S.dense_chol_solve(A, np.ones(n))
    if rc != k: bad += 1; print(f"indefinite n = {n}: info {rc}, expected {k}  <-- BAD")
print(f"one-tile: 400 cases, {bad} bad, worst error / cond {worst:.2e}")
# run-to-run bit identity of a 14-camera and a 50-camera run_sfm
for m, n in ((14, 1500), (50, 10000)):
    s = B.synth_ba(m, n, 10)
    vm = B.dense_vmask(n, m, s["rowptr"], s["colidx"])
    opt = B.default_options(verbose=0)
    outs = []
    for _ in range(3):
        c2 = B.copy_cameras(s["cams"]); p2 = s["pts"].copy()
        B.run_sfm(n, m, 0, vm, s["proj"], 1, 0, 1, 1, c2, p2, eps2=1e-12, options=opt)
        outs.append(p2.copy())
    same = all(np.array_equal(outs[0], o) for o in outs[1:])
    print(f"{m} cameras: three runs bit-identical: {same}")
    if not same: bad += 1
sys.exit(1 if bad else 0)
