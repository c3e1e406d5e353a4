"""How closely the LM trajectory of the known-intrinsics fixture scene (8 cameras / 60 points, one-tile reduced system) follows the reference at fixed
iteration indices, for the one-tile dataflow solve (default) and the round-2 tile kernel (BSFM_CHOL=streams): cost and parameter differences."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bundler_sfm_amd as B
import oracle_util as O
X = np.load(os.path.join(ROOT, "tests", "golden", "known_golden.npz"))
m, n = len(X["cam_f"]), len(X["pts"]) // 3
cams = O.arrays_to_cams(X["cam_R"], X["cam_t"], X["cam_f"], X["cam_k"])
for j in range(m):
    cams[j].known_intrinsics = int(X["cam_known"][j])
    for q in range(9): cams[j].K_known[q] = float(X["cam_K_known"][j][q])
    for q in range(5): cams[j].k_known[q] = float(X["cam_k_known"][j][q])
vm = B.dense_vmask(n, m, X["rowptr"], X["colidx"])
REF_OPTS = [1e-3, 1e-10, 1e-12, 1e-12, 0.0, 4e-2]
for it in (5, 10, 15, 20, 25, 30):
    r = O.ref_sba(n, m, vm, X["proj"], cams, X["pts"], itmax=it, jac_mode=0)
    pb = B.Problem(n, m, X["rowptr"], X["colidx"], X["proj"], cams, X["pts"], options=B.default_options(jacobian=B.JAC_FD, verbose=0, itmax=it, opts=REF_OPTS))
    rc, info = pb.solve(); p = pb.download(want_cams=False)[0]; pb.close()
    print(os.environ.get("BSFM_CHOL", "flow"), "it", it, "cost rel diff %.2e" % (abs(info[1] - r["info"][1]) / r["info"][1]), "p rel diff %.2e" % (np.abs(p - r["p"]).max() / np.abs(r["p"]).max()),
          "counters", list(info[5:10]) == list(r["info"][5:10]), "mu ratio-1 %.1e" % (info[4] / r["info"][4] - 1 if r["info"][4] else 0))
