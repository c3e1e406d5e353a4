#!/bin/bash
# round 6, call M: laxity 2.0 against 0 at other sizes and on a dense-valued matrix
ulimit -c 0
cd /root/repo
O=gpurun_out/r6m; mkdir -p $O
run() { echo "== n=$N $*"; env "$@" BSFM_CHOL_REPS=6 timeout 120 python scripts/r4/chol_reps.py $N 2>&1 | grep "rep [1-5]" | awk '{printf "%s ", $(NF-1)}'; echo; }
{ for N in 5400 6400 7700 9000 12000; do run X=1; run BSFM_FLOW_LAXITY=2; done; } 2>&1 | tee $O/laxity_sizes.txt
python - <<'PY' 2>&1 | tee -a $O/laxity_sizes.txt
import os, numpy as np
import bundler_sfm_amd.sfm as S
rng = np.random.default_rng(3); n = 9000
G = rng.standard_normal((n, 256)); A = G @ G.T; A[np.diag_indices(n)] += n; b = rng.standard_normal(n)
for lx in ("0", "2"):
    os.environ["BSFM_FLOW_LAXITY"] = lx
    rc, x, ms, fm, gf = S.dense_chol_solve_timed(A, b, reps=5)
    print("dense-valued n=9000 laxity", lx, "solve ms", [round(v, 3) for v in ms], "kernel", round(fm, 3))
PY
