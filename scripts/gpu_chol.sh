#!/bin/bash
# Cholesky-only GPU check: small sizes first (each under its own timeout so that a hang cannot eat the call), then the full test file.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/chol
for n in 1 127 129 300 1000; do
timeout 60 python - <<PY
import sys, numpy as np, time
sys.path.insert(0, "$ROOT"); sys.path.insert(0, "$ROOT/tests")
import bundler_sfm_amd as B
n = $n
rng = np.random.default_rng(n)
A = rng.standard_normal((n, n)); A = A @ A.T + n * np.eye(n)
b = rng.standard_normal(n)
t = time.time(); rc, x = B.dense_chol_solve(A, b); dt = time.time() - t
ref = np.linalg.solve(A, b)
print("n", n, "rc", rc, "err", np.abs(x - ref).max() / np.abs(ref).max(), "secs %.2f" % dt, flush=True)
PY
echo "exit $?"
done
timeout 600 python -m pytest tests/test_chol_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5
