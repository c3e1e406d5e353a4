"""Does the dense solve take longer on a matrix whose tiles all hold numbers than on one that is mostly zeros (same task list)?
BSFM_CHOL_REPS=4 python scripts/r4/dense_vs_zero_tiles.py  -- the library prints the device time of every repetition on stderr."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bundler_sfm_amd.sfm as B

n = 9000
rng = np.random.default_rng(5)
G = rng.standard_normal((n, n + 64))
A = G @ G.T + n * np.eye(n)                      # every tile full
b = rng.standard_normal(n)
print("fully dense", flush=True); sys.stderr.write("## fully dense\n"); sys.stderr.flush()
rc, x = B.dense_chol_solve(A, b)
print("residual", np.abs(A @ x - b).max())
Z = np.zeros_like(A)
for g in range(100):                             # the headline scene's pattern: 100 dense 90 x 90 blocks, zeros elsewhere
    Z[90 * g:90 * (g + 1), 90 * g:90 * (g + 1)] = A[90 * g:90 * (g + 1), 90 * g:90 * (g + 1)]
print("block diagonal, solved as dense", flush=True); sys.stderr.write("## block diagonal values, dense task list\n"); sys.stderr.flush()
rc, x = B.dense_chol_solve(Z, b)
print("residual", np.abs(Z @ x - b).max())
