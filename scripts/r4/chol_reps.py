"""BSFM_CHOL_REPS=n python scripts/r4/chol_reps.py [n_unknowns]: repeated dense solves of the headline scene's pattern (100 dense 90 x 90
blocks, zeros elsewhere, through the dense task list); the library prints the device time of every repetition on stderr."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bundler_sfm_amd.sfm as B
n = int(sys.argv[1]) if len(sys.argv) > 1 else 9000
rng = np.random.default_rng(5)
Z = np.zeros((n, n))
for g in range((n + 89) // 90):
    lo, hi = 90 * g, min(n, 90 * (g + 1))
    G = rng.standard_normal((hi - lo, hi - lo + 8))
    Z[lo:hi, lo:hi] = G @ G.T + (hi - lo) * np.eye(hi - lo)
b = rng.standard_normal(n)
rc, x = B.dense_chol_solve(Z, b)
print("rc", rc, "residual", np.abs(Z @ x - b).max())
