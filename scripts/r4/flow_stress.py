"""Randomised stress of the tile-dataflow Cholesky through the library's test entry (bsfm_dense_chol_solve): many sizes, dense and
random tile envelopes (backend 2 derives the envelope from the zero pattern), every solution against numpy, every info word checked.
python scripts/r4/flow_stress.py [cases] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bundler_sfm_amd.sfm as B

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
worst, t0, bad = 0.0, time.time(), 0
for c in range(cases):
    n = int(rng.choice([rng.integers(129, 400), rng.integers(400, 1400), rng.integers(1400, 3300), rng.integers(3300, 5200)], p=[0.35, 0.35, 0.22, 0.08]))
    T = (n + 127) // 128
    env = rng.random() < 0.5
    A = np.zeros((n, n))
    if env:
        # random block profile: tile row I starts at tile column first[I] (non-decreasing bandwidth is not required)
        first = [max(0, I - int(rng.integers(0, max(1, min(T, 8))))) for I in range(T)]
        for I in range(T):
            r0, r1 = 128 * I, min(n, 128 * (I + 1))
            c0 = 128 * first[I]
            A[r0:r1, c0:r1] = rng.standard_normal((r1 - r0, r1 - c0))
        A = np.tril(A); A = A + A.T
    else:
        G = rng.standard_normal((n, min(n, 96)))
        A = G @ G.T
    A[np.diag_indices(n)] = np.abs(A).sum(axis=1) + 1.0
    b = rng.standard_normal(n)
    rc, x = B.dense_chol_solve(A, b, backend=2 if env else 0)
    res = np.abs(A @ x - b).max() / (np.abs(A).max() * max(np.abs(x).max(), 1e-300))
    worst = max(worst, res)
    if rc != 0 or not np.isfinite(res) or res > 1e-12:
        bad += 1
        print(f"case {c}: n = {n}, envelope = {env}: rc {rc}, scaled residual {res:.2e}  <-- BAD", flush=True)
print(f"{cases} cases, {bad} bad, worst scaled residual {worst:.2e}, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
