"""A single case of scripts/r4/flow_stress.py (same random stream), solved alone under the current BSFM_FLOW_DATA_FLAGS:  dfl_case.py <seed> <case>"""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bundler_sfm_amd.sfm as B
seed, want = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for c in range(want + 1):
    n = int(rng.choice([rng.integers(129, 400), rng.integers(400, 1400), rng.integers(1400, 3300), rng.integers(3300, 5200)], p=[0.35, 0.35, 0.22, 0.08]))
    T = (n + 127) // 128
    env = rng.random() < 0.5
    A = np.zeros((n, n))
    if env:
        first = [max(0, I - int(rng.integers(0, max(1, min(T, 8))))) for I in range(T)]
        for I in range(T):
            r0, r1 = 128 * I, min(n, 128 * (I + 1)); c0 = 128 * first[I]
            A[r0:r1, c0:r1] = rng.standard_normal((r1 - r0, r1 - c0))
        A = np.tril(A); A = A + A.T
    else:
        G = rng.standard_normal((n, min(n, 96)))
        A = G @ G.T
    A[np.diag_indices(n)] = np.abs(A).sum(axis=1) + 1.0
    b = rng.standard_normal(n)
print("case", want, "n", n, "T", T, "env", env, "first", first if env else None)
for rep in range(4):
    rc, x = B.dense_chol_solve(A, b, backend=2 if env else 0)
    res = np.abs(A @ x - b).max() / (np.abs(A).max() * max(np.abs(x).max(), 1e-300))
    print("  rep", rep, "rc", rc, "res %.2e" % res)
