"""n <= 128: the one-launch solve on the dataflow tile factorisation (default) against the round-2 kernel (BSFM_CHOL=streams) and LAPACK."""
import os, sys, subprocess
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
if len(sys.argv) > 1:
    import bundler_sfm_amd.sfm as B
    rng = np.random.default_rng(11)
    for n in (7, 63, 90, 126, 128):
        for cond in (1e2, 1e8):
            Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
            d = np.logspace(0, np.log10(cond), n)
            A = (Q * d) @ Q.T; A = 0.5 * (A + A.T)
            b = rng.standard_normal(n)
            rc, x = B.dense_chol_solve(A, b)
            xr = np.linalg.solve(A, b)
            print(f"n {n:4d} cond {cond:.0e} rc {rc} |x - x_lapack| / |x| = {np.abs(x - xr).max() / np.abs(xr).max():.2e}   residual {np.abs(A @ x - b).max() / (np.abs(A).max() * np.abs(x).max()):.2e}")
else:
    for mode in ("flow", "streams"):
        env = dict(os.environ); env["BSFM_CHOL"] = mode
        print("BSFM_CHOL =", mode, flush=True)
        subprocess.run([sys.executable, __file__, "child"], env=env)
