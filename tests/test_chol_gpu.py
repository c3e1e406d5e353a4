"""Dense FP64 Cholesky solve on the GPU (own MFMA-f64 tiled potrf) vs numpy / the oracle's dpotrf restatement."""
import numpy as np
import pytest

import oracle_util as O

pytestmark = pytest.mark.gpu


def spd(n, seed, cond_shift=1.0):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((n, n))
    A = A @ A.T + cond_shift * n * np.eye(n)
    # asymmetric-looking content (the guide's transposed-fragment trap): scale rows/cols differently
    d = 1.0 + np.arange(n) / n
    return A * d[:, None] * d[None, :], rng.standard_normal(n)


@pytest.mark.parametrize("n", [1, 7, 127, 128, 129, 450, 1000, 1799])
def test_solve_matches_numpy(gpu_bsfm, n):
    A, b = spd(n, n)
    rc, x = gpu_bsfm.dense_chol_solve(A, b)
    assert rc == 0
    ref = np.linalg.solve(A, b)
    assert np.abs(x - ref).max() <= 1e-10 * np.abs(ref).max()
    r = A @ x - b
    assert np.abs(r).max() <= 1e-9 * np.abs(b).max()


def test_matches_oracle_restatement(gpu_bsfm):
    import ctypes as C
    dp = C.POINTER(C.c_double)
    A, b = spd(300, 5)
    x0 = np.zeros(300)
    assert O.port().oracle_chol_solve(300, A.ctypes.data_as(dp), b.ctypes.data_as(dp), x0.ctypes.data_as(dp)) == 0
    rc, x = gpu_bsfm.dense_chol_solve(A, b)
    assert rc == 0 and np.abs(x - x0).max() <= 1e-11 * np.abs(x0).max()


@pytest.mark.parametrize("n,bad", [(50, 10), (300, 200), (300, 0)])
def test_not_positive_definite_reports_leading_minor(gpu_bsfm, n, bad):
    """dpotrf's info = order of the first non-positive-definite leading minor (sba_lapack.c:436-439)."""
    A, b = spd(n, 9)
    A[bad, bad] = -1.0
    rc, _ = gpu_bsfm.dense_chol_solve(A, b)
    assert rc == bad + 1


@pytest.mark.skipif(__import__("os").environ.get("BSFM_TEST_ROCSOLVER") != "1",
                    reason="rocSOLVER cross-check costs ~2 min of library initialisation; set BSFM_TEST_ROCSOLVER=1")
def test_cross_check_backend_agrees(gpu_bsfm):
    A, b = spd(700, 3)
    rc0, x0 = gpu_bsfm.dense_chol_solve(A, b, 0)
    rc1, x1 = gpu_bsfm.dense_chol_solve(A, b, 1)
    assert rc0 == 0 and rc1 == 0
    assert np.abs(x0 - x1).max() <= 1e-11 * np.abs(x1).max()
