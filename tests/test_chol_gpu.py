"""Dense FP64 Cholesky solve on the GPU (own MFMA-f64 tiled potrf) vs numpy / the oracle's dpotrf restatement."""
import os

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


def test_solution_is_bit_identical_from_run_to_run(gpu_bsfm):
    """The tile-dataflow factorisation hands its tasks to whichever workgroup is free, but the ORDER of the floating-point
    operations on every tile is fixed by the schedule (csrc/chol_flow_sched.h): repeated solves must agree to the last bit.  (A stale
    read of a panel tile -- the failure the agent-scope loads / stores of chol_flow.hip.h exist to prevent -- would show up here as an
    occasional difference.)"""
    A, b = spd(2900, 11)
    ref = None
    for _ in range(6):
        rc, x = gpu_bsfm.dense_chol_solve(A, b)
        assert rc == 0
        if ref is None:
            ref = x.copy()
        assert np.array_equal(x, ref)


@pytest.mark.parametrize("n,bad", [(50, 10), (300, 200), (300, 0), (1000, 700), (1000, 128), (1000, 999)])
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


@pytest.mark.timeout(900)
def test_solve_at_the_headline_order_9000(gpu_bsfm):
    """n = 9 000 = 71 tile columns: the size of the reduced camera system of BASELINE.json configs[2] -- one k_chol_flow launch
    of ~42 000 tile tasks (csrc/chol_flow.hip.h), persistent backward substitution over 71 workgroups."""
    n = 9000
    rng = np.random.default_rng(9000)
    # low-rank-plus-diagonal SPD matrix with a wide spectrum (building A A^T at this order on the host would take minutes)
    F = rng.standard_normal((n, 64))
    d = np.exp(rng.uniform(np.log(1e-2), np.log(1e2), n))
    A = (F * np.linspace(0.5, 2.0, 64)) @ F.T
    A[np.diag_indices(n)] += d
    b = rng.standard_normal(n)
    rc, x = gpu_bsfm.dense_chol_solve(A, b)
    assert rc == 0
    r = A @ x - b
    assert np.abs(r).max() <= 1e-11 * (np.abs(A).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max())     # backward error
    # forward check through Woodbury: A = D + G G^T with G = F diag(sqrt(w))
    G = F * np.sqrt(np.linspace(0.5, 2.0, 64))
    Dib = b / d; DiG = G / d[:, None]
    xw = Dib - DiG @ np.linalg.solve(np.eye(64) + G.T @ DiG, G.T @ Dib)
    assert np.abs(x - xw).max() <= 1e-9 * np.abs(xw).max()


@pytest.mark.skipif(os.environ.get("BSFM_TEST_HUGE") != "1", reason="31 232 unknowns = 244 tile columns: 7.8 GB on the host, minutes; set BSFM_TEST_HUGE=1")
@pytest.mark.timeout(1800)
def test_more_than_240_tile_columns(gpu_bsfm):
    """Rounds 1-3 refused systems of more than 240 tile columns (3 413 cameras): the persistent backward substitution needs one
    resident workgroup per tile column.  The dataflow path (round 4) runs it in waves of 240 columns."""
    n = 244 * 128
    rng = np.random.default_rng(31232)
    F = rng.standard_normal((n, 32))
    d = np.exp(rng.uniform(np.log(1e-1), np.log(1e1), n))
    w = np.linspace(0.5, 2.0, 32)
    A = (F * w) @ F.T
    A[np.diag_indices(n)] += d
    b = rng.standard_normal(n)
    rc, x = gpu_bsfm.dense_chol_solve(A, b)
    assert rc == 0
    G = F * np.sqrt(w)
    Dib = b / d; DiG = G / d[:, None]
    xw = Dib - DiG @ np.linalg.solve(np.eye(32) + G.T @ DiG, G.T @ Dib)       # Woodbury
    assert np.abs(x - xw).max() <= 1e-9 * np.abs(xw).max()


@pytest.mark.timeout(600)
def test_ill_conditioned_reduced_camera_system(gpu_bsfm):
    """The matrix the factorisation really sees: the reduced camera system of a CONNECTED scene (banded visibility, no camera
    held fixed, so the undamped S has the 7-dimensional gauge null space) with the small first-iteration damping
    mu = 1e-3 * max diag (sba_levmar.c:1124-1128).  The panel uses an explicit inverse of the diagonal factor tile
    (potrf.hip.h), so the check is the backward error a backward-stable dpotrf/dpotrs pair would deliver, and agreement with
    LAPACK's solution to within cond * eps."""
    import scipy.linalg as sl
    B = gpu_bsfm
    m, n = 400, 40000
    s = B.synth_ba(m, n, 8, banded=True)
    pb = B.Problem(n, m, s["rowptr"], s["colidx"], s["proj"], s["cams"], s["pts"], options=B.default_options(jacobian=1, verbose=0))
    ne0 = pb.normal_equations(mu=0.0)
    maxdiag = max(np.abs(np.einsum("jkk->jk", ne0["U"])).max(), np.abs(np.einsum("ikk->ik", ne0["V"])).max())
    # tau = 1e-3 (sfm.c:705) gives cond(S) ~ 1e3 on this scene; LM shrinks mu by up to 3x per accepted step (sba_levmar.c:1546-1549),
    # so later iterations factor systems many orders worse: take the smallest damping that still leaves S positive definite in FP64
    for mu_rel in (1e-9, 1e-8, 1e-7, 1e-6):
        ne = pb.normal_equations(mu=mu_rel * maxdiag)
        S, E = ne["S"], ne["E"]
        assert np.abs(S - S.T).max() <= 1e-12 * np.abs(S).max()
        S = np.tril(S) + np.tril(S, -1).T                # the factorisation reads the lower triangle only
        w = np.linalg.eigvalsh(S)
        if w[0] > 0 and w[-1] / w[0] < 1e12:
            break
    pb.close()
    cond = w[-1] / w[0]
    assert w[0] > 0 and cond > 1e6, cond                 # genuinely ill-conditioned, still positive definite
    rc, x = B.dense_chol_solve(S, E)
    assert rc == 0
    ref = sl.cho_solve(sl.cho_factor(S, lower=True), E)
    nrm = np.abs(S).sum(axis=1).max()
    assert np.abs(S @ x - E).max() <= 1e-11 * (nrm * np.abs(x).max() + np.abs(E).max())
    assert np.abs(x - ref).max() <= 20 * cond * np.finfo(float).eps * np.abs(ref).max()


def test_info_word_of_a_normal_and_of_an_indefinite_solve(gpu_bsfm):
    """dpotrf's info through the word the persistent kernels' time-outs are folded into (sba_lapack.c:374-485: info = k for the first
    non-positive leading minor): 0 for SPD systems of one to twenty tile columns, k for a planted negative pivot."""
    for n in (1, 129, 1000, 2500):
        A, b = spd(n, 40 + n)
        rc0, x0 = gpu_bsfm.dense_chol_solve(A, b)
        assert rc0 == 0
        ref = np.linalg.solve(A, b)
        assert np.abs(x0 - ref).max() <= 1e-10 * np.abs(ref).max()
        if n > 200:
            A[150, 150] = -1.0
            assert gpu_bsfm.dense_chol_solve(A, b)[0] == 151


def test_one_tile_solve_random_sizes(gpu_bsfm):
    """The one-tile launch (n <= 128: factorisation and both substitutions in one workgroup, the substitutions reading the inverse factor from LDS --
    round 6): random sizes against numpy, a failing pivot at the first / an inner / the last position, every system twice with the same bits.
    scripts/r6/one_tile_stress.py is the long form."""
    rng = np.random.default_rng(31)
    for c in range(40):
        n = int(rng.integers(1, 129))
        G = rng.standard_normal((n, n + 3))
        A = G @ G.T + (1e-3 + n * rng.random()) * np.eye(n)
        b = rng.standard_normal(n)
        rc, x = gpu_bsfm.dense_chol_solve(A, b)
        rc2, x2 = gpu_bsfm.dense_chol_solve(A, b)
        assert rc == 0 and rc2 == 0 and np.array_equal(x, x2), (c, n)
        ref = np.linalg.solve(A, b)
        assert np.abs(x - ref).max() <= 1e-13 * np.linalg.cond(A) * max(np.abs(ref).max(), 1e-300), (c, n)
    for n, k in ((40, 17), (128, 128), (100, 1), (1, 1)):
        A = np.eye(n) * 4.0
        A[k - 1, k - 1] = -1.0
        assert gpu_bsfm.dense_chol_solve(A, np.ones(n))[0] == k, (n, k)


def test_flow_stress_bit_identity(gpu_bsfm):
    """Random sizes and random tile envelopes through the dataflow launch (the dynamic check of the hand-off invariant stated at flow_tri,
    chol_flow.hip.h: write-once lines have one writer, rewritten data is only touched with agent-scope accesses): every solution
    against its scaled residual, and every system solved twice -- the same bits both times.  scripts/r4/flow_stress.py is the long
    form (150 cases)."""
    rng = np.random.default_rng(23)
    for c in range(24):
        n = int(rng.choice([rng.integers(129, 400), rng.integers(400, 1400), rng.integers(1400, 3000)], p=[0.4, 0.4, 0.2]))
        T = (n + 127) // 128
        env = c % 2 == 1
        if env:
            A = np.zeros((n, n))
            first = [max(0, I - int(rng.integers(0, max(1, min(T, 8))))) for I in range(T)]
            for I in range(T):
                r0, r1 = 128 * I, min(n, 128 * (I + 1))
                A[r0:r1, 128 * first[I]:r1] = rng.standard_normal((r1 - r0, r1 - 128 * first[I]))
            A = np.tril(A); A = A + A.T
        else:
            G = rng.standard_normal((n, min(n, 96)))
            A = G @ G.T
        A[np.diag_indices(n)] = np.abs(A).sum(axis=1) + 1.0
        b = rng.standard_normal(n)
        rc, x = gpu_bsfm.dense_chol_solve(A, b, backend=2 if env else 0)
        rc2, x2 = gpu_bsfm.dense_chol_solve(A, b, backend=2 if env else 0)
        assert rc == 0 and rc2 == 0 and x.tobytes() == x2.tobytes(), (c, n, env)
        assert np.abs(A @ x - b).max() <= 1e-12 * np.abs(A).max() * max(np.abs(x).max(), 1e-300), (c, n, env)


def test_info_is_the_first_failing_minor_even_when_diagonal_tiles_run_concurrently(gpu_bsfm):
    """Block-diagonal S (a legal envelope: tile columns with nothing below their diagonal tile): the POTRFs of independent diagonal
    tiles run at the same time on different chain workgroups, so the failure that happens first in TIME need not be the first in the
    MATRIX.  dpotrf reports the first failing leading minor (sba_lapack.c:374-485 passes it on): the smaller index wins."""
    n, blk = 1536, 384
    rng = np.random.default_rng(9)
    i, j = np.indices((n, n))
    A = rng.standard_normal((n, n)); A[(i // blk) != (j // blk)] = 0.0
    A = np.tril(A); A = A + A.T
    A[np.arange(n), np.arange(n)] = np.abs(A).sum(axis=1) + 1.0
    b = rng.standard_normal(n)
    for planted in ((1300, 70), (70, 1300), (900, 500, 1400), (1535, 385)):
        Ab = A.copy()
        for q in planted:
            Ab[q, q] = -1.0
        for rep in range(3):
            assert gpu_bsfm.dense_chol_solve(Ab, b, backend=2)[0] == min(planted) + 1, (planted, rep)


@pytest.mark.parametrize("hook,value", [("BSFM_FLOW_TEST_STALL", "9"), ("BSFM_FLOW_TEST_STALL_BWD", "3")], ids=["forward-task-never-signals", "backward-column-never-signals"])
def test_starved_dataflow_launch_falls_back_to_the_stream_schedule(gpu_bsfm, monkeypatch, capfd, hook, value):
    """The tile-dataflow launch (chol_flow.hip.h) waits on counters other workgroups of the same launch signal; on a GPU shared with
    another job those workgroups may not be resident in time.  The test hooks make ONE task never signal -- a bulk task of the
    factorisation, or one column of the backward substitution -- with the spin limit cut from 0.4 s to 20 ms: the waiters give up, the
    launch ends with the time-out word set, and the library repeats the solve on the stream-ordered schedule of rounds 1-3 (ordinary
    launches, nothing to wait for but stream order) with a warning instead of failing.  The answer is LAPACK's; a
    planted negative pivot still comes back as dpotrf's info."""
    import scipy.linalg as sl
    monkeypatch.setenv(hook, value); monkeypatch.setenv("BSFM_FLOW_SPIN_MS", "20")
    A, b = spd(1500, 77)
    rc, x = gpu_bsfm.dense_chol_solve(A, b)
    err = capfd.readouterr().err
    assert rc == 0 and "timed out" in err and "stream-ordered" in err, err[-500:]
    ref = sl.cho_solve(sl.cho_factor(A, lower=True), b)
    assert np.abs(x - ref).max() <= 1e-11 * np.abs(ref).max()
    A[700, 700] = -1.0
    assert gpu_bsfm.dense_chol_solve(A, b)[0] == 701
    monkeypatch.delenv(hook)
    capfd.readouterr()
    A, b = spd(1500, 78)
    rc, x = gpu_bsfm.dense_chol_solve(A, b)
    assert rc == 0 and "timed out" not in capfd.readouterr().err          # and without the hook nothing times out


def test_starved_dataflow_launch_inside_the_lm_loop(gpu_bsfm, monkeypatch, capfd):
    """The same inside bsfm_lm_iterate (what run_sfm runs): the attempt whose solve timed out is repeated from the point inversion on
    with the stream-ordered schedule -- no LM counter has moved -- and the problem stays on that schedule.  Same iterations, stop code,
    linear-system count and (to rounding: another summation order inside the tiles) the same parameters as an undisturbed run."""
    B = gpu_bsfm
    m, n = 72, 6000
    s = B.synth_ba(m, n, 8, banded=True)

    def run():
        pb = B.Problem(n, m, s["rowptr"], s["colidx"], s["proj"], s["cams"], s["pts"], options=B.default_options(jacobian=B.JAC_ANALYTIC, verbose=0, itmax=6))
        rc, info = pb.solve()
        p, _, _ = pb.download()
        fb = pb.phase_ms("flow_fallbacks")
        pb.close()
        return rc, np.array(info), p, fb

    ref = run()
    assert ref[3] == 0
    monkeypatch.setenv("BSFM_FLOW_TEST_STALL", "5"); monkeypatch.setenv("BSFM_FLOW_SPIN_MS", "20")
    capfd.readouterr()
    got = run()
    err = capfd.readouterr().err
    assert got[3] == 1 and "falling back to the stream-ordered schedule" in err, err[-400:]
    assert got[0] == ref[0] and list(got[1][5:10]) == list(ref[1][5:10])
    assert abs(got[1][1] - ref[1][1]) <= 1e-10 * ref[1][1]
    assert np.abs(got[2] - ref[2]).max() <= 1e-9 * np.abs(ref[2]).max()


@pytest.mark.parametrize("n,half_band,seed", [(700, 90, 1), (2500, 300, 2), (1500, 64, 3), (900, 1000, 4), (2000, 1, 5), (1536, 40, 6), (1300, 200, 7)])
def test_envelope_factorisation_of_a_banded_system(gpu_bsfm, n, half_band, seed):
    """The opt-in envelope solver's factorisation (potrf.hip.h with PotrfWorkspace::env_rows: every step of the tiled schedule only
    touches the tile rows inside the envelope of the matrix; Cholesky without pivoting creates no fill outside it) on banded SPD
    systems -- narrow bands (steps with ONE or NO tile row below the diagonal tile: the decoupled-column path), a band wider than the
    matrix (= dense), a ragged envelope -- against LAPACK and against the dense schedule on the same matrix; dpotrf's failure code
    is kept.  Test entry: bsfm_dense_chol_solve(..., backend = 2) derives the tile envelope from the zero pattern of A."""
    import scipy.linalg as sl
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((n, n))
    i, j = np.indices((n, n))
    A[np.abs(i - j) > half_band] = 0.0
    if seed == 3:                                                        # ragged: a few long rows far from the diagonal
        A[1200:1210, 100:140] = rng.standard_normal((10, 40))
    if seed in (6, 7):                                                   # block diagonal: tile columns with NOTHING below the diagonal tile
        blk = 512 if seed == 6 else 384                                  # (6: blocks end on tile boundaries; 7: they do not)
        A[(i // blk) != (j // blk)] = 0.0
    A = np.tril(A); A = A + A.T
    A[np.arange(n), np.arange(n)] = np.abs(A).sum(axis=1) + 1.0          # diagonally dominant => SPD, same pattern
    b = rng.standard_normal(n)
    rc2, x2 = gpu_bsfm.dense_chol_solve(A, b, backend=2)
    rc0, x0 = gpu_bsfm.dense_chol_solve(A, b, backend=0)
    assert rc2 == 0 and rc0 == 0
    ref = sl.cho_solve(sl.cho_factor(A, lower=True), b)
    assert np.abs(x2 - ref).max() <= 1e-11 * np.abs(ref).max()
    assert np.abs(x2 - x0).max() <= 1e-12 * np.abs(ref).max()
    if n > 400:
        A[333, 333] = -1.0
        assert gpu_bsfm.dense_chol_solve(A, b, backend=2)[0] == 334


# ---- the dynamic bulk (csrc/chol_dyn.hip.h, chol_dyn_plan.h; opt-in BSFM_FLOW_SCHED=dynamic): same roles, served from per-half-tile
# state words (scan, claim by atomic OR, release by store) instead of the host-simulated ticket order.  Round 6 measured it slower than the
# static order (profiles/r06_dynamic_bulk_*.txt), so it is not the default; it stays correct and tested.
@pytest.mark.parametrize("n", [129, 300, 450, 1000, 1799, 2900])
def test_dynamic_bulk_matches_numpy_and_is_independent_of_the_number_of_workgroups(gpu_bsfm, monkeypatch, n):
    """Which workgroup applies which panels to a tile, and in how many passes, is decided at run time -- but every role applies the panels
    of a tile in ascending order through the same accumulation chains, and which role applies which panel is fixed on the host: the bits
    must not depend on the batching.  Same system with 512, 96 and 40 workgroups (different interleavings, different batch sizes)."""
    monkeypatch.setenv("BSFM_FLOW_SCHED", "dynamic")
    A, b = spd(n, 100 + n)
    ref = np.linalg.solve(A, b)
    first = None
    for wgs in ("512", "96", "40"):
        monkeypatch.setenv("BSFM_FLOW_WGS", wgs)
        rc, x = gpu_bsfm.dense_chol_solve(A, b)
        assert rc == 0
        assert np.abs(x - ref).max() <= 1e-10 * np.abs(ref).max()
        if first is None:
            first = x.copy()
        assert x.tobytes() == first.tobytes(), (n, wgs)


def test_dynamic_bulk_envelopes_info_and_starved_launch(gpu_bsfm, monkeypatch, capfd):
    import scipy.linalg as sl
    monkeypatch.setenv("BSFM_FLOW_SCHED", "dynamic")
    rng = np.random.default_rng(31)
    for n, half_band, blk in ((2500, 300, 0), (1500, 64, 0), (1536, 40, 512), (1300, 200, 384)):
        A = rng.standard_normal((n, n))
        i, j = np.indices((n, n))
        A[np.abs(i - j) > half_band] = 0.0
        if blk:
            A[(i // blk) != (j // blk)] = 0.0
        A = np.tril(A); A = A + A.T
        A[np.arange(n), np.arange(n)] = np.abs(A).sum(axis=1) + 1.0
        b = rng.standard_normal(n)
        rc, x = gpu_bsfm.dense_chol_solve(A, b, backend=2)
        ref = sl.cho_solve(sl.cho_factor(A, lower=True), b)
        assert rc == 0 and np.abs(x - ref).max() <= 1e-11 * np.abs(ref).max(), (n, half_band, blk)
        A[333, 333] = -1.0
        assert gpu_bsfm.dense_chol_solve(A, b, backend=2)[0] == 334
    # a claim that is never released: the waiters give up, the solve is repeated on the stream-ordered schedule
    monkeypatch.setenv("BSFM_FLOW_TEST_STALL", "9"); monkeypatch.setenv("BSFM_FLOW_SPIN_MS", "20")
    A, b = spd(1500, 77)
    capfd.readouterr()
    rc, x = gpu_bsfm.dense_chol_solve(A, b)
    err = capfd.readouterr().err
    assert rc == 0 and "timed out" in err and "stream-ordered" in err, err[-500:]
    ref = sl.cho_solve(sl.cho_factor(A, lower=True), b)
    assert np.abs(x - ref).max() <= 1e-11 * np.abs(ref).max()
