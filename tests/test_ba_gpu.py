"""GPU parity tests of the bundle-adjustment hot path: every check calls the HIP kernels through the C-ABI
(libbsfm_hip.so) and compares with the CPU oracle (oracle/liboracle_port.so, pinned to the reference in
tests/test_oracle.py) and with the committed reference fixtures (tests/golden/*.npz).

Tolerances (FP64 everywhere; SURVEY/BASELINE parity gates):
  cost            <= 1e-9  relative
  first-step dp   <= 1e-7  relative            (oracle mode B: same Jacobian on both sides)
  block values    <= 1e-10 relative to the block-array maximum
  index / counters (iterations, solves, stop code) bit-exact for the first iterations
"""
import os
import sys

import numpy as np
import pytest

import oracle_util as O
from test_oracle import G, K, CASES, load_case

pytestmark = pytest.mark.gpu


def make_problem(B, c, jac, **kw):
    opt = B.default_options(jacobian=jac, verbose=0, **kw)
    pb = B.Problem(c["n"], c["m"], c["rowptr"], c["colidx"], c["proj"], c["cams"], c["pts"], mcon=c["ncons"],
                   est_focal_length=c["est"], undistort=c["und"], use_constraints=c["cons"], options=opt)
    return pb


@pytest.mark.parametrize("name", CASES)
def test_residuals_and_cost(gpu_bsfm, name):
    B = gpu_bsfm
    c = load_case(name)
    pb = make_problem(B, c, B.JAC_ANALYTIC)
    e, cost = pb.residuals()
    q = O.port_run_sfm(c["n"], c["m"], c["vm"], c["proj"], c["cams"], c["pts"], itmax=1, jac_mode=1, ncons=c["ncons"],
                       est_focal=c["est"], undistort=c["und"], use_constraints=c["cons"])
    assert abs(cost - q["info"][0]) <= 1e-12 * q["info"][0]
    assert abs(cost - G[f"{name}_an_it1_info"][0]) <= 1e-12 * cost        # reference's initial error
    assert abs(np.dot(e, e) - (cost if not c["cons"] else np.dot(e, e))) <= 1e-9 * cost
    pb.close()


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("jac", [0, 1])
def test_normal_equation_blocks(gpu_bsfm, name, jac):
    """J, U, ea, V, eb, S, E at the initial p with the first damping value, vs the oracle's dump."""
    B = gpu_bsfm
    c = load_case(name)
    q = O.port_run_sfm(c["n"], c["m"], c["vm"], c["proj"], c["cams"], c["pts"], itmax=1, jac_mode=jac, ncons=c["ncons"],
                       est_focal=c["est"], undistort=c["und"], use_constraints=c["cons"], want_dumps=True)
    mu = float(q["mu"][0])
    pb = make_problem(B, c, jac)
    ne = pb.normal_equations(mu, want_J=True)
    tolJ = 1e-12 if jac == 1 else 2e-9      # FD columns amplify 1-ulp projection differences by 1/d = 1e4..1e6
    tol = 1e-11 if jac == 1 else 5e-9
    for key, t in (("J", tolJ), ("U", tol), ("ea", tol), ("V", tol), ("eb", tol), ("S", tol), ("E", tol)):
        ref = q[key]; got = ne[key].reshape(ref.shape)
        if key == "U" and c["ncons"]:
            ref = ref[c["ncons"]:]; got = got[c["ncons"]:]
        if key == "ea" and c["ncons"]:
            ref = ref[c["ncons"]:]; got = got[c["ncons"]:]
        scale = np.abs(ref).max()
        assert np.abs(got - ref).max() <= t * scale, (key, np.abs(got - ref).max() / scale)
    # S symmetric and E consistent with a dense host solve of the same system
    assert np.abs(ne["S"] - ne["S"].T).max() <= 1e-12 * np.abs(ne["S"]).max()
    pb.close()


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("tag,jac", [("fd", 0), ("an", 1)])
def test_first_iterations_match_reference_fixture(gpu_bsfm, name, tag, jac):
    """One and three LM iterations against the REFERENCE's own iterates (golden fixture)."""
    B = gpu_bsfm
    c = load_case(name)
    for it in (1, 3):
        pb = make_problem(B, c, jac, itmax=it)
        rc, info = pb.solve()
        p, _, _ = pb.download()
        gp, gi = G[f"{name}_{tag}_it{it}_p"], G[f"{name}_{tag}_it{it}_info"]
        assert rc == it
        assert info[5] == gi[5] and info[6] == gi[6] and info[9] == gi[9]           # counters: bit-exact
        assert info[7] == gi[7] and info[8] == gi[8]
        assert abs(info[0] - gi[0]) <= 1e-12 * gi[0]
        assert abs(info[1] - gi[1]) <= 1e-9 * gi[1]
        dp_ref = gp - G[f"{name}_{tag}_it1_p"] * 0
        tol = 1e-7 if jac == 1 else 1e-6
        assert np.abs(p - gp).max() <= tol * np.abs(gp).max(), np.abs(p - gp).max() / np.abs(gp).max()
        pb.close()


@pytest.mark.parametrize("name", CASES)
def test_full_lm_converges_to_reference_solution(gpu_bsfm, name):
    B = gpu_bsfm
    c = load_case(name)
    for tag, jac in (("fd", 0), ("an", 1)):
        pb = make_problem(B, c, jac)
        rc, info = pb.solve()
        gi = G[f"{name}_{tag}_it150_info"]
        assert rc >= 0
        assert abs(info[1] - gi[1]) <= 1e-6 * gi[1], (info, gi)
        assert abs(info[5] - gi[5]) <= 1          # stop rule 4 fires on rounding noise in the reference
        pb.close()


@pytest.mark.parametrize("name", ["s9", "s9c", "s7", "s6"])
def test_run_sfm_drop_in_matches_reference_run_sfm(gpu_bsfm, name):
    """The drop-in boundary itself: same arguments as lib/sfm-driver/sfm.h:68-86, cameras/points updated in place,
    against the reference run_sfm's outputs (fixture).  The BA gauge is free (no fixed camera), so parameters are
    compared after both solutions have converged: reprojection cost and per-camera focal/distortion/centre."""
    B = gpu_bsfm
    c = load_case(name)
    cams = B.copy_cameras(c["cams"]); pts = c["pts"].copy()
    rc, info = B.run_sfm(c["n"], c["m"], c["ncons"], c["vm"], c["proj"], c["est"], 0, c["und"], 1, cams, pts,
                         use_constraints=c["cons"], eps2=1e-12, options=B.default_options(verbose=0))
    assert rc >= 0
    ref_f = G[f"{name}_run_cam_f"]; ref_t = G[f"{name}_run_cam_t"]; ref_pts = G[f"{name}_run_pts"]
    f = np.array([cm.f for cm in cams]); t = np.array([list(cm.t) for cm in cams])
    assert np.abs(f - ref_f).max() <= 1e-4 * np.abs(ref_f).max()
    assert np.abs(t - ref_t).max() <= 1e-4 * max(1.0, np.abs(ref_t).max())
    assert np.abs(pts - ref_pts).max() <= 1e-4 * max(1.0, np.abs(ref_pts).max())
    for cm in cams:
        Rm = np.array(list(cm.R)).reshape(3, 3)
        assert np.abs(Rm @ Rm.T - np.eye(3)).max() < 1e-12
        assert cm.f_scale == 1.0 and cm.k_scale == 1.0      # reset on exit, sfm.c:918-921


def test_kermit_replay(gpu_bsfm):
    """The reference's only golden artefact (examples/kermit/results.example/bundle.out) replayed through the GPU core."""
    B = gpu_bsfm
    m, n = len(K["cam_f"]), len(K["pts"]) // 3
    cams = O.arrays_to_cams(K["cam_R"], K["cam_t"], K["cam_f"], K["cam_k"], K["cam_constrained"],
                            K["cam_constraints"], K["cam_weights"])
    for tag, jac in (("fd", 0), ("an", 1)):
        for it in (1, 3, 150):
            opt = B.default_options(jacobian=jac, verbose=0, itmax=it)
            opt.opts[2] = 1e-12
            pb = B.Problem(n, m, K["rowptr"], K["colidx"], K["proj"], cams, K["pts"], use_constraints=1, options=opt)
            rc, info = pb.solve()
            p, _, _ = pb.download()
            gi, gp = K[f"{tag}_it{it}_info"], K[f"{tag}_it{it}_p"]
            assert abs(info[1] - gi[1]) <= 1e-8 * gi[1], (info, gi)
            assert info[6] == gi[6] and info[5] == gi[5]
            tol = 1e-7 if it < 150 else 1e-4
            assert np.abs(p - gp).max() <= tol * np.abs(gp).max()
            pb.close()
    # drop-in run_sfm from the perturbed state lands on the golden reconstruction's focal lengths
    vm = B.dense_vmask(n, m, K["rowptr"], K["colidx"])
    c2 = B.copy_cameras(cams); pts = K["pts"].copy()
    rc, info = B.run_sfm(n, m, 0, vm, K["proj"], 1, 0, 1, 1, c2, pts, use_constraints=1, eps2=1e-12,
                         options=B.default_options(verbose=0))
    assert rc == 11 and info[6] == 2                      # same as the reference run (fixture)
    assert np.abs(np.array([cm.f for cm in c2]) - K["run_cam_f"]).max() <= 1e-5 * K["run_cam_f"].max()
    assert np.abs(pts - K["run_pts"]).max() <= 1e-5 * np.abs(K["run_pts"]).max()


def test_config2_single_iteration_vs_oracle(gpu_bsfm):
    """BASELINE.json configs[1]: 50 cameras / 10 000 points / 100 000 observations, one LM iteration."""
    B = gpu_bsfm
    m, n = 50, 10000
    s = B.synth_ba(m, n, 10)
    vm = B.dense_vmask(n, m, s["rowptr"], s["colidx"])
    for jac in (1, 0):
        q = O.port_run_sfm(n, m, vm, s["proj"], s["cams"], s["pts"], itmax=1, jac_mode=jac)
        pb = B.Problem(n, m, s["rowptr"], s["colidx"], s["proj"], s["cams"], s["pts"],
                       options=B.default_options(jacobian=jac, verbose=0, itmax=1))
        rc, info = pb.solve()
        p, _, _ = pb.download()
        assert info[5] == 1 and info[9] == q["info"][9]
        assert abs(info[0] - q["info"][0]) <= 1e-12 * q["info"][0]
        assert abs(info[1] - q["info"][1]) <= 1e-9 * q["info"][1]
        p0 = np.concatenate([np.zeros(0)])
        dp_ref = q["p"]
        tol = 1e-7 if jac == 1 else 1e-6
        assert np.abs(p - q["p"]).max() <= tol * np.abs(q["p"]).max()
        pb.close()


def test_two_pass_backsub_below_256_cameras_vs_oracle(gpu_bsfm):
    """120 cameras / 45 000 points / 450 000 observations, one LM iteration against the oracle: the size class where the back-substitution is two
    passes (from 400 000 observations) AND the trial point's camera table is built by the workgroup appended to k_backsub (below 256 cameras) --
    a combination neither the 50-camera case above nor the 1 000-camera fixtures reach."""
    B = gpu_bsfm
    m, n = 120, 45000
    s = B.synth_ba(m, n, 10)
    vm = B.dense_vmask(n, m, s["rowptr"], s["colidx"])
    q = O.port_run_sfm(n, m, vm, s["proj"], s["cams"], s["pts"], itmax=1, jac_mode=1)
    pb = B.Problem(n, m, s["rowptr"], s["colidx"], s["proj"], s["cams"], s["pts"], options=B.default_options(jacobian=1, verbose=0, itmax=1))
    rc, info = pb.solve()
    p, _, _ = pb.download()
    pb.close()
    assert info[5] == 1 and info[9] == q["info"][9]
    assert abs(info[0] - q["info"][0]) <= 1e-12 * q["info"][0]
    assert abs(info[1] - q["info"][1]) <= 1e-9 * q["info"][1]
    assert np.abs(p - q["p"]).max() <= 1e-7 * np.abs(q["p"]).max()


def test_large_problem_properties(gpu_bsfm):
    """Size-independent properties at a size the CPU oracle cannot follow (200 cams / 50k pts / 500k obs):
    cost decreases monotonically over accepted steps, S is symmetric with U on its diagonal blocks when no point is
    shared, and solving twice from the same state is bit-identical (no atomics anywhere in the path)."""
    B = gpu_bsfm
    m, n = 200, 50000
    s = B.synth_ba(m, n, 10)
    outs = []
    for _ in range(2):
        pb = B.Problem(n, m, s["rowptr"], s["colidx"], s["proj"], s["cams"], s["pts"],
                       options=B.default_options(jacobian=1, verbose=0, itmax=6))
        pb.lm_begin()
        costs = []
        for _ in range(6):
            pb.lm_iterate(1)
            rc, info = pb.lm_finish()
            costs.append(info[1])
        p, _, _ = pb.download()
        outs.append((p, costs))
        assert all(b <= a for a, b in zip(costs, costs[1:]))
        assert costs[-1] < 0.05 * info[0]
        pb.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1]


def test_edge_cases(gpu_bsfm):
    B = gpu_bsfm
    # fewer measurements than unknowns: SBA_ERROR (sba_levmar.c:647-650)
    s = B.synth_ba(6, 10, 2)
    pb = B.Problem(10, 6, s["rowptr"], s["colidx"], s["proj"], s["cams"], s["pts"], options=B.default_options(verbose=0))
    assert pb.lm_begin() == -1
    pb.close()
    # a camera without any observation and ragged rows (2..5 views)
    rng = np.random.default_rng(3)
    m, n = 7, 80
    base = B.synth_ba(m, n, 5)
    rows = []; proj = []
    ci = base["colidx"].reshape(n, 5); pr = base["proj"].reshape(n, 5, 2)
    for i in range(n):
        keep = [q for q in range(5) if ci[i, q] != 3]
        keep = keep[: rng.integers(2, len(keep) + 1)]
        rows.append(ci[i, keep]); proj.append(pr[i, keep])
    rowptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    colidx = np.concatenate(rows).astype(np.int32); proj = np.concatenate(proj).ravel()
    vm = B.dense_vmask(n, m, rowptr, colidx)
    q = O.port_run_sfm(n, m, vm, proj, base["cams"], base["pts"], itmax=2, jac_mode=1)
    pb = B.Problem(n, m, rowptr, colidx, proj, base["cams"], base["pts"], options=B.default_options(jacobian=1, verbose=0, itmax=2))
    rc, info = pb.solve()
    p, _, _ = pb.download()
    assert info[5] == q["info"][5] and info[9] == q["info"][9]
    assert abs(info[1] - q["info"][1]) <= 1e-9 * q["info"][1]
    assert np.abs(p - q["p"]).max() <= 1e-7 * np.abs(q["p"]).max()
    pb.close()
    # an underdetermined problem (SBA's "fewer measurements than unknowns", sba_levmar.c:724-728) fails loudly and leaves
    # the inputs untouched
    cams = B.copy_cameras(base["cams"]); pts = base["pts"][:3].copy()
    k1 = int(rowptr[1])
    rc, _ = B.run_sfm(1, m, 0, vm[:1], proj[:2 * k1], 1, 0, 1, 1, cams, pts, options=B.default_options(verbose=0))
    assert rc == -1 and np.array_equal(pts, base["pts"][:3])
    assert all(list(a.t) == list(b.t) and a.f == b.f for a, b in zip(cams, base["cams"]))


def test_point_constraints(gpu_bsfm):
    B = gpu_bsfm
    c = load_case("s9")
    pc = np.zeros(3 * c["n"]); pc[:30] = c["pts"][:30] + 0.001
    q = O.port_run_sfm(c["n"], c["m"], c["vm"], c["proj"], c["cams"], c["pts"], itmax=3, jac_mode=1,
                       point_constraints=pc, point_w=0.5)
    pb = B.Problem(c["n"], c["m"], c["rowptr"], c["colidx"], c["proj"], c["cams"], c["pts"], point_constraints=pc,
                   point_constraint_weight=0.5, options=B.default_options(jacobian=1, verbose=0, itmax=3))
    rc, info = pb.solve()
    p, _, _ = pb.download()
    assert abs(info[0] - q["info"][0]) <= 1e-12 * q["info"][0]
    assert abs(info[1] - q["info"][1]) <= 1e-9 * q["info"][1]
    assert np.abs(p - q["p"]).max() <= 1e-7 * np.abs(q["p"]).max()
    pb.close()


@pytest.mark.parametrize("name", ["s9", "s9c", "s7"])
def test_covariance_export_matches_reference(gpu_bsfm, name):
    """run_sfm's optional Vout/Sout/Uout/Wout (sfm.h:68-86; filled by sba_levmar.c:1633-2026 at the solution,
    undamped) against the reference's own export after the same 3 analytic-Jacobian iterations
    (tests/golden/export_golden.npz, generated by tests/golden/make_golden.py:exports from oracle/_ref)."""
    B = gpu_bsfm
    X = np.load(os.path.join(os.path.dirname(__file__), "golden", "export_golden.npz"))
    c = load_case(name)
    m, n = c["m"], c["n"]
    cnp = 6 + c["est"] + 2 * c["und"]
    cams = B.copy_cameras(c["cams"]); pts = c["pts"].copy()
    V = np.zeros((n, 3, 3)); S = np.zeros((m * cnp, m * cnp)); U = np.zeros((m, cnp, cnp)); W = np.zeros((m * cnp, 3 * n))
    rc, info = B.run_sfm(n, m, c["ncons"], c["vm"], c["proj"], c["est"], 0, c["und"], 1, cams, pts,
                         use_constraints=c["cons"], eps2=1e-12, Vout=V, Sout=S, Uout=U, Wout=W,
                         options=B.default_options(jacobian=B.JAC_ANALYTIC, verbose=0, itmax=3))
    assert rc >= 0
    assert abs(info[1] - X[f"{name}_info"][1]) <= 1e-9 * X[f"{name}_info"][1]
    for got, key in ((U, "U"), (V, "V"), (S, "S"), (W, "W")):
        ref = X[f"{name}_{key}"]
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 1e-8 * np.abs(ref).max(), key


@pytest.mark.parametrize("name", ["s9", "s9c", "s9m", "s7", "s6"])
def test_sba_level_sibling_entry_matches_reference_sba(gpu_bsfm, name):
    """SURVEY 8(b) secondary boundary: bsfm_sba_motstr_levmar takes sba_motstr_levmar's argument list (sba.h:95-108 +
    Bundler's constraint / V,S,U,W arguments) with a camera-model id in place of the host callbacks.  Called exactly
    like oracle/ref_harness.c:ref_sba_motstr calls the reference (scaled parameter vector, scaled constraints) it must
    reproduce the reference's parameter vector after 3 analytic-Jacobian iterations (fixture <case>_an_it3_p)."""
    import ctypes as C
    from bundler_sfm_amd import _lib as L
    B = gpu_bsfm
    c = load_case(name)
    m, n, est, und = c["m"], c["n"], c["est"], c["und"]
    cnp = 6 + est + 2 * und
    ca = O.cams_to_arrays(c["cams"])
    p = np.zeros(m * cnp + 3 * n)
    for j in range(m):
        a = p[j * cnp:(j + 1) * cnp]
        a[0:3] = ca["t"][j]
        col = 6
        if est:
            a[6] = ca["f"][j] * 0.001; col = 7
        if und:
            a[col:col + 2] = ca["k"][j] * 5.0
    p[m * cnp:] = c["pts"]
    p0_packed = p.copy()
    Rinit = np.ascontiguousarray(ca["R"].reshape(m, 9)); finit = np.ascontiguousarray(ca["f"], np.float64)
    ptsc = np.ascontiguousarray(c["pts"], np.float64)
    md = L.SnavelyModel(est, und, 1, Rinit.ctypes.data_as(C.POINTER(C.c_double)), finit.ctypes.data_as(C.POINTER(C.c_double)),
                        ptsc.ctypes.data_as(C.POINTER(C.c_double)))
    cons = None; keep = []
    if c["cons"]:
        cons = (L.CameraConstraints * m)()
        for j in range(m):
            con = np.ascontiguousarray(ca["constrained"][j][:cnp], np.uint8)
            val = np.array(ca["constraints"][j][:cnp], np.float64); w = np.array(ca["weights"][j][:cnp], np.float64)
            if est:
                val[6] *= 0.001; w[6] *= 1.0 / (0.001 * 0.001)          # sfm.c:721-754, as ref_harness.c does
            if und:
                val[7:9] *= 5.0; w[7:9] *= 1.0 / 25.0
            keep += [con, val, w]
            cons[j].constrained = C.cast(con.ctypes.data, C.POINTER(C.c_char))
            cons[j].constraints = val.ctypes.data_as(C.POINTER(C.c_double)); cons[j].weights = w.ctypes.data_as(C.POINTER(C.c_double))
    opts = (C.c_double * 6)(1e-3, 1e-10, 1e-12, 1e-12, 0.0, 4e-2)
    info = np.zeros(10)
    os.environ["BSFM_JACOBIAN"] = "analytic"
    try:
        fn = B.lib.bsfm_sba_motstr_levmar
        fn.restype = C.c_int
        vm = np.ascontiguousarray(c["vm"], np.uint8); proj = np.ascontiguousarray(c["proj"], np.float64)
        rc = fn(n, m, c["ncons"], vm.ctypes.data_as(C.c_char_p), p.ctypes.data_as(C.POINTER(C.c_double)), cnp, 3,
                proj.ctypes.data_as(C.POINTER(C.c_double)), None, 2, 1, C.byref(md), 3, 0, opts,
                info.ctypes.data_as(C.POINTER(C.c_double)), 1 if cons is not None else 0, cons, 0, None, None, None, None, None)
    finally:
        del os.environ["BSFM_JACOBIAN"]
    gi, gp = G[f"{name}_an_it3_info"], G[f"{name}_an_it3_p"]
    assert rc == int(gi[5]) == 3
    assert abs(info[1] - gi[1]) <= 1e-9 * gi[1]
    assert np.abs(p - gp).max() <= 1e-8 * np.abs(gp).max()
    # camera-only sibling (sba_mot_levmar's argument list) against the reference's mot fixture
    if name in ("s9", "s9c", "s9m", "s7"):
        pm = p0_packed[:m * cnp].copy()
        os.environ["BSFM_JACOBIAN"] = "analytic"
        try:
            fm = B.lib.bsfm_sba_mot_levmar
            fm.restype = C.c_int
            rcm = fm(n, m, c["ncons"], vm.ctypes.data_as(C.c_char_p), pm.ctypes.data_as(C.POINTER(C.c_double)), cnp,
                     proj.ctypes.data_as(C.POINTER(C.c_double)), None, 2, 1, C.byref(md), 3, 0, opts,
                     info.ctypes.data_as(C.POINTER(C.c_double)), 1 if cons is not None else 0, cons)
        finally:
            del os.environ["BSFM_JACOBIAN"]
        assert rcm == 3 and list(info[5:10]) == list(MOT[f"{name}_an_it3_info"][5:10])
        assert np.abs(pm - MOT[f"{name}_an_it3_p"]).max() <= 1e-8 * np.abs(MOT[f"{name}_an_it3_p"]).max()
    # refusals leave p untouched
    p2 = p.copy()
    assert fn(n, m, 0, vm.ctypes.data_as(C.c_char_p), p2.ctypes.data_as(C.POINTER(C.c_double)), cnp, 3,
              proj.ctypes.data_as(C.POINTER(C.c_double)), None, 2, 7, C.byref(md), 3, 0, opts, None, 0, None, 0, None,
              None, None, None, None) == -1
    assert np.array_equal(p2, p)


def _iround(x):
    return int(x + 0.5)


@pytest.mark.parametrize("with_pcons", [False, True])
def test_post_solve_outlier_statistics(gpu_bsfm, with_pcons):
    """the statistics RunSFM_SBA computes after every run_sfm (src/Bundle.cpp:659-913) from the resident
    problem.  CPU side: every observation projected with the oracle's camera model, then the reference's formulas
    restated in numpy (kth_element_copy = k-th smallest, 0.0 when k >= n; thresh = clamp(2.4 * kth80, 8, 16); a point is
    an outlier if one observation exceeds its camera's threshold; constrained points (x != 0) are exempt)."""
    import ctypes as C
    B = gpu_bsfm
    c = load_case("band")
    m, n = c["m"], c["n"]
    rng = np.random.default_rng(5)
    proj = c["proj"].copy().reshape(-1, 2)
    bad = rng.choice(len(proj), 40, replace=False)
    proj[bad] += rng.normal(0, 30.0, (40, 2))                     # a few gross outliers
    pcons = None
    if with_pcons:
        pcons = np.zeros((n, 3)); ids = np.repeat(np.arange(n), np.diff(c["rowptr"]))[bad[:10]]
        pcons[ids] = c["pts"].reshape(-1, 3)[ids] + 0.01           # exempt some of the outliers' points
    opt = B.default_options(jacobian=B.JAC_ANALYTIC, verbose=0, itmax=5)
    pb = B.Problem(n, m, c["rowptr"], c["colidx"], proj.ravel(), c["cams"], c["pts"], est_focal_length=c["est"],
                   undistort=c["und"], use_constraints=c["cons"], point_constraints=pcons, point_constraint_weight=1.0,
                   options=opt)
    pb.solve()
    st = pb.outlier_stats(8.0, 16.0)
    p, cams, pts = pb.download()
    pb.close()
    # CPU restatement
    cnp = 9
    ca = O.cams_to_arrays(c["cams"])
    dp = C.POINTER(C.c_double)
    dist = np.zeros(len(proj))
    cam_of = c["colidx"]; pt_of = np.repeat(np.arange(n), np.diff(c["rowptr"]))
    x = np.zeros(2)
    for k in range(len(proj)):
        j, i = cam_of[k], pt_of[k]
        a = np.ascontiguousarray(p[j * cnp:(j + 1) * cnp]); b = np.ascontiguousarray(p[m * cnp + 3 * i:m * cnp + 3 * i + 3])
        Rj = np.ascontiguousarray(ca["R"][j])
        O.port().oracle_project(c["est"], c["und"], 1, Rj.ctypes.data_as(dp), float(ca["f"][j]), a.ctypes.data_as(dp),
                                b.ctypes.data_as(dp), x.ctypes.data_as(dp))
        dist[k] = np.hypot(proj[k, 0] - x[0], proj[k, 1] - x[1])
    thresh = np.zeros(m)
    for j in range(m):
        d = np.sort(dist[cam_of == j]); nj = len(d)
        k80, k50 = _iround(0.8 * nj), _iround(0.5 * nj)
        v80 = d[k80] if k80 < nj else 0.0
        v50 = d[k50] if k50 < nj else 0.0
        thresh[j] = min(max(2.4 * v80, 8.0), 16.0)
        assert st["nobs"][j] == nj
        assert abs(st["kth80"][j] - v80) <= 1e-9 * max(1.0, v80) and abs(st["kth50"][j] - v50) <= 1e-9 * max(1.0, v50)
        assert abs(st["thresh"][j] - thresh[j]) <= 1e-9 * thresh[j]
        assert abs(st["mean"][j] - d.mean()) <= 1e-9 * max(1.0, d.mean())
    assert abs(st["global_mean"] - dist.mean()) <= 1e-9 * dist.mean()
    flag = np.zeros(n, np.uint8); err = np.zeros(n)
    for i in range(n):
        if pcons is not None and pcons[i, 0] != 0.0:
            continue
        for k in range(c["rowptr"][i], c["rowptr"][i + 1]):
            if dist[k] > thresh[cam_of[k]]:
                flag[i] = 1; err[i] = dist[k]; break
    margin = np.abs(dist - thresh[cam_of]).min()
    assert margin > 1e-6                                          # no observation sits on a threshold: flags must agree exactly
    assert np.array_equal(st["outlier"], flag)
    assert flag.sum() >= 10 and (pcons is None or flag.sum() < 40)
    assert np.abs(st["err"] - err).max() <= 1e-9 * max(1.0, err.max())


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("with_pcons", [False, True])
def test_post_solve_outlier_statistics_vs_reference_routines(gpu_bsfm, with_pcons):
    """SURVEY 8(f).1 pinned to the REFERENCE'S OWN routines (oracle/_ref): the statistics loop of RunSFM_SBA (src/Bundle.cpp:659-913)
    driven, statement for statement, through the reference's sfm_project_rd (lib/sfm-driver/sfm.c:302-380, called as
    Bundle.cpp:726-739 calls it: K = diag(f, f, 1), the camera's own R / t / k) and kth_element_copy
    (lib/imagelib/qsort.c:152-203) -- not through a numpy re-statement: per-camera 80th-percentile distance,
    thresh = CLAMP(1.2 * NUM_STDDEV * med, 8, 16), mean, median, outlier list in camera-then-key order with the
    constrained-point exemption, first recorded reprojection error per point."""
    B = gpu_bsfm
    c = load_case("band")
    m, n = c["m"], c["n"]
    rng = np.random.default_rng(11)
    proj = c["proj"].copy().reshape(-1, 2)
    bad = rng.choice(len(proj), 40, replace=False)
    proj[bad] += rng.normal(0, 30.0, (40, 2))
    pt_of = np.repeat(np.arange(n), np.diff(c["rowptr"]))
    pcons = None
    if with_pcons:
        pcons = np.zeros((n, 3)); ids = pt_of[bad[:10]]
        pcons[ids] = c["pts"].reshape(-1, 3)[ids] + 0.01
    opt = B.default_options(jacobian=B.JAC_ANALYTIC, verbose=0, itmax=5)
    pb = B.Problem(n, m, c["rowptr"], c["colidx"], proj.ravel(), c["cams"], c["pts"], est_focal_length=c["est"],
                   undistort=c["und"], use_constraints=c["cons"], point_constraints=pcons, point_constraint_weight=1.0,
                   options=opt)
    pb.solve()
    st = pb.outlier_stats(8.0, 16.0)
    _, cams, pts = pb.download()               # cameras as run_sfm hands them back: R <- dR(w) R, t = centre, f, k, scales 1
    pb.close()
    pts = pts.reshape(-1, 3)
    cam_of = c["colidx"]
    outliers, errors = [], []
    total, count = 0.0, 0
    for j in range(m):                         # "for (int i = 0; i < num_cameras; i++)", Bundle.cpp:669
        ks = np.nonzero(cam_of == j)[0]        # the camera's keys with a point, in key order (= ascending point index here)
        dists = np.zeros(len(ks))
        for t, k in enumerate(ks):
            pr = O.ref_project_rd(cams[j], pts[pt_of[k]], undistort=c["und"], explicit_centers=1)       # Bundle.cpp:736-739
            dx = pr[0] - proj[k, 0]; dy = pr[1] - proj[k, 1]
            dists[t] = np.sqrt(dx * dx + dy * dy)
        npp = len(ks)
        med = O.ref_kth_element_copy(dists, _iround(0.8 * npp)) if npp else 0.0                        # Bundle.cpp:761-763
        thresh = min(max(1.2 * 2.0 * med, 8.0), 16.0)                                                   # :767-770
        ssum = 0.0
        for v in dists:
            ssum += v
        assert st["nobs"][j] == npp
        assert abs(st["kth80"][j] - med) <= 1e-9 * max(1.0, med)
        if npp:
            assert abs(st["mean"][j] - ssum / npp) <= 1e-9 * max(1.0, ssum / npp)
            assert abs(st["kth50"][j] - O.ref_kth_element_copy(dists, _iround(0.5 * npp))) <= 1e-9 * max(1.0, med)
        assert abs(st["thresh"][j] - thresh) <= 1e-9 * thresh
        total += ssum; count += npp
        for t, k in enumerate(ks):                                                                      # :792-821
            i = pt_of[k]
            if pcons is not None and pcons[i, 0] != 0.0:
                continue
            if dists[t] > thresh and i not in outliers:
                assert abs(dists[t] - thresh) > 1e-6
                outliers.append(i); errors.append(dists[t])
    assert abs(st["global_mean"] - total / count) <= 1e-9 * total / count
    flag = np.zeros(n, np.uint8); flag[outliers] = 1
    assert np.array_equal(st["outlier"], flag) and len(outliers) >= 10
    # the reference records the error of the first offending observation in CAMERA order; the library reports the first in the
    # point's own (camera-ascending) view list -- the same observation
    err = np.zeros(n); err[outliers] = errors
    assert np.abs(st["err"] - err).max() <= 1e-9 * max(1.0, err.max())


@pytest.mark.parametrize("thr", [2.0, 3.5])
def test_ray_angle_pruning(gpu_bsfm, thr):
    """SURVEY 8(f).1, second half, pinned to the REFERENCE'S OWN BundlerApp::RemoveBadPointsAndCameras (src/Bundle.cpp:4190-4261,
    compiled from where it lies by oracle/ref_prune.cpp -> oracle/_ref/libpruneref.so): committed fixture
    tests/golden/prune_golden.npz (tests/golden/make_golden.py prune) and, where oracle/_ref is present, a live call on the same
    scene.  Some points are pulled far away so that their rays become nearly parallel.  Flags and the count must agree exactly; the
    largest ray angle per point to 1e-9 degrees (acos near 1 amplifies the last ulp of the dot product)."""
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import prune_scene
    B = gpu_bsfm
    c, pts = prune_scene()
    m, n = c["m"], c["n"]
    pb = B.Problem(n, m, c["rowptr"], c["colidx"], c["proj"], c["cams"], pts.ravel(), est_focal_length=c["est"],
                   undistort=c["und"], options=B.default_options(verbose=0))
    st = pb.ray_angles(thr)
    pb.close()
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "prune_golden.npz"))
    refs = [(int(G[f"num_pruned_{thr}"][0]), G[f"prune_{thr}"], G[f"angle_deg_{thr}"])]
    if O.have_pruneref():
        refs.append(O.ref_remove_bad_points(n, m, c["rowptr"], c["colidx"], c["cams"], pts, thr))
        assert refs[1][0] == refs[0][0] and np.array_equal(refs[1][1], refs[0][1]) and np.array_equal(refs[1][2], refs[0][2])
    for k, prune, ang in refs:
        assert np.abs(st["angle_deg"] - ang).max() <= 1e-9
        assert np.abs(ang - 0.5 * thr).min() > 1e-6               # nobody sits on the threshold: flags must agree exactly
        assert np.array_equal(st["prune"], prune) and st["num_pruned"] == k == int(prune.sum())
        assert 0 < prune.sum() < n


@pytest.mark.parametrize("scene", ["A", "B"])
@pytest.mark.parametrize("tag", ["an", "fd"])
def test_lm_failure_branches_vs_reference(gpu_bsfm, scene, tag):
    """The two failure branches of the LM loop, forced INSIDE an LM run, against the reference itself (fixture
    tests/golden/failure_golden.npz from oracle/_ref; tests/golden/make_golden.py failures) and the live CPU oracle:
      A: dpotrf fails (info > 0) => issolved = 0 => more damping (sba_levmar.c:1368-1377, 1584-1611) -- eight rejected linear systems
         before the first accepted step (an indefinite U_3 through a constraint of weight -1e8);
      B: singular V*_i => more damping without a linear system (sba_levmar.c:1138-1162), mu stuck at 0 (tau = 0) until nu overflows:
         stop 6, ||dp|| = DBL_MAX, parameters untouched.
    Counters (iterations, stop code, function / Jacobian evaluations, linear systems) bit-exact, costs 1e-9, parameters 1e-7 / 1e-6."""
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import failure_scenes
    B = gpu_bsfm
    sc = failure_scenes()[scene]
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "failure_golden.npz"))
    jac = B.JAC_ANALYTIC if tag == "an" else B.JAC_FD
    tol = 1e-7 if tag == "an" else 1e-6
    vm = B.dense_vmask(sc["n"], sc["m"], sc["rowptr"], sc["colidx"])
    O.port().oracle_set_tau.argtypes = [__import__("ctypes").c_double]
    for it in (1, 6):
        opt = B.default_options(jacobian=jac, verbose=0, itmax=it, opts=[sc["tau"], 1e-10, 1e-12, 1e-12, 0.0, 4e-2])
        pb = B.Problem(sc["n"], sc["m"], sc["rowptr"], sc["colidx"], sc["proj"], sc["cams"], sc["pts"],
                       use_constraints=sc["use_constraints"], options=opt)
        rc, info = pb.solve()
        p = pb.download(want_cams=False)[0]
        pb.close()
        O.port().oracle_set_tau(sc["tau"])
        q = O.port_run_sfm(sc["n"], sc["m"], vm, sc["proj"], sc["cams"], sc["pts"], itmax=it, jac_mode=1 if tag == "an" else 0,
                           use_constraints=sc["use_constraints"])
        O.port().oracle_set_tau(-1.0)
        for gi, gp, grc in ((G[f"{scene}_{tag}_{it}_info"], G[f"{scene}_{tag}_{it}_p"], int(G[f"{scene}_{tag}_{it}_rc"][0])),
                            (q["info"], q["p"], q["rc"])):
            assert rc == grc, (rc, grc, info, gi)
            assert list(info[5:10]) == list(gi[5:10]), (info, gi)          # iterations, stop, nfev, njev, nlss
            assert abs(info[0] - gi[0]) <= 1e-12 * abs(gi[0]) and abs(info[1] - gi[1]) <= 1e-9 * abs(gi[1])
            assert info[3] == gi[3] or abs(info[3] - gi[3]) <= 1e-5 * abs(gi[3])      # ||dp||^2 (DBL_MAX when no step was ever computed)
            assert np.abs(p - gp).max() <= tol * np.abs(gp).max()
        if scene == "A" and it == 6:
            assert info[9] == 14 and info[5] == 6            # 8 rejected + 6 accepted systems
        if scene == "B" and it == 6:
            assert info[6] == 6 and info[9] == 0 and np.array_equal(p[sc["m"] * 9:], np.asarray(sc["pts"]).ravel())


@pytest.mark.parametrize("solver", ["envelope", "auto"])
def test_envelope_reduced_solve_matches_dense_on_a_connected_scene(gpu_bsfm, solver):
    """Opt-in envelope solver (BSFM_SOLVER_ENVELOPE; BSFM_SOLVER_AUTO picks it when the camera graph is connected): cameras renumbered by
    reverse Cuthill-McKee, S assembled in that numbering, the tiled Cholesky restricted to the tile envelope, the step mapped back.  On a
    connected (banded-visibility) scene big enough for several tile columns the LM trajectory must be the dense path's to rounding: same
    counters, costs to 1e-12, parameters to 1e-9 of max |p|, after 1 and after 4 iterations; S is exported in the natural numbering by
    both."""
    B = gpu_bsfm
    m, n = 120, 6000                                                     # 1 080 unknowns = 9 tile columns
    s = B.synth_ba(m, n, 8, banded=True)
    res = {}
    for tag, rs in (("dense", B.SOLVER_DENSE), (solver, B.SOLVER_ENVELOPE if solver == "envelope" else B.SOLVER_AUTO)):
        for it in (1, 4):
            pb = B.Problem(n, m, s["rowptr"], s["colidx"], s["proj"], s["cams"], s["pts"],
                           options=B.default_options(jacobian=B.JAC_ANALYTIC, verbose=0, itmax=it, reduced_solver=rs,
                                                     opts=[1e-3, 1e-10, 1e-12, 1e-12, 0.0, 4e-2]))
            rc, info = pb.solve()
            p = pb.download(want_cams=False)[0]
            ne = pb.normal_equations(mu=0.0) if it == 1 else None
            pb.close()
            res[(tag, it)] = (rc, info, p, ne)
    for it in (1, 4):
        rc0, i0, p0, n0 = res[("dense", it)]; rc1, i1, p1, n1 = res[(solver, it)]
        assert rc0 == rc1 and list(i0[5:10]) == list(i1[5:10])
        assert abs(i0[1] - i1[1]) <= 1e-12 * i0[1]
        assert np.abs(p0 - p1).max() <= 1e-9 * np.abs(p0).max()
        if n0 is not None:            # exported in the natural numbering by both (at parameters that agree to rounding)
            assert np.abs(n0["S"] - n1["S"]).max() <= 1e-9 * np.abs(n0["S"]).max()


MOT = np.load(os.path.join(os.path.dirname(__file__), "golden", "mot_golden.npz"))
REF_OPTS = [1e-3, 1e-10, 1e-12, 1e-12, 0.0, 4e-2]     # what run_sfm passes (sfm.c:705-714, eps2 = 1e-12 as in the fixtures)


@pytest.mark.parametrize("name", ["s9", "s9c", "s9m", "s7"])
def test_mot_camera_only_refinement_matches_reference(gpu_bsfm, name):
    """SURVEY 8(f).3 / run_sfm's fix_points: camera-only LM (sba_mot_levmar_x, sba_levmar.c:2090-2690) against the
    reference's own runs (tests/golden/mot_golden.npz from oracle/_ref): parameter vector after 1 and 3 iterations,
    counters (iterations, function / Jacobian evaluations, linear systems = attempts x free cameras), final cost."""
    B = gpu_bsfm
    c = load_case(name)
    m, n = c["m"], c["n"]
    cnp = 6 + c["est"] + 2 * c["und"]
    for tag, jac, tol in (("an", B.JAC_ANALYTIC, 1e-8), ("fd", B.JAC_FD, 1e-6)):
        for it in (1, 3):
            opt = B.default_options(jacobian=jac, verbose=0, itmax=it, opts=REF_OPTS)
            pb = B.Problem(n, m, c["rowptr"], c["colidx"], c["proj"], c["cams"], c["pts"], mcon=c["ncons"],
                           est_focal_length=c["est"], undistort=c["und"], use_constraints=c["cons"], options=opt, fix_points=1)
            rc, info = pb.solve()
            p = pb.download(want_cams=False)[0]
            pb.close()
            gi, gp = MOT[f"{name}_{tag}_it{it}_info"], MOT[f"{name}_{tag}_it{it}_p"]
            assert rc == it and list(info[5:10]) == list(gi[5:10]), (info, gi)
            assert abs(info[0] - gi[0]) <= 1e-12 * gi[0] and abs(info[1] - gi[1]) <= 1e-9 * gi[1]
            assert np.abs(p[:m * cnp] - gp).max() <= tol * np.abs(gp).max()
            assert np.array_equal(p[m * cnp:], c["pts"])                      # points are constants
        pb = B.Problem(n, m, c["rowptr"], c["colidx"], c["proj"], c["cams"], c["pts"], mcon=c["ncons"],
                       est_focal_length=c["est"], undistort=c["und"], use_constraints=c["cons"],
                       options=B.default_options(jacobian=jac, verbose=0, itmax=150, opts=REF_OPTS), fix_points=1)
        rc, info = pb.solve()
        pb.close()
        gi = MOT[f"{name}_{tag}_it150_info"]
        assert abs(info[1] - gi[1]) <= 1e-7 * gi[1]
        # the tail is rounding noise: with eps4 = 0 stop rule 4 (sba_levmar.c:2625) fires as soon as the relative cost
        # change drops below ~1e-8 -- the reference itself ends after 24 (FD) or 28 (analytic) iterations on s9
        assert info[6] in (2.0, 4.0) and abs(info[5] - gi[5]) <= 6


def test_run_sfm_fix_points_matches_reference_run_sfm(gpu_bsfm):
    """The boundary itself with fix_points = 1 (EstimateIgnoredCameras, src/Bundle.cpp:1917,1969)."""
    B = gpu_bsfm
    for name in ("s9", "s9c", "s7"):
        c = load_case(name)
        cams = B.copy_cameras(c["cams"]); pts = c["pts"].copy()
        rc, info = B.run_sfm(c["n"], c["m"], c["ncons"], c["vm"], c["proj"], c["est"], 0, c["und"], 1, cams, pts,
                             use_constraints=c["cons"], fix_points=1, eps2=1e-12, options=B.default_options(verbose=0))
        assert rc >= 0
        assert np.array_equal(pts, c["pts"])
        f = np.array([cm.f for cm in cams]); t = np.array([list(cm.t) for cm in cams])
        R = np.array([list(cm.R) for cm in cams])
        assert np.abs(f - MOT[f"{name}_run_cam_f"]).max() <= 1e-5 * np.abs(f).max()
        assert np.abs(t - MOT[f"{name}_run_cam_t"]).max() <= 1e-5 * max(1.0, np.abs(t).max())
        assert np.abs(R - MOT[f"{name}_run_cam_R"]).max() <= 1e-5


def test_resident_problem_reuse_and_caller_stream(gpu_bsfm):
    """Resident API across outer rounds: bsfm_problem_reset_params restarts the same device problem from
    new parameters and must reproduce a fresh problem bit for bit (all reductions are fixed-order); a caller-provided
    HIP stream (bsfm_problem_set_stream) gives the same result as the library's own."""
    import ctypes as C
    B = gpu_bsfm
    c = load_case("band")
    opt = B.default_options(jacobian=B.JAC_ANALYTIC, verbose=0, itmax=4, opts=REF_OPTS)

    def fresh(cams, pts):
        pb = B.Problem(c["n"], c["m"], c["rowptr"], c["colidx"], c["proj"], cams, pts, est_focal_length=c["est"],
                       undistort=c["und"], use_constraints=c["cons"], options=opt)
        rc, info = pb.solve()
        p = pb.download(want_cams=False)[0]
        return pb, p, info

    pb, p1, info1 = fresh(c["cams"], c["pts"])
    # second round: perturbed start, same structure, same problem object
    rng = np.random.default_rng(3)
    pts2 = c["pts"] + 1e-3 * rng.standard_normal(c["pts"].shape)
    assert pb.reset_params(c["cams"], pts2) == 0
    rc, info2 = pb.solve()
    p2 = pb.download(want_cams=False)[0]
    pb.close()
    pbf, p2f, info2f = fresh(c["cams"], pts2)
    pbf.close()
    assert np.array_equal(p2, p2f) and np.array_equal(info2, info2f)
    assert not np.array_equal(p1, p2)
    # caller-owned stream
    st = C.c_void_p()
    hip = C.CDLL("libamdhip64.so")
    if hip.hipStreamCreate(C.byref(st)) != 0:
        # only when another test of the same process has imported torch first: dlopen("libamdhip64.so") then answers with torch's bundled
        # copy of the runtime, which has no device initialised (hipErrorNoDevice) -- the library's own runtime is unaffected
        pytest.skip("a second HIP runtime (torch's bundled copy) answered dlopen('libamdhip64.so')")
    pbs = B.Problem(c["n"], c["m"], c["rowptr"], c["colidx"], c["proj"], c["cams"], c["pts"], est_focal_length=c["est"],
                    undistort=c["und"], use_constraints=c["cons"], options=opt)
    pbs.set_stream(st.value)
    rc, info3 = pbs.solve()
    p3 = pbs.download(want_cams=False)[0]
    pbs.close()
    hip.hipStreamDestroy(st)
    assert np.array_equal(p3, p1) and np.array_equal(info3, info1)


def test_known_intrinsics_cameras_match_reference(gpu_bsfm):
    """cameras with known intrinsics go through the 5-parameter Brown model and their own K
    (sfm_project_rd, lib/sfm-driver/sfm.c:339-358).  Fixture = the reference's sba_motstr_levmar (FD Jacobian) on a scene
    where 3 of 8 cameras have known intrinsics (tests/golden/known_golden.npz)."""
    B = gpu_bsfm
    X = np.load(os.path.join(os.path.dirname(__file__), "golden", "known_golden.npz"))
    m, n = len(X["cam_f"]), len(X["pts"]) // 3
    cams = O.arrays_to_cams(X["cam_R"], X["cam_t"], X["cam_f"], X["cam_k"])
    for j in range(m):
        cams[j].known_intrinsics = int(X["cam_known"][j])
        for q in range(9):
            cams[j].K_known[q] = float(X["cam_K_known"][j][q])
        for q in range(5):
            cams[j].k_known[q] = float(X["cam_k_known"][j][q])
    assert X["cam_known"].sum() == 3
    for it in (1, 3):
        # asking for the analytic Jacobian must fall back to forward differences for such scenes
        opt = B.default_options(jacobian=B.JAC_ANALYTIC if it == 3 else B.JAC_FD, verbose=0, itmax=it, opts=REF_OPTS)
        pb = B.Problem(n, m, X["rowptr"], X["colidx"], X["proj"], cams, X["pts"], options=opt)
        rc, info = pb.solve()
        p = pb.download(want_cams=False)[0]
        pb.close()
        gi, gp = X[f"fd_it{it}_info"], X[f"fd_it{it}_p"]
        assert rc == it and list(info[5:10]) == list(gi[5:10])
        assert abs(info[0] - gi[0]) <= 1e-11 * gi[0] and abs(info[1] - gi[1]) <= 1e-7 * gi[1]
        assert np.abs(p - gp).max() <= 2e-6 * np.abs(gp).max()
    # ... and at FIXED iteration indices inside the run (10 and 15 of the reference's 38): the converged run below ends on a plateau where
    # one iteration more or less is rounding -- the iterate after exactly 10 iterations is not, and a changed summation order cannot hide here.  (Not
    # deeper: from ~12 iterations on this scene's trajectory separates from the reference's under ANY other arithmetic, the round-2 LAPACK-style kernel
    # included: profiles/r06_known_intrinsics_trajectory.txt.)
    for it, tol_cost, tol_p in ((10, 1e-9, 1e-7), (15, 2e-6, 1e-4)):
        opt = B.default_options(jacobian=B.JAC_FD, verbose=0, itmax=it, opts=REF_OPTS)
        pb = B.Problem(n, m, X["rowptr"], X["colidx"], X["proj"], cams, X["pts"], options=opt)
        rc, info = pb.solve()
        p = pb.download(want_cams=False)[0]
        pb.close()
        gi, gp = X[f"fd_it{it}_info"], X[f"fd_it{it}_p"]
        assert rc == it and list(info[5:10]) == list(gi[5:10])
        assert abs(info[1] - gi[1]) <= tol_cost * gi[1], (it, info[1], gi[1])
        assert np.abs(p - gp).max() <= tol_p * np.abs(gp).max(), it
    # the drop-in boundary accepts them as well (it used to refuse)
    c2 = B.copy_cameras(cams); pts = X["pts"].copy()
    vm = B.dense_vmask(n, m, X["rowptr"], X["colidx"])
    rc, info = B.run_sfm(n, m, 0, vm, X["proj"], 1, 0, 1, 1, c2, pts, eps2=1e-12, options=B.default_options(verbose=0))
    gi = X["fd_it150_info"]
    # The reference stops after 38 iterations on its relative-step test, on a plateau: with the one-tile solve of round 5 (same backward
    # error, different rounding: scripts/r5/one_tile_accuracy.py) the test fires one iteration earlier, 1.5e-4 above the reference's cost;
    # with the round-2 tile kernel it fired at the same iteration, 1e-5 away.  One iteration either way, cost to 5e-4.
    assert rc >= 0 and abs(int(info[5]) - int(gi[5])) <= 1 and abs(info[1] - gi[1]) <= 5e-4 * gi[1]


@pytest.mark.gpu
@pytest.mark.parametrize("und", [1, 0])
def test_fisheye_projection_matches_reference(gpu_bsfm, und):
    """run_sfm(optimize_for_fisheye=1): sfm_project_point2_fisheye (lib/sfm-driver/sfm.c:448-492) = pinhole without the
    radial term + the equidistant map of the cameras flagged fisheye (sfm_fisheye_distort, sfm.c:426-446).  Fixture = the
    reference's sba_motstr_levmar driven by that callback (tests/golden/fisheye_golden.npz); with undistort=1 the k1,k2
    columns of the Jacobian are identically zero, as in the reference."""
    B = gpu_bsfm
    X = np.load(os.path.join(os.path.dirname(__file__), "golden", "fisheye_golden.npz"))
    m, n = len(X["cam_f"]), len(X["pts"]) // 3
    cams = O.arrays_to_cams(X["cam_R"], X["cam_t"], X["cam_f"], X["cam_k"])
    for j in range(m):
        cams[j].fisheye = int(X["cam_fisheye"][j])
        cams[j].f_cx, cams[j].f_cy, cams[j].f_rad, cams[j].f_angle, cams[j].f_focal = [float(v) for v in X["cam_fparams"][j]]
    assert X["cam_fisheye"].sum() == 3
    for it in (1, 3):
        opt = B.default_options(jacobian=B.JAC_ANALYTIC if it == 3 else B.JAC_FD, verbose=0, itmax=it, opts=REF_OPTS)
        pb = B.Problem(n, m, X["rowptr"], X["colidx"], X["proj"], cams, X["pts"], undistort=und, options=opt,
                       optimize_for_fisheye=1)
        rc, info = pb.solve()
        p = pb.download(want_cams=False)[0]
        pb.close()
        gi, gp = X[f"u{und}_it{it}_info"], X[f"u{und}_it{it}_p"]
        assert rc == it and list(info[5:10]) == list(gi[5:10])
        assert abs(info[0] - gi[0]) <= 1e-11 * gi[0] and abs(info[1] - gi[1]) <= 1e-7 * gi[1]
        assert np.abs(p - gp).max() <= 2e-6 * np.abs(gp).max()
    # the drop-in boundary against the verbatim reference run_sfm
    c2 = B.copy_cameras(cams); pts = X["pts"].copy()
    vm = B.dense_vmask(n, m, X["rowptr"], X["colidx"])
    rc, info = B.run_sfm(n, m, 0, vm, X["proj"], 1, 0, und, 1, c2, pts, optimize_for_fisheye=1, eps2=1e-12,
                         options=B.default_options(verbose=0))
    gi = X[f"u{und}_it150_info"]
    assert rc >= 0 and info[6] == gi[6] and abs(info[5] - gi[5]) <= 2 and abs(info[1] - gi[1]) <= 1e-6 * gi[1]
    got = O.cams_to_arrays(c2)
    assert np.abs(got["f"] - X[f"u{und}_run_f"]).max() <= 1e-5 * np.abs(X[f"u{und}_run_f"]).max()
    assert np.abs(got["R"] - X[f"u{und}_run_R"]).max() <= 1e-6
    assert np.abs(got["t"] - X[f"u{und}_run_t"]).max() <= 1e-5 * max(1.0, np.abs(X[f"u{und}_run_t"]).max())
    assert np.abs(pts - X[f"u{und}_run_pts"]).max() <= 1e-5 * max(1.0, np.abs(X[f"u{und}_run_pts"]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("mcon", [0, 3])
def test_group_by_group_reduced_solve_matches_dense(gpu_bsfm, mcon):
    """Opt-in reduced solver (compsolve.hip.h): the SURVEY 8(d) generator makes each point visible in the cameras
    (j0 + d m/deg) mod m, so the cameras fall into m/deg groups that share no point and S is block diagonal up to a
    permutation.  Solving group by group must reproduce the dense Cholesky path (same LM trajectory, solution equal to
    rounding) and the oracle; a connected scene must silently stay on the dense path."""
    B = gpu_bsfm
    s = B.synth_ba(40, 400, 4)                       # 10 groups of 4 cameras
    vm = B.dense_vmask(400, 40, s["rowptr"], s["colidx"])
    groups = {tuple(sorted(set(s["colidx"][s["rowptr"][i]:s["rowptr"][i + 1]] % 10))) for i in range(400)}
    assert all(len(g) == 1 for g in groups)          # every point stays inside one residue class mod 10
    res = {}
    for mode in (B.SOLVER_DENSE, B.SOLVER_AUTO):
        opt = B.default_options(jacobian=B.JAC_ANALYTIC, verbose=0, itmax=6, reduced_solver=mode)
        pb = B.Problem(400, 40, s["rowptr"], s["colidx"], s["proj"], s["cams"], s["pts"], mcon=mcon, options=opt)
        rc, info = pb.solve()
        res[mode] = (rc, info.copy(), pb.download(want_cams=False)[0], pb.phase_ms("groups"))
        pb.close()
    (rc0, i0, p0, g0), (rc1, i1, p1, g1) = res[B.SOLVER_DENSE], res[B.SOLVER_AUTO]
    assert g0 == 0 and g1 == 10                      # 10 groups of 4 (mcon = 3: three of them lose their fixed camera)
    assert rc0 == rc1 and list(i0[5:10]) == list(i1[5:10])
    assert abs(i0[1] - i1[1]) <= 1e-10 * i0[1]
    assert np.abs(p0 - p1).max() <= 1e-9 * np.abs(p0).max()
    q = O.port_run_sfm(400, 40, vm, s["proj"], s["cams"], s["pts"], itmax=6, jac_mode=1, ncons=mcon)
    assert abs(i1[1] - q["info"][1]) <= 1e-9 * q["info"][1] and np.abs(p1 - q["p"]).max() <= 1e-7 * np.abs(q["p"]).max()
    # connected scene (banded visibility): no small groups -- auto falls through to the envelope solver (round 3; it was the dense path):
    # the same trajectory to rounding
    c = load_case("band")
    out = []
    for mode in (B.SOLVER_DENSE, B.SOLVER_AUTO):
        opt = B.default_options(jacobian=B.JAC_ANALYTIC, verbose=0, itmax=3, reduced_solver=mode)
        pb = B.Problem(c["n"], c["m"], c["rowptr"], c["colidx"], c["proj"], c["cams"], c["pts"], est_focal_length=c["est"],
                       undistort=c["und"], use_constraints=c["cons"], options=opt)
        pb.solve()
        assert pb.phase_ms("groups") == 0
        out.append(pb.download(want_cams=False)[0]); pb.close()
    assert np.abs(out[0] - out[1]).max() <= 1e-11 * np.abs(out[0]).max()


@pytest.mark.gpu
def test_group_solver_with_a_camera_that_sees_nothing(gpu_bsfm):
    """A camera without observations is a group of its own (S_jj = mu I from the diagonal fill): the group-by-group solver
    must treat it exactly like the dense path does."""
    B = gpu_bsfm
    m, n = 9, 120
    base = B.synth_ba(m, n, 3)                       # cameras j0 + 3 d: three groups
    ci = base["colidx"].reshape(n, 3); pr = base["proj"].reshape(n, 3, 2)
    rows, proj, keep_pt = [], [], []
    for i in range(n):
        sel = [q for q in range(3) if ci[i, q] != 4]  # camera 4 loses every observation
        if len(sel) >= 2:
            rows.append(ci[i, sel]); proj.append(pr[i, sel]); keep_pt.append(i)
    pts = base["pts"].reshape(-1, 3)[keep_pt].ravel()
    rowptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    colidx = np.concatenate(rows).astype(np.int32); proj = np.concatenate(proj).ravel()
    out = []
    for mode in (B.SOLVER_DENSE, B.SOLVER_AUTO):
        opt = B.default_options(jacobian=B.JAC_ANALYTIC, verbose=0, itmax=4, reduced_solver=mode)
        pb = B.Problem(len(rows), m, rowptr, colidx, proj, base["cams"], pts, options=opt)
        rc, info = pb.solve()
        out.append((rc, list(info[5:10]), info[1], pb.download(want_cams=False)[0], pb.phase_ms("groups")))
        pb.close()
    assert out[0][4] == 0 and out[1][4] == 4         # three camera groups + the lonely camera
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1]
    assert abs(out[0][2] - out[1][2]) <= 1e-12 * out[0][2] and np.abs(out[0][3] - out[1][3]).max() <= 1e-12


@pytest.mark.gpu
def test_block_cache_reuse_and_trim_leave_results_bit_identical(gpu_bsfm):
    """Resident problems take their device buffers from a process-wide block cache and hand them back dirty (devcache.h: hipFree
    synchronises the device ~60 times per tear-down).  The same run_sfm call must give the same bits whether its blocks are fresh
    (after bsfm_device_cache_trim), reused from an identical problem, or reused from DIFFERENT problems (other sizes, other data)."""
    B = gpu_bsfm
    opt = B.default_options(verbose=0)

    def run(m, n, deg):
        s = B.synth_ba(m, n, deg)
        vm = B.dense_vmask(n, m, s["rowptr"], s["colidx"])
        cams = B.copy_cameras(s["cams"]); pts = s["pts"].copy()
        rc, info = B.run_sfm(n, m, 0, vm, s["proj"], 1, 0, 1, 1, cams, pts, eps2=1e-12, options=opt)
        return rc, np.array(info), np.array([[c.f, c.k[0], c.k[1], *list(c.t), *list(c.R)] for c in cams]), pts

    B.lib.bsfm_device_cache_trim()
    ref = run(20, 900, 6)                         # fresh blocks
    again = run(20, 900, 6)                       # every block comes back out of the cache
    run(33, 2000, 5); run(9, 300, 4)              # other shapes leave their own dirt in the size classes around
    third = run(20, 900, 6)
    B.lib.bsfm_device_cache_trim()
    fourth = run(20, 900, 6)                      # fresh again
    for other in (again, third, fourth):
        assert other[0] == ref[0] and np.array_equal(other[1], ref[1])
        assert np.array_equal(other[2], ref[2]) and np.array_equal(other[3], ref[3])


@pytest.mark.parametrize("banded", [False, True], ids=["cliques", "connected"])
def test_schur_launch_order_does_not_change_a_bit(gpu_bsfm, monkeypatch, banded):
    """The Schur tasks are LAUNCHED in an order chosen for the L2s -- clustered (default: point slice, then the two cameras in
    breadth-first numbering of the co-visibility graph, index_build.hip), by block, or by first point (rounds 1-3) -- while their
    partial sums keep the block order the reference visits them in (sba_levmar.c:1218-1268).  So S and E must be the same BITS under
    all three, every order must be a permutation of the same task list, and the LM run must not notice."""
    B = gpu_bsfm
    m, n = 130, 30000          # 16 groups of 8 cameras with 10 tasks per block / ~3 200 blocks of ~2 tasks
    s = B.synth_ba(m, n, 8, banded=banded)
    res = {}
    for order in ("", "block", "point"):
        if order:
            monkeypatch.setenv("BSFM_SCHUR_ORDER", order)
        else:
            monkeypatch.delenv("BSFM_SCHUR_ORDER", raising=False)
        pb = B.Problem(n, m, s["rowptr"], s["colidx"], s["proj"], s["cams"], s["pts"],
                       options=B.default_options(jacobian=B.JAC_ANALYTIC, verbose=0, itmax=4))
        ne = pb.normal_equations(mu=0.37)
        sc = pb.export_schur()
        rc, info = pb.solve()
        p, _, _ = pb.download()
        pb.close()
        live = sc["tasks"][sc["tasks"][:, 3] >= 0]
        res[order] = dict(S=ne["S"], E=ne["E"], tasks=live, info=np.array(info), p=p, ntasks=sc["ntasks"])
    ref = res[""]
    for order in ("block", "point"):
        r = res[order]
        assert r["S"].tobytes() == ref["S"].tobytes() and r["E"].tobytes() == ref["E"].tobytes(), order
        assert r["info"].tobytes() == ref["info"].tobytes() and r["p"].tobytes() == ref["p"].tobytes(), order
        assert r["ntasks"] == ref["ntasks"] == len(r["tasks"]) == len(ref["tasks"])
        key = lambda t: t[np.lexsort((t[:, 0], t[:, 3]))]
        assert np.array_equal(key(r["tasks"]), key(ref["tasks"])), order          # the same tasks, another order
    assert np.array_equal(np.sort(ref["tasks"][:, 3]), np.arange(ref["ntasks"]))     # every output slot exactly once
    assert not np.array_equal(ref["tasks"][:, 3], res["block"]["tasks"][:, 3])      # ... and the orders do differ


@pytest.mark.parametrize("est,und,ncons,banded", [(1, 1, 0, True), (1, 1, 3, False), (1, 0, 0, True), (0, 0, 2, True)],
                         ids=["cnp9-connected", "cnp9-3fixed-cliques", "cnp7-connected", "cnp6-2fixed-connected"])
def test_midsize_run_sfm_matches_the_reference_live(gpu_bsfm, est, und, ncons, banded):
    """The drop-in boundary at the sizes incremental Bundler spends its time at (72 cameras: 4-5 tile columns of the tile-dataflow
    Cholesky, chol_flow.hip.h), every camera model (cnp 9 / 7 / 6) and fixed cameras in front of the reduced system, against the
    reference's own run_sfm called LIVE on the same inputs (oracle/_ref, lib/sfm-driver/sfm.c:592-1003): per-camera focal length,
    distortion, centre and rotation, and every point, after both runs stopped by their own rules."""
    import oracle_util as O
    if not O.have_ref():
        pytest.skip("oracle/_ref not built")
    B = gpu_bsfm
    m, n = 72, 6000
    s = B.synth_ba(m, n, 8, banded=banded)
    vm = B.dense_vmask(n, m, s["rowptr"], s["colidx"])
    if not und:
        for cm in s["cams"]:
            cm.k[0] = 0.0; cm.k[1] = 0.0
        # (the generator projected WITH its small distortion: both implementations fit the same slightly wrong model to the same data)
    cams = B.copy_cameras(s["cams"]); pts = s["pts"].copy()
    rc, info = B.run_sfm(n, m, ncons, vm, s["proj"], est, 0, und, 1, cams, pts, eps2=1e-12, options=B.default_options(verbose=0))
    assert rc >= 0
    rcams, rpts = O.ref_run_sfm(n, m, vm, s["proj"], s["cams"], s["pts"], ncons=ncons, est_focal=est, undistort=und, explicit=1, eps2=1e-12)
    f = np.array([c.f for c in cams]); rf = np.array([c.f for c in rcams])
    k = np.array([list(c.k) for c in cams]); rk = np.array([list(c.k) for c in rcams])
    t = np.array([list(c.t) for c in cams]); rt = np.array([list(c.t) for c in rcams])
    R = np.array([list(c.R) for c in cams]); rR = np.array([list(c.R) for c in rcams])
    d = dict(f=np.abs(f - rf).max() / np.abs(rf).max(), k=np.abs(k - rk).max() / max(np.abs(rk).max(), 1e-3), t=np.abs(t - rt).max() / np.abs(rt).max(),
             R=np.abs(R - rR).max(), pts=np.abs(pts - rpts).max() / np.abs(rpts).max())
    print("\n[midsize vs live reference]", {q: f"{v:.1e}" for q, v in d.items()}, "iterations", int(info[5]), "stop", int(info[6]))
    # measured: 0 .. 2e-11 (cnp 6 / 7), 4e-7 (cnp 9, connected), 3e-5 at most on the clique scene with fixed cameras (the second distortion coefficient of the clique scene is barely determined: values up to 2, Snavely's
    # 4 % rule stops both runs on the way there); the bounds are those of the fixture tests above
    assert d["f"] <= 1e-5 and d["k"] <= 1e-4 and d["t"] <= 1e-5 and d["R"] <= 1e-5 and d["pts"] <= 1e-5, d
    if ncons:
        assert np.array_equal(f[:ncons], np.array([c.f for c in s["cams"]][:ncons])) and np.array_equal(t[:ncons], np.array([list(c.t) for c in s["cams"]][:ncons]))
