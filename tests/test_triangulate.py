"""batched multi-view triangulation (bsfm_triangulate_batch) against the reference's
triangulate_n / triangulate_n_refine / triangulate (lib/imagelib/triangulate.c).

Fixture tests/golden/triangulate_golden.npz was produced by the reference itself (oracle/_ref, one call per point,
tests/golden/make_golden.py::triangulation).  Tolerance: the polish is MINPACK's lmdif stopped at tol = 1e-5 (1e-10 for the
two-view variant), i.e. the reference's own answer is only that close to the minimiser; following the same iterates the GPU
result agrees to <= 1e-6 of max(1, |X|) (observed: 1e-7 on the distant, nearly-parallel-ray points, 1e-10 typical) and the
rms reprojection error to 1e-10."""
import os

import numpy as np
import pytest

import oracle_util as O

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "triangulate_golden.npz"))


def _rel(X, Y):
    X = X.reshape(-1, 3); Y = Y.reshape(-1, 3)
    return np.abs(X - Y).max(axis=1) / np.maximum(1.0, np.abs(Y).max(axis=1))


@pytest.mark.parametrize("mode,tag", [(0, "n"), (1, "refine")])
def test_batch_matches_reference_fixture(gpu_bsfm, mode, tag):
    B = gpu_bsfm
    X, err, info = B.triangulate_batch(mode, G["view_ptr"], G["p"], G["R"], G["t"], X=G["X0"], view_cam=G["view_cam"])
    d = _rel(X, G[f"{tag}_X"])
    assert d.max() <= 1e-6 and np.median(d) <= 1e-9
    assert np.abs(err - G[f"{tag}_err"]).max() <= 1e-10
    assert set(np.unique(info)) <= {1, 2, 3}                      # lmdif1: converged on ftol / xtol / both
    # the reference's own argument layout (one R, t per VIEW) gives the same bits as the indexed form
    Rv = G["R"].reshape(-1, 9)[G["view_cam"]].ravel(); tv = G["t"].reshape(-1, 3)[G["view_cam"]].ravel()
    X2, err2, _ = B.triangulate_batch(mode, G["view_ptr"], G["p"], Rv, tv, X=G["X0"])
    assert np.array_equal(X, X2) and np.array_equal(err, err2)


def test_two_view_variant_matches_reference_fixture(gpu_bsfm):
    B = gpu_bsfm
    sel = G["pair_sel"]
    p2 = G["p"].reshape(-1, 2)[sel].ravel(); cam2 = G["view_cam"][sel]
    X, err, _ = B.triangulate_batch(B.TRI_PAIR, G["pair_ptr"], p2, G["R"], G["t"], view_cam=cam2)
    d = _rel(X, G["pair_X"])
    assert d.max() <= 2e-6 and np.median(d) <= 1e-9
    assert np.abs(err - G["pair_err"]).max() <= 1e-12             # sum of squares, ~1e-6 in magnitude


def test_refused_inputs_leave_outputs_untouched(gpu_bsfm):
    """dgelsy_driver / lmdif_driver refuse fewer equations than unknowns (lib/matrix/matrix.c:463-466,790-793)."""
    B = gpu_bsfm
    vp = np.array([0, 3, 4], np.int32)                             # second point has a single view
    with pytest.raises(RuntimeError):
        B.triangulate_batch(B.TRI_N, vp, G["p"][:8], G["R"], G["t"], view_cam=G["view_cam"][:4])
    with pytest.raises(RuntimeError):                             # the two-view variant wants exactly two
        B.triangulate_batch(B.TRI_PAIR, np.array([0, 3], np.int32), G["p"][:6], G["R"], G["t"], view_cam=G["view_cam"][:3])
    with pytest.raises(RuntimeError):                             # camera index out of range
        B.triangulate_batch(B.TRI_N, np.array([0, 2], np.int32), G["p"][:4], G["R"][:18], G["t"][:6], view_cam=np.array([0, 5], np.int32))
    X, err, info = B.triangulate_batch(B.TRI_N, np.array([0], np.int32), np.zeros(0), G["R"], G["t"], view_cam=np.zeros(0, np.int32))
    assert X.size == 0 and err.size == 0                          # empty batch is fine


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_against_live_reference_on_a_fresh_batch(gpu_bsfm):
    """New random batch, long tracks included, reference called point by point on the spot."""
    B = gpu_bsfm
    rng = np.random.default_rng(2024)
    s = B.synth_ba(60, 600, 4)
    ca = O.cams_to_arrays(s["cams"])
    Rc = ca["R"].reshape(-1, 3, 3); tc = np.einsum("mij,mj->mi", Rc, -ca["t"])
    npts = 120
    Xt = rng.uniform(-1, 1, (npts, 3))
    deg = rng.integers(2, 41, npts)
    vp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    cam = np.concatenate([np.sort(rng.choice(60, d, replace=False)) for d in deg]).astype(np.int32)
    P = np.einsum("vij,vj->vi", Rc[cam], np.repeat(Xt, deg, axis=0)) + tc[cam]
    p = P[:, :2] / P[:, 2:3] + rng.normal(0, 1e-3, (len(cam), 2))
    X, err, _ = B.triangulate_batch(B.TRI_N, vp, p.ravel(), ca["R"].ravel(), tc.ravel(), view_cam=cam)
    for i in range(npts):
        v = slice(vp[i], vp[i + 1])
        Xr, er = O.ref_triangulate(0, p[v], ca["R"][cam[v]], tc[cam[v]])
        assert np.abs(X[3 * i:3 * i + 3] - Xr).max() <= 1e-6 * max(1.0, np.abs(Xr).max())
        assert abs(err[i] - er) <= 1e-10
