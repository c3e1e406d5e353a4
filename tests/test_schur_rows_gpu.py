"""GPU parity of the row-wise Schur kernel (schur.hip.h: k_schur_rows; plan: schur_rows.h) -- the reduced camera system
S_jk = [j == k](U_j + mu I) - sum_i Y_ij W_ik^T and e_j = ea_j - sum_i Y_ij eb_i of lib/sba-1.5/sba_levmar.c:1182-1339.

The row kernel replaces the task kernel for the DENSE blocks of large problems; here it is forced onto small ones
(BSFM_SCHUR_ROWS=1, short segments, a low density bar) so that
  * S and E agree with the CPU oracle's dump (same tolerance as tests/test_ba_gpu.py::test_normal_equation_blocks) for every camera
    model (cnp 9 / 7 / 6), with constrained cameras in front, and with the task kernel to rounding (another summation order);
  * the plan resident on the device is the host plan of schur_rows.h, and the slots of sparse blocks still belong to the task kernel;
  * results are bit-identical from run to run and the LM run takes the same decisions with either kernel.
"""
import numpy as np
import pytest

import oracle_util as O
from test_oracle import CASES, load_case

pytestmark = pytest.mark.gpu


def rows_env(monkeypatch, on, L=None, dense_min=None, wg_min=1, tri_max=None):
    monkeypatch.setenv("BSFM_SCHUR_ROWS", "1" if on else "0")
    for key, v in (("BSFM_SCHUR_ROW_L", L), ("BSFM_SCHUR_ROW_MIN", dense_min), ("BSFM_SCHUR_ROW_WGMIN", wg_min), ("BSFM_SCHUR_ROW_TRIMAX", tri_max)):
        if v is None:
            monkeypatch.delenv(key, raising=False)
        else:
            monkeypatch.setenv(key, str(v))


def make_problem(B, c, jac, **kw):
    opt = B.default_options(jacobian=jac, verbose=0, **kw)
    return B.Problem(c["n"], c["m"], c["rowptr"], c["colidx"], c["proj"], c["cams"], c["pts"], mcon=c["ncons"],
                     est_focal_length=c["est"], undistort=c["und"], use_constraints=c["cons"], options=opt)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("L,dense_min", [(16, 1), (48, 3)])
def test_rows_reduced_system_vs_oracle(gpu_bsfm, monkeypatch, name, L, dense_min):
    B = gpu_bsfm
    c = load_case(name)
    q = O.port_run_sfm(c["n"], c["m"], c["vm"], c["proj"], c["cams"], c["pts"], itmax=1, jac_mode=1, ncons=c["ncons"],
                       est_focal=c["est"], undistort=c["und"], use_constraints=c["cons"], want_dumps=True)
    mu = float(q["mu"][0])
    rows_env(monkeypatch, True, L, dense_min)
    pb = make_problem(B, c, 1)
    assert pb.phase_ms("row_wgs") > 0, "the row kernel was not planned for this scene"
    ne = pb.normal_equations(mu)
    pb.close()
    rows_env(monkeypatch, False)
    pb = make_problem(B, c, 1)
    assert pb.phase_ms("row_wgs") == 0
    ne0 = pb.normal_equations(mu)
    pb.close()
    for key in ("S", "E"):
        ref = q[key]; got = ne[key].reshape(ref.shape)
        scale = np.abs(ref).max()
        assert np.abs(got - ref).max() <= 1e-11 * scale, (key, np.abs(got - ref).max() / scale)
        assert np.abs(got - ne0[key].reshape(ref.shape)).max() <= 1e-12 * scale, key
    assert np.abs(ne["S"] - ne["S"].T).max() <= 1e-12 * np.abs(ne["S"]).max()


def bfs_rank(blk_j, blk_k, mm, mcon):
    adj = [[] for _ in range(mm)]
    for a, c in zip(blk_j - mcon, blk_k - mcon):
        if a != c:
            adj[a].append(c); adj[c].append(a)
    seen = np.zeros(mm, bool); q = []
    for s0 in range(mm):
        if seen[s0]:
            continue
        seen[s0] = True; q.append(s0); h = len(q) - 1
        while h < len(q):
            for v in adj[q[h]]:
                if not seen[v]:
                    seen[v] = True; q.append(v)
            h += 1
    rank = np.zeros(mm, np.int32); rank[np.array(q)] = np.arange(mm)
    return rank


@pytest.mark.parametrize("banded,L,dense_min,mcon,wg_min,tri_max", [(False, 96, 24, 0, None, None), (False, 32, 8, 2, 1, 128), (True, 16, 2, 0, 1, None),
                                                                     (True, 64, 6, 1, 1, None), (False, 128, 24, 0, None, 512)],
                         ids=["cliques-default", "cliques-L32-2fixed-split", "connected-L16", "connected-L64-mixed", "cliques-L128-split"])
def test_rows_midsize_plan_bits_and_lm(gpu_bsfm, monkeypatch, banded, L, dense_min, mcon, wg_min, tri_max):
    B = gpu_bsfm
    m, n = 130, 30000
    s = B.synth_ba(m, n, 8, banded=banded)

    def run(on):
        rows_env(monkeypatch, on, L, dense_min, wg_min, tri_max)
        pb = B.Problem(n, m, s["rowptr"], s["colidx"], s["proj"], s["cams"], s["pts"], mcon=mcon,
                       options=B.default_options(jacobian=B.JAC_ANALYTIC, verbose=0, itmax=4))
        ne = pb.normal_equations(mu=0.37)
        sc = pb.export_schur(); rw = pb.export_rows(); ix = pb.export_index()
        rc, info = pb.solve()
        p, _, _ = pb.download()
        pb.close()
        return dict(S=ne["S"], E=ne["E"], sc=sc, rw=rw, ix=ix, info=np.array(info), p=p, rc=rc)

    a = run(True); a2 = run(True); off = run(False)
    assert len(a["rw"]["wgs"]) > 0
    # run to run: the same bits
    assert a["S"].tobytes() == a2["S"].tobytes() and a["E"].tobytes() == a2["E"].tobytes() and a["p"].tobytes() == a2["p"].tobytes()
    # against the task kernel: rounding only (the blocks are summed segment by segment instead of 192 triples at a time)
    scale = np.abs(off["S"]).max()
    assert np.abs(a["S"] - off["S"]).max() <= 1e-12 * scale
    assert np.abs(a["E"] - off["E"]).max() <= 1e-12 * np.abs(off["E"]).max()
    assert a["rc"] == off["rc"] and list(a["info"][5:10]) == list(off["info"][5:10])          # iterations, stop code, evaluations, solves
    assert abs(a["info"][1] - off["info"][1]) <= 1e-9 * off["info"][1]
    assert np.abs(a["p"] - off["p"]).max() <= 1e-8 * np.abs(off["p"]).max()
    # the resident plan is the host plan (schur_rows.h) of the resident structure
    sc, rw, ix = a["sc"], a["rw"], a["ix"]
    assert rw["L"] == L
    blk_start = np.zeros(len(sc["blk_j"]) + 1, np.int32)
    cam = ix["cam_cam"]
    key = cam[sc["triples"][:, 0]].astype(np.int64) * m + cam[sc["triples"][:, 1]]
    starts = np.flatnonzero(np.concatenate([[True], np.diff(key) != 0]))
    blk_start[:-1] = starts; blk_start[-1] = len(key)
    rank = bfs_rank(sc["blk_j"], sc["blk_k"], m - mcon, mcon)
    host = B.sfm.schur_row_plan(m, mcon, sc["blk_j"], sc["blk_k"], blk_start, sc["triples"][:, 0], ix["camptr"], rank, L, dense_min,
                                wg_min or 0, tri_max or 0, sc["ntasks"])
    assert np.array_equal(host["wgs"], rw["wgs"]) and np.array_equal(host["pieces"], rw["pieces"]) and np.array_equal(host["blk_row0"], rw["blk_row0"])
    assert host["nslots"] == rw["nslots"] and host["ntri"] == len(rw["row_tri"])
    # the kernel's own triple array = the plan's copy list applied to the resident triples (k_row_fill)
    exp = np.zeros((host["ntri"], 2), np.int32)
    for src, cnt, dst, rec0 in host["fills"]:
        padded = (cnt + 15) // 16 * 16
        t = np.minimum(np.arange(padded), cnt - 1)
        exp[dst:dst + padded, 0] = (sc["triples"][src + t, 0] - rec0) | np.where(np.arange(padded) < cnt, 0, B.sfm.ROW_DEAD)
        exp[dst:dst + padded, 1] = sc["triples"][src + t, 1]
    assert np.array_equal(exp, rw["row_tri"])
    if tri_max:
        assert (16 * rw["wgs"][:, 3] <= max(tri_max, 128)).all() and len(np.unique(rw["wgs"][:, 0])) < len(rw["wgs"]), "meant to split workgroups"
    dense = np.diff(rw["blk_row0"]) > 0
    exp_range = np.stack([sc["blk_task0"][:-1], sc["blk_task0"][1:]], axis=1)
    exp_range[dense] = sc["ntasks"] + np.stack([rw["blk_row0"][:-1], rw["blk_row0"][1:]], axis=1)[dense]
    assert np.array_equal(rw["blk_range"], exp_range)
    # the task kernel's launch list: the tasks of dense blocks are padding, the others untouched
    tl, full = rw["tasks_launch"], sc["tasks"]
    blk_of_slot = np.repeat(np.arange(len(dense)), np.diff(sc["blk_task0"]))
    live = full[:, 3] >= 0
    to_rows = np.zeros(len(full), bool); to_rows[live] = dense[blk_of_slot[full[live, 3]]]
    assert np.array_equal(tl[~to_rows], full[~to_rows]) and (tl[to_rows, 3] == -1).all()
    if banded and dense_min >= 6:
        assert dense.any() and (~dense).any(), "this case is meant to mix both kernels"
    # with the row kernel off nothing of it exists
    assert len(off["rw"]["wgs"]) == 0 and np.array_equal(off["rw"]["tasks_launch"], off["sc"]["tasks"])


def test_rows_first_iterations_match_reference_fixture(gpu_bsfm, monkeypatch):
    """The reference's own first iterations (tests/golden, generated by oracle/_ref) with the row kernel forced on."""
    from test_oracle import G
    B = gpu_bsfm
    rows_env(monkeypatch, True, 16, 1)
    for name in ("s9", "s7", "s6"):
        c = load_case(name)
        for tag, jac, tol in (("an", 1, 1e-7), ("fd", 0, 1e-6)):
            pb = make_problem(B, c, jac, itmax=3)
            assert pb.phase_ms("row_wgs") > 0
            rc, info = pb.solve()
            p, _, _ = pb.download()
            pb.close()
            ref_info = G[f"{name}_{tag}_it3_info"]; ref_p = G[f"{name}_{tag}_it3_p"]
            assert list(info[5:10]) == list(ref_info[5:10])
            assert abs(info[1] - ref_info[1]) <= 1e-9 * ref_info[1]
            assert np.abs(p - ref_p).max() <= tol * np.abs(ref_p).max()
