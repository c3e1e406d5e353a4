"""Plan of the row-wise Schur kernel (bundler_sfm_amd/csrc/schur_rows.h), checked on the CPU.

The kernel (schur.hip.h: k_schur_rows) accumulates the reduced camera system S_jk -= sum_i Y_ij W_ik^T of
lib/sba-1.5/sba_levmar.c:1195-1302 for the DENSE blocks: a workgroup owns (camera j, a segment of <= L consecutive camera-major
records), keeps those records in LDS and its four waves walk PIECES of the partner blocks' triple lists.  What must hold for that to
be the reference's sum, and is tested here without a GPU through the host-only entry bsfm_schur_row_plan:
  * coverage: every triple of a dense block lies in exactly one piece, no triple of a sparse block in any;
  * locality: the j-side record of every triple of a piece lies inside its workgroup's segment (it is read from the slab);
  * slots: the pieces of a block hold consecutive slots [blk_row0[b], blk_row0[b+1]) and follow the block's triple order, so that
    adding the slots in order adds the triples in the reference's (point) order;
  * balance: the waves of a workgroup carry the same number of 16-triple passes (+- 1);
  * semantics: a numpy replay that only touches what the kernel touches (segment rows through the slab index, k-side records
    through the triple list) reproduces sum_i A_ij^T (C_ij B_ik^T) A_ik block by block.
"""
import numpy as np
import pytest

import bundler_sfm_amd.sfm as B


def random_scene(rng, n, m, mcon, mode):
    """CRS of a visibility mask: 'clique' = groups of cameras that see the same points, 'banded' = a window of neighbours, 'random'."""
    rows = []
    for i in range(n):
        if mode == "clique":
            g = rng.integers(0, max(1, m // 6))
            cams = np.arange(g, m, max(1, m // 6))
        elif mode == "banded":
            j0 = rng.integers(0, m)
            cams = np.unique((j0 + rng.choice(min(12, m), size=min(5, m), replace=False)) % m)
        else:
            cams = np.flatnonzero(rng.random(m) < 0.3)
            if len(cams) < 2:
                cams = np.array([0, m - 1])
        rows.append(np.sort(cams))
    rowptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    colidx = np.concatenate(rows).astype(np.int32)
    return rowptr, colidx


def structure(rowptr, colidx, m, mcon):
    """camera-major order, triples grouped by block in (j, k) order and point order inside a block (index_build.hip)."""
    nvis = len(colidx)
    camobs = np.argsort(colidx, kind="stable").astype(np.int32)
    campos = np.empty(nvis, np.int32); campos[camobs] = np.arange(nvis, dtype=np.int32)
    camptr = np.concatenate([[0], np.cumsum(np.bincount(colidx, minlength=m))]).astype(np.int32)
    keys, tx, ty = [], [], []
    mm = m - mcon
    for i in range(len(rowptr) - 1):
        ks = [k for k in range(rowptr[i], rowptr[i + 1]) if colidx[k] >= mcon]
        for a in range(len(ks)):
            for b in range(a, len(ks)):
                keys.append((colidx[ks[a]] - mcon) * mm + (colidx[ks[b]] - mcon)); tx.append(campos[ks[a]]); ty.append(campos[ks[b]])
    keys = np.array(keys, np.int64); order = np.argsort(keys, kind="stable")
    keys = keys[order]; tx = np.array(tx, np.int32)[order]; ty = np.array(ty, np.int32)[order]
    ukeys, starts = np.unique(keys, return_index=True)
    blk_start = np.append(starts, len(keys)).astype(np.int32)
    return dict(camptr=camptr, campos=campos, tx=tx, ty=ty, blk_j=(mcon + ukeys // mm).astype(np.int32),
                blk_k=(mcon + ukeys % mm).astype(np.int32), blk_start=blk_start)


def row_entries(st, plan):
    """The row kernel's own triple array as index_build.hip:k_row_fill builds it from the plan's copy list."""
    out = np.zeros((plan["ntri"], 2), np.int32)
    for src, cnt, dst, rec0 in plan["fills"]:
        padded = (cnt + 15) // 16 * 16
        t = np.minimum(np.arange(padded), cnt - 1)
        out[dst:dst + padded, 0] = (st["tx"][src + t] - rec0) | np.where(np.arange(padded) < cnt, 0, B.ROW_DEAD)
        out[dst:dst + padded, 1] = st["ty"][src + t]
    return out


def walk(plan):
    """Yields (workgroup, wave, piece index, first entry, passes) exactly as k_schur_rows walks its flat pass list."""
    wgs, pieces = plan["wgs"], plan["pieces"]
    for w in range(len(wgs)):
        rec0, nrec, tri0, npass, piece0, npieces = wgs[w, :6]
        wpass = list(wgs[w, 6:10]) + [npass]; wpiece = wgs[w, 10:14]
        assert wpass[0] == 0 and all(wpass[q] <= wpass[q + 1] for q in range(4))
        covered = 0
        for wave in range(4):
            g, g1, pi = wpass[wave], wpass[wave + 1], wpiece[wave]
            while g < g1:
                assert pi < npieces
                np_ = pieces[piece0 + pi, 0]
                assert g + np_ <= g1, "a piece never straddles two waves"
                yield w, wave, piece0 + pi, tri0 + 16 * g, np_
                g += np_; pi += 1; covered += np_
            if wave < 3 and wpass[wave] < wpass[wave + 1]:
                assert wpiece[wave] <= wpiece[wave + 1] <= npieces
        assert covered == npass == pieces[piece0:piece0 + npieces, 0].sum()


@pytest.mark.parametrize("mode,n,m,mcon,L,dense_min,wg_min,tri_max,seed", [
    ("clique", 900, 18, 0, 32, 8, 1, 0, 1), ("clique", 2500, 24, 2, 128, 24, 0, 0, 2), ("banded", 1500, 30, 0, 16, 2, 1, 0, 3),
    ("random", 700, 14, 1, 48, 6, 1, 128, 4), ("random", 300, 9, 0, 16, 1, 1, 0, 5), ("clique", 400, 7, 0, 64, 1000000, 1, 0, 6),
    ("clique", 1500, 18, 0, 96, 12, 0, 192, 7), ("banded", 2000, 20, 0, 32, 4, 200, 0, 8)])
def test_row_plan_covers_dense_blocks_exactly_once(mode, n, m, mcon, L, dense_min, wg_min, tri_max, seed):
    rng = np.random.default_rng(seed)
    rowptr, colidx = random_scene(rng, n, m, mcon, mode)
    st = structure(rowptr, colidx, m, mcon)
    nblk = len(st["blk_j"])
    base = 1000
    plan = B.schur_row_plan(m, mcon, st["blk_j"], st["blk_k"], st["blk_start"], st["tx"], st["camptr"], None, L, dense_min, wg_min, tri_max, base)
    wgs, pieces, row0 = plan["wgs"], plan["pieces"], plan["blk_row0"]
    counts = np.diff(st["blk_start"])
    nrec = np.diff(st["camptr"])
    nseg = (nrec + L - 1) // L
    dense = (counts >= dense_min * np.maximum(nseg[st["blk_j"]], 1)) & (nseg[st["blk_j"]] > 0)
    row_dense = np.zeros(m, np.int64)
    np.add.at(row_dense, st["blk_j"][dense], counts[dense])
    wmin = wg_min if wg_min > 0 else (5 * L) // 2
    dense &= row_dense[st["blk_j"]] >= wmin * nseg[st["blk_j"]]
    if not dense.any():
        assert len(wgs) == 0 and len(pieces) == 0 and plan["nslots"] == 0 and not row0.any() and plan["ntri"] == 0
        return
    tmax = max(tri_max if tri_max > 0 else 12 * L, 128)
    ent = row_entries(st, plan)
    blk_of = np.repeat(np.arange(nblk), counts)
    covered = np.zeros(len(st["tx"]), np.int32)
    slot_seen = np.zeros(plan["nslots"], np.int32)
    cam_of_rec = np.repeat(np.arange(m), nrec)
    fills = plan["fills"]
    passes_of = {}
    for w, wave, pc, e0, np_ in walk(plan):
        rec0, nr = wgs[w, 0], wgs[w, 1]
        j = cam_of_rec[rec0]
        assert 1 <= nr <= L and cam_of_rec[rec0 + nr - 1] == j and (rec0 - st["camptr"][j]) % L == 0
        assert nr == min(L, st["camptr"][j + 1] - rec0)
        assert 16 * wgs[w, 3] <= tmax and wgs[w, 5] <= 32
        npass, diag, out, _ = pieces[pc]
        src, cnt, dst, frec0 = fills[pc]
        assert dst == e0 and frec0 == rec0 and npass == np_ == (cnt + 15) // 16 and cnt >= 1
        e = ent[e0:e0 + 16 * np_]
        live = e[:, 0] < B.ROW_DEAD
        assert live.sum() == cnt and live[:cnt].all()
        li = e[:, 0] & (B.ROW_DEAD - 1)
        assert (li < nr).all()                                              # the j side comes out of the slab (padding too)
        tri = np.arange(src, src + cnt)
        assert np.array_equal(li[:cnt] + rec0, st["tx"][tri]) and np.array_equal(e[:cnt, 1], st["ty"][tri])
        b = blk_of[src]
        assert (blk_of[tri] == b).all() and dense[b] and st["blk_j"][b] == j
        assert diag == int(st["blk_j"][b] == st["blk_k"][b])
        covered[tri] += 1
        assert base + row0[b] <= out < base + row0[b + 1]
        slot_seen[out - base] += 1
        passes_of.setdefault(w, [0, 0, 0, 0])[wave] += np_
    for w, ps in passes_of.items():
        assert max(ps) - min(ps) <= 1, ps
    assert len(passes_of) == len(wgs)
    assert (covered[dense[blk_of]] == 1).all() and (covered[~dense[blk_of]] == 0).all()
    assert (slot_seen == 1).all()
    # slots of a block follow its triple order
    slot_src = {out - base: src for (npass, diag, out, _), (src, cnt, dst, r0) in zip(pieces, fills)}
    for b in np.flatnonzero(dense):
        s = [slot_src[t] for t in range(row0[b], row0[b + 1])]
        assert s == sorted(s) and len(s) >= 1
    assert not (np.diff(row0)[~dense]).any()
    # the entries of the workgroups tile the array without gaps
    order = np.argsort(wgs[:, 2])
    assert wgs[order[0], 2] == 0 and np.array_equal(wgs[order, 2][1:], np.cumsum(16 * wgs[order, 3])[:-1]) and plan["ntri"] == 16 * wgs[:, 3].sum()


def test_row_plan_replay_reproduces_the_block_sums():
    rng = np.random.default_rng(11)
    n, m, mcon, L, cnp = 1200, 16, 1, 32, 9
    rowptr, colidx = random_scene(rng, n, m, mcon, "clique")
    st = structure(rowptr, colidx, m, mcon)
    nvis = len(colidx)
    A = rng.standard_normal((nvis, 2, cnp)); Bm = rng.standard_normal((nvis, 2, 3)); Cm = rng.standard_normal((nvis, 2, 3))
    nblk = len(st["blk_j"])
    ref = np.zeros((nblk, cnp, cnp))
    for b in range(nblk):
        for t in range(st["blk_start"][b], st["blk_start"][b + 1]):
            x, y = st["tx"][t], st["ty"][t]
            ref[b] += A[x].T @ (Cm[x] @ Bm[y].T) @ A[y]
    plan = B.schur_row_plan(m, mcon, st["blk_j"], st["blk_k"], st["blk_start"], st["tx"], st["camptr"], None, L, 4, 1, 160, 0)
    wgs, pieces, row0 = plan["wgs"], plan["pieces"], plan["blk_row0"]
    ent = row_entries(st, plan)
    partial = np.zeros((plan["nslots"], cnp, cnp))
    for w, wave, pc, e0, np_ in walk(plan):
        rec0, nr = wgs[w, 0], wgs[w, 1]
        slabA, slabC = A[rec0:rec0 + nr], Cm[rec0:rec0 + nr]            # what the workgroup streams into LDS
        acc = np.zeros((cnp, cnp))
        for e in ent[e0:e0 + 16 * np_]:
            if e[0] < B.ROW_DEAD:                                        # padding contributes zeros
                acc += slabA[e[0]].T @ (slabC[e[0]] @ Bm[e[1]].T) @ A[e[1]]
        partial[pieces[pc, 2]] = acc
    ndense = 0
    for b in range(nblk):
        if row0[b + 1] > row0[b]:
            ndense += 1
            got = partial[row0[b]:row0[b + 1]].sum(axis=0)
            assert np.abs(got - ref[b]).max() <= 1e-9 * max(1.0, np.abs(ref[b]).max())
    assert ndense > nblk // 2
