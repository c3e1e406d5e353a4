"""Index bookkeeping, bit-exact.

The PRODUCT's arrays -- the CRS run_sfm derives from the dense vmask, and the camera-major index / co-visibility structure the
device-side construction (bundler_sfm_amd/csrc/index_build.hip) leaves in HBM -- are compared with the REFERENCE'S OWN
struct sba_crsm (lib/sba-1.5/sba_levmar.c:653-663) and its camera-major traversal sba_crsm_col_elmidxs
(lib/sba-1.5/sba_crsm.c:183-212), compiled into oracle/_ref/libsfmref.so (oracle/ref_harness.c:ref_crsm_index).
"""
import ctypes as C

import numpy as np
import pytest

import oracle_util as O

needs_ref = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")


def random_vmask(rng, n, m, density, mcon_rows=True):
    vm = (rng.random((n, m)) < density).astype(np.uint8)
    if n > 4:
        vm[1, :] = 0                       # a point nobody sees
        vm[2, :] = 1                       # a point everybody sees
    if m > 3:
        vm[:, m - 1] = 0                   # a camera without observations
    return vm


def product_crs(B, vm):
    n, m = vm.shape
    ip = C.POINTER(C.c_int)
    nvis = B.lib.bsfm_crs_from_vmask(n, m, vm.ctypes.data_as(C.c_char_p), None, None)
    rp = np.zeros(n + 1, np.int32); ci = np.zeros(max(nvis, 1), np.int32)
    assert B.lib.bsfm_crs_from_vmask(n, m, vm.ctypes.data_as(C.c_char_p), rp.ctypes.data_as(ip), ci.ctypes.data_as(ip)) == nvis
    return rp, ci[:nvis]


@needs_ref
@pytest.mark.parametrize("n,m,density,seed", [(300, 40, 0.2, 1), (57, 9, 0.6, 2), (1, 5, 1.0, 3), (1000, 130, 0.03, 4)])
def test_crs_from_vmask_is_the_reference_crsm(bsfm, n, m, density, seed):
    vm = random_vmask(np.random.default_rng(seed), n, m, density)
    rp, ci = product_crs(bsfm, vm)
    r = O.ref_crsm_index(n, m, vm)
    assert np.array_equal(rp, r["rowptr"]) and np.array_equal(ci, r["colidx"])
    assert np.array_equal(r["val"], np.arange(len(ci)))          # val[k] = k: the observation ordering contract


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("n,m,density,seed", [(300, 40, 0.2, 1), (57, 9, 0.6, 2), (1, 5, 1.0, 3), (1000, 130, 0.03, 4), (4096, 16, 0.5, 5),
                                              (33, 1, 0.5, 6), (700, 1000, 0.01, 7), (5, 3, 0.0, 8), (64, 64, 1.0, 9)])
def test_device_crs_from_vmask_is_the_reference_crsm(gpu_bsfm, n, m, density, seed):
    """run_sfm's vmask -> CRS for masks of scale runs ON THE DEVICE (index_build.hip:crs_from_vmask_device: flat byte string, SWAR
    non-zero test, piece counts scanned by rocPRIM, compaction): bit-identical to the reference's struct sba_crsm
    (lib/sba-1.5/sba_levmar.c:653-663 via oracle/_ref) and to the host loop, for row lengths below / at / above the 16-byte lane
    width, empty rows, an empty mask, a full mask, and total sizes that are and are not multiples of 16."""
    B = gpu_bsfm
    vm = random_vmask(np.random.default_rng(seed), n, m, density)
    if density == 0.0:
        vm[:] = 0
    if seed == 4:
        vm[10:20] = 0                                               # empty rows
    ip = C.POINTER(C.c_int)
    rp = np.full(n + 1, -7, np.int32); ci = np.full(max(int((vm != 0).sum()), 1), -7, np.int32)
    nvis = B.lib.bsfm_crs_from_vmask_device(n, m, vm.ctypes.data_as(C.c_char_p), rp.ctypes.data_as(ip), ci.ctypes.data_as(ip), None)
    assert nvis == int((vm != 0).sum())
    hrp, hci = product_crs(B, vm)
    assert np.array_equal(rp, hrp) and np.array_equal(ci[:nvis], hci)
    r = O.ref_crsm_index(n, m, vm)
    assert np.array_equal(rp, r["rowptr"]) and np.array_equal(ci[:nvis], r["colidx"])


@pytest.mark.gpu
def test_run_sfm_through_the_device_crs_equals_the_host_crs(gpu_bsfm):
    """The drop-in run_sfm with the mask turned into the CRS on the device (forced: BSFM_VMASK_DEVICE_MIN=0 would be read once per
    process, so the scene is made large enough instead: 2 200 points x 500 cameras = 1.1 MB) gives bit-identical results to the
    resident-problem API fed with the host CRS."""
    B = gpu_bsfm
    m, n = 500, 2200
    s = B.synth_ba(m, n, 8)
    vm = B.dense_vmask(n, m, s["rowptr"], s["colidx"])
    assert vm.size >= (1 << 20)
    cams = B.copy_cameras(s["cams"]); pts = s["pts"].copy()
    opt = B.default_options(verbose=0, itmax=3)
    rc, info = B.run_sfm(n, m, 0, vm, s["proj"], 1, 0, 1, 1, cams, pts, eps2=1e-12, options=opt)
    assert B.lib.bsfm_run_sfm_last_ms(b"crs_on_device") == 1.0 and B.lib.bsfm_run_sfm_last_ms(b"total") > 0.0
    opt2 = B.default_options(verbose=0, itmax=3); opt2.opts[2] = 1e-12
    pb = B.Problem(n, m, s["rowptr"], s["colidx"], s["proj"], s["cams"], s["pts"], options=opt2)
    rc2, info2 = pb.solve()
    p, cams2, pts2 = pb.download()
    pb.close()
    assert rc == rc2 and list(info) == list(info2)
    assert np.array_equal(pts, pts2)


def schur_restatement(rowptr, colidx, campos, mcon, m):
    """numpy restatement of the order the reference visits the co-visibility pairs in (sba_levmar.c:1182-1268): block (j, k),
    j <= k, and inside a block ascending point index."""
    keys, vals, pts = [], [], []
    mm = m - mcon
    for i in range(len(rowptr) - 1):
        ks = [k for k in range(rowptr[i], rowptr[i + 1]) if colidx[k] >= mcon]
        for a in range(len(ks)):
            for b in range(a, len(ks)):
                keys.append((colidx[ks[a]] - mcon) * mm + (colidx[ks[b]] - mcon)); vals.append((campos[ks[a]], campos[ks[b]])); pts.append(i)
    keys = np.array(keys, np.int64); order = np.argsort(keys, kind="stable")
    return keys[order], np.array(vals, np.int32).reshape(-1, 2)[order], np.array(pts, np.int32)[order]


def expected_launch(by_slot, sc, pts, n, m, mcon, nvis):
    """numpy restatement of the launch order of the Schur tasks (index_build.hip: build_schur, k_task_keys_clustered, k_launch_order)."""
    ntasks = len(by_slot)
    if nvis * 272 <= (4 << 20):
        return by_slot                                     # block order, no padding slots
    mm = m - mcon
    adj = [[] for _ in range(mm)]
    for a, c in zip(sc["blk_j"] - mcon, sc["blk_k"] - mcon):
        if a != c:
            adj[a].append(c); adj[c].append(a)
    seen = np.zeros(mm, bool); q = []
    for s0 in range(mm):                                   # breadth-first numbering, components one after the other
        if seen[s0]:
            continue
        seen[s0] = True; q.append(s0); h = len(q) - 1
        while h < len(q):
            for v in adj[q[h]]:
                if not seen[v]:
                    seen[v] = True; q.append(v)
            h += 1
    rank = np.zeros(mm, np.int64); rank[np.array(q)] = np.arange(mm)
    ntask = np.diff(sc["blk_task0"])
    slices = min(max(1, int(ntask.max())), 4096)
    blk_of = np.repeat(np.arange(len(ntask)), ntask)
    ra, rb = rank[sc["blk_j"][blk_of] - mcon], rank[sc["blk_k"][blk_of] - mcon]
    first = pts[by_slot[:, 0]].astype(np.int64)
    key = ((first * slices // max(n, 1)) << 48) | (np.minimum(ra, rb) << 24) | np.maximum(ra, rb)
    order = np.argsort(key, kind="stable")
    nwg = (ntasks + 3) // 4
    out = np.zeros((4 * nwg, 4), np.int32); out[:, 3] = -1
    for wg in range(nwg):
        x = wg & 7
        nxt = (wg >> 3) + sum((nwg - r + 7) // 8 for r in range(x) if nwg > r)
        for w in range(4):
            src = 4 * nxt + w
            if src < ntasks:
                out[4 * wg + w] = by_slot[order[src]]
    return out


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("n,m,density,mcon,seed", [(300, 40, 0.2, 0, 1), (57, 9, 0.6, 2, 2), (400, 64, 0.1, 0, 5), (2000, 130, 0.03, 3, 4),
                                                   (2600, 70, 0.1, 1, 7), (6000, 12, 0.5, 0, 8)])   # the last two are past the block-order limit (one / eight tasks per block)
def test_device_index_is_the_reference_index(gpu_bsfm, n, m, density, mcon, seed):
    B = gpu_bsfm
    rng = np.random.default_rng(seed)
    vm = random_vmask(rng, n, m, density)
    rp, ci = product_crs(B, vm)
    nvis = len(ci)
    s = B.synth_ba(m, n, 3)
    # nobs >= nvars is not needed to build the index; the LM is never started here
    pb = B.Problem(n, m, rp, ci, rng.standard_normal(2 * nvis), s["cams"], s["pts"], mcon=mcon, options=B.default_options(verbose=0))
    ix = pb.export_index()
    r = O.ref_crsm_index(n, m, vm)
    assert np.array_equal(ix["rowptr"], r["rowptr"]) and np.array_equal(ix["colidx"], r["colidx"])
    assert np.array_equal(ix["camptr"], r["camptr"])
    assert np.array_equal(ix["camobs"], r["camobs"])             # traversal order of sba_crsm_col_elmidxs, camera by camera
    assert np.array_equal(ix["cam_pt"], r["campt"])
    assert np.array_equal(ix["campos"][ix["camobs"]], np.arange(nvis))
    assert np.array_equal(ix["obs_pt"], np.repeat(np.arange(n), np.diff(rp)))
    assert np.array_equal(ix["cam_cam"], ci[ix["camobs"]])
    # co-visibility structure
    sc = pb.export_schur()
    keys, vals, pts = schur_restatement(rp, ci, ix["campos"], mcon, m)
    assert np.array_equal(sc["triples"], vals) and np.array_equal(sc["tri_pt"], pts)
    ukeys, starts = np.unique(keys, return_index=True)
    mm = m - mcon
    assert np.array_equal(sc["blk_j"], mcon + ukeys // mm) and np.array_equal(sc["blk_k"], mcon + ukeys % mm)
    counts = np.diff(np.append(starts, len(keys)))
    CH = int(B.lib.bsfm_schur_chunk())            # triples per task (index_build.hip:schur_chunk, BSFM_SCHUR_CHUNK)
    assert CH % 16 == 0 and 16 <= CH <= 192
    ntask = (counts + CH - 1) // CH
    assert np.array_equal(sc["blk_task0"], np.concatenate([[0], np.cumsum(ntask)]))
    tasks = sc["tasks"]; real = tasks[tasks[:, 3] >= 0]
    assert len(real) == sc["ntasks"] == int(ntask.sum())
    by_slot = real[np.argsort(real[:, 3])]
    assert np.array_equal(by_slot[:, 3], np.arange(len(real)))
    exp = []
    for b in range(len(ukeys)):
        for t in range(ntask[b]):
            s0 = starts[b] + CH * t
            exp.append((s0, min(CH, starts[b] + counts[b] - s0), int(sc["blk_j"][b] == sc["blk_k"][b])))
    assert np.array_equal(by_slot[:, :3], np.array(exp, np.int32).reshape(-1, 3))
    # launch order (index_build.hip): block order when the records fit one L2, else (point slice, cameras in breadth-first numbering),
    # one contiguous stretch of that order per XCD (workgroup % 8)
    assert np.array_equal(tasks, expected_launch(by_slot, sc, pts, n, m, mcon, nvis))
    pb.close()


@pytest.mark.gpu
def test_device_index_at_the_headline_size(gpu_bsfm):
    """1 000 cameras / 500 000 points / 5 M observations: camera-major order = stable sort by camera (what
    sba_crsm_col_elmidxs enumerates), Schur structure checked through its invariants; also reports the build time."""
    B = gpu_bsfm
    m, n = 1000, 500000
    s = B.synth_ba(m, n, 10)
    pb = B.Problem(n, m, s["rowptr"], s["colidx"], s["proj"], s["cams"], s["pts"], options=B.default_options(verbose=0))
    ix = pb.export_index()
    order = np.argsort(s["colidx"], kind="stable").astype(np.int32)
    assert np.array_equal(ix["camobs"], order)
    assert np.array_equal(ix["camptr"], np.concatenate([[0], np.cumsum(np.bincount(s["colidx"], minlength=m))]))
    assert np.array_equal(ix["campos"][order], np.arange(len(order)))
    sc = pb.export_schur()
    assert len(sc["triples"]) == n * 55
    cam = ix["cam_cam"]
    j = cam[sc["triples"][:, 0]].astype(np.int64); k = cam[sc["triples"][:, 1]].astype(np.int64)
    key = j * m + k
    assert (j <= k).all() and (np.diff(key) >= 0).all()                       # grouped by block in (j, k) order
    same = np.diff(key) == 0
    assert (np.diff(sc["tri_pt"])[same] > 0).all()                            # point order inside a block
    assert np.array_equal(ix["cam_pt"][sc["triples"][:, 0]], sc["tri_pt"]) and np.array_equal(ix["cam_pt"][sc["triples"][:, 1]], sc["tri_pt"])
    print("index build: %.2f ms on the device, problem_create %.1f ms wall (upload %.1f, index %.1f, allocation %.1f)" % (
        pb.phase_ms("index_build"), pb.phase_ms("create_total"), pb.phase_ms("create_upload"), pb.phase_ms("create_index"),
        pb.phase_ms("create_alloc")))
    assert pb.phase_ms("index_build") < 100.0
    pb.close()
