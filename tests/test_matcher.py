"""Matcher parity.  CPU part: the oracle's exhaustive 2-NN (oracle/match_oracle.c) against the reference's own
MatchKeys output (fixture generated from oracle/_ref with the ANN visit cap disabled = exact search, and with the
shipped 200-visit cap).  GPU part: the HIP brute-force kernel must equal the exact reference search bit for bit."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_util as O

HERE = os.path.dirname(os.path.abspath(__file__))
MG = np.load(os.path.join(HERE, "golden", "match_golden.npz"))
U = C.POINTER(C.c_ubyte)


def synth_keys(B, n, seed, dup=None):
    k = np.zeros((n, 128), np.uint8)
    B.lib.bsfm_synth_keys(n, seed, None if dup is None else dup.ctypes.data_as(U), 0 if dup is None else len(dup),
                          k.ctypes.data_as(U))
    return k


def gpu_match(B, k1, k2, ratio=0.6):
    """Runs BOTH scan kernels of the matcher (k_match_l2: exact running top-2; k_match_bound: one running maximum per slot, bounds,
    exact rescan of the winning slot -- match_l2.hip) and insists that they agree, so every parity test below covers both."""
    res = []
    for mode in (1, 2):
        old = B.lib.bsfm_match_kernel(mode)
        try:
            out = np.zeros((len(k1), 2), np.int32)
            cnt = B.lib.bsfm_match_keys_l2(len(k1), k1.ctypes.data_as(U), len(k2), k2.ctypes.data_as(U), ratio,
                                           out.ctypes.data_as(C.POINTER(C.c_int)), len(k1))
        finally:
            B.lib.bsfm_match_kernel(old)
        res.append((cnt, out[:max(cnt, 0)]))
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1]), "top-2 and rescan kernels disagree"
    return res[0]


@pytest.mark.parametrize("name", ["a", "b", "tiny"])
def test_oracle_equals_reference_exact_search(name):
    got = O.port_match(MG[f"{name}_k1"], MG[f"{name}_k2"])
    assert np.array_equal(got, MG[f"{name}_exact"])
    # the shipped approximate search may only MISS matches relative to the exact one, never invent different ones
    approx = {tuple(r) for r in MG[f"{name}_ann200"]}
    assert approx <= {tuple(r) for r in got} or len(approx - {tuple(r) for r in got}) <= len(approx) // 20


def test_integer_ratio_test_is_equivalent_to_the_double_compare():
    # (double)d0 < 0.6*0.6*(double)d1  <=>  25*d0 < 9*d1 on the whole int32 range used
    rng = np.random.default_rng(0)
    d1 = rng.integers(0, 128 * 255 * 255 + 1, 200000)
    thr = np.floor(0.36 * d1).astype(np.int64)
    for delta in (-1, 0, 1):
        d0 = np.clip(thr + delta, 0, None)
        lhs = d0.astype(np.float64) < (0.6 * 0.6) * d1.astype(np.float64)
        assert np.array_equal(lhs, 25 * d0 < 9 * d1)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["a", "b", "tiny"])
def test_gpu_matches_reference_fixture(gpu_bsfm, name):
    cnt, got = gpu_match(gpu_bsfm, MG[f"{name}_k1"], MG[f"{name}_k2"])
    assert cnt == len(MG[f"{name}_exact"])
    assert np.array_equal(got, MG[f"{name}_exact"])


@pytest.mark.gpu
@pytest.mark.parametrize("n1,n2", [(1, 2), (63, 17), (64, 64), (65, 129), (500, 333), (2049, 1000)])
def test_gpu_matches_oracle_ragged_sizes(gpu_bsfm, n1, n2):
    B = gpu_bsfm
    k2 = synth_keys(B, n2, 100 + n2)
    k1 = synth_keys(B, n1, 200 + n1, dup=k2)
    cnt, got = gpu_match(B, k1, k2)
    ref = O.port_match(k1, k2)
    assert cnt == len(ref) and np.array_equal(got, ref)
    # extreme descriptors: all-zero / all-255 rows and exact duplicates (d0 == 0)
    k1[0] = 0; k1[-1] = 255; k2[0] = 255
    if n2 > 2:
        k2[1] = k1[n1 // 2]
    cnt, got = gpu_match(B, k1, k2)
    ref = O.port_match(k1, k2)
    assert cnt == len(ref) and np.array_equal(got, ref)


@pytest.mark.gpu
def test_gpu_rejects_what_the_reference_cannot_search(gpu_bsfm):
    k = synth_keys(gpu_bsfm, 4, 1)
    cnt, _ = gpu_match(gpu_bsfm, k, k[:1])
    assert cnt == -1        # ANN aborts when asked for 2 neighbours among < 2 points


@pytest.mark.gpu
def test_key_match_full_text_is_byte_identical(gpu_bsfm, tmp_path):
    """KeyMatchFull driver: 6 images, all pairs, output text vs the same loop driven by the CPU oracle
    (src/KeyMatchFull.cpp:105-151: pair header 'j i', count, 'idx_j idx_i' lines, only pairs with >= 16 matches)."""
    B = gpu_bsfm
    sizes = [300, 0, 257, 400, 64, 350]
    keys = []
    prev = None
    for i, nk in enumerate(sizes):
        k = synth_keys(B, nk, 1000 + i, dup=prev) if nk else np.zeros((0, 128), np.uint8)
        keys.append(k)
        if nk:
            prev = k
    arr = (U * len(sizes))(*[k.ctypes.data_as(U) for k in keys])
    nks = np.array(sizes, np.int32)
    out = tmp_path / "matches.init.txt"
    rc = B.lib.bsfm_key_match_full(len(sizes), nks.ctypes.data_as(C.POINTER(C.c_int)), arr, 0.6, -1, str(out).encode())
    expect = []
    npairs = 0
    for i in range(len(sizes)):
        if sizes[i] == 0:
            continue
        for j in range(i):
            if sizes[j] == 0:
                continue
            mt = O.port_match(keys[j], keys[i])
            if len(mt) >= 16:
                npairs += 1
                expect.append(f"{j} {i}\n{len(mt)}\n" + "".join(f"{a} {b}\n" for a, b in mt))
    assert rc == npairs and npairs > 0
    assert out.read_text() == "".join(expect)


@pytest.mark.gpu
def test_many_small_images_block_to_pair_lookup(gpu_bsfm, tmp_path):
    """200 images of 100..169 keys (one or two query blocks each): the launch for database image i carries up to 199 pairs, so
    the kernel's block -> pair search (match_l2.hip, head of k_match_l2) runs over a long, ragged table; text vs the oracle loop."""
    B = gpu_bsfm
    sizes = [100 + (7 * i) % 70 for i in range(200)]
    keys = []
    prev = None
    for i, nk in enumerate(sizes):
        k = synth_keys(B, nk, 5000 + i, dup=prev)
        keys.append(k)
        prev = k
    arr = (U * len(sizes))(*[k.ctypes.data_as(U) for k in keys])
    nks = np.array(sizes, np.int32)
    out = tmp_path / "matches.init.txt"
    rc = B.lib.bsfm_key_match_full(len(sizes), nks.ctypes.data_as(C.POINTER(C.c_int)), arr, 0.6, -1, str(out).encode())
    expect = []
    for i in range(len(sizes)):
        for j in range(i):
            mt = O.port_match(keys[j], keys[i])
            if len(mt) >= 16:
                expect.append(f"{j} {i}\n{len(mt)}\n" + "".join(f"{a} {b}\n" for a, b in mt))
    assert rc == len(expect) and len(expect) >= 60
    assert out.read_text() == "".join(expect)


def _write_key_file(path, keys, gz=False):
    """Lowe's ASCII .key format (keys2a.cpp:183-190): 'num 128', then per key 'row col scale ori' and the 128 bytes
    on 7 lines (6 x 20 + 8)."""
    import gzip
    lines = [f"{len(keys)} 128\n"]
    for q, d in enumerate(keys):
        lines.append(f"{10.5 + q:.2f} {20.25 + q:.2f} {1.5:.2f} {-0.733:.3f}\n")
        for s in range(0, 120, 20):
            lines.append(" " + " ".join(str(int(v)) for v in d[s:s + 20]) + "\n")
        lines.append(" " + " ".join(str(int(v)) for v in d[120:128]) + "\n")
    data = "".join(lines).encode()
    if gz:
        with gzip.open(str(path) + ".gz", "wb") as f:
            f.write(data)
    else:
        with open(path, "wb") as f:
            f.write(data)


@pytest.mark.gpu
def test_keymatchfull_cli_same_process_boundary_as_reference(gpu_bsfm, tmp_path):
    """The command-line tool (tools/KeyMatchFull.cpp; reference usage KeyMatchFull.cpp:64-76): list file -> key files
    (plain, gzip fallback, a missing one) -> matches file, byte-identical to the in-process driver and to the oracle loop;
    also the window_radius argument."""
    import subprocess
    B = gpu_bsfm
    exe = os.path.join(os.path.dirname(HERE), "bundler_sfm_amd", "bin", "KeyMatchFull")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    sizes = [280, 300, 0, 190, 320]
    keys, prev = [], None
    for i, nk in enumerate(sizes):
        k = synth_keys(B, nk, 4000 + i, dup=prev) if nk else np.zeros((0, 128), np.uint8)
        keys.append(k)
        if nk:
            prev = k
    names = []
    for i, k in enumerate(keys):
        p = tmp_path / f"img{i}.key"
        names.append(str(p))
        if sizes[i] == 0:
            continue                          # image 2: no file at all -> "Could not open file", 0 keys, skipped
        _write_key_file(p, k, gz=(i == 3))    # image 3 only exists as .key.gz
    (tmp_path / "list.txt").write_text("\n".join(names) + "\n")

    def expect(window):
        out = []
        for i in range(len(sizes)):
            for j in range(max(i - window, 0) if window > 0 else 0, i):
                if sizes[i] == 0 or sizes[j] == 0:
                    continue
                mt = O.port_match(keys[j], keys[i])
                if len(mt) >= 16:
                    out.append(f"{j} {i}\n{len(mt)}\n" + "".join(f"{a} {b}\n" for a, b in mt))
        return "".join(out)

    for window, extra in ((-1, []), (1, ["1"])):
        outp = tmp_path / f"matches{window}.txt"
        r = subprocess.run([exe, str(tmp_path / "list.txt"), str(outp)] + extra, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "Could not open file" in r.stdout and "[KeyMatchFull] Matching took" in r.stdout
        text = outp.read_text()
        assert text == expect(window)
        assert text.count("\n") > 50          # something was actually matched
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode != 0 and "Usage:" in r.stdout


@pytest.mark.gpu
def test_sharded_matcher_merges_to_the_single_run_file(gpu_bsfm, tmp_path):
    """SURVEY 8(e), matcher: pairs are split over ranks by database image (no collective); the merged per-rank files
    equal the single-run file byte for byte."""
    B = gpu_bsfm
    sizes = [260, 300, 0, 190, 320, 280, 310]
    keys, prev = [], None
    for i, nk in enumerate(sizes):
        k = synth_keys(B, nk, 7000 + i, dup=prev) if nk else np.zeros((0, 128), np.uint8)
        keys.append(k)
        if nk:
            prev = k
    arr = (U * len(sizes))(*[k.ctypes.data_as(U) for k in keys])
    nks = np.array(sizes, np.int32)
    ip = nks.ctypes.data_as(C.POINTER(C.c_int))
    single = tmp_path / "single.txt"
    n_single = B.lib.bsfm_key_match_full(len(sizes), ip, arr, C.c_double(0.6), -1, str(single).encode())
    assert n_single > 3
    world = 3
    paths, total = [], 0
    for r in range(world):
        p = tmp_path / f"rank{r}.txt"
        paths.append(str(p).encode())
        got = B.lib.bsfm_key_match_full_sharded(len(sizes), ip, arr, C.c_double(0.6), -1, paths[-1], r, world)
        assert got >= 0
        total += got
    assert total == n_single
    merged = tmp_path / "merged.txt"
    cp = (C.c_char_p * world)(*paths)
    assert B.lib.bsfm_merge_match_files(world, cp, str(merged).encode()) == n_single
    assert merged.read_bytes() == single.read_bytes()
    assert B.lib.bsfm_key_match_full_sharded(len(sizes), ip, arr, C.c_double(0.6), -1, paths[0], 3, 3) < 0     # bad rank


# ---- BASELINE.json configs[4] shape (5 000 keys per image) and database images beyond one 8 192-key scan segment -------------
# Fixtures: the reference's exact search (tests/golden/make_match_cfg5_golden.py); descriptors regenerated from their seeds.
CFG5 = {"p5000a": (5000, 502, 5000, 501), "p5000b": (5000, 504, 5000, 503), "seg2": (2000, 508, 9000, 507),
        "seg3": (3000, 506, 17000, 505)}
_cfg5_path = os.path.join(HERE, "golden", "match_cfg5_golden.npz")
MG5 = np.load(_cfg5_path) if os.path.exists(_cfg5_path) else None


def cfg5_keys(B, name):
    n1, s1, n2, s2 = CFG5[name]
    k2 = synth_keys(B, n2, s2)
    return synth_keys(B, n1, s1, dup=k2), k2


def test_oracle_equals_reference_exact_search_at_5000_keys(bsfm):
    assert MG5 is not None, "tests/golden/match_cfg5_golden.npz missing"
    k1, k2 = cfg5_keys(bsfm, "p5000a")
    assert np.array_equal(O.port_match(k1, k2), MG5["p5000a_exact"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CFG5))
def test_gpu_matches_reference_exact_search_at_config5_sizes(gpu_bsfm, name):
    """Bit-identical to MatchKeys(..., max_pts_visit = 0) (src/keys2a.cpp:347-372) for two 5 000 x 5 000 pairs and for database
    images of 9 000 and 17 000 keys (2 and 3 segments of the kernel's 8 192-key scan loop, match_l2.hip)."""
    assert MG5 is not None, "tests/golden/match_cfg5_golden.npz missing"
    k1, k2 = cfg5_keys(gpu_bsfm, name)
    cnt, got = gpu_match(gpu_bsfm, k1, k2)
    ref = MG5[f"{name}_exact"]
    assert cnt == len(ref) and np.array_equal(got, ref)
    if name.startswith("seg"):          # matches must come from every segment of the database image
        segs = set((ref[:, 1] // 8192).tolist())
        assert segs == set(range((CFG5[name][2] + 8191) // 8192))


@pytest.mark.gpu
def test_gpu_best_and_second_best_in_different_segments(gpu_bsfm):
    """The running top-2 is kept per 8 192-key segment and merged afterwards: plant the nearest and the second nearest
    neighbour of a query in different segments (and a tie across segments) and compare with the exhaustive oracle."""
    B = gpu_bsfm
    k2 = synth_keys(B, 20000, 77)
    k1 = synth_keys(B, 64, 78)
    rng = np.random.default_rng(5)
    for q in range(64):
        a, b = int(rng.integers(0, 8192)), int(rng.integers(8192, 20000))
        if q % 2:
            a, b = b, a
        near = k1[q].astype(np.int32)
        k2[a] = np.clip(near + rng.integers(-2, 3, 128), 0, 255).astype(np.uint8)             # nearest
        k2[b] = np.clip(near + rng.integers(-(3 + q % 5), 4 + q % 5, 128), 0, 255).astype(np.uint8)   # runner-up: ratio test decides
    k2[19999] = k2[100]                                                                      # exact duplicate across segments
    cnt, got = gpu_match(B, k1, k2)
    ref = O.port_match(k1, k2)
    assert cnt == len(ref) and np.array_equal(got, ref)


@pytest.mark.gpu
def test_gpu_runner_up_inside_the_winning_slot(gpu_bsfm):
    """The rescan kernel keeps one running minimum per (query row, column class mod 64) slot and measures the winning slot's other
    columns exactly only for rows that pass the ratio test against the bound: plant nearest and runner-up in the SAME slot (columns
    64 t apart, same 8 192-key segment), at distances on both sides of the ratio test, also in the last partial tile."""
    B = gpu_bsfm
    n2 = 9000 + 37
    k2 = synth_keys(B, n2, 81)
    k1 = synth_keys(B, 300, 82)
    rng = np.random.default_rng(7)
    for q in range(300):
        cls = int(rng.integers(0, 64))
        seg = int(rng.integers(0, 2))
        lo, hi = (0, 8192) if seg == 0 else (8192, n2)
        cols = np.arange(lo + (cls - lo) % 64, hi, 64)
        a, b = rng.choice(cols, 2, replace=False)
        if q % 7 == 0:
            a = cols[-1]                                   # the last (partial) tile of the segment
            b = cols[0] if b == a else b
        near = k1[q].astype(np.int32)
        k2[a] = np.clip(near + rng.integers(-2, 3, 128), 0, 255).astype(np.uint8)
        w = 3 + q % 6                                      # runner-up from "fails the test" to "passes it clearly"
        k2[b] = np.clip(near + rng.integers(-w, w + 1, 128), 0, 255).astype(np.uint8)
    cnt, got = gpu_match(B, k1, k2)
    ref = O.port_match(k1, k2)
    assert cnt == len(ref) and np.array_equal(got, ref)
    assert 30 < cnt < 290                                  # both outcomes of the test occur


def _with_sq_dist(base, rng, sq):
    """a copy of `base` (values kept inside 40..200) whose squared distance to it is exactly `sq` (sum of 25s, then 9 / 4 / 1 steps)"""
    k = base.astype(np.int32).copy()
    pos = list(rng.permutation(128))
    for step in (5, 3, 2, 1):
        while sq >= step * step and pos:
            i = pos.pop()
            k[i] += step if k[i] < 128 else -step
            sq -= step * step
    assert sq == 0
    return k.astype(np.uint8)


@pytest.mark.gpu
def test_gpu_ratio_test_on_the_parity_bit_and_ties(gpu_bsfm):
    """k_match_bound knows every slot's best distance only to within one unit (the parity bit of the column term): plant nearest /
    second-nearest pairs in DIFFERENT slots whose ratio test is decided by that unit (25 d0 vs 9 d1 one apart on either side, for
    d1 of both parities), exact duplicates (two slots tie at distance 0) and equal nearest distances in two slots -- the rows the
    kernel has to measure against the whole image.  Reference semantics: src/keys2a.cpp:362, strict inequality in double."""
    B = gpu_bsfm
    n2 = 3000
    rng = np.random.default_rng(11)
    k2 = synth_keys(B, n2, 83)
    k1 = np.clip(synth_keys(B, 120, 84), 40, 200)
    cols = rng.permutation(n2)
    expect_pass = {}
    for q in range(120):
        a, b = int(cols[2 * q]), int(cols[2 * q + 1])
        if (a - b) % 64 == 0:
            b = (b + 1) % n2
        m = 4 + q % 20
        kind = q % 6
        if kind < 4:             # d0 = 9 m; 25 d0 = 225 m against 9 d1 with d1 = 25 m + delta: passes iff delta >= 1
            delta = (-1, 0, 1, 2)[kind]
            d0, d1 = 9 * m, 25 * m + delta
            expect_pass[q] = 25 * d0 < 9 * d1
        elif kind == 4:          # exact duplicates in two slots: 0 < 0.36 * 0 is false
            d0, d1 = 0, 0
            expect_pass[q] = False
        else:                    # the same non-zero nearest distance in two slots
            d0 = d1 = 9 * m + 1
            expect_pass[q] = False
        k2[a] = _with_sq_dist(k1[q], rng, d0)
        k2[b] = _with_sq_dist(k1[q], rng, d1)
    cnt, got = gpu_match(B, k1, k2)
    ref = O.port_match(k1, k2)
    assert cnt == len(ref) and np.array_equal(got, ref)
    accepted = set(got[:, 0].tolist())
    for q, ok in expect_pass.items():
        assert (q in accepted) == ok, (q, ok)


@pytest.mark.gpu
def test_auto_kernel_choice_follows_the_acceptance_rate(gpu_bsfm, tmp_path):
    """auto starts with the rescan kernel and moves to the top-2 kernel once the launches it has seen accept more than 5 % of their
    queries; whatever it picks, matches.init.txt is the same bytes."""
    B = gpu_bsfm
    images, nk = 14, 700
    base = synth_keys(B, nk, 90)
    rng = np.random.default_rng(3)
    keys = []
    for i in range(images):
        k = synth_keys(B, nk, 91 + i)
        if i >= 4:                                         # the first images share nothing, the later ones half of their keys
            sel = rng.permutation(nk)[: nk // 2]
            k[sel] = np.clip(base[sel].astype(np.int32) + rng.integers(-6, 7, (len(sel), 128)), 0, 255).astype(np.uint8)
        keys.append(k)
    arr = (U * images)(*[k.ctypes.data_as(U) for k in keys])
    nks = np.full(images, nk, np.int32)
    ms = B.lib.bsfm_match_set_create(images, nks.ctypes.data_as(C.POINTER(C.c_int)), arr)
    assert ms
    texts, mix = {}, {}
    try:
        for mode, name in ((1, "top2"), (2, "rescan"), (0, "auto")):
            old = B.lib.bsfm_match_kernel(mode)
            try:
                path = str(tmp_path / f"m_{name}.txt").encode()
                assert B.lib.bsfm_match_set_run(ms, 0.6, -1, path, 0, 1) >= 0
            finally:
                B.lib.bsfm_match_kernel(old)
            texts[name] = open(path, "rb").read()
            kms, dist, npairs, nl = C.c_double(), C.c_double(), C.c_longlong(), C.c_int()
            B.lib.bsfm_match_set_stats(ms, C.byref(kms), C.byref(dist), C.byref(npairs), C.byref(nl))
            mix[name] = (B.lib.bsfm_match_set_rescan_launches(ms), nl.value)
    finally:
        B.lib.bsfm_match_set_destroy(ms)
    assert texts["top2"] == texts["rescan"] == texts["auto"] and len(texts["auto"]) > 1000
    assert mix["top2"][0] == 0 and mix["rescan"][0] == mix["rescan"][1] == images - 1
    assert 0 < mix["auto"][0] < mix["auto"][1], mix       # it began with rescan and switched


@pytest.mark.gpu
def test_match_table_in_memory_equals_the_text_file_and_feeds_the_track_builder(gpu_bsfm, tmp_path):
    """bsfm_match_set_run_table hands over what matches.init.txt would hold (BaseApp::LoadMatchTable's layout,
    src/BundleIO.cpp:112-166) without the text round trip, and bsfm_compute_tracks takes it as is."""
    import oracle_util as O2
    B = gpu_bsfm
    sizes = [300, 0, 257, 400, 64, 350, 290]
    keys, prev = [], None
    for i, nk in enumerate(sizes):
        k = synth_keys(B, nk, 2000 + i, dup=prev) if nk else np.zeros((0, 128), np.uint8)
        keys.append(k)
        if nk:
            prev = k
    arr = (U * len(sizes))(*[k.ctypes.data_as(U) for k in keys])
    nks = np.array(sizes, np.int32)
    out = tmp_path / "matches.init.txt"
    rc = B.lib.bsfm_key_match_full(len(sizes), nks.ctypes.data_as(C.POINTER(C.c_int)), arr, 0.6, -1, str(out).encode())
    pi_t, pj_t, ptr_t, mt_t = O2.read_match_table(str(out))
    pi, pj, ptr, mt = B.match_table(keys, 0.6)
    assert rc == len(pi) > 0
    assert np.array_equal(pi, pi_t) and np.array_equal(pj, pj_t) and np.array_equal(ptr, ptr_t) and np.array_equal(mt, mt_t)
    assert (pi < pj).all()
    # straight into the track builder (after the reference's PruneDoubleMatches, src/MatchTracks.cpp:394-440)
    ptr_p, mt_p = O2.prune_double_matches(ptr, mt)
    tp, vw = B.compute_tracks(nks, pi, pj, ptr_p, mt_p)
    assert len(tp) - 1 > 0 and tp[-1] == len(vw)
    if O2.have_tracksref():
        rtp, rvw = O2.ref_compute_tracks(nks, pi, pj, ptr_p, mt_p)
        assert np.array_equal(tp, rtp) and np.array_equal(vw, rvw)
