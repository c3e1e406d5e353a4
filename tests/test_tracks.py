"""Track building: bsfm_compute_tracks against the reference's own BundlerApp::ComputeTracks -- src/ComputeTracks.cpp
compiled verbatim against a minimal class context (oracle/ref_tracks.cpp -> oracle/_ref/libtracksref.so) -- bit for bit: the same tracks,
in the same numbering, with the same view order.  Inputs: the kermit example's own matches.init.txt (the reference's only match fixture)
after PruneDoubleMatches, and synthetic match graphs that force the order-dependent cases (two keys of one image inside one connected
component, chains that only close through a third image)."""
import os

import numpy as np
import pytest

import oracle_util as O

HERE = os.path.dirname(os.path.abspath(__file__))
KERMIT = os.path.join(HERE, "golden", "kermit_matches_init.txt")
needs_ref = pytest.mark.skipif(not O.have_tracksref(), reason="oracle/_ref/libtracksref.so not built")


def kermit_table():
    pi, pj, ptr, mt = O.read_match_table(KERMIT)
    ptr, mt = O.prune_double_matches(ptr, mt)
    nimg = int(max(pi.max(), pj.max())) + 1
    nk = np.zeros(nimg, np.int32)
    for p in range(len(pi)):
        seg = mt[ptr[p]:ptr[p + 1]]
        if len(seg):
            nk[pi[p]] = max(nk[pi[p]], seg[:, 0].max() + 1); nk[pj[p]] = max(nk[pj[p]], seg[:, 1].max() + 1)
    return nk, pi, pj, ptr, mt


def synth_table(seed, nimg=12, nkeys=60, ntracks=150, noise=40):
    """Ground-truth tracks observed in random image subsets + wrong matches that glue tracks together (so that components hold several keys
    of one image and the search order decides who gets which)."""
    rng = np.random.default_rng(seed)
    owner = {}
    pairs = {}
    for t in range(ntracks):
        imgs = np.sort(rng.choice(nimg, int(rng.integers(2, 7)), replace=False))
        keys = [int(rng.integers(0, nkeys)) for _ in imgs]
        for a in range(len(imgs)):
            for b in range(a + 1, len(imgs)):
                if rng.random() < 0.7:
                    pairs.setdefault((int(imgs[a]), int(imgs[b])), []).append((keys[a], keys[b]))
    for _ in range(noise):
        i, j = sorted(rng.choice(nimg, 2, replace=False).tolist())
        pairs.setdefault((i, j), []).append((int(rng.integers(0, nkeys)), int(rng.integers(0, nkeys))))
    pi, pj, ptr, mt = [], [], [0], []
    for (i, j) in sorted(pairs, key=lambda q: (q[1], q[0])):        # KeyMatchFull's block order: database image outer, query inner
        seen1, seen2, lst = set(), set(), []
        for a, b in pairs[(i, j)]:
            if a not in seen1 and b not in seen2:                   # unique on both sides (matcher + PruneDoubleMatches)
                seen1.add(a); seen2.add(b); lst.append((a, b))
        pi.append(i); pj.append(j); mt += lst; ptr.append(len(mt))
    return (np.full(nimg, nkeys, np.int32), np.array(pi, np.int32), np.array(pj, np.int32), np.array(ptr, np.int32),
            np.array(mt, np.int32).reshape(-1, 2))


@needs_ref
def test_reference_compute_tracks_on_the_kermit_match_table():
    """Pins the oracle wiring: every match of the table ends up inside one track or is dropped because its image is already in the
    track; tracks hold one key per image; numbering follows the first feature."""
    nk, pi, pj, ptr, mt = kermit_table()
    tp, vw = O.ref_compute_tracks(nk, pi, pj, ptr, mt)
    assert len(tp) - 1 == 737 and len(vw) == 2313                   # what the reference computes on its own example
    firsts = vw[tp[:-1]]
    order = firsts[:, 0].astype(np.int64) * 100000 + firsts[:, 1]
    assert (np.diff(order) > 0).all()
    for t in range(len(tp) - 1):
        imgs = vw[tp[t]:tp[t + 1], 0]
        assert len(imgs) >= 2 and len(set(imgs.tolist())) == len(imgs)
    assert len({tuple(v) for v in vw.tolist()}) == len(vw)          # a key belongs to at most one track


@pytest.mark.gpu
@needs_ref
def test_gpu_tracks_equal_the_reference_on_kermit(gpu_bsfm):
    nk, pi, pj, ptr, mt = kermit_table()
    tp, vw = gpu_bsfm.compute_tracks(nk, pi, pj, ptr, mt)
    rtp, rvw = O.ref_compute_tracks(nk, pi, pj, ptr, mt)
    assert np.array_equal(tp, rtp) and np.array_equal(vw, rvw)


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("seed", range(8))
def test_gpu_tracks_equal_the_reference_on_conflicting_match_graphs(gpu_bsfm, seed):
    nk, pi, pj, ptr, mt = synth_table(seed, nimg=10 + seed, nkeys=40 + 5 * seed, ntracks=120 + 30 * seed, noise=30 + 10 * seed)
    rtp, rvw = O.ref_compute_tracks(nk, pi, pj, ptr, mt)
    tp, vw = gpu_bsfm.compute_tracks(nk, pi, pj, ptr, mt)
    assert np.array_equal(tp, rtp) and np.array_equal(vw, rvw)
    # the interesting case is present: some connected component of the match graph holds two keys of one image
    sizes = np.diff(rtp)
    assert sizes.max() >= 4


@pytest.mark.gpu
def test_gpu_tracks_edge_cases(gpu_bsfm, capfd):
    B = gpu_bsfm
    nk = np.array([3, 3, 3], np.int32)
    tp, vw = B.compute_tracks(nk, np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(1, np.int32), np.zeros((0, 2), np.int32))
    assert len(tp) == 1 and len(vw) == 0
    # a chain 0:0 - 1:1 - 2:2 and a second key of image 0 matched to 2:2 -> claimed by the first search, (0,1) stays alone
    pi = np.array([0, 1, 0], np.int32); pj = np.array([1, 2, 2], np.int32)
    tp, vw = B.compute_tracks(nk, pi, pj, np.array([0, 1, 2, 3], np.int32), np.array([[0, 1], [1, 2], [1, 2]], np.int32))
    assert tp.tolist() == [0, 3] and vw.tolist() == [[0, 0], [1, 1], [2, 2]]
    with pytest.raises(RuntimeError):           # key matched twice into the same image: refused
        B.compute_tracks(nk, np.array([0], np.int32), np.array([1], np.int32), np.array([0, 2], np.int32), np.array([[0, 1], [0, 2]], np.int32))
    assert "twice" in capfd.readouterr().err
