"""the matcher's exact running top-2 (3 VALU instructions per distance) against "slot-local minimum only
(2 VALU) + exact rescan of the winning slot for the rows that pass the ratio test against the bound" (BSFM_MATCH_KERNEL=rescan; 256 queries per workgroup, hand-pipelined loads) and the default per-launch choice (auto).
Two key sets of IMAGES x 5000 keys: `chain` = the benchmark's generator (20 % of an image's keys are noisy copies of keys of
the previous image: few pairs have matches) and `common` = 20 % of EVERY image's keys are noisy copies of one base set (every pair
has ~1000 accepted matches: many rows to rescan), `dense` = half of every image's keys are noisy copies of the same base keys (~2 500 matches per pair: video-like).  Prints k_match_l2's HIP-event time per image pair and the SHA-1
of matches.init.txt (must be identical between the modes)."""
import ctypes as C, hashlib, os, subprocess, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
IMAGES = int(os.environ.get("AB_IMAGES", "160")); NKEYS = 5000


def worker(kind):
    import bundler_sfm_amd as B
    U = C.POINTER(C.c_ubyte)
    keys, prev = [], None
    base = np.zeros((NKEYS, 128), np.uint8)
    B.lib.bsfm_synth_keys(NKEYS, 777, None, 0, base.ctypes.data_as(U))
    rng = np.random.default_rng(5)
    for i in range(IMAGES):
        k = np.zeros((NKEYS, 128), np.uint8)
        if kind == "dense":          # half of every image's keys are noisy copies of the SAME base keys: ~2500 true matches per pair
            B.lib.bsfm_synth_keys(NKEYS, 9000 + i, None, 0, k.ctypes.data_as(U))
            sel = rng.permutation(NKEYS)[: NKEYS // 2]
            k[sel] = np.clip(base[sel].astype(np.int32) + rng.integers(-8, 9, (len(sel), 128)), 0, 255).astype(np.uint8)
            keys.append(k); continue
        src = prev if kind == "chain" else base
        B.lib.bsfm_synth_keys(NKEYS, 9000 + i, None if src is None else src.ctypes.data_as(U), 0 if src is None else len(src), k.ctypes.data_as(U))
        keys.append(k); prev = k
    arr = (U * IMAGES)(*[k.ctypes.data_as(U) for k in keys])
    nks = np.full(IMAGES, NKEYS, np.int32)
    ms = B.lib.bsfm_match_set_create(IMAGES, nks.ctypes.data_as(C.POINTER(C.c_int)), arr)
    out = f"/dev/shm/bsfm_ab_{os.getpid()}.txt".encode()
    B.lib.bsfm_match_set_run(ms, 0.6, 3, out, 0, 1)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        B.lib.bsfm_match_set_run(ms, 0.6, -1, out, 0, 1)
        B.lib.bsfm_device_synchronize()
        el = time.perf_counter() - t0
        kms, dist, npairs, nl = C.c_double(), C.c_double(), C.c_longlong(), C.c_int()
        B.lib.bsfm_match_set_stats(ms, C.byref(kms), C.byref(dist), C.byref(npairs), C.byref(nl))
        if best is None or kms.value < best[0]: best = (kms.value, el, npairs.value)
    B.lib.bsfm_match_set_destroy(ms)
    data = open(out, "rb").read(); os.unlink(out)
    nmatch = sum(1 for ln in data.split(b"\n") if ln.count(b" ") == 1) - best[2]
    print(f"{kind:7s} kernel={os.environ.get('BSFM_MATCH_KERNEL', 'auto'):6s}: kernel {best[0]:8.2f} ms = {1e3 * best[0] / best[2]:.3f} us per image pair "
          f"({best[2]} pairs, wall {1e3 * best[1]:.1f} ms), accepted matches/pair {nmatch / best[2]:.1f}, sha1 {hashlib.sha1(data).hexdigest()[:16]}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        worker(sys.argv[1])
    else:
        for kind in ("chain", "common", "dense"):
            for mode in ("top2", "rescan", "auto"):
                subprocess.run([sys.executable, os.path.abspath(__file__), kind], env=dict(os.environ, BSFM_MATCH_KERNEL=mode), check=True)
