import sys, time, ctypes as C, numpy as np, tempfile, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import bundler_sfm_amd as B
import oracle_util as O
U = C.POINTER(C.c_ubyte)
nimg, nk = int(sys.argv[1]) if len(sys.argv) > 1 else 24, 5000
keys = []; prev = None
for i in range(nimg):
    k = np.zeros((nk, 128), np.uint8)
    B.lib.bsfm_synth_keys(nk, 5000 + i, None if prev is None else prev.ctypes.data_as(U), 0 if prev is None else nk, k.ctypes.data_as(U))
    keys.append(k); prev = k
arr = (U * nimg)(*[k.ctypes.data_as(U) for k in keys]); nks = np.full(nimg, nk, np.int32)
out = os.path.join(tempfile.gettempdir(), "m.txt")
B.lib.bsfm_key_match_full(2, nks.ctypes.data_as(C.POINTER(C.c_int)), arr, 0.6, -1, out.encode())   # warm-up
t = time.time(); rc = B.lib.bsfm_key_match_full(nimg, nks.ctypes.data_as(C.POINTER(C.c_int)), arr, 0.6, -1, out.encode()); t = time.time() - t
npairs = nimg * (nimg - 1) // 2
print(f"GPU KeyMatchFull: {nimg} images x {nk} keys, {npairs} pairs in {t:.3f}s = {npairs/t:.1f} pairs/s ({1e3*t/npairs:.3f} ms/pair), {rc} pairs written, file {os.path.getsize(out)} B")
print(f"   extrapolated 500 images (124750 pairs): {124750*t/npairs:.1f} s")
if O.have_ref():
    m200, s200 = O.ref_match(keys[0], keys[1], 0.6, 200)
    mex, sex = O.ref_match(keys[0], keys[1], 0.6, 0)
    cnt = B.lib.bsfm_match_keys_l2(nk, keys[0].ctypes.data_as(U), nk, keys[1].ctypes.data_as(U), 0.6, None, 0)
    print(f"reference MatchKeys one pair: ANN-200 {s200:.3f}s ({len(m200)} matches), exact {sex:.3f}s ({len(mex)} matches); GPU count {cnt}")
