"""k_match_l2 time per image pair as a function of the keys per image (HIP events of the launches, bsfm_match_set_stats):
   python scripts/match_scaling.py [images] [keys ...]   -- separates the per-workgroup overhead from the per-tile cost."""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bundler_sfm_amd as B  # noqa: E402

U = C.POINTER(C.c_ubyte)
nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sizes = [int(a) for a in sys.argv[2:]] or [1280, 2560, 5120, 8192, 10240]
out = os.path.join(tempfile.gettempdir(), "match_scaling.txt").encode()
for nk in sizes:
    keys = []; prev = None
    for i in range(nimg):
        k = np.zeros((nk, 128), np.uint8)
        B.lib.bsfm_synth_keys(nk, 7000 + i, None if prev is None else prev.ctypes.data_as(U), 0 if prev is None else nk, k.ctypes.data_as(U))
        keys.append(k); prev = k
    arr = (U * nimg)(*[k.ctypes.data_as(U) for k in keys]); nks = np.full(nimg, nk, np.int32)
    ms = B.lib.bsfm_match_set_create(nimg, nks.ctypes.data_as(C.POINTER(C.c_int)), arr)
    B.lib.bsfm_match_set_run(ms, 0.6, -1, out, 0, 1)
    B.lib.bsfm_match_set_run(ms, 0.6, -1, out, 0, 1)
    kms, dist, npairs, nl = C.c_double(), C.c_double(), C.c_longlong(), C.c_int()
    B.lib.bsfm_match_set_stats(ms, C.byref(kms), C.byref(dist), C.byref(npairs), C.byref(nl))
    B.lib.bsfm_match_set_destroy(ms)
    us = 1e3 * kms.value / npairs.value
    wg = (nk + 127) // 128; tiles = (nk + 63) // 64
    print(f"{nk:6d} keys: {us:8.3f} us/pair  {us / wg * 1e3:8.1f} ns per workgroup  {us / wg / tiles * 1e3:7.2f} ns per (workgroup, tile)  "
          f"{2 * 128 * dist.value / (kms.value * 1e-3) / 1e12:7.1f} Top/s")
