#!/usr/bin/env python3
"""The config-5 match file (500 images x 5 000 keys, all 124 750 pairs) against the reference on a 200-pair random sample (missing #6):
the key set and the file of bench.py's matcher leg, the reference's exact MatchKeys(.., 0) of oracle/_ref on every host core.
usage: python scripts/r6/match_file_sample_check.py [pairs=200] [workers=all cores]"""
import ctypes as C, json, os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
import bundler_sfm_amd as B
npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
workers = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
images, nkeys = 500, 5000
U = C.POINTER(C.c_ubyte)
keys = bench.synth_key_set(B, images, nkeys)
arr = (U * images)(*[k.ctypes.data_as(U) for k in keys])
nks = np.full(images, nkeys, np.int32)
ms = B.lib.bsfm_match_set_create(images, nks.ctypes.data_as(C.POINTER(C.c_int)), arr)
assert ms
out = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir(), "bsfm_match_check_%d.txt" % os.getpid())
t0 = time.time()
blocks = B.lib.bsfm_match_set_run(ms, 0.6, -1, out.encode(), 0, 1)
B.lib.bsfm_device_synchronize()
t_gpu = time.time() - t0
B.lib.bsfm_match_set_destroy(ms)
r = bench.check_match_file_sample(out, keys, npairs, seed=20260930, workers=workers)
r.update({"images": images, "keys_per_image": nkeys, "pair_blocks_written": int(blocks), "file_bytes": os.path.getsize(out), "gpu_pass_s": round(t_gpu, 2), "host_cpus": os.cpu_count()})
os.unlink(out)
print(json.dumps(r))
sys.exit(0 if r.get("identical") else 1)
