"""Timings of the SURVEY 8(f) rows at the benchmark size (1 000 cameras / 500 000 points / 5 000 000 observations), next
to the reference on this host where the reference has the function: post-solve outlier statistics, ray-angle pruning,
camera-only refinement (fix_points), batched triangulation.  Usage: python scripts/gpu_widen_bench.py [cams points deg]"""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import bundler_sfm_amd as B
import oracle_util as O

m, n, deg = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (1000, 500000, 10)
s = B.synth_ba(m, n, deg)
nvis = int(s["rowptr"][-1])
sync = B.lib.bsfm_device_synchronize


def timed(f, reps=5):
    f(); sync()
    t = time.perf_counter()
    for _ in range(reps):
        f()
    sync()
    return (time.perf_counter() - t) / reps


opt = B.default_options(verbose=0, itmax=3)
pb = B.Problem(n, m, s["rowptr"], s["colidx"], s["proj"], s["cams"], s["pts"], options=opt)
pb.solve()
t_out = timed(lambda: pb.outlier_stats(8.0, 16.0))
t_ray = timed(lambda: pb.ray_angles(2.0))
print(f"outlier statistics (RunSFM_SBA, Bundle.cpp:659-913): {1e3 * t_out:.2f} ms for {nvis} observations incl. download of "
      f"{m} camera rows + {n} point flags = {nvis / t_out / 1e9:.2f} G obs/s")
print(f"ray-angle pruning (RemoveBadPointsAndCameras, Bundle.cpp:4190-4261): {1e3 * t_ray:.2f} ms for {n} points / "
      f"{n * deg * (deg - 1) // 2} ray pairs")
# one pass of the outlier loop on the resident problem: drop 2 000 flagged points (and their observations) on the device
flags = np.zeros(n, np.uint8); flags[np.random.default_rng(0).choice(n, 2000, replace=False)] = 1
sync(); t = time.perf_counter(); got, _ = pb.remove_points(flags); sync(); t_rm = time.perf_counter() - t
print(f"bsfm_problem_remove_points: {got} of {n} points (+ {nvis - pb.nvis} observations) dropped, index rebuilt on the device: "
      f"{1e3 * t_rm:.1f} ms (the reference's caller rebuilds vmask / projections on the host and run_sfm re-derives every index: "
      f"see problem_create in the bench line)")
pb.close()

# camera-only refinement (sba_mot_levmar): per-iteration time through the resident API
opt = B.default_options(verbose=0, itmax=1000, opts=[1e-3, 0.0, 0.0, 0.0, 0.0, -1.0])
pb = B.Problem(n, m, s["rowptr"], s["colidx"], s["proj"], s["cams"], s["pts"], options=opt, fix_points=1)
pb.lm_begin(); pb.lm_iterate(2); sync()
t = time.perf_counter(); pb.lm_iterate(10); sync(); t_mot = (time.perf_counter() - t) / 10
print(f"camera-only LM iteration (fix_points, sba_mot_levmar): {1e3 * t_mot:.3f} ms")
pb.close()

# triangulation: every point of the scene from its observations (normalised by the true intrinsics, noise kept)
ca = O.cams_to_arrays(s["cams"])
Rc = ca["R"].reshape(-1, 3, 3); tc = np.einsum("mij,mj->mi", Rc, -ca["t"])
cam = s["colidx"]
pts = s["pts"].reshape(-1, 3)
P = np.einsum("vij,vj->vi", Rc[cam], np.repeat(pts, np.diff(s["rowptr"]), axis=0)) + tc[cam]
p = P[:, :2] / P[:, 2:3] + np.random.default_rng(1).normal(0, 5e-4, (nvis, 2))
args = (B.TRI_N, s["rowptr"], p.ravel(), ca["R"].ravel(), tc.ravel())
B.triangulate_batch(*args, view_cam=cam)
t = time.perf_counter(); X, err, info = B.triangulate_batch(*args, view_cam=cam); t_tri = time.perf_counter() - t
print(f"triangulate_n batch: {n} points x {deg} views in {1e3 * t_tri:.1f} ms incl. upload/download = {n / t_tri / 1e6:.2f} M points/s; "
      f"median rms error {np.median(err):.2e}; lmdif codes {np.bincount(info)}")
if O.have_ref():
    k = 2000
    t = time.perf_counter()
    for i in range(k):
        v = slice(s["rowptr"][i], s["rowptr"][i + 1])
        O.ref_triangulate(0, p[v], ca["R"][cam[v]], tc[cam[v]])
    t_ref = (time.perf_counter() - t) / k
    print(f"reference triangulate_n on this host (through ctypes): {1e6 * t_ref:.1f} us per point = {1 / t_ref / 1e6:.4f} M points/s")
    small = B.synth_ba(200, 50000, 10)
    vm = B.dense_vmask(50000, 200, small["rowptr"], small["colidx"])
    r = O.ref_sba_mot(50000, 200, vm, small["proj"], small["cams"], small["pts"], itmax=3, jac_mode=0)
    print(f"reference sba_mot_levmar, 200 cams / 50 000 pts / 500 000 obs: {1e3 * r['secs'] / max(r['info'][5], 1):.0f} ms per iteration")

# epipolar geometry: EstimateFMatrix over a batch of image pairs (2048 trials, threshold 9 as Bundler runs it)
rng = np.random.default_rng(5)
def _pair(nm, out_frac):
    X = rng.uniform(-1, 1, (nm, 3)) + [0, 0, 5]
    th = rng.uniform(0.1, 0.4)
    Rm = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    P2 = (Rm @ X.T).T + [-1.0, 0.1, 0.1]
    a = 800 * X[:, :2] / X[:, 2:3] + rng.normal(0, 0.7, (nm, 2)); b = 800 * P2[:, :2] / P2[:, 2:3] + rng.normal(0, 0.7, (nm, 2))
    k = int(out_frac * nm); b[:k] = rng.uniform(-400, 400, (k, 2))
    return a, b
npairs = 512
prs = [_pair(int(rng.integers(50, 400)), rng.uniform(0.1, 0.5)) for _ in range(npairs)]
ptr = np.concatenate([[0], np.cumsum([len(a) for a, _ in prs])]).astype(np.int32)
K1 = np.concatenate([a for a, _ in prs]).ravel(); K2 = np.concatenate([b for _, b in prs]).ravel()
B.estimate_fmatrix_batch(ptr[:3], K1[:2 * ptr[2]], K2[:2 * ptr[2]], 2048, 9.0, B.Rand(1))
t = time.perf_counter(); F, cnt, inl, info = B.estimate_fmatrix_batch(ptr, K1, K2, 2048, 9.0, B.Rand(1)); t_fm = time.perf_counter() - t
print(f"EstimateFMatrix batch: {npairs} pairs ({ptr[-1]} matches, 2048 trials each) in {1e3 * t_fm:.1f} ms = {1e3 * t_fm / npairs:.3f} ms/pair; "
      f"mean inlier share {cnt.sum() / ptr[-1]:.2f}")
if O.have_fmref():
    k = 24
    t = time.perf_counter()
    for q in range(k):
        O.ref_fm_estimate(1, prs[q][0], prs[q][1], 2048, 9.0)
    t_ref = (time.perf_counter() - t) / k
    print(f"reference EstimateFMatrix sequence on this host: {1e3 * t_ref:.2f} ms/pair")
