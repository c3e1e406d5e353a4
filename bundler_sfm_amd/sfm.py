"""Host-side mirror of the reference's sfm-driver interface on top of the C-ABI.

`run_sfm` has the argument meaning of lib/sfm-driver/sfm.h:68-86 (numpy arrays instead of raw pointers);
`Problem` wraps the resident-problem API (include/bsfm.h section 3) used by bench.py and the multi-GPU path.
All compute happens in libbsfm_hip.so on the GPU.
"""
import ctypes as C
import numpy as np

from . import _lib
from ._lib import CameraParams, Options, ProblemDesc, lib

SYNTH_SEED = 88172645463325252


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int))


def default_options(**kw):
    o = Options()
    lib.bsfm_default_options(C.byref(o))
    for k, v in kw.items():
        if k == "opts":
            for i, x in enumerate(v):
                o.opts[i] = x
        else:
            setattr(o, k, v)
    return o


def make_cameras(m):
    return (CameraParams * m)()


def copy_cameras(cams):
    out = (CameraParams * len(cams))()
    C.memmove(out, cams, C.sizeof(out))
    return out


def synth_ba(m, n, deg=10, seed=SYNTH_SEED, banded=False):
    """Deterministic synthetic scene of SURVEY.md 8(d). Returns dict(rowptr, colidx, proj, cams, pts)."""
    rowptr = np.zeros(n + 1, np.int32)
    colidx = np.zeros(n * deg, np.int32)
    proj = np.zeros(2 * n * deg, np.float64)
    pts = np.zeros(3 * n, np.float64)
    cams = make_cameras(m)
    rc = lib.bsfm_synth_ba(m, n, deg, seed, int(banded), _ip(rowptr), _ip(colidx), _dp(proj), cams, _dp(pts))
    if rc != 0:
        raise ValueError("bsfm_synth_ba rejected the configuration")
    return dict(m=m, n=n, rowptr=rowptr, colidx=colidx, proj=proj, cams=cams, pts=pts)


def dense_vmask(n, m, rowptr, colidx):
    vm = np.zeros((n, m), np.uint8)
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    vm[rows, colidx] = 1
    return vm


def run_sfm(num_pts, num_cameras, ncons, vmask, projections, est_focal_length, const_focal_length, undistort,
            explicit_camera_centers, cameras, pts, use_constraints=0, use_point_constraints=0,
            points_constraints=None, point_constraint_weight=0.0, fix_points=0, optimize_for_fisheye=0,
            eps2=1e-12, Vout=None, Sout=None, Uout=None, Wout=None, options=None, want_info=True):
    """Reference signature (sfm.h:68-86). cameras (ctypes array) and pts (float64 3n) are updated in place.
    Returns (rc, info[10])."""
    vm = np.ascontiguousarray(vmask, dtype=np.uint8)
    info = np.zeros(_lib.INFOSZ)
    rc = lib.bsfm_run_sfm_ex(num_pts, num_cameras, ncons, vm.ctypes.data_as(C.c_char_p), _dp(projections),
                             est_focal_length, const_focal_length, undistort, explicit_camera_centers, cameras,
                             _dp(pts), use_constraints, use_point_constraints, _dp(points_constraints),
                             point_constraint_weight, fix_points, optimize_for_fisheye, eps2,
                             _dp(Vout), _dp(Sout), _dp(Uout), _dp(Wout),
                             C.byref(options) if options is not None else None, _dp(info))
    return rc, info


TRI_N, TRI_N_REFINE, TRI_PAIR = 0, 1, 2


def triangulate_batch(mode, view_ptr, p, R, t, X=None, view_cam=None):
    """Batched triangulate_n / triangulate_n_refine / triangulate (lib/imagelib/triangulate.c) on the GPU.
    view_ptr (npoints+1), p (2 per view), R / t per view (view_cam None) or per camera.  Returns (X, error, info)."""
    view_ptr = np.ascontiguousarray(view_ptr, np.int32)
    npts = len(view_ptr) - 1
    p = np.ascontiguousarray(p, np.float64); R = np.ascontiguousarray(R, np.float64); t = np.ascontiguousarray(t, np.float64)
    X = np.zeros(3 * npts) if X is None else np.array(X, np.float64, copy=True).ravel()
    err = np.zeros(npts); info = np.zeros(npts, np.int32)
    cam = None if view_cam is None else np.ascontiguousarray(view_cam, np.int32)
    rc = lib.bsfm_triangulate_batch(mode, npts, _ip(view_ptr), _dp(p), _ip(cam), 0 if cam is None else R.size // 9,
                                    _dp(R), _dp(t), _dp(X), _dp(err), _ip(info))
    if rc != 0:
        raise RuntimeError("bsfm_triangulate_batch failed")
    return X, err, info


class Rand:
    """glibc rand() restated (bsfm_rand_t): Rand(seed).next() == srand(seed); rand()."""

    def __init__(self, seed=1):
        self.st = _lib.RandState()
        lib.bsfm_rand_seed(C.byref(self.st), seed)

    def next(self):
        return lib.bsfm_rand_next(C.byref(self.st))


def fmatrix_ransac_batch(match_ptr, a_xy, b_xy, num_trials, threshold, success_ratio, rng):
    """estimate_fmatrix_ransac_matches (lib/imagelib/fmatrix.c:293-475) over a batch of pairs. Returns (F (npairs, 9), inliers)."""
    match_ptr = np.ascontiguousarray(match_ptr, np.int32)
    npairs = len(match_ptr) - 1
    a = np.ascontiguousarray(a_xy, np.float64); b = np.ascontiguousarray(b_xy, np.float64)
    F = np.zeros((npairs, 9)); cnt = np.zeros(npairs, np.int32)
    rc = lib.bsfm_fmatrix_ransac_batch(npairs, _ip(match_ptr), _dp(a), _dp(b), num_trials, threshold, success_ratio,
                                       C.byref(rng.st), _dp(F), _ip(cnt))
    if rc != 0:
        raise RuntimeError("bsfm_fmatrix_ransac_batch failed")
    return F, cnt


def estimate_fmatrix_batch(match_ptr, k1_xy, k2_xy, num_trials, threshold, rng):
    """EstimateFMatrix (src/Epipolar.cpp:118-237) over a batch of pairs. Returns (F (npairs, 9), num_inliers, inlier flags, lm info)."""
    match_ptr = np.ascontiguousarray(match_ptr, np.int32)
    npairs = len(match_ptr) - 1
    k1 = np.ascontiguousarray(k1_xy, np.float64); k2 = np.ascontiguousarray(k2_xy, np.float64)
    F = np.zeros((npairs, 9)); cnt = np.zeros(npairs, np.int32); inl = np.zeros(max(int(match_ptr[-1]), 1), np.uint8)
    info = np.zeros(npairs, np.int32)
    rc = lib.bsfm_estimate_fmatrix_batch(npairs, _ip(match_ptr), _dp(k1), _dp(k2), num_trials, threshold, C.byref(rng.st), _dp(F),
                                         _ip(cnt), inl.ctypes.data_as(C.POINTER(C.c_ubyte)), _ip(info))
    if rc != 0:
        raise RuntimeError("bsfm_estimate_fmatrix_batch failed")
    return F, cnt, inl[:int(match_ptr[-1])], info


def match_table(keys, ratio=0.6, window_radius=-1):
    """bsfm_match_set_create + bsfm_match_set_run_table: the KeyMatchFull search with the match table in memory.
    keys: list of (n_i, 128) uint8 arrays.  Returns (pair_i, pair_j, match_ptr, matches[n, 2])."""
    U = C.POINTER(C.c_ubyte)
    keys = [np.ascontiguousarray(k, np.uint8) for k in keys]
    arr = (U * len(keys))(*[k.ctypes.data_as(U) for k in keys])
    nks = np.array([len(k) for k in keys], np.int32)
    ms = lib.bsfm_match_set_create(len(keys), _ip(nks), arr)
    if not ms:
        raise RuntimeError("bsfm_match_set_create failed")
    IP = C.POINTER(C.c_int)
    pi, pj, ptr, mt = IP(), IP(), IP(), IP()
    rc = lib.bsfm_match_set_run_table(ms, float(ratio), int(window_radius), 0, 1, C.byref(pi), C.byref(pj), C.byref(ptr), C.byref(mt))
    lib.bsfm_match_set_destroy(ms)
    if rc < 0:
        raise RuntimeError("bsfm_match_set_run_table failed")
    out_ptr = np.ctypeslib.as_array(ptr, (rc + 1,)).copy()
    nm = int(out_ptr[-1])
    res = (np.ctypeslib.as_array(pi, (max(rc, 1),))[:rc].copy(), np.ctypeslib.as_array(pj, (max(rc, 1),))[:rc].copy(), out_ptr,
           np.ctypeslib.as_array(mt, (max(2 * nm, 1),))[:2 * nm].copy().reshape(nm, 2))
    for q in (pi, pj, ptr, mt):
        lib.bsfm_free(q)
    return res


def compute_tracks(num_keys, pair_i, pair_j, match_ptr, matches, new_image_start=0):
    """bsfm_compute_tracks (BundlerApp::ComputeTracks).  Returns (track_ptr, views[nviews, 2] = (image, key))."""
    nk = np.ascontiguousarray(num_keys, np.int32); pi = np.ascontiguousarray(pair_i, np.int32); pj = np.ascontiguousarray(pair_j, np.int32)
    mp = np.ascontiguousarray(match_ptr, np.int32); mt = np.ascontiguousarray(matches, np.int32)
    nv = C.c_int()
    nt = lib.bsfm_compute_tracks(len(nk), _ip(nk), len(pi), _ip(pi), _ip(pj), _ip(mp), _ip(mt), new_image_start, None, None, 0, 0, C.byref(nv))
    if nt < 0:
        raise RuntimeError("bsfm_compute_tracks failed")
    tp = np.zeros(nt + 1, np.int32); vw = np.zeros((max(nv.value, 1), 2), np.int32)
    got = lib.bsfm_compute_tracks(len(nk), _ip(nk), len(pi), _ip(pi), _ip(pj), _ip(mp), _ip(mt), new_image_start, _ip(tp), _ip(vw), nt, nv.value,
                                  C.byref(nv))
    if got != nt:
        raise RuntimeError("bsfm_compute_tracks failed")
    return tp, vw[:nv.value]


class Problem:
    """Device-resident BA problem (sparse CRS boundary)."""

    def __init__(self, n, m, rowptr, colidx, proj, cams, pts, mcon=0, est_focal_length=1, undistort=1,
                 explicit_camera_centers=1, use_constraints=0, point_constraints=None, point_constraint_weight=0.0,
                 options=None, world_size=1, rank=0, nvis_global=0, nvars_global=0, fix_points=0,
                 optimize_for_fisheye=0):
        self._keep = (np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(colidx, np.int32),
                      np.ascontiguousarray(proj, np.float64), cams, np.ascontiguousarray(pts, np.float64),
                      None if point_constraints is None else np.ascontiguousarray(point_constraints, np.float64))
        d = ProblemDesc()
        d.n, d.m, d.mcon = n, m, mcon
        d.rowptr, d.colidx, d.projections = _ip(self._keep[0]), _ip(self._keep[1]), _dp(self._keep[2])
        d.est_focal_length, d.undistort, d.explicit_camera_centers = est_focal_length, undistort, explicit_camera_centers
        d.fix_points = fix_points
        d.optimize_for_fisheye = optimize_for_fisheye
        d.cameras = C.cast(cams, C.POINTER(CameraParams))
        d.points = _dp(self._keep[4])
        d.use_constraints = use_constraints
        d.use_point_constraints = 0 if point_constraints is None else 1
        d.point_constraints = _dp(self._keep[5])
        d.point_constraint_weight = point_constraint_weight
        d.world_size, d.rank, d.nvis_global, d.nvars_global = world_size, rank, nvis_global, nvars_global
        self.options = options if options is not None else default_options()
        self.n, self.m, self.mcon = n, m, mcon
        self.h = lib.bsfm_problem_create(C.byref(d), C.byref(self.options))
        if not self.h:
            raise RuntimeError("bsfm_problem_create failed (no HIP device or invalid input); there is no CPU fallback")
        self.cnp = lib.bsfm_problem_cnp(self.h)
        self.nvis = int(lib.bsfm_problem_nvis(self.h))
        self._hook = None

    def close(self):
        if self.h:
            lib.bsfm_problem_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_allreduce(self, fn):
        self._hook = _lib.ALLREDUCE_FN(fn)
        lib.bsfm_problem_set_allreduce(self.h, self._hook, None)

    def set_stream(self, stream_ptr):
        lib.bsfm_problem_set_stream(self.h, C.c_void_p(stream_ptr))

    def reset_params(self, cams, pts):
        pts = np.ascontiguousarray(pts, np.float64)
        return lib.bsfm_problem_reset_params(self.h, cams, _dp(pts))

    def append(self, new_cams, new_pts, add_pt, add_cam, add_xy):
        """bsfm_problem_append: grow the resident problem (new cameras / points / observations); parameters stay in HBM."""
        new_pts = np.ascontiguousarray(new_pts, np.float64).ravel()
        add_pt = np.ascontiguousarray(add_pt, np.int32); add_cam = np.ascontiguousarray(add_cam, np.int32)
        add_xy = np.ascontiguousarray(add_xy, np.float64).ravel()
        ncam = 0 if new_cams is None else len(new_cams)
        rc = lib.bsfm_problem_append(self.h, ncam, new_cams, len(new_pts) // 3, _dp(new_pts) if len(new_pts) else None, len(add_pt),
                                     _ip(add_pt) if len(add_pt) else None, _ip(add_cam) if len(add_cam) else None,
                                     _dp(add_xy) if len(add_xy) else None)
        if rc == 0:
            self.m += ncam; self.n += len(new_pts) // 3
            self.nvis = int(lib.bsfm_problem_nvis(self.h))
            if ncam:        # download() copies the caller's camera array as the template of the non-parameter fields
                merged = make_cameras(self.m)
                C.memmove(merged, self._keep[3], C.sizeof(self._keep[3]))
                C.memmove(C.byref(merged, C.sizeof(self._keep[3])), new_cams, C.sizeof(new_cams))
                self._keep = self._keep[:3] + (merged,) + self._keep[4:]
        return rc

    def remove_points(self, remove):
        """bsfm_problem_remove_points: drop the flagged points and all their observations on the device (the outlier loop of
        RunSFM_SBA); returns (number removed, remap: new index of every old point or -1)."""
        remove = np.ascontiguousarray(remove, np.uint8)
        assert len(remove) == self.n
        remap = np.zeros(self.n, np.int32)
        rc = lib.bsfm_problem_remove_points(self.h, remove.ctypes.data_as(C.POINTER(C.c_ubyte)), _ip(remap))
        if rc > 0:
            self.n = int(lib.bsfm_problem_num_points(self.h))
            self.nvis = int(lib.bsfm_problem_nvis(self.h))
        return rc, remap

    def lm_begin(self):
        return lib.bsfm_lm_begin(self.h)

    def lm_iterate(self, iters):
        return lib.bsfm_lm_iterate(self.h, iters)

    def lm_finish(self):
        info = np.zeros(_lib.INFOSZ)
        rc = lib.bsfm_lm_finish(self.h, _dp(info))
        return rc, info

    def solve(self, itmax=None):
        if itmax is not None:
            self.options.itmax = itmax
        rc = self.lm_begin()
        if rc == 0:
            self.lm_iterate(self.options.itmax)
        return self.lm_finish()

    def attempts(self):
        return lib.bsfm_lm_solve_attempts(self.h)

    def phase_ms(self, name):
        return lib.bsfm_lm_last_kernel_ms(self.h, name.encode())

    def download(self, want_cams=True):
        p = np.zeros(self.m * self.cnp + 3 * self.n)
        cams = make_cameras(self.m) if want_cams else None
        if want_cams:
            C.memmove(cams, self._keep[3], C.sizeof(cams))
        pts = np.zeros(3 * self.n)
        rc = lib.bsfm_problem_download(self.h, _dp(p), cams, _dp(pts))
        if rc != 0:
            raise RuntimeError("download failed")
        return p, cams, pts

    def outlier_stats(self, min_thr=8.0, max_thr=16.0):
        """RunSFM_SBA's post-solve statistics (src/Bundle.cpp:659-913) at the current parameters."""
        m, n = self.m, self.n
        out = dict(nobs=np.zeros(m, np.int32), mean=np.zeros(m), kth80=np.zeros(m), kth50=np.zeros(m), thresh=np.zeros(m),
                   outlier=np.zeros(n, np.uint8), err=np.zeros(n))
        gm = C.c_double()
        rc = lib.bsfm_problem_outlier_stats(self.h, float(min_thr), float(max_thr), _ip(out["nobs"]), _dp(out["mean"]),
                                            _dp(out["kth80"]), _dp(out["kth50"]), _dp(out["thresh"]),
                                            out["outlier"].ctypes.data_as(C.POINTER(C.c_ubyte)), _dp(out["err"]), C.byref(gm))
        if rc != 0:
            raise RuntimeError("bsfm_problem_outlier_stats failed")
        out["global_mean"] = gm.value
        return out

    def ray_angles(self, ray_angle_threshold=2.0):
        """RemoveBadPointsAndCameras' ray-angle test (src/Bundle.cpp:4190-4261) at the current parameters.
        Returns dict(angle_deg (n), prune (n, uint8), num_pruned)."""
        out = dict(angle_deg=np.zeros(self.n), prune=np.zeros(self.n, np.uint8))
        cnt = C.c_int()
        rc = lib.bsfm_problem_ray_angles(self.h, float(ray_angle_threshold), _dp(out["angle_deg"]),
                                         out["prune"].ctypes.data_as(C.POINTER(C.c_ubyte)), C.byref(cnt))
        if rc != 0:
            raise RuntimeError("bsfm_problem_ray_angles failed")
        out["num_pruned"] = cnt.value
        return out

    def residuals(self):
        e = np.zeros(2 * self.nvis)
        cost = C.c_double()
        if lib.bsfm_eval_residuals(self.h, _dp(e), C.byref(cost)) != 0:
            raise RuntimeError("bsfm_eval_residuals failed")
        return e, cost.value

    def normal_equations(self, mu=0.0, want_J=False):
        cnp, m, n = self.cnp, self.m, self.n
        U = np.zeros((m, cnp, cnp)); ea = np.zeros((m, cnp)); V = np.zeros((n, 3, 3)); eb = np.zeros((n, 3))
        J = np.zeros((self.nvis, 2 * cnp + 6)) if want_J else None
        sd = (m - self.mcon) * cnp          # the library writes S / E for the free cameras only
        S = np.zeros((sd, sd)); E = np.zeros(sd)
        rc = lib.bsfm_eval_normal_equations(self.h, mu, _dp(U), _dp(ea), _dp(V), _dp(eb), _dp(J), _dp(S), _dp(E))
        if rc != 0:
            raise RuntimeError("bsfm_eval_normal_equations failed")
        return dict(U=U, ea=ea, V=V, eb=eb, J=J, S=S, E=E)

    def export_index(self):
        """The index arrays the kernels use, downloaded from HBM."""
        nv = self.nvis
        out = dict(rowptr=np.zeros(self.n + 1, np.int32), colidx=np.zeros(nv, np.int32), obs_pt=np.zeros(nv, np.int32),
                   camptr=np.zeros(self.m + 1, np.int32), camobs=np.zeros(nv, np.int32), campos=np.zeros(nv, np.int32),
                   cam_pt=np.zeros(nv, np.int32), cam_cam=np.zeros(nv, np.int32))
        rc = lib.bsfm_problem_export_index(self.h, *[_ip(out[k]) for k in ("rowptr", "colidx", "obs_pt", "camptr", "camobs", "campos",
                                                                       "cam_pt", "cam_cam")])
        if rc != 0:
            raise RuntimeError("bsfm_problem_export_index failed")
        return out

    def export_schur(self):
        """Co-visibility triples / blocks / tasks of the Schur complement as resident on the device."""
        sz = [C.c_int() for _ in range(4)]
        lib.bsfm_problem_schur_sizes(self.h, *[C.byref(v) for v in sz])
        nt, nb, ntask, nslots = [v.value for v in sz]
        out = dict(triples=np.zeros((nt, 2), np.int32), tri_pt=np.zeros(nt, np.int32), blk_j=np.zeros(nb, np.int32),
                   blk_k=np.zeros(nb, np.int32), blk_task0=np.zeros(nb + 1, np.int32), tasks=np.zeros((nslots, 4), np.int32))
        rc = lib.bsfm_problem_export_schur(self.h, *[_ip(out[k]) for k in ("triples", "tri_pt", "blk_j", "blk_k", "blk_task0", "tasks")])
        if rc != 0:
            raise RuntimeError("bsfm_problem_export_schur failed")
        out["ntasks"] = ntask
        return out



FLOW_TASK_DTYPE = np.dtype([("type", "u1"), ("np", "u1"), ("part", "u1"), ("nwait", "u1"), ("i", "<u2"), ("j", "<u2"), ("p0", "<u2"),
                            ("pad", "<u2"), ("sig", "<u4"), ("w", "<u4", (3, 2))])


def chol_flow_schedule(nblk, last=None, np_max=0, slots=0):
    """Static task order of the tile-dataflow Cholesky (csrc/chol_flow_sched.h); host only.  Returns (tasks, simulated microseconds)."""
    lastp = None
    if last is not None:
        last = np.ascontiguousarray(last, np.int32)
        lastp = last.ctypes.data_as(C.POINTER(C.c_int))
    sim = C.c_double(0.0)
    n = lib.bsfm_chol_flow_schedule(nblk, lastp, np_max, slots, None, 0, C.byref(sim))
    if n < 0:
        raise RuntimeError("bsfm_chol_flow_schedule failed its dependency check")
    tasks = np.zeros(n, FLOW_TASK_DTYPE)
    assert tasks.itemsize == 40
    n2 = lib.bsfm_chol_flow_schedule(nblk, lastp, np_max, slots, tasks.ctypes.data_as(C.c_void_p), n, C.byref(sim))
    assert n2 == n
    return tasks, sim.value


def chol_dyn_plan(nblk, last=None):
    """Host-side plan of the dynamic tile-dataflow Cholesky (csrc/chol_dyn_plan.h); host only.
    Returns dict(chain, potrf, init, ofs_c32, ofs_c10, ofs_wd, ofs_rd, ofs_tw)."""
    lastp = None
    if last is not None:
        last = np.ascontiguousarray(last, np.int32)
        lastp = last.ctypes.data_as(C.POINTER(C.c_int))
    meta = np.zeros(8, np.int32)
    mp = meta.ctypes.data_as(C.POINTER(C.c_int))
    if lib.bsfm_chol_dyn_plan(nblk, lastp, None, 0, None, 0, None, 0, mp) != 0:
        raise RuntimeError("bsfm_chol_dyn_plan failed")
    chain = np.zeros(int(meta[0]), FLOW_TASK_DTYPE); potrf = np.zeros(int(meta[1]), FLOW_TASK_DTYPE); init = np.zeros(int(meta[2]), np.uint32)
    rc = lib.bsfm_chol_dyn_plan(nblk, lastp, chain.ctypes.data_as(C.c_void_p), len(chain), potrf.ctypes.data_as(C.c_void_p), len(potrf),
                                init.ctypes.data_as(C.POINTER(C.c_uint)), len(init), mp)
    assert rc == 0
    return dict(chain=chain, potrf=potrf, init=init, ofs_c32=int(meta[3]), ofs_c10=int(meta[4]), ofs_wd=int(meta[5]), ofs_rd=int(meta[6]),
                ofs_tw=int(meta[7]))


def dense_chol_solve_timed(A, b, reps=3, backend=0):
    """dense_chol_solve `reps` times; returns (rc, x, ms per repetition, mean k_chol_flow launch ms, Gflop scheduled per launch)."""
    A = np.ascontiguousarray(A, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    x = np.zeros_like(b)
    ms = np.zeros(max(1, reps)); fm = C.c_double(-1.0); gf = C.c_double(-1.0)
    rc = lib.bsfm_dense_chol_solve_timed(A.shape[0], _dp(A), _dp(b), _dp(x), backend, reps, _dp(ms), C.byref(fm), C.byref(gf))
    return rc, x, ms, fm.value, gf.value


def dense_chol_solve_dist(comm, A, b, backend=0):
    """bsfm_dense_chol_solve_dist: the ranks of `comm` (a bsfm_comm_t* as c_void_p) factor A together; collective."""
    A = np.ascontiguousarray(A, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    x = np.zeros_like(b)
    rc = lib.bsfm_dense_chol_solve_dist(comm, A.shape[0], _dp(A), _dp(b), _dp(x), backend)
    return rc, x


def dense_chol_solve(A, b, backend=0):
    A = np.ascontiguousarray(A, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    x = np.zeros_like(b)
    rc = lib.bsfm_dense_chol_solve(A.shape[0], _dp(A), _dp(b), _dp(x), backend)
    return rc, x
