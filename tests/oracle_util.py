"""Test-side access to the CPU oracles under oracle/ (TEST INFRASTRUCTURE, never imported by the product).

  ref  = oracle/_ref/libsfmref.so      the reference's own sources compiled by oracle/Makefile
  port = oracle/liboracle_port.so      our plain-C restatement (oracle/sba_oracle.c)
"""
import ctypes as C
import contextlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bundler_sfm_amd._lib import CameraParams  # noqa: E402  (layout mirror only)

REF_PATH = os.path.join(ROOT, "oracle", "_ref", "libsfmref.so")
REF_KM_PATH = os.path.join(ROOT, "oracle", "_ref", "libkeymatchref.so")
PORT_PATH = os.path.join(ROOT, "oracle", "liboracle_port.so")

_dp = C.POINTER(C.c_double)
_cp = C.POINTER(CameraParams)


def have_ref():
    return os.path.exists(REF_PATH)


FMREF_PATH = os.path.join(os.path.dirname(REF_PATH), "libfmref.so")
_fmref = None


def have_fmref():
    return os.path.exists(FMREF_PATH)


def fmref():
    global _fmref
    if _fmref is None:
        _fmref = C.CDLL(FMREF_PATH)
        _fmref.ref_fm_ransac.restype = C.c_int
        _fmref.ref_fm_estimate.restype = C.c_int
        _fmref.ref_fm_residual.restype = C.c_double
    return _fmref


def have_port():
    return os.path.exists(PORT_PATH)


@contextlib.contextmanager
def quiet_stdout():
    """The reference prints every LM iteration at verbosity 3 (sfm.c:815): silence fd 1 (and 2)."""
    sys.stdout.flush(); sys.stderr.flush()
    saved = os.dup(1), os.dup(2)
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 1); os.dup2(devnull, 2)
    try:
        yield
    finally:
        os.dup2(saved[0], 1); os.dup2(saved[1], 2)
        os.close(saved[0]); os.close(saved[1]); os.close(devnull)


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


_ref = None


def ref():
    global _ref
    if _ref is None:
        lib = C.CDLL(REF_PATH)
        lib.ref_sizeof_camera_params.restype = C.c_int
        run_args = [C.c_int, C.c_int, C.c_int, C.c_char_p, _dp, C.c_int, C.c_int, C.c_int, C.c_int, _cp, _dp,
                    C.c_int, C.c_int, _dp, C.c_double, C.c_int, C.c_int, C.c_double, _dp, _dp, _dp, _dp]
        lib.ref_run_sfm.argtypes = run_args
        lib.ref_run_sfm.restype = None
        lib.ref_sba_motstr.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, _dp, C.c_int, C.c_int, C.c_int, _cp, _dp,
                                       C.c_int, C.c_int, _dp, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int,
                                       _dp, _dp, _dp, _dp, _dp, _dp, _dp]
        lib.ref_sba_motstr.restype = C.c_int
        lib.ref_project_point.argtypes = [C.c_int, C.c_int, C.c_int, _cp, _dp, _dp, _dp]
        lib.ref_project_point.restype = None
        assert lib.ref_sizeof_camera_params() == C.sizeof(CameraParams)
        _ref = lib
    return _ref


def copy_cams(cams):
    out = (CameraParams * len(cams))()
    C.memmove(out, cams, C.sizeof(out))
    return out


def ref_run_sfm(n, m, vmask, proj, cams, pts, ncons=0, est_focal=1, undistort=1, explicit=1, use_constraints=0,
                point_constraints=None, point_w=0.0, eps2=1e-12, quiet=True, fix_points=0, optimize_for_fisheye=0):
    """Verbatim reference run_sfm (oracle mode A). Returns (cams_out, pts_out)."""
    cams = copy_cams(cams)
    pts = np.array(pts, np.float64, copy=True)
    vm = np.ascontiguousarray(vmask, np.uint8)
    proj = np.ascontiguousarray(proj, np.float64)
    ctx = quiet_stdout() if quiet else contextlib.nullcontext()
    with ctx:
        ref().ref_run_sfm(n, m, ncons, vm.ctypes.data_as(C.c_char_p), _d(proj), est_focal, 0, undistort, explicit,
                          cams, _d(pts), use_constraints, 0 if point_constraints is None else 1,
                          _d(point_constraints), point_w, fix_points, optimize_for_fisheye, eps2, None, None, None, None)
    return cams, pts


def ref_sba_mot(n, m, vmask, proj, cams, pts, itmax, jac_mode, ncons=0, est_focal=1, undistort=1, explicit=1,
                use_constraints=0, eps2=1e-12, quiet=True):
    """Reference sba_mot_levmar (camera-only refinement, what run_sfm calls with fix_points != 0) with chosen itmax /
    Jacobian.  Returns dict(rc, info, p (m*cnp), secs)."""
    cnp = (7 if est_focal else 6) + (2 if undistort else 0)
    cams = copy_cams(cams)
    pts = np.array(pts, np.float64, copy=True)
    vm = np.ascontiguousarray(vmask, np.uint8)
    proj = np.ascontiguousarray(proj, np.float64)
    info = np.zeros(10)
    p = np.zeros(m * cnp)
    secs = C.c_double()
    fn = ref().ref_sba_mot
    fn.restype = C.c_int
    ctx = quiet_stdout() if quiet else contextlib.nullcontext()
    with ctx:
        rc = fn(n, m, ncons, vm.ctypes.data_as(C.c_char_p), _d(proj), est_focal, undistort, explicit, cams, _d(pts),
                use_constraints, C.c_double(eps2), itmax, jac_mode, 0 if quiet else 3, _d(info), _d(p), C.byref(secs))
    return dict(rc=rc, info=info, p=p, secs=secs.value, cnp=cnp)


def ref_sba(n, m, vmask, proj, cams, pts, itmax, jac_mode, ncons=0, est_focal=1, undistort=1, explicit=1,
            use_constraints=0, point_constraints=None, point_w=0.0, eps2=1e-12, want_blocks=False, quiet=True, fisheye=False):
    """Reference sba_motstr_levmar with chosen itmax / Jacobian (0 = reference FD, 1 = our analytic via projac);
    fisheye=True projects with sfm_project_point2_fisheye (sfm.c:448-492), as run_sfm(optimize_for_fisheye=1) does.
    Returns dict(rc, info, p, secs[, U, V, S, W])."""
    cnp = (7 if est_focal else 6) + (2 if undistort else 0)
    cams = copy_cams(cams)
    pts = np.array(pts, np.float64, copy=True)
    vm = np.ascontiguousarray(vmask, np.uint8)
    proj = np.ascontiguousarray(proj, np.float64)
    info = np.zeros(10)
    p = np.zeros(m * cnp + 3 * n)
    secs = C.c_double()
    U = V = S = W = None
    if want_blocks:
        U = np.zeros((m, cnp, cnp)); V = np.zeros((n, 3, 3)); S = np.zeros((m * cnp, m * cnp))
        W = np.zeros((m * cnp, 3 * n))
    ctx = quiet_stdout() if quiet else contextlib.nullcontext()
    with ctx:
        ref().ref_set_fisheye(1 if fisheye else 0)
        rc = ref().ref_sba_motstr(n, m, ncons, vm.ctypes.data_as(C.c_char_p), _d(proj), est_focal, undistort, explicit,
                                  cams, _d(pts), use_constraints, 0 if point_constraints is None else 1,
                                  _d(point_constraints), point_w, eps2, itmax, jac_mode, 0 if quiet else 3,
                                  _d(info), _d(p), _d(V), _d(S), _d(U), _d(W), C.byref(secs))
        ref().ref_set_fisheye(0)
    return dict(rc=rc, info=info, p=p, secs=secs.value, U=U, V=V, S=S, W=W, cnp=cnp)


def cams_from_packed(cams0, p, m, cnp, est_focal=1, undistort=1):
    """Cameras whose parameter block equals the packed vector p but with w folded into R (sfm.c:876-922):
    lets a solver be restarted at the state another solver stopped at."""
    out = copy_cams(cams0)
    for j in range(m):
        a = p[j * cnp:(j + 1) * cnp]
        R0 = np.array(out[j].R).reshape(3, 3)
        w = a[3:6]
        th = np.linalg.norm(w)
        if th > 0:
            nn = w / th
            nx = np.array([[0, -nn[2], nn[1]], [nn[2], 0, -nn[0]], [-nn[1], nn[0], 0]])
            dR = np.eye(3) + np.sin(th) * nx + (1 - np.cos(th)) * nx @ nx
            R0 = dR @ R0
        for q in range(9):
            out[j].R[q] = R0.flat[q]
        for q in range(3):
            out[j].t[q] = a[q]
        c = 6
        if est_focal:
            out[j].f = a[6] / 0.001
            c = 7
        if undistort:
            out[j].k[0] = a[c] / 5.0
            out[j].k[1] = a[c + 1] / 5.0
    return out


# ---------------------------------------------------------------------------------------------------
# port = oracle/liboracle_port.so
class OracleDumps(C.Structure):
    _fields_ = [(k, _dp) for k in ("J", "U", "ea", "V", "eb", "S", "E", "dp", "mu")]


_port = None


def port():
    global _port
    if _port is None:
        lib = C.CDLL(PORT_PATH)
        lib.oracle_sizeof_camera.restype = C.c_int
        assert lib.oracle_sizeof_camera() == C.sizeof(CameraParams)
        lib.oracle_run_sfm.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, _dp, C.c_int, C.c_int, C.c_int, _cp, _dp,
                                       C.c_int, C.c_int, _dp, C.c_double, C.c_double, C.c_int, C.c_int, _dp, _dp,
                                       C.POINTER(OracleDumps)]
        lib.oracle_run_sfm.restype = C.c_int
        lib.oracle_crs_from_vmask.argtypes = [C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.oracle_crs_from_vmask.restype = C.c_int
        lib.oracle_project.argtypes = [C.c_int, C.c_int, C.c_int, _dp, C.c_double, _dp, _dp, _dp]
        lib.oracle_project.restype = None
        lib.oracle_jacobian.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _dp, C.c_double, _dp, _dp, _dp, _dp]
        lib.oracle_jacobian.restype = None
        lib.oracle_chol_solve.argtypes = [C.c_int, _dp, _dp, _dp]
        lib.oracle_chol_solve.restype = C.c_int
        lib.oracle_match_keys.argtypes = [C.c_int, C.POINTER(C.c_ubyte), C.c_int, C.POINTER(C.c_ubyte), C.c_double,
                                          C.POINTER(C.c_int), C.c_int]
        lib.oracle_match_keys.restype = C.c_int
        _port = lib
    return _port


def port_run_sfm(n, m, vmask, proj, cams, pts, itmax=150, jac_mode=0, ncons=0, est_focal=1, undistort=1, explicit=1,
                 use_constraints=0, point_constraints=None, point_w=0.0, eps2=1e-12, want_dumps=False):
    """Restated run_sfm. Returns dict(rc, info, p, cams, pts[, dumps...])."""
    cnp = (7 if est_focal else 6) + (2 if undistort else 0)
    cams = copy_cams(cams)
    pts = np.array(pts, np.float64, copy=True)
    vm = np.ascontiguousarray(vmask, np.uint8)
    proj = np.ascontiguousarray(proj, np.float64)
    nvis = int(vm.sum())
    info = np.zeros(10)
    p = np.zeros(m * cnp + 3 * n)
    out = {}
    dumps = None
    if want_dumps:
        sd = (m - ncons) * cnp
        out = dict(J=np.zeros((nvis, 2 * cnp + 6)), U=np.zeros((m, cnp, cnp)), ea=np.zeros((m, cnp)),
                   V=np.zeros((n, 3, 3)), eb=np.zeros((n, 3)), S=np.zeros((sd, sd)), E=np.zeros(sd),
                   dp=np.zeros(m * cnp + 3 * n), mu=np.zeros(1))
        dumps = OracleDumps(*[_d(out[k]) for k in ("J", "U", "ea", "V", "eb", "S", "E", "dp", "mu")])
    rc = port().oracle_run_sfm(n, m, ncons, vm.ctypes.data_as(C.c_char_p), _d(proj), est_focal, undistort, explicit,
                               cams, _d(pts), use_constraints, 0 if point_constraints is None else 1,
                               _d(point_constraints), point_w, eps2, itmax, jac_mode, _d(info), _d(p),
                               C.byref(dumps) if dumps is not None else None)
    out.update(rc=rc, info=info, p=p, cams=cams, pts=pts, cnp=cnp)
    return out


# ---------------------------------------------------------------------------------------------------
def cams_to_arrays(cams):
    m = len(cams)
    out = dict(R=np.zeros((m, 9)), t=np.zeros((m, 3)), f=np.zeros(m), k=np.zeros((m, 2)),
               constrained=np.zeros((m, 9), np.uint8), constraints=np.zeros((m, 9)), weights=np.zeros((m, 9)))
    for j in range(m):
        out["R"][j] = list(cams[j].R); out["t"][j] = list(cams[j].t); out["f"][j] = cams[j].f
        out["k"][j] = list(cams[j].k)
        out["constrained"][j] = [1 if cams[j].constrained[q] else 0 for q in range(9)]
        out["constraints"][j] = list(cams[j].constraints); out["weights"][j] = list(cams[j].weights)
    return out


def arrays_to_cams(R, t, f, k, constrained=None, constraints=None, weights=None):
    m = len(f)
    cams = (CameraParams * m)()
    for j in range(m):
        for q in range(9):
            cams[j].R[q] = R[j][q]
        for q in range(3):
            cams[j].t[q] = t[j][q]
        cams[j].f = f[j]; cams[j].k[0] = k[j][0]; cams[j].k[1] = k[j][1]
        cams[j].f_scale = 1.0; cams[j].k_scale = 1.0
        if constrained is not None:
            for q in range(9):
                cams[j].constrained[q] = int(constrained[j][q])
                cams[j].constraints[q] = constraints[j][q]; cams[j].weights[q] = weights[j][q]
    return cams


def set_bundler_constraints(cams, focal_weight=1e-4, distortion_weight=100.0):
    """RunBundler's settings: --constrain_focal with weight 0.0001 (RunBundler.sh:55,122; SetFocalConstraint,
    src/Bundle.cpp:977-984) and distortion parameters constrained to 0 with weight 100 (src/Bundle.cpp:942-974)."""
    for c in cams:
        for q in range(9):
            c.constrained[q] = 1 if q >= 6 else 0
        c.constraints[6] = c.f; c.weights[6] = focal_weight
        c.constraints[7] = 0.0; c.weights[7] = distortion_weight
        c.constraints[8] = 0.0; c.weights[8] = distortion_weight
    return cams


_km = None


def ref_km():
    global _km
    if _km is None:
        lib = C.CDLL(REF_KM_PATH)
        lib.ref_match_keys.argtypes = [C.c_int, C.POINTER(C.c_ubyte), C.c_int, C.POINTER(C.c_ubyte), C.c_double, C.c_int,
                                       C.POINTER(C.c_int), C.c_int, _dp]
        lib.ref_match_keys.restype = C.c_int
        _km = lib
    return _km


def ref_match(k1, k2, ratio=0.6, max_pts_visit=0):
    k1 = np.ascontiguousarray(k1, np.uint8); k2 = np.ascontiguousarray(k2, np.uint8)
    out = np.zeros((len(k1), 2), np.int32)
    secs = C.c_double()
    u = C.POINTER(C.c_ubyte)
    cnt = ref_km().ref_match_keys(len(k1), k1.ctypes.data_as(u), len(k2), k2.ctypes.data_as(u), ratio, max_pts_visit,
                                  out.ctypes.data_as(C.POINTER(C.c_int)), len(k1), C.byref(secs))
    return out[:cnt].copy(), secs.value


def port_match(k1, k2, ratio=0.6):
    k1 = np.ascontiguousarray(k1, np.uint8); k2 = np.ascontiguousarray(k2, np.uint8)
    out = np.zeros((len(k1), 2), np.int32)
    u = C.POINTER(C.c_ubyte)
    cnt = port().oracle_match_keys(len(k1), k1.ctypes.data_as(u), len(k2), k2.ctypes.data_as(u), ratio,
                                   out.ctypes.data_as(C.POINTER(C.c_int)), len(k1))
    return out[:cnt].copy()


def ref_triangulate(mode, p, R, t, X0=None):
    """The reference's triangulate_n (mode 0) / triangulate_n_refine (1) / triangulate (2) on ONE point
    (lib/imagelib/triangulate.c).  p: (nviews, 2), R: (nviews, 9), t: (nviews, 3).  Returns (X, err)."""
    p = np.ascontiguousarray(p, np.float64).copy(); R = np.ascontiguousarray(R, np.float64).copy()
    t = np.ascontiguousarray(t, np.float64).copy()
    X = np.zeros(3) if X0 is None else np.array(X0, np.float64, copy=True)
    err = np.zeros(1)
    with quiet_stdout():
        ref().ref_triangulate(int(mode), len(p), _d(p), _d(R), _d(t), _d(X), _d(err))
    return X, err[0]


def _xy1(xy):
    xy = np.asarray(xy, np.float64).reshape(-1, 2)
    return np.concatenate([xy, np.ones((len(xy), 1))], axis=1).ravel().copy()


def ref_fm_ransac(seed, a_xy, b_xy, num_trials, threshold, success_ratio):
    """srand(seed); estimate_fmatrix_ransac_matches (lib/imagelib/fmatrix.c:293-475). Returns (inliers_max, F)."""
    a = _xy1(a_xy); b = _xy1(b_xy); F = np.zeros(9)
    with quiet_stdout():
        c = fmref().ref_fm_ransac(C.c_uint(seed), len(a) // 3, _d(a), _d(b), num_trials, C.c_double(threshold), C.c_double(success_ratio), _d(F))
    return c, F


def ref_fm_estimate(seed, k1_xy, k2_xy, num_trials, threshold):
    """srand(seed); the call sequence of EstimateFMatrix (src/Epipolar.cpp:118-237). Returns (inlier indices, F_ransac, F)."""
    k1 = _xy1(k1_xy); k2 = _xy1(k2_xy); n = len(k1) // 3
    Fr = np.zeros(9); F = np.zeros(9); il = np.zeros(max(n, 1), np.int32)
    with quiet_stdout():
        c = fmref().ref_fm_estimate(C.c_uint(seed), n, _d(k1), _d(k2), num_trials, C.c_double(threshold), _d(Fr), _d(F),
                                    il.ctypes.data_as(C.POINTER(C.c_int)))
    return il[:c].copy(), Fr, F


def ref_rand_sequence(seed, n):
    out = np.zeros(n, np.int32)
    fmref().ref_rand_sequence(C.c_uint(seed), n, out.ctypes.data_as(C.POINTER(C.c_int)))
    return out


def ref_crsm_index(n, m, vmask):
    """The reference's own visibility index (oracle/ref_harness.c:ref_crsm_index): struct sba_crsm filled as
    lib/sba-1.5/sba_levmar.c:653-663 and the camera-major traversal of sba_crsm_col_elmidxs (lib/sba-1.5/sba_crsm.c:183-212)."""
    vm = np.ascontiguousarray(vmask, np.uint8)
    nvis = int((vm != 0).sum())
    ip = C.POINTER(C.c_int)
    out = dict(rowptr=np.zeros(n + 1, np.int32), colidx=np.zeros(max(nvis, 1), np.int32), val=np.zeros(max(nvis, 1), np.int32),
               camptr=np.zeros(m + 1, np.int32), camobs=np.zeros(max(nvis, 1), np.int32), campt=np.zeros(max(nvis, 1), np.int32))
    fn = ref().ref_crsm_index
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.c_int, C.c_char_p, ip, ip, ip, ip, ip, ip]
    got = fn(n, m, vm.ctypes.data_as(C.c_char_p), *[out[k].ctypes.data_as(ip) for k in ("rowptr", "colidx", "val", "camptr", "camobs", "campt")])
    assert got == nvis
    for k in ("colidx", "val", "camobs", "campt"):
        out[k] = out[k][:nvis]
    return out


def ref_kth_element_copy(arr, k):
    """kth_element_copy of lib/imagelib/qsort.c:152-203 (compiled into oracle/_ref/libsfmref.so)."""
    a = np.ascontiguousarray(arr, np.float64).copy()
    fn = ref().ref_kth_element_copy
    fn.restype = C.c_double
    fn.argtypes = [C.c_int, C.c_int, _dp]
    return fn(len(a), int(k), _d(a))


def ref_project_rd(cam, b, undistort=1, explicit_centers=1):
    """sfm_project_rd (lib/sfm-driver/sfm.c:302-380) called as src/Bundle.cpp:726-739 calls it: K = diag(f, f, 1), dt = the
    camera's t (centre), the camera's own R and k."""
    K = np.array([cam.f, 0, 0, 0, cam.f, 0, 0, 0, 1.0])
    k = np.array(list(cam.k), np.float64); R = np.array(list(cam.R), np.float64); dt = np.array(list(cam.t), np.float64)
    bb = np.array(b, np.float64); p = np.zeros(2)
    fn = ref().ref_sfm_project_rd
    fn.restype = None
    fn.argtypes = [_cp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_int, C.c_int]
    one = (CameraParams * 1)()
    C.memmove(one, C.byref(cam), C.sizeof(CameraParams))
    fn(one, _d(K), _d(k), _d(R), _d(dt), _d(bb), _d(p), undistort, explicit_centers)
    return p


TRACKSREF_PATH = os.path.join(os.path.dirname(REF_PATH), "libtracksref.so")


def have_tracksref():
    return os.path.exists(TRACKSREF_PATH)


def read_match_table(path):
    """matches.init.txt as BaseApp::LoadMatchTable reads it (src/BundleIO.cpp:112-166): blocks "i1 i2 / n / k1 k2 ...".
    Returns (pair_i, pair_j, match_ptr, matches[n, 2])."""
    tok = open(path).read().split()
    pi, pj, ptr, mt = [], [], [0], []
    q = 0
    while q < len(tok):
        i1, i2, n = int(tok[q]), int(tok[q + 1]), int(tok[q + 2]); q += 3
        pi.append(i1); pj.append(i2)
        mt.append(np.array(tok[q:q + 2 * n], np.int32).reshape(n, 2)); q += 2 * n
        ptr.append(ptr[-1] + n)
    return (np.array(pi, np.int32), np.array(pj, np.int32), np.array(ptr, np.int32),
            np.concatenate(mt) if mt else np.zeros((0, 2), np.int32))


def prune_double_matches(match_ptr, matches):
    """PruneDoubleMatches (src/MatchTracks.cpp:394-440), the part that edits the lists: inside a pair, a match whose second index was
    already seen is dropped (first occurrence wins)."""
    ptr, out = [0], []
    for p in range(len(match_ptr) - 1):
        seen, keep = set(), []
        for a, b in matches[match_ptr[p]:match_ptr[p + 1]]:
            if int(b) not in seen:
                seen.add(int(b)); keep.append((a, b))
        out += keep; ptr.append(len(out))
    return np.array(ptr, np.int32), np.array(out, np.int32).reshape(-1, 2)


def ref_compute_tracks(num_keys, pair_i, pair_j, match_ptr, matches, new_image_start=0):
    """The reference's BundlerApp::ComputeTracks (oracle/ref_tracks.cpp).  Returns (track_ptr, views[nviews, 2] = (image, key))."""
    lib = C.CDLL(TRACKSREF_PATH)
    ip = C.POINTER(C.c_int)
    nk = np.ascontiguousarray(num_keys, np.int32)
    total = int(nk.sum())
    tp = np.zeros(total + 2, np.int32); vw = np.zeros((max(total, 1), 2), np.int32)
    mt = np.ascontiguousarray(matches, np.int32)
    with quiet_stdout():
        nt = lib.ref_compute_tracks(len(nk), nk.ctypes.data_as(ip), len(pair_i), np.ascontiguousarray(pair_i, np.int32).ctypes.data_as(ip),
                                    np.ascontiguousarray(pair_j, np.int32).ctypes.data_as(ip),
                                    np.ascontiguousarray(match_ptr, np.int32).ctypes.data_as(ip), mt.ctypes.data_as(ip), new_image_start,
                                    tp.ctypes.data_as(ip), vw.ctypes.data_as(ip), total + 1, max(total, 1))
    assert nt >= 0
    return tp[:nt + 1].copy(), vw[:tp[nt]].copy()


PRUNEREF_PATH = os.path.join(os.path.dirname(REF_PATH), "libpruneref.so")


def have_pruneref():
    return os.path.exists(PRUNEREF_PATH)


def ref_remove_bad_points(n, m, rowptr, colidx, cams, pts, threshold):
    """The reference's BundlerApp::RemoveBadPointsAndCameras (src/Bundle.cpp:4190-4261, compiled from where it lies into
    oracle/_ref/libpruneref.so by oracle/ref_prune.cpp).  Returns (num_pruned, prune[n] uint8, max ray angle per point in degrees)."""
    lib = C.CDLL(PRUNEREF_PATH)
    ip = C.POINTER(C.c_int)
    rp = np.ascontiguousarray(rowptr, np.int32); ci = np.ascontiguousarray(colidx, np.int32)
    P = np.ascontiguousarray(pts, np.float64).ravel()
    prune = np.zeros(n, np.uint8); ang = np.zeros(n)
    lib.ref_remove_bad_points.restype = C.c_int
    lib.ref_remove_bad_points.argtypes = [C.c_int, C.c_int, ip, ip, _cp, _dp, C.c_double, C.POINTER(C.c_ubyte), _dp]
    cc = copy_cams(cams)
    with quiet_stdout():
        k = lib.ref_remove_bad_points(n, m, rp.ctypes.data_as(ip), ci.ctypes.data_as(ip), cc, _d(P), float(threshold),
                                      prune.ctypes.data_as(C.POINTER(C.c_ubyte)), _d(ang))
    return k, prune, ang
