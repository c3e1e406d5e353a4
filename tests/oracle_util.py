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
                point_constraints=None, point_w=0.0, eps2=1e-12, quiet=True):
    """Verbatim reference run_sfm (oracle mode A). Returns (cams_out, pts_out)."""
    cams = copy_cams(cams)
    pts = np.array(pts, np.float64, copy=True)
    vm = np.ascontiguousarray(vmask, np.uint8)
    proj = np.ascontiguousarray(proj, np.float64)
    ctx = quiet_stdout() if quiet else contextlib.nullcontext()
    with ctx:
        ref().ref_run_sfm(n, m, ncons, vm.ctypes.data_as(C.c_char_p), _d(proj), est_focal, 0, undistort, explicit,
                          cams, _d(pts), use_constraints, 0 if point_constraints is None else 1,
                          _d(point_constraints), point_w, 0, 0, eps2, None, None, None, None)
    return cams, pts


def ref_sba(n, m, vmask, proj, cams, pts, itmax, jac_mode, ncons=0, est_focal=1, undistort=1, explicit=1,
            use_constraints=0, point_constraints=None, point_w=0.0, eps2=1e-12, want_blocks=False, quiet=True):
    """Reference sba_motstr_levmar with chosen itmax / Jacobian (0 = reference FD, 1 = our analytic via projac).
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
        rc = ref().ref_sba_motstr(n, m, ncons, vm.ctypes.data_as(C.c_char_p), _d(proj), est_focal, undistort, explicit,
                                  cams, _d(pts), use_constraints, 0 if point_constraints is None else 1,
                                  _d(point_constraints), point_w, eps2, itmax, jac_mode, 0 if quiet else 3,
                                  _d(info), _d(p), _d(V), _d(S), _d(U), _d(W), C.byref(secs))
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
