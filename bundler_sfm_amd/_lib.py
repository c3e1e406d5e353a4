"""ctypes binding of libbsfm_hip.so (the C-ABI declared in include/bsfm.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C bundler_sfm_amd/csrc`.
There is no Python or CPU fallback: if the shared object is missing, importing this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbsfm_hip.so")

INFOSZ = 10
JAC_FD, JAC_ANALYTIC = 0, 1
SOLVER_DENSE, SOLVER_AUTO, SOLVER_ENVELOPE = 0, 1, 2


class CameraParams(C.Structure):
    """Layout-identical to camera_params_t (reference lib/sfm-driver/sfm.h:32-51)."""
    _fields_ = [
        ("R", C.c_double * 9), ("t", C.c_double * 3), ("f", C.c_double), ("k", C.c_double * 2),
        ("k_inv", C.c_double * 6), ("constrained", C.c_ubyte * 9), ("constraints", C.c_double * 9),
        ("weights", C.c_double * 9), ("K_known", C.c_double * 9), ("k_known", C.c_double * 5),
        ("fisheye", C.c_ubyte), ("known_intrinsics", C.c_ubyte),
        ("f_cx", C.c_double), ("f_cy", C.c_double), ("f_rad", C.c_double), ("f_angle", C.c_double),
        ("f_focal", C.c_double), ("f_scale", C.c_double), ("k_scale", C.c_double),
    ]


class Options(C.Structure):
    _fields_ = [("jacobian", C.c_int), ("itmax", C.c_int), ("verbose", C.c_int),
                ("opts", C.c_double * 6), ("potrf_backend", C.c_int), ("reduced_solver", C.c_int), ("num_gpus", C.c_int)]


class RandState(C.Structure):
    """bsfm_rand_t: glibc random() TYPE_3 state."""
    _fields_ = [("s", C.c_uint * 31), ("fi", C.c_int), ("ri", C.c_int)]


class ProblemDesc(C.Structure):
    _fields_ = [
        ("n", C.c_int), ("m", C.c_int), ("mcon", C.c_int),
        ("rowptr", C.POINTER(C.c_int)), ("colidx", C.POINTER(C.c_int)), ("projections", C.POINTER(C.c_double)),
        ("est_focal_length", C.c_int), ("undistort", C.c_int), ("explicit_camera_centers", C.c_int),
        ("cameras", C.POINTER(CameraParams)), ("points", C.POINTER(C.c_double)),
        ("use_constraints", C.c_int), ("use_point_constraints", C.c_int),
        ("point_constraints", C.POINTER(C.c_double)), ("point_constraint_weight", C.c_double),
        ("world_size", C.c_int), ("rank", C.c_int), ("nvis_global", C.c_longlong), ("nvars_global", C.c_longlong),
        ("p_packed", C.POINTER(C.c_double)), ("constraints_prescaled", C.c_int), ("fix_points", C.c_int),
        ("optimize_for_fisheye", C.c_int), ("arrays_on_device", C.c_int),
    ]


class SnavelyModel(C.Structure):
    _fields_ = [("est_focal_length", C.c_int), ("undistort", C.c_int), ("explicit_camera_centers", C.c_int),
                ("R_init", C.POINTER(C.c_double)), ("f_init", C.POINTER(C.c_double)), ("points", C.POINTER(C.c_double))]


class CameraConstraints(C.Structure):     # lib/sba-1.5/sba.h:80-84
    _fields_ = [("constrained", C.POINTER(C.c_char)), ("constraints", C.POINTER(C.c_double)), ("weights", C.POINTER(C.c_double))]


class PointConstraints(C.Structure):      # lib/sba-1.5/sba.h:86-90
    _fields_ = [("constrained", C.c_char), ("constraints", C.c_double * 3), ("weight", C.c_double)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)

# every symbol include/bsfm.h declares (tests/test_abi.py checks the header against this list)
SYMBOLS = [
    "bsfm_default_options", "run_sfm", "bsfm_run_sfm_ex", "bsfm_sba_motstr_levmar", "bsfm_sba_mot_levmar", "bsfm_problem_create", "bsfm_problem_destroy",
    "bsfm_comm_create_from_env", "bsfm_comm_create_all", "bsfm_comm_destroy", "bsfm_comm_rank", "bsfm_comm_world", "bsfm_comm_transport",
    "bsfm_comm_allreduce", "bsfm_comm_allreduce_host", "bsfm_comm_barrier", "bsfm_comm_idfile_exchange", "bsfm_problem_set_comm",
    "bsfm_problem_set_allreduce", "bsfm_problem_set_stream", "bsfm_problem_reset_params", "bsfm_problem_append", "bsfm_problem_remove_points", "bsfm_lm_begin",
    "bsfm_lm_iterate", "bsfm_lm_finish", "bsfm_lm_solve_attempts", "bsfm_lm_last_kernel_ms",
    "bsfm_problem_download", "bsfm_problem_export_index", "bsfm_problem_schur_sizes", "bsfm_problem_export_schur", "bsfm_crs_from_vmask", "bsfm_crs_from_vmask_device", "bsfm_run_sfm_last_ms", "bsfm_schur_chunk", "bsfm_device_cache_trim",
    "bsfm_problem_cnp", "bsfm_problem_num_cameras", "bsfm_problem_num_points", "bsfm_problem_nvis", "bsfm_eval_residuals", "bsfm_problem_outlier_stats", "bsfm_problem_ray_angles", "bsfm_triangulate_batch", "bsfm_rand_seed", "bsfm_rand_next", "bsfm_fmatrix_ransac_batch",
    "bsfm_estimate_fmatrix_batch", "bsfm_compute_tracks",
    "bsfm_eval_normal_equations", "bsfm_dense_chol_solve", "bsfm_dense_chol_solve_timed", "bsfm_chol_flow_schedule", "bsfm_chol_dyn_plan", "bsfm_comm_share", "bsfm_comm_unshare", "bsfm_dense_chol_solve_dist", "bsfm_match_keys_l2", "bsfm_key_match_full",
    "bsfm_key_match_full_sharded", "bsfm_merge_match_files", "bsfm_match_set_create", "bsfm_match_set_run", "bsfm_match_set_run_table", "bsfm_free", "bsfm_match_set_stats", "bsfm_match_kernel", "bsfm_match_set_rescan_launches",
    "bsfm_match_set_destroy",
    "bsfm_device_count", "bsfm_version", "bsfm_device_synchronize", "bsfm_synth_ba", "bsfm_synth_keys",
]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). bundler_sfm_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
    cp = C.POINTER(CameraParams)
    lib.bsfm_default_options.argtypes = [C.POINTER(Options)]
    lib.bsfm_default_options.restype = None
    run_args = [C.c_int, C.c_int, C.c_int, C.c_char_p, dp, C.c_int, C.c_int, C.c_int, C.c_int, cp, dp,
                C.c_int, C.c_int, dp, C.c_double, C.c_int, C.c_int, C.c_double, dp, dp, dp, dp]
    lib.run_sfm.argtypes = run_args
    lib.run_sfm.restype = None
    lib.bsfm_run_sfm_ex.argtypes = run_args + [C.POINTER(Options), dp]
    lib.bsfm_run_sfm_ex.restype = C.c_int
    lib.bsfm_problem_create.argtypes = [C.POINTER(ProblemDesc), C.POINTER(Options)]
    lib.bsfm_problem_create.restype = vp
    lib.bsfm_problem_destroy.argtypes = [vp]
    lib.bsfm_problem_destroy.restype = None
    lib.bsfm_problem_set_allreduce.argtypes = [vp, ALLREDUCE_FN, vp]
    lib.bsfm_problem_set_allreduce.restype = None
    lib.bsfm_problem_set_stream.argtypes = [vp, vp]
    lib.bsfm_problem_set_stream.restype = None
    lib.bsfm_problem_reset_params.argtypes = [vp, cp, dp]
    lib.bsfm_problem_reset_params.restype = C.c_int
    lib.bsfm_problem_append.argtypes = [vp, C.c_int, cp, C.c_int, dp, C.c_int, ip, ip, dp]
    lib.bsfm_problem_append.restype = C.c_int
    lib.bsfm_problem_remove_points.argtypes = [vp, C.POINTER(C.c_ubyte), ip]
    lib.bsfm_problem_remove_points.restype = C.c_int
    lib.bsfm_lm_begin.argtypes = [vp]
    lib.bsfm_lm_begin.restype = C.c_int
    lib.bsfm_lm_iterate.argtypes = [vp, C.c_int]
    lib.bsfm_lm_iterate.restype = C.c_int
    lib.bsfm_lm_finish.argtypes = [vp, dp]
    lib.bsfm_lm_finish.restype = C.c_int
    lib.bsfm_lm_solve_attempts.argtypes = [vp]
    lib.bsfm_lm_solve_attempts.restype = C.c_int
    lib.bsfm_lm_last_kernel_ms.argtypes = [vp, C.c_char_p]
    lib.bsfm_lm_last_kernel_ms.restype = C.c_double
    lib.bsfm_problem_download.argtypes = [vp, dp, cp, dp]
    lib.bsfm_problem_download.restype = C.c_int
    lib.bsfm_problem_cnp.argtypes = [vp]
    lib.bsfm_problem_cnp.restype = C.c_int
    lib.bsfm_problem_num_cameras.argtypes = [vp]
    lib.bsfm_problem_num_cameras.restype = C.c_int
    lib.bsfm_problem_num_points.argtypes = [vp]
    lib.bsfm_problem_num_points.restype = C.c_int
    lib.bsfm_problem_nvis.argtypes = [vp]
    lib.bsfm_problem_nvis.restype = C.c_longlong
    lib.bsfm_eval_residuals.argtypes = [vp, dp, dp]
    lib.bsfm_eval_residuals.restype = C.c_int
    lib.bsfm_problem_outlier_stats.argtypes = [vp, C.c_double, C.c_double, C.POINTER(C.c_int), dp, dp, dp, dp, C.POINTER(C.c_ubyte), dp, dp]
    lib.bsfm_problem_outlier_stats.restype = C.c_int
    lib.bsfm_problem_ray_angles.argtypes = [vp, C.c_double, dp, C.POINTER(C.c_ubyte), C.POINTER(C.c_int)]
    lib.bsfm_problem_ray_angles.restype = C.c_int
    lib.bsfm_triangulate_batch.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), dp, C.POINTER(C.c_int), C.c_int, dp, dp, dp, dp,
                                           C.POINTER(C.c_int)]
    lib.bsfm_triangulate_batch.restype = C.c_int
    lib.bsfm_rand_seed.argtypes = [C.POINTER(RandState), C.c_uint]
    lib.bsfm_rand_seed.restype = None
    lib.bsfm_rand_next.argtypes = [C.POINTER(RandState)]
    lib.bsfm_rand_next.restype = C.c_int
    lib.bsfm_fmatrix_ransac_batch.argtypes = [C.c_int, C.POINTER(C.c_int), dp, dp, C.c_int, C.c_double, C.c_double,
                                              C.POINTER(RandState), dp, C.POINTER(C.c_int)]
    lib.bsfm_fmatrix_ransac_batch.restype = C.c_int
    lib.bsfm_estimate_fmatrix_batch.argtypes = [C.c_int, C.POINTER(C.c_int), dp, dp, C.c_int, C.c_double, C.POINTER(RandState), dp,
                                                C.POINTER(C.c_int), C.POINTER(C.c_ubyte), C.POINTER(C.c_int)]
    lib.bsfm_estimate_fmatrix_batch.restype = C.c_int
    lib.bsfm_compute_tracks.argtypes = [C.c_int, ip, C.c_int, ip, ip, ip, ip, C.c_int, ip, ip, C.c_int, C.c_int, ip]
    lib.bsfm_compute_tracks.restype = C.c_int
    lib.bsfm_eval_normal_equations.argtypes = [vp, C.c_double, dp, dp, dp, dp, dp, dp, dp]
    lib.bsfm_eval_normal_equations.restype = C.c_int
    lib.bsfm_dense_chol_solve.argtypes = [C.c_int, dp, dp, dp, C.c_int]
    lib.bsfm_dense_chol_solve.restype = C.c_int
    lib.bsfm_dense_chol_solve_dist.argtypes = [vp, C.c_int, dp, dp, dp, C.c_int]
    lib.bsfm_dense_chol_solve_dist.restype = C.c_int
    lib.bsfm_comm_share.argtypes = [vp, vp, C.POINTER(vp)]
    lib.bsfm_comm_share.restype = C.c_int
    lib.bsfm_comm_unshare.argtypes = [vp, C.POINTER(vp)]
    lib.bsfm_comm_unshare.restype = C.c_int
    lib.bsfm_comm_idfile_exchange.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_ubyte)]
    lib.bsfm_comm_idfile_exchange.restype = C.c_int
    lib.bsfm_chol_flow_schedule.argtypes = [C.c_int, ip, C.c_int, C.c_int, vp, C.c_int, dp]
    lib.bsfm_chol_flow_schedule.restype = C.c_int
    lib.bsfm_chol_dyn_plan.argtypes = [C.c_int, ip, vp, C.c_int, vp, C.c_int, C.POINTER(C.c_uint), C.c_int, ip]
    lib.bsfm_chol_dyn_plan.restype = C.c_int
    ucp = C.POINTER(C.c_ubyte)
    lib.bsfm_match_keys_l2.argtypes = [C.c_int, ucp, C.c_int, ucp, C.c_double, ip, C.c_int]
    lib.bsfm_match_keys_l2.restype = C.c_int
    lib.bsfm_key_match_full.argtypes = [C.c_int, ip, C.POINTER(ucp), C.c_double, C.c_int, C.c_char_p]
    lib.bsfm_key_match_full.restype = C.c_int
    lib.bsfm_key_match_full_sharded.argtypes = [C.c_int, ip, C.POINTER(ucp), C.c_double, C.c_int, C.c_char_p, C.c_int, C.c_int]
    lib.bsfm_key_match_full_sharded.restype = C.c_int
    lib.bsfm_merge_match_files.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_char_p]
    lib.bsfm_merge_match_files.restype = C.c_int
    lib.bsfm_match_set_create.argtypes = [C.c_int, ip, C.POINTER(ucp)]
    lib.bsfm_match_set_create.restype = vp
    lib.bsfm_match_set_run.argtypes = [vp, C.c_double, C.c_int, C.c_char_p, C.c_int, C.c_int]
    lib.bsfm_match_set_run.restype = C.c_int
    ipp = C.POINTER(C.POINTER(C.c_int))
    lib.bsfm_match_set_run_table.argtypes = [vp, C.c_double, C.c_int, C.c_int, C.c_int, ipp, ipp, ipp, ipp]
    lib.bsfm_match_set_run_table.restype = C.c_int
    lib.bsfm_free.argtypes = [vp]
    lib.bsfm_free.restype = None
    lib.bsfm_match_set_stats.argtypes = [vp, dp, dp, C.POINTER(C.c_longlong), ip]
    lib.bsfm_match_set_stats.restype = C.c_int
    lib.bsfm_match_kernel.argtypes = [C.c_int]
    lib.bsfm_match_kernel.restype = C.c_int
    lib.bsfm_match_set_rescan_launches.argtypes = [vp]
    lib.bsfm_match_set_rescan_launches.restype = C.c_int
    lib.bsfm_match_set_destroy.argtypes = [vp]
    lib.bsfm_match_set_destroy.restype = None
    lib.bsfm_comm_create_from_env.argtypes = []
    lib.bsfm_comm_create_from_env.restype = vp
    lib.bsfm_comm_create_all.argtypes = [C.c_int, ip, C.POINTER(vp)]
    lib.bsfm_comm_create_all.restype = C.c_int
    lib.bsfm_comm_destroy.argtypes = [vp]
    lib.bsfm_comm_destroy.restype = None
    lib.bsfm_comm_rank.argtypes = [vp]
    lib.bsfm_comm_rank.restype = C.c_int
    lib.bsfm_comm_world.argtypes = [vp]
    lib.bsfm_comm_world.restype = C.c_int
    lib.bsfm_comm_transport.argtypes = [vp]
    lib.bsfm_comm_transport.restype = C.c_char_p
    lib.bsfm_comm_allreduce.argtypes = [vp, vp, C.c_size_t, C.c_int, vp]
    lib.bsfm_comm_allreduce.restype = C.c_int
    lib.bsfm_comm_allreduce_host.argtypes = [vp, dp, C.c_int, C.c_int]
    lib.bsfm_comm_allreduce_host.restype = C.c_int
    lib.bsfm_comm_barrier.argtypes = [vp]
    lib.bsfm_comm_barrier.restype = C.c_int
    lib.bsfm_problem_set_comm.argtypes = [vp, vp]
    lib.bsfm_problem_set_comm.restype = None
    ccp, pcp = C.POINTER(CameraConstraints), C.POINTER(PointConstraints)
    lib.bsfm_sba_motstr_levmar.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, dp, C.c_int, C.c_int, dp, dp, C.c_int, C.c_int, vp,
                                           C.c_int, C.c_int, dp, dp, C.c_int, ccp, C.c_int, pcp, dp, dp, dp, dp]
    lib.bsfm_sba_motstr_levmar.restype = C.c_int
    lib.bsfm_sba_mot_levmar.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, dp, C.c_int, dp, dp, C.c_int, C.c_int, vp,
                                        C.c_int, C.c_int, dp, dp, C.c_int, ccp]
    lib.bsfm_sba_mot_levmar.restype = C.c_int
    lib.bsfm_problem_export_index.argtypes = [vp, ip, ip, ip, ip, ip, ip, ip, ip]
    lib.bsfm_problem_export_index.restype = C.c_int
    lib.bsfm_problem_schur_sizes.argtypes = [vp, ip, ip, ip, ip]
    lib.bsfm_problem_schur_sizes.restype = C.c_int
    lib.bsfm_problem_export_schur.argtypes = [vp, ip, ip, ip, ip, ip, ip]
    lib.bsfm_problem_export_schur.restype = C.c_int
    lib.bsfm_dense_chol_solve_timed.argtypes = [C.c_int, dp, dp, dp, C.c_int, C.c_int, dp, dp, dp]
    lib.bsfm_dense_chol_solve_timed.restype = C.c_int
    lib.bsfm_crs_from_vmask.argtypes = [C.c_int, C.c_int, C.c_char_p, ip, ip]
    lib.bsfm_crs_from_vmask.restype = C.c_int
    lib.bsfm_crs_from_vmask_device.argtypes = [C.c_int, C.c_int, C.c_char_p, ip, ip, dp]
    lib.bsfm_crs_from_vmask_device.restype = C.c_int
    lib.bsfm_schur_chunk.argtypes = []
    lib.bsfm_schur_chunk.restype = C.c_int
    lib.bsfm_device_cache_trim.argtypes = []
    lib.bsfm_device_cache_trim.restype = None
    lib.bsfm_run_sfm_last_ms.argtypes = [C.c_char_p]
    lib.bsfm_run_sfm_last_ms.restype = C.c_double
    lib.bsfm_device_count.argtypes = []
    lib.bsfm_device_count.restype = C.c_int
    lib.bsfm_device_synchronize.argtypes = []
    lib.bsfm_device_synchronize.restype = C.c_int
    lib.bsfm_version.argtypes = []
    lib.bsfm_version.restype = C.c_char_p
    lib.bsfm_synth_ba.argtypes = [C.c_int, C.c_int, C.c_int, C.c_ulonglong, C.c_int, ip, ip, dp, cp, dp]
    lib.bsfm_synth_ba.restype = C.c_int
    lib.bsfm_synth_keys.argtypes = [C.c_int, C.c_ulonglong, ucp, C.c_int, ucp]
    lib.bsfm_synth_keys.restype = C.c_int
    return lib


lib = _load()
