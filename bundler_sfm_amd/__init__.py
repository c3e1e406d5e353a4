"""bundler_sfm_amd -- MI355X-native sparse bundle-adjustment core (and SIFT matcher) behind Bundler's
run_sfm()/KeyMatchFull boundaries.  The product is libbsfm_hip.so (C-ABI, include/bsfm.h); this package
is the thin host-side mirror used by tests and bench.py."""
from ._lib import lib, CameraParams, Options, ProblemDesc, JAC_FD, JAC_ANALYTIC, SOLVER_DENSE, SOLVER_AUTO, SOLVER_ENVELOPE, LIB_PATH  # noqa: F401
from .sfm import (Problem, run_sfm, synth_ba, dense_vmask, default_options, make_cameras,  # noqa: F401
                  copy_cameras, dense_chol_solve, SYNTH_SEED, triangulate_batch, TRI_N, TRI_N_REFINE, TRI_PAIR,
                  Rand, fmatrix_ransac_batch, estimate_fmatrix_batch, compute_tracks, match_table)
