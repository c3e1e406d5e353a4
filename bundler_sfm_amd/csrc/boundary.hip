// boundary.hip -- the drop-in C-ABI boundary: run_sfm with the reference's exact signature
// (lib/sfm-driver/sfm.h:68-86, body lib/sfm-driver/sfm.c:592-1003) on top of the resident-problem API.
//
// What happens here, in the reference's order:
//   1. dense vmask -> CRS (the ordering contract of lib/sba-1.5/sba_levmar.c:653-663: k-th set bit of
//      vmask in row-major order == k-th measurement) -- integer bookkeeping, bit-exact;
//   2. cnp selection (sfm.c:637-643), parameter packing / scaling and constraints (inside
//      bsfm_problem_create, sfm.c:649-781);
//   3. LM on the GPU (bsfm_lm_*), itmax = 150, verbose summary lines as sfm.c:872-873;
//   4. unpack cameras / points in place (sfm.c:876-929);
//   5. optional Vout/Sout/Uout/Wout export at the solution (lib/sba-1.5/sba_levmar.c:1633-2026).
// Modes the GPU core does not cover (fix_points -> sba_mot_levmar, fisheye, known intrinsics) fail
// loudly and leave every input untouched: there is deliberately no CPU fallback in this library.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/bsfm.h"

extern "C" {

int bsfm_run_sfm_ex(int num_pts, int num_cameras, int ncons, char* vmask, double* projections,
                    int est_focal_length, int const_focal_length, int undistort, int explicit_camera_centers,
                    bsfm_camera_params_t* init_camera_params, bsfm_v3_t* init_pts,
                    int use_constraints, int use_point_constraints,
                    bsfm_v3_t* points_constraints, double point_constraint_weight,
                    int fix_points, int optimize_for_fisheye, double eps2,
                    double* Vout, double* Sout, double* Uout, double* Wout,
                    const bsfm_options_t* opt_in, double info[BSFM_INFOSZ])
{
    bsfm_options_t opt;
    if (opt_in) opt = *opt_in; else bsfm_default_options(&opt);
    opt.opts[2] = eps2;   // sfm.c:707
    double linfo[BSFM_INFOSZ];
    for (int i = 0; i < BSFM_INFOSZ; ++i) linfo[i] = 0.0;
    if (!info) info = linfo;

    if (fix_points || optimize_for_fisheye) {
        fprintf(stderr, "[bsfm] run_sfm: fix_points / fisheye modes are not implemented on the GPU core "
                        "(reference: sba_mot_levmar, sfm.c:839-855); inputs left untouched\n");
        return BSFM_ERROR;
    }
    if (est_focal_length && const_focal_length)
        printf("Error: case of constant focal length has not been implemented.\n");   // sfm.c:521-523
    for (int j = 0; j < num_cameras; ++j)
        if (init_camera_params[j].known_intrinsics) {
            fprintf(stderr, "[bsfm] run_sfm: known-intrinsics cameras (sfm.c:339-358) are not implemented on the GPU core\n");
            return BSFM_ERROR;
        }

    // 1. vmask -> CRS
    std::vector<int> rowptr((size_t)num_pts + 1), colidx;
    {
        size_t nvis = 0;
        const size_t tot = (size_t)num_pts * num_cameras;
        for (size_t q = 0; q < tot; ++q) nvis += (vmask[q] != 0);
        colidx.resize(nvis);
        size_t k = 0;
        for (int i = 0; i < num_pts; ++i) {
            rowptr[i] = (int)k;
            const char* row = vmask + (size_t)i * num_cameras;
            for (int j = 0; j < num_cameras; ++j) if (row[j]) colidx[k++] = j;
        }
        rowptr[num_pts] = (int)k;
    }

    bsfm_problem_desc_t d;
    memset(&d, 0, sizeof(d));
    d.n = num_pts; d.m = num_cameras; d.mcon = ncons;
    d.rowptr = rowptr.data(); d.colidx = colidx.data(); d.projections = projections;
    d.est_focal_length = est_focal_length; d.undistort = undistort; d.explicit_camera_centers = explicit_camera_centers;
    d.cameras = init_camera_params; d.points = reinterpret_cast<const double*>(init_pts);
    d.use_constraints = use_constraints; d.use_point_constraints = use_point_constraints;
    d.point_constraints = reinterpret_cast<const double*>(points_constraints);
    d.point_constraint_weight = point_constraint_weight;
    d.world_size = 1; d.rank = 0;

    bsfm_problem_t* pb = bsfm_problem_create(&d, &opt);
    if (!pb) return BSFM_ERROR;
    int rc = bsfm_lm_begin(pb);
    if (rc == 0) bsfm_lm_iterate(pb, opt.itmax);
    rc = bsfm_lm_finish(pb, info);
    if (opt.verbose >= 1) {
        printf("[run_sfm] Number of iterations: %d\n", (int)info[5]);   // sfm.c:872-873
        printf("info[6] = %0.3f\n", info[6]);
    }
    const int cnp = bsfm_problem_cnp(pb);
    if (rc != BSFM_ERROR || info[5] > 0) {
        // the reference copies the parameter vector back unconditionally (sfm.c:876-929)
        bsfm_problem_download(pb, nullptr, init_camera_params, reinterpret_cast<double*>(init_pts));
    }
    if (Sout || Uout || Vout || Wout) {
        const size_t nvis = colidx.size();
        std::vector<double> J;
        if (Wout) J.resize(nvis * (size_t)(2 * cnp + 6));
        bsfm_eval_normal_equations(pb, 0.0, Uout, nullptr, Vout, nullptr, Wout ? J.data() : nullptr, Sout, nullptr);
        if (Wout) {   // Wout[(j*cnp+ii)*3*n + 3*i + jj] = (A_ij^T B_ij)[ii][jj]  (sba_levmar.c:1836-1846)
            const int js = 2 * cnp + 6;
            for (int i = 0; i < num_pts; ++i)
                for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
                    const int j = colidx[k];
                    const double* A = &J[(size_t)k * js]; const double* B = A + 2 * cnp;
                    for (int ii = 0; ii < cnp; ++ii)
                        for (int jj = 0; jj < 3; ++jj)
                            Wout[((size_t)j * cnp + ii) * 3 * num_pts + (size_t)3 * i + jj] =
                                (j < ncons) ? 0.0 : A[ii] * B[jj] + A[cnp + ii] * B[3 + jj];
                }
        }
    }
    bsfm_problem_destroy(pb);
    return rc;
}

void run_sfm(int num_pts, int num_cameras, int ncons, char* vmask, double* projections,
             int est_focal_length, int const_focal_length, int undistort, int explicit_camera_centers,
             bsfm_camera_params_t* init_camera_params, bsfm_v3_t* init_pts,
             int use_constraints, int use_point_constraints,
             bsfm_v3_t* points_constraints, double point_constraint_weight,
             int fix_points, int optimize_for_fisheye, double eps2,
             double* Vout, double* Sout, double* Uout, double* Wout)
{
    const int rc = bsfm_run_sfm_ex(num_pts, num_cameras, ncons, vmask, projections, est_focal_length,
                                   const_focal_length, undistort, explicit_camera_centers, init_camera_params,
                                   init_pts, use_constraints, use_point_constraints, points_constraints,
                                   point_constraint_weight, fix_points, optimize_for_fisheye, eps2,
                                   Vout, Sout, Uout, Wout, nullptr, nullptr);
    if (rc == BSFM_ERROR && bsfm_device_count() <= 0) {
        // the reference's run_sfm cannot fail silently either (fatal paths exit(1), sba_levmar.c:72-83)
        fprintf(stderr, "[bsfm] run_sfm: no HIP device and no CPU fallback -- aborting\n");
        exit(1);
    }
}

}  // extern "C"
