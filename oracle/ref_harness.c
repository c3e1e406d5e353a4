/* oracle/ref_harness.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Thin harness around the REFERENCE'S OWN sources (compiled from /root/reference by oracle/Makefile
 * into oracle/_ref/libsfmref.so; nothing from the reference is copied into this repository).
 * It #includes lib/sfm-driver/sfm.c so that the file-static projection callback
 * sfm_project_point3 (sfm.c:503) and the caches global_last_ws/global_last_Rs (sfm.c:382-383) are
 * reachable, because run_sfm hard-codes MAX_ITERS=150 (sfm.c:814) and passes projac=NULL
 * (sfm.c:820-828).  Exports:
 *   ref_run_sfm        verbatim reference run_sfm (oracle mode A).
 *   ref_sba_motstr     reference sba_motstr_levmar (lib/sba-1.5/sba_levmar_wrap.c:599) with a chosen
 *                      itmax and either the reference forward-difference Jacobian (jac_mode 0) or OUR
 *                      analytic Jacobian supplied through SBA's projac hook (jac_mode 1, oracle mode B);
 *                      itmax == 0 runs the reference's own Jacobian checker on it
 *                      (lib/sba-1.5/sba_levmar.c:769-773).
 *   ref_set_fisheye    switch ref_sba_motstr to the fisheye projection callback (sfm.c:448-492).
 *   ref_triangulate    the reference's triangulate_n / triangulate_n_refine / triangulate
 *                      (lib/imagelib/triangulate.c:181-272, 133-178, 281-338), one point per call.
 *   ref_sizeof_camera_params   layout check for the ctypes mirror of camera_params_t (sfm.h:32-51).
 */
#include "sfm.c"          /* resolved through -I$(REF)/lib/sfm-driver */
#include "snavely_model.h"

#include <sys/time.h>

/* f2c's exit_.c wants libf2c's I/O shutdown hook; no Fortran I/O is linked here */
void f_exit(void) {}

typedef struct {
    sfm_global_t *globs;
    sm_config cfg;
} harness_adata_t;

/* sfm_project_point3 receives `adata` as sfm_global_t*; SBA hands the same pointer to projac, so the
 * config for the analytic Jacobian lives in a file-static next to it. */
static sm_config g_cfg;

static void harness_projac(int j, int i, double *aj, double *bi, double *Aij, double *Bij, void *adata)
{
    sfm_global_t *globs = (sfm_global_t *) adata;
    (void) i;
    sm_jacobian(&g_cfg, globs->init_params[j].R, globs->init_params[j].f, aj, bi, Aij, Bij);
}

int ref_sizeof_camera_params(void) { return (int) sizeof(camera_params_t); }

/* ref_sba_motstr projects with sfm_project_point2_fisheye (sfm.c:448-492) instead of sfm_project_point3 while this is
 * set -- what run_sfm does for optimize_for_fisheye != 0 (sfm.c:829-836); the Jacobian is then always the reference's FD. */
static int g_fisheye = 0;
void ref_set_fisheye(int on) { g_fisheye = on; }
/* eps1 of the next ref_sba_motstr calls (run_sfm's value 1e-10, sfm.c:706, unless set).  A huge eps1 makes the reference stop
 * with code 1 before its first step (lib/sba-1.5/sba_levmar.c:1117-1121) -- with Sout != NULL it then exports U, V and the
 * UNDAMPED reduced camera system at the INITIAL parameters: the fixture for the Schur complement at the headline size. */
static double g_eps1 = 1.0e-10;
void ref_set_eps1(double eps1) { g_eps1 = eps1; }
/* tau (opts[0]) of the next ref_sba_motstr calls; run_sfm's value 1e-3 (sfm.c:705) unless set.  tau = 0 gives mu = 0: with a point
 * nobody observes V*_i is the zero matrix, dsytrf reports it, and the reference walks its "singular V*_i => more damping" branch
 * (lib/sba-1.5/sba_levmar.c:1156-1161, 1584-1611) until nu overflows (stop 6). */
static double g_tau = 1.0e-3;
void ref_set_tau(double tau) { g_tau = tau; }

void ref_run_sfm(int num_pts, int num_cameras, int ncons, char *vmask, double *projections,
                 int est_focal_length, int const_focal_length, int undistort, int explicit_camera_centers,
                 camera_params_t *init_camera_params, v3_t *init_pts, int use_constraints,
                 int use_point_constraints, v3_t *pt_constraints, double pt_constraint_weight,
                 int fix_points, int optimize_for_fisheye, double eps2,
                 double *Vout, double *Sout, double *Uout, double *Wout)
{
    run_sfm(num_pts, num_cameras, ncons, vmask, projections, est_focal_length, const_focal_length,
            undistort, explicit_camera_centers, init_camera_params, init_pts, use_constraints,
            use_point_constraints, pt_constraints, pt_constraint_weight, fix_points,
            optimize_for_fisheye, eps2, Vout, Sout, Uout, Wout);
}

/* Packs the parameter vector the way run_sfm does (sfm.c:649-703), sets up constraints
 * (sfm.c:721-781) and the rotation cache (sfm.c:796-811), then calls the reference LM.
 * cams are NOT modified (f_scale/k_scale are restored); the final packed parameter vector is
 * returned in p_out (m*cnp + 3n doubles).  Returns the SBA return code; wall seconds in *secs. */
int ref_sba_motstr(int n, int m, int mcon, char *vmask, double *projections,
                   int est_focal, int undistort, int explicit_centers,
                   camera_params_t *cams, double *pts,
                   int use_constraints, int use_point_constraints, double *pt_constraints,
                   double pt_constraint_weight, double eps2,
                   int itmax, int jac_mode, int verbose,
                   double *info, double *p_out,
                   double *Vout, double *Sout, double *Uout, double *Wout, double *secs)
{
    const double f_scale = 0.001, k_scale = 5.0;
    int cnp = (est_focal ? 7 : 6) + (undistort ? 2 : 0);
    int nvars = cnp * m + 3 * n, i, j, rc;
    double opts[6];
    double *params = (double *) malloc(sizeof(double) * nvars);
    camera_constraints_t *cons = NULL;
    point_constraints_t *pcons = NULL;
    sfm_global_t globs;
    struct timeval t0, t1;

    for (j = 0; j < m; j++) {
        double *a = params + cnp * j;
        int c = 6;
        cams[j].f_scale = f_scale; cams[j].k_scale = k_scale;
        a[0] = cams[j].t[0]; a[1] = cams[j].t[1]; a[2] = cams[j].t[2];
        a[3] = a[4] = a[5] = 0.0;
        if (est_focal) { a[6] = cams[j].f * f_scale; c = 7; }
        if (undistort) { a[c] = cams[j].k[0] * k_scale; a[c + 1] = cams[j].k[1] * k_scale; }
    }
    memcpy(params + cnp * m, pts, sizeof(double) * 3 * n);

    opts[0] = g_tau; opts[1] = g_eps1; opts[2] = eps2; opts[3] = 1.0e-12; opts[4] = 0.0; opts[5] = 4.0e-2;

    if (use_constraints) {
        cons = (camera_constraints_t *) malloc(m * sizeof(camera_constraints_t));
        for (j = 0; j < m; j++) {
            cons[j].constrained = (char *) malloc(cnp);
            cons[j].constraints = (double *) malloc(sizeof(double) * cnp);
            cons[j].weights = (double *) malloc(sizeof(double) * cnp);
            memcpy(cons[j].constrained, cams[j].constrained, cnp);
            memcpy(cons[j].constraints, cams[j].constraints, cnp * sizeof(double));
            memcpy(cons[j].weights, cams[j].weights, cnp * sizeof(double));
            if (est_focal) { cons[j].constraints[6] *= f_scale; cons[j].weights[6] *= 1.0 / (f_scale * f_scale); }
            if (undistort) {
                cons[j].constraints[7] *= k_scale; cons[j].weights[7] *= 1.0 / (k_scale * k_scale);
                cons[j].constraints[8] *= k_scale; cons[j].weights[8] *= 1.0 / (k_scale * k_scale);
            }
        }
    }
    if (use_point_constraints) {
        pcons = (point_constraints_t *) malloc(n * sizeof(point_constraints_t));
        for (i = 0; i < n; i++) {
            double *pc = pt_constraints + 3 * i;
            int on = !(pc[0] == 0.0 && pc[1] == 0.0 && pc[2] == 0.0);
            pcons[i].constrained = (char) on;
            pcons[i].weight = on ? pt_constraint_weight : 0.0;
            pcons[i].constraints[0] = on ? pc[0] : 0.0;
            pcons[i].constraints[1] = on ? pc[1] : 0.0;
            pcons[i].constraints[2] = on ? pc[2] : 0.0;
        }
    }

    memset(&globs, 0, sizeof(globs));
    globs.num_cameras = m; globs.num_points = n; globs.num_params_per_camera = cnp;
    globs.est_focal_length = est_focal; globs.const_focal_length = 0;
    globs.estimate_distortion = undistort; globs.explicit_camera_centers = explicit_centers;
    globs.global_params.f = 1.0; globs.init_params = cams; globs.points = (v3_t *) pts;

    global_last_ws = (double *) calloc(3 * m, sizeof(double));
    global_last_Rs = (double *) malloc(9 * m * sizeof(double));
    for (j = 0; j < m; j++) memcpy(global_last_Rs + 9 * j, cams[j].R, 9 * sizeof(double));

    g_cfg.cnp = cnp; g_cfg.est_focal = est_focal; g_cfg.undistort = undistort;
    g_cfg.explicit_centers = explicit_centers; g_cfg.f_scale = f_scale; g_cfg.k_scale = k_scale;

    gettimeofday(&t0, NULL);
    rc = sba_motstr_levmar(n, m, mcon, vmask, params, cnp, 3, projections, NULL, 2,
                           g_fisheye ? sfm_project_point2_fisheye : sfm_project_point3,
                           (jac_mode && !g_fisheye) ? harness_projac : NULL, (void *) &globs,
                           itmax, verbose, opts, info, use_constraints, cons,
                           use_point_constraints, pcons, Vout, Sout, Uout, Wout);
    gettimeofday(&t1, NULL);
    if (secs) *secs = (t1.tv_sec - t0.tv_sec) + 1e-6 * (t1.tv_usec - t0.tv_usec);

    if (p_out) memcpy(p_out, params, sizeof(double) * nvars);
    for (j = 0; j < m; j++) { cams[j].f_scale = 1.0; cams[j].k_scale = 1.0; }

    if (cons) { for (j = 0; j < m; j++) { free(cons[j].constrained); free(cons[j].constraints); free(cons[j].weights); } free(cons); }
    free(pcons); free(params); free(global_last_ws); free(global_last_Rs);
    global_last_ws = global_last_Rs = NULL;
    return rc;
}

/* Camera-only refinement (run_sfm with fix_points != 0: lib/sfm-driver/sfm.c:839-846 -> sba_mot_levmar,
 * lib/sba-1.5/sba_levmar_wrap.c:707-800, expert driver sba_levmar.c:2090-2690): same set-up as ref_sba_motstr but the
 * parameter vector holds the m*cnp camera parameters only; points stay what `pts` says.  jac_mode 1 hands OUR analytic
 * d x / d a to SBA through its projac hook.  p_out gets m*cnp doubles. */
static void harness_projac_mot(int j, int i, double *aj, double *Aij, void *adata)
{
    sfm_global_t *globs = (sfm_global_t *) adata;
    double Bij[6];
    sm_jacobian(&g_cfg, globs->init_params[j].R, globs->init_params[j].f, aj, globs->points[i].p, Aij, Bij);
}

int ref_sba_mot(int n, int m, int mcon, char *vmask, double *projections,
                int est_focal, int undistort, int explicit_centers,
                camera_params_t *cams, double *pts, int use_constraints, double eps2,
                int itmax, int jac_mode, int verbose, double *info, double *p_out, double *secs)
{
    const double f_scale = 0.001, k_scale = 5.0;
    int cnp = (est_focal ? 7 : 6) + (undistort ? 2 : 0);
    int nvars = cnp * m, j, rc;
    double opts[6];
    double *params = (double *) malloc(sizeof(double) * nvars);
    camera_constraints_t *cons = NULL;
    sfm_global_t globs;
    struct timeval t0, t1;

    for (j = 0; j < m; j++) {
        double *a = params + cnp * j;
        int c = 6;
        cams[j].f_scale = f_scale; cams[j].k_scale = k_scale;
        a[0] = cams[j].t[0]; a[1] = cams[j].t[1]; a[2] = cams[j].t[2];
        a[3] = a[4] = a[5] = 0.0;
        if (est_focal) { a[6] = cams[j].f * f_scale; c = 7; }
        if (undistort) { a[c] = cams[j].k[0] * k_scale; a[c + 1] = cams[j].k[1] * k_scale; }
    }
    opts[0] = 1.0e-3; opts[1] = 1.0e-10; opts[2] = eps2; opts[3] = 1.0e-12; opts[4] = 0.0; opts[5] = 4.0e-2;
    if (use_constraints) {
        cons = (camera_constraints_t *) malloc(m * sizeof(camera_constraints_t));
        for (j = 0; j < m; j++) {
            cons[j].constrained = (char *) malloc(cnp);
            cons[j].constraints = (double *) malloc(sizeof(double) * cnp);
            cons[j].weights = (double *) malloc(sizeof(double) * cnp);
            memcpy(cons[j].constrained, cams[j].constrained, cnp);
            memcpy(cons[j].constraints, cams[j].constraints, cnp * sizeof(double));
            memcpy(cons[j].weights, cams[j].weights, cnp * sizeof(double));
            if (est_focal) { cons[j].constraints[6] *= f_scale; cons[j].weights[6] *= 1.0 / (f_scale * f_scale); }
            if (undistort) {
                cons[j].constraints[7] *= k_scale; cons[j].weights[7] *= 1.0 / (k_scale * k_scale);
                cons[j].constraints[8] *= k_scale; cons[j].weights[8] *= 1.0 / (k_scale * k_scale);
            }
        }
    }
    memset(&globs, 0, sizeof(globs));
    globs.num_cameras = m; globs.num_points = n; globs.num_params_per_camera = cnp;
    globs.est_focal_length = est_focal; globs.const_focal_length = 0;
    globs.estimate_distortion = undistort; globs.explicit_camera_centers = explicit_centers;
    globs.global_params.f = 1.0; globs.init_params = cams; globs.points = (v3_t *) pts;
    global_last_ws = (double *) calloc(3 * m, sizeof(double));
    global_last_Rs = (double *) malloc(9 * m * sizeof(double));
    for (j = 0; j < m; j++) memcpy(global_last_Rs + 9 * j, cams[j].R, 9 * sizeof(double));
    g_cfg.cnp = cnp; g_cfg.est_focal = est_focal; g_cfg.undistort = undistort;
    g_cfg.explicit_centers = explicit_centers; g_cfg.f_scale = f_scale; g_cfg.k_scale = k_scale;

    gettimeofday(&t0, NULL);
    rc = sba_mot_levmar(n, m, mcon, vmask, params, cnp, projections, NULL, 2,
                        sfm_project_point3_mot, jac_mode ? harness_projac_mot : NULL, (void *) &globs,
                        itmax, verbose, opts, info, use_constraints, cons);
    gettimeofday(&t1, NULL);
    if (secs) *secs = (t1.tv_sec - t0.tv_sec) + 1e-6 * (t1.tv_usec - t0.tv_usec);
    if (p_out) memcpy(p_out, params, sizeof(double) * nvars);
    for (j = 0; j < m; j++) { cams[j].f_scale = 1.0; cams[j].k_scale = 1.0; }
    if (cons) { for (j = 0; j < m; j++) { free(cons[j].constrained); free(cons[j].constraints); free(cons[j].weights); } free(cons); }
    free(params); free(global_last_ws); free(global_last_Rs);
    global_last_ws = global_last_Rs = NULL;
    return rc;
}

/* Reference projection of one observation with the packed parameters (for model parity tests). */
void ref_project_point(int est_focal, int undistort, int explicit_centers, camera_params_t *cam,
                       double *aj, double *bi, double *xij)
{
    sfm_global_t globs;
    double ws[3] = { 1e300, 1e300, 1e300 }, Rs[9];
    double fs = cam->f_scale, ks = cam->k_scale;
    memset(&globs, 0, sizeof(globs));
    globs.num_cameras = 1; globs.num_points = 1;
    globs.est_focal_length = est_focal; globs.estimate_distortion = undistort;
    globs.explicit_camera_centers = explicit_centers; globs.init_params = cam;
    cam->f_scale = 0.001; cam->k_scale = 5.0;
    global_last_ws = ws; global_last_Rs = Rs;
    sfm_project_point3(0, 0, aj, bi, xij, &globs);
    global_last_ws = global_last_Rs = NULL;
    cam->f_scale = fs; cam->k_scale = ks;
}

/* ---- multi-view triangulation ---------------------------------------------------------------------- */
#include "triangulate.h"

/* mode 0 = triangulate_n, 1 = triangulate_n_refine (X holds the start), 2 = triangulate (two views; error = sum of squares).
 * p: 2*nviews, R: 9*nviews, t: 3*nviews; X: 3 (in/out); err: 1. */
void ref_triangulate(int mode, int nviews, double *p, double *R, double *t, double *X, double *err)
{
    v3_t r;
    if (mode == 0) r = triangulate_n(nviews, (v2_t *) p, R, t, err);
    else if (mode == 1) r = triangulate_n_refine(v3_new(X[0], X[1], X[2]), nviews, (v2_t *) p, R, t, err);
    else r = triangulate(v2_new(p[0], p[1]), v2_new(p[2], p[3]), R, t, R + 9, t + 3, err);
    X[0] = Vx(r); X[1] = Vy(r); X[2] = Vz(r);
}

/* ---- index bookkeeping --------------------------------------------------------------------------
 * The reference's own visibility index: sba_crsm_alloc + the fill loop of sba_motstr_levmar_x (lib/sba-1.5/sba_levmar.c:653-663,
 * restated here because it is inline in that function), then the camera-major traversal every U_j / Q / Jacobian loop uses,
 * sba_crsm_col_elmidxs (lib/sba-1.5/sba_crsm.c:183-212): for camera j the list of (val index, point) pairs in ascending point
 * order.  Outputs: rowptr (n+1), colidx (nvis), val (nvis), camptr (m+1), camobs (nvis: idxij.val[rcidxs[.]] in traversal
 * order), campt (nvis: rcsubs in traversal order).  Returns nvis. */
int ref_crsm_index(int n, int m, char *vmask, int *rowptr, int *colidx, int *val, int *camptr, int *camobs, int *campt)
{
    struct sba_crsm idxij;
    int i, j, k, ii, nvis, jj, nnz, t = 0;
    int *rcidxs, *rcsubs;
    for (i = nvis = 0, jj = n * m; i < jj; ++i) nvis += (vmask[i] != 0);
    sba_crsm_alloc(&idxij, n, m, nvis);
    for (i = k = 0; i < n; ++i) {
        idxij.rowptr[i] = k;
        ii = i * m;
        for (j = 0; j < m; ++j)
            if (vmask[ii + j]) { idxij.val[k] = k; idxij.colidx[k++] = j; }
    }
    idxij.rowptr[n] = nvis;
    memcpy(rowptr, idxij.rowptr, (n + 1) * sizeof(int));
    memcpy(colidx, idxij.colidx, nvis * sizeof(int));
    memcpy(val, idxij.val, nvis * sizeof(int));
    rcidxs = (int *) malloc((n > m ? n : m) * sizeof(int) + sizeof(int));
    rcsubs = (int *) malloc((n > m ? n : m) * sizeof(int) + sizeof(int));
    for (j = 0; j < m; ++j) {
        camptr[j] = t;
        nnz = sba_crsm_col_elmidxs(&idxij, j, rcidxs, rcsubs);
        for (i = 0; i < nnz; ++i) { camobs[t] = idxij.val[rcidxs[i]]; campt[t] = rcsubs[i]; ++t; }
    }
    camptr[m] = t;
    free(rcidxs); free(rcsubs);
    sba_crsm_free(&idxij);
    return nvis;
}

/* ---- post-solve statistics of RunSFM_SBA ----------------------------------------------------------------
 * The two reference routines the statistics loop of src/Bundle.cpp:659-913 is built on: the final projection
 * sfm_project_rd (lib/sfm-driver/sfm.c:302-380; Bundle.cpp:736 calls it with the camera's R, its k, 1/f scaling done by the
 * caller) and kth_element_copy (lib/imagelib/qsort.c:152-203, compiled into this library by oracle/Makefile). */
extern double kth_element_copy(int n, int k, double *arr);
double ref_kth_element_copy(int n, int k, double *arr) { return kth_element_copy(n, k, arr); }
void ref_sfm_project_rd(camera_params_t *cam, double *K, double *k, double *R, double *dt, double *b, double *p,
                        int undistort, int explicit_centers)
{
    sfm_project_rd(cam, K, k, R, dt, b, p, undistort, explicit_centers);
}
