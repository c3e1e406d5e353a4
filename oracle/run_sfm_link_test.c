/* oracle/run_sfm_link_test.c -- TEST INFRASTRUCTURE ONLY (built into oracle/_ref/ by oracle/Makefile).
 *
 * The link recipe of INTEGRATION.md section 1, compiled for real: this translation unit #includes the REFERENCE'S OWN
 * lib/sfm-driver/sfm.h (camera_params_t, v3_t, run_sfm, sfm_project_final exactly as Bundler's src/Bundle.cpp sees
 * them) and is linked with -lbsfm_hip ahead of the reference's sfm.c built with -Drun_sfm=run_sfm_cpu_reference, so
 *   run_sfm                -> libbsfm_hip.so  (the GPU core, no CPU fallback)
 *   sfm_project_final      -> the reference's object (lib/sfm-driver/sfm.c:220-300)
 *   run_sfm_cpu_reference  -> the reference's run_sfm under its new name, callable side by side.
 * It builds a small synthetic scene, calls run_sfm the way RunSFM_SBA does (src/Bundle.cpp:645-652), and prints the mean
 * squared reprojection error before / after through the reference's sfm_project_final, for both implementations.
 * tests/test_boundary_link.py runs it on the GPU box and checks the two summary lines run_sfm prints (sfm.c:872-873).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "sfm.h"

void run_sfm_cpu_reference(int num_pts, int num_cameras, int ncons, char *vmask, double *projections,
                           int est_focal_length, int const_focal_length, int undistort, int explicit_camera_centers,
                           camera_params_t *init_camera_params, v3_t *init_pts, int use_constraints,
                           int use_point_constraints, v3_t *points_constraints, double point_constraint_weight,
                           int fix_points, int optimize_for_fisheye, double eps2,
                           double *Vout, double *Sout, double *Uout, double *Wout);

void f_exit(void) {}   /* f2c's exit_.c hook */

static unsigned long long rs = 88172645463325252ULL;
static double urand(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (double)(rs >> 11) / 9007199254740992.0; }
static double nrand(void) { double u = urand() + 1e-300, v = urand(); return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v); }

static double mean_sq_error(int n, int m, const char *vmask, const double *proj, camera_params_t *cams, v3_t *pts)
{
    double sum = 0.0; int cnt = 0, i, j, k = 0;
    for (i = 0; i < n; i++)
        for (j = 0; j < m; j++)
            if (vmask[i * m + j]) {
                v2_t pr = sfm_project_final(cams + j, pts[i], 1, 1);      /* the REFERENCE'S projection */
                double dx = Vx(pr) - proj[2 * k], dy = Vy(pr) - proj[2 * k + 1];
                sum += dx * dx + dy * dy; ++cnt; ++k;
            }
    return sum / (cnt ? cnt : 1);
}

int main(void)
{
    enum { M = 8, N = 120, DEG = 4 };
    camera_params_t cams[M], cams_gpu[M], cams_cpu[M];
    v3_t pts[N], pts_gpu[N], pts_cpu[N];
    char *vmask = (char *) calloc(N * M, 1);
    double *proj = (double *) malloc(sizeof(double) * 2 * N * DEG);
    int i, j, k = 0, d;
    memset(cams, 0, sizeof(cams));
    for (j = 0; j < M; j++) {       /* ring of cameras looking at the origin, as the synthetic generator of SURVEY 8(d) */
        double a = 6.283185307179586 * j / M, c[3], z[3], x[3], y[3], up[3] = { 0, 0, 1 }, nz, nx;
        c[0] = 6 * cos(a); c[1] = 6 * sin(a); c[2] = 0.3 * sin(3 * a);
        nz = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
        for (d = 0; d < 3; d++) z[d] = c[d] / nz;                                   /* camera looks down -z */
        x[0] = up[1] * z[2] - up[2] * z[1]; x[1] = up[2] * z[0] - up[0] * z[2]; x[2] = up[0] * z[1] - up[1] * z[0];
        nx = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
        for (d = 0; d < 3; d++) x[d] /= nx;
        y[0] = z[1] * x[2] - z[2] * x[1]; y[1] = z[2] * x[0] - z[0] * x[2]; y[2] = z[0] * x[1] - z[1] * x[0];
        for (d = 0; d < 3; d++) { cams[j].R[d] = x[d]; cams[j].R[3 + d] = y[d]; cams[j].R[6 + d] = z[d]; cams[j].t[d] = c[d]; }
        cams[j].f = 1000.0 + 100.0 * urand(); cams[j].k[0] = -0.05 * urand(); cams[j].k[1] = 0.01 * urand();
        cams[j].f_scale = 1.0; cams[j].k_scale = 1.0;
    }
    for (i = 0; i < N; i++) {
        int j0 = (int)(urand() * M) % M;
        pts[i] = v3_new(2 * urand() - 1, 2 * urand() - 1, 2 * urand() - 1);
        for (d = 0; d < DEG; d++) vmask[i * M + (j0 + d * (M / DEG)) % M] = 1;
    }
    for (i = 0; i < N; i++)
        for (j = 0; j < M; j++)
            if (vmask[i * M + j]) {
                v2_t pr = sfm_project_final(cams + j, pts[i], 1, 1);
                proj[2 * k] = Vx(pr) + 0.5 * nrand(); proj[2 * k + 1] = Vy(pr) + 0.5 * nrand(); ++k;
            }
    for (i = 0; i < N; i++) pts[i] = v3_new(Vx(pts[i]) + 0.01 * nrand(), Vy(pts[i]) + 0.01 * nrand(), Vz(pts[i]) + 0.01 * nrand());
    for (j = 0; j < M; j++) { for (d = 0; d < 3; d++) cams[j].t[d] += 0.01 * nrand(); cams[j].f *= 1.0 + 0.01 * nrand(); }

    memcpy(cams_gpu, cams, sizeof(cams)); memcpy(cams_cpu, cams, sizeof(cams));
    memcpy(pts_gpu, pts, sizeof(pts)); memcpy(pts_cpu, pts, sizeof(pts));
    printf("before: %.9e\n", mean_sq_error(N, M, vmask, proj, cams, pts));
    fflush(stdout);
    printf("== gpu\n"); fflush(stdout);
    run_sfm(N, M, 0, vmask, proj, 1, 0, 1, 1, cams_gpu, pts_gpu, 0, 0, NULL, 0.0, 0, 0, 1.0e-12, NULL, NULL, NULL, NULL);
    fflush(stdout);
    printf("== cpu\n"); fflush(stdout);
    run_sfm_cpu_reference(N, M, 0, vmask, proj, 1, 0, 1, 1, cams_cpu, pts_cpu, 0, 0, NULL, 0.0, 0, 0, 1.0e-12, NULL, NULL, NULL, NULL);
    fflush(stdout);
    printf("== end\n");
    printf("after_gpu: %.9e\n", mean_sq_error(N, M, vmask, proj, cams_gpu, pts_gpu));
    printf("after_cpu: %.9e\n", mean_sq_error(N, M, vmask, proj, cams_cpu, pts_cpu));
    {
        double dmax = 0.0;
        for (j = 0; j < M; j++) { double e = fabs(cams_gpu[j].f - cams_cpu[j].f) / cams_cpu[j].f; if (e > dmax) dmax = e; }
        printf("max_rel_focal_diff: %.3e\n", dmax);
        printf("scales: %g %g\n", cams_gpu[0].f_scale, cams_gpu[0].k_scale);
    }
    free(vmask); free(proj);
    return 0;
}
