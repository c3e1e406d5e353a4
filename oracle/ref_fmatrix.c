/* oracle/ref_fmatrix.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Harness around the REFERENCE'S OWN epipolar-geometry sources (lib/imagelib/fmatrix.c, compiled where it lies by
 * oracle/Makefile into oracle/_ref/libfmref.so; nothing is copied into this repository).  SURVEY 8(f).4.  Exports:
 *   ref_fm_ransac     srand(seed) + estimate_fmatrix_ransac_matches (fmatrix.c:293-475)
 *   ref_fm_linear     estimate_fmatrix_linear (fmatrix.c:729-890)
 *   ref_fm_residual   fmatrix_compute_residual (fmatrix.c:63-87)
 *   ref_fm_refine     refine_fmatrix_nonlinear_matches (fmatrix.c:637-659)
 *   ref_fm_estimate   the call sequence of EstimateFMatrix (src/Epipolar.cpp:118-237, a C++ function around the C calls
 *                     above): RANSAC, inliers of its F, non-linear refinement on them, inliers of the refined F
 *   ref_rand_sequence srand(seed) + n calls of rand(): pins the generator restatement the product uses to draw samples
 * The image / polynomial helpers fmatrix.c links against but never reaches from these entry points are stubbed in
 * oracle/ref_fmatrix_stubs.c. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fmatrix.h"
#include "matrix.h"
#include "vector.h"


int ref_fm_ransac(unsigned seed, int num_pts, double *a_pts, double *b_pts, int num_trials, double threshold,
                  double success_ratio, double *F)
{
    srand(seed);
    return estimate_fmatrix_ransac_matches(num_pts, (v3_t *) a_pts, (v3_t *) b_pts, num_trials, threshold, success_ratio, 0, F);
}

int ref_fm_linear(int num_pts, double *r_pts, double *l_pts, double *F, double *e1, double *e2)
{
    return estimate_fmatrix_linear(num_pts, (v3_t *) r_pts, (v3_t *) l_pts, 0, F, e1, e2);
}

double ref_fm_residual(double *F, double *r, double *l)
{
    return fmatrix_compute_residual(F, v3_new(r[0], r[1], r[2]), v3_new(l[0], l[1], l[2]));
}

void ref_fm_refine(int num_pts, double *r_pts, double *l_pts, double *F0, double *Fout)
{
    refine_fmatrix_nonlinear_matches(num_pts, (v3_t *) r_pts, (v3_t *) l_pts, F0, Fout);
}

/* EstimateFMatrix (src/Epipolar.cpp:118-237) for essential == false.  k1 / k2: 3 doubles per match (x, y, 1).
 * Returns the number of final inliers, their indices in `inliers`, the RANSAC matrix in F_ransac, the final one in F. */
int ref_fm_estimate(unsigned seed, int num_pts, double *k1, double *k2, int num_trials, double threshold,
                    double *F_ransac, double *F, int *inliers)
{
    v3_t *k1_pts = (v3_t *) k1, *k2_pts = (v3_t *) k2;
    v3_t *k1_in, *k2_in;
    int i, n = 0;
    double F0[9];
    if (num_pts < 20) return 0;                                        /* Epipolar.cpp:127-130 */
    srand(seed);
    estimate_fmatrix_ransac_matches(num_pts, k2_pts, k1_pts, num_trials, threshold, 0.95, 0, F);
    memcpy(F_ransac, F, sizeof(double) * 9);
    k1_in = (v3_t *) malloc(sizeof(v3_t) * num_pts); k2_in = (v3_t *) malloc(sizeof(v3_t) * num_pts);
    for (i = 0; i < num_pts; i++)
        if (fmatrix_compute_residual(F, k2_pts[i], k1_pts[i]) < threshold) { k1_in[n] = k1_pts[i]; k2_in[n] = k2_pts[i]; n++; }
    memcpy(F0, F, sizeof(double) * 9);
    refine_fmatrix_nonlinear_matches(n, k2_in, k1_in, F0, F);
    n = 0;
    for (i = 0; i < num_pts; i++)
        if (fmatrix_compute_residual(F, k2_pts[i], k1_pts[i]) < threshold) inliers[n++] = i;
    free(k1_in); free(k2_in);
    return n;
}

void ref_rand_sequence(unsigned seed, int n, int *out)
{
    int i;
    srand(seed);
    for (i = 0; i < n; i++) out[i] = rand();
}
