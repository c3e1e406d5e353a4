/* oracle/snavely_model.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Plain-C restatement of Bundler's per-observation camera model and its derivatives.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
 *
 * What it restates (paths relative to the reference tree):
 *   - rot_update                 lib/sfm-driver/sfm.c:77-116   (Rodrigues update R = dR(w) * R_init)
 *   - sfm_project_point3         lib/sfm-driver/sfm.c:503-552  (parameter unpacking, f = a[6]/f_scale)
 *   - sfm_project_rd             lib/sfm-driver/sfm.c:302-380  (P = R(b-c); p = -f*P.xy/P.z; radial factor)
 *   - SnavelyReprojectionError   include/snavely_reprojection_error.h:57-92 (same model, angle-axis form)
 *   - forward differences        lib/sba-1.5/sba_levmar_wrap.c:163-259 (d = max(1e-4*|p|, 1e-6))
 * The analytic Jacobian has no counterpart in the reference (Bundler passes projac=NULL,
 * lib/sfm-driver/sfm.c:820-828); it is derived here and validated against the reference's own
 * checker sba_motstr_chkjac_x (lib/sba-1.5/sba_chkjac.c:91-210) via oracle/ref_harness.c.
 *
 * Parameter layout of one camera a_j (lib/sfm-driver/sfm.c:652-696):
 *   a[0..2] = camera centre c (explicit centres) or translation t
 *   a[3..5] = incremental rotation w (restarts at 0 on every run_sfm call)
 *   a[6]    = f * f_scale            (only if est_focal)
 *   a[6+est_focal .. +1] = k1*k_scale, k2*k_scale   (only if undistort)
 */
#ifndef BSFM_ORACLE_SNAVELY_MODEL_H
#define BSFM_ORACLE_SNAVELY_MODEL_H

#include <math.h>
#include <string.h>

typedef struct {
    int cnp;              /* 6, 7, 8 or 9 */
    int est_focal;        /* a[6] carries the scaled focal length */
    int undistort;        /* two scaled radial coefficients follow */
    int explicit_centers; /* a[0..2] is the camera centre (Bundler always passes 1) */
    double f_scale;       /* 0.001 inside run_sfm (sfm.c:634) */
    double k_scale;       /* 5.0   inside run_sfm (sfm.c:635) */
} sm_config;

/* R = dR(w) * Rinit ; theta == 0 short-circuits to Rinit (sfm.c:91-94). */
static void sm_rot_update(const double *Rinit, const double *w, double *R)
{
    double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    double n[3], nx[9], nxsq[9], dR[9], s, c1;
    int r, c, k;
    if (th == 0.0) { memcpy(R, Rinit, 9 * sizeof(double)); return; }
    n[0] = w[0] / th; n[1] = w[1] / th; n[2] = w[2] / th;
    nx[0] = 0.0;   nx[1] = -n[2]; nx[2] = n[1];
    nx[3] = n[2];  nx[4] = 0.0;   nx[5] = -n[0];
    nx[6] = -n[1]; nx[7] = n[0];  nx[8] = 0.0;
    for (r = 0; r < 3; r++) for (c = 0; c < 3; c++) {
        double acc = 0.0;
        for (k = 0; k < 3; k++) acc += nx[3 * r + k] * nx[3 * k + c];
        nxsq[3 * r + c] = acc;
    }
    s = sin(th); c1 = 1.0 - cos(th);
    for (k = 0; k < 9; k++) dR[k] = ((k % 4 == 0) ? 1.0 : 0.0) + s * nx[k] + c1 * nxsq[k];
    for (r = 0; r < 3; r++) for (c = 0; c < 3; c++) {
        double acc = 0.0;
        for (k = 0; k < 3; k++) acc += dR[3 * r + k] * Rinit[3 * k + c];
        R[3 * r + c] = acc;
    }
}

/* Projection of point b through camera a with current rotation R (already updated). */
static void sm_project_R(const sm_config *cfg, const double *R, double f_init,
                         const double *a, const double *b, double *x)
{
    double P[3], d[3], f, p0, p1;
    if (cfg->explicit_centers) {
        d[0] = b[0] - a[0]; d[1] = b[1] - a[1]; d[2] = b[2] - a[2];
        P[0] = R[0] * d[0] + R[1] * d[1] + R[2] * d[2];
        P[1] = R[3] * d[0] + R[4] * d[1] + R[5] * d[2];
        P[2] = R[6] * d[0] + R[7] * d[1] + R[8] * d[2];
    } else {
        P[0] = R[0] * b[0] + R[1] * b[1] + R[2] * b[2] + a[0];
        P[1] = R[3] * b[0] + R[4] * b[1] + R[5] * b[2] + a[1];
        P[2] = R[6] * b[0] + R[7] * b[1] + R[8] * b[2] + a[2];
    }
    f = cfg->est_focal ? a[6] / cfg->f_scale : f_init;
    p0 = -P[0] * f / P[2];
    p1 = -P[1] * f / P[2];
    if (cfg->undistort) {
        const double *ks = a + (cfg->est_focal ? 7 : 6);
        double k1 = ks[0] / cfg->k_scale, k2 = ks[1] / cfg->k_scale;
        double rsq = (p0 * p0 + p1 * p1) / (f * f);
        double factor = 1.0 + k1 * rsq + k2 * rsq * rsq;
        p0 *= factor; p1 *= factor;
    }
    x[0] = p0; x[1] = p1;
}

static void sm_project(const sm_config *cfg, const double *Rinit, double f_init,
                       const double *a, const double *b, double *x)
{
    double R[9];
    sm_rot_update(Rinit, a + 3, R);
    sm_project_R(cfg, R, f_init, a, b, x);
}

/* Analytic A = dx/da (2 x cnp, row-major), B = dx/db (2 x 3, row-major).
 * d(dR(w) v)/dw = -R [v]x M(w),  M = a I + (1-a) n n^T - b [n]x,  a = sin(th)/th, b = (1-cos th)/th
 * (exponential-map derivative; at th == 0 it reduces to -[v]x with v = P). */
static void sm_jacobian(const sm_config *cfg, const double *Rinit, double f_init,
                        const double *a, const double *b, double *A, double *B)
{
    const int cnp = cfg->cnp;
    double R[9], d[3], P[3], f, iz, u0, u1, rsq, g, dg, k1 = 0.0, k2 = 0.0;
    double D[6];   /* dx/dP, 2x3 */
    double Jw[9];  /* dP/dw, 3x3 */
    double th;
    const double *w = a + 3;
    int r, c, k;

    sm_rot_update(Rinit, w, R);
    if (cfg->explicit_centers) {
        d[0] = b[0] - a[0]; d[1] = b[1] - a[1]; d[2] = b[2] - a[2];
    } else {
        d[0] = b[0]; d[1] = b[1]; d[2] = b[2];
    }
    for (r = 0; r < 3; r++) P[r] = R[3 * r] * d[0] + R[3 * r + 1] * d[1] + R[3 * r + 2] * d[2];
    if (!cfg->explicit_centers) { P[0] += a[0]; P[1] += a[1]; P[2] += a[2]; }

    f = cfg->est_focal ? a[6] / cfg->f_scale : f_init;
    if (cfg->undistort) {
        const double *ks = a + (cfg->est_focal ? 7 : 6);
        k1 = ks[0] / cfg->k_scale; k2 = ks[1] / cfg->k_scale;
    }
    iz = 1.0 / P[2];
    u0 = -P[0] * f * iz; u1 = -P[1] * f * iz;
    rsq = (P[0] * P[0] + P[1] * P[1]) * iz * iz;
    g = 1.0 + k1 * rsq + k2 * rsq * rsq;
    dg = k1 + 2.0 * k2 * rsq;
    {
        /* du/dP */
        double du[6] = { -f * iz, 0.0, f * P[0] * iz * iz,
                         0.0, -f * iz, f * P[1] * iz * iz };
        /* drsq/dP */
        double dr[3] = { 2.0 * P[0] * iz * iz, 2.0 * P[1] * iz * iz, -2.0 * rsq * iz };
        for (c = 0; c < 3; c++) {
            D[c]     = g * du[c]     + u0 * dg * dr[c];
            D[3 + c] = g * du[3 + c] + u1 * dg * dr[c];
        }
    }
    /* B = D R ; dx/dc = -D R (explicit centres) or D (translation) */
    for (r = 0; r < 2; r++) for (c = 0; c < 3; c++) {
        double acc = 0.0;
        for (k = 0; k < 3; k++) acc += D[3 * r + k] * R[3 * k + c];
        B[3 * r + c] = acc;
        A[cnp * r + c] = cfg->explicit_centers ? -acc : D[3 * r + c];
    }
    /* dP/dw: rotated vector is v = Rinit*d, P_rot = dR v  (translation, if any, is added after) */
    {
        double Pr[3];   /* dR(w) v */
        for (r = 0; r < 3; r++) Pr[r] = R[3 * r] * d[0] + R[3 * r + 1] * d[1] + R[3 * r + 2] * d[2];
        th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
        if (th == 0.0) {
            /* -[Pr]x */
            Jw[0] = 0.0;    Jw[1] = Pr[2];  Jw[2] = -Pr[1];
            Jw[3] = -Pr[2]; Jw[4] = 0.0;    Jw[5] = Pr[0];
            Jw[6] = Pr[1];  Jw[7] = -Pr[0]; Jw[8] = 0.0;
        } else {
            double n[3] = { w[0] / th, w[1] / th, w[2] / th };
            double sa = sin(th) / th, hs = sin(0.5 * th), sb = 2.0 * hs * hs / th;
            double M[9], Rv[9], v[3], vx[9];
            /* v = Rinit d  =  dR^T Pr */
            for (r = 0; r < 3; r++) v[r] = Rinit[3 * r] * d[0] + Rinit[3 * r + 1] * d[1] + Rinit[3 * r + 2] * d[2];
            M[0] = sa + (1.0 - sa) * n[0] * n[0];
            M[4] = sa + (1.0 - sa) * n[1] * n[1];
            M[8] = sa + (1.0 - sa) * n[2] * n[2];
            M[1] = (1.0 - sa) * n[0] * n[1] + sb * n[2];
            M[3] = (1.0 - sa) * n[0] * n[1] - sb * n[2];
            M[2] = (1.0 - sa) * n[0] * n[2] - sb * n[1];
            M[6] = (1.0 - sa) * n[0] * n[2] + sb * n[1];
            M[5] = (1.0 - sa) * n[1] * n[2] + sb * n[0];
            M[7] = (1.0 - sa) * n[1] * n[2] - sb * n[0];
            vx[0] = 0.0;   vx[1] = -v[2]; vx[2] = v[1];
            vx[3] = v[2];  vx[4] = 0.0;   vx[5] = -v[0];
            vx[6] = -v[1]; vx[7] = v[0];  vx[8] = 0.0;
            /* dR = R Rinit^T ; Jw = -dR [v]x M */
            {
                double dR[9];
                for (r = 0; r < 3; r++) for (c = 0; c < 3; c++) {
                    double acc = 0.0;
                    for (k = 0; k < 3; k++) acc += R[3 * r + k] * Rinit[3 * c + k];
                    dR[3 * r + c] = acc;
                }
                for (r = 0; r < 3; r++) for (c = 0; c < 3; c++) {
                    double acc = 0.0;
                    for (k = 0; k < 3; k++) acc += dR[3 * r + k] * vx[3 * k + c];
                    Rv[3 * r + c] = acc;
                }
            }
            for (r = 0; r < 3; r++) for (c = 0; c < 3; c++) {
                double acc = 0.0;
                for (k = 0; k < 3; k++) acc += Rv[3 * r + k] * M[3 * k + c];
                Jw[3 * r + c] = -acc;
            }
        }
    }
    for (r = 0; r < 2; r++) for (c = 0; c < 3; c++) {
        double acc = 0.0;
        for (k = 0; k < 3; k++) acc += D[3 * r + k] * Jw[3 * k + c];
        A[cnp * r + 3 + c] = acc;
    }
    {
        int col = 6;
        if (cfg->est_focal) {
            /* x = g * u, u proportional to f, rsq independent of f: dx/df = x / f ; a6 = f * f_scale */
            A[col] = g * u0 / f / cfg->f_scale;
            A[cnp + col] = g * u1 / f / cfg->f_scale;
            col++;
        }
        if (cfg->undistort) {
            A[col] = u0 * rsq / cfg->k_scale;           A[cnp + col] = u1 * rsq / cfg->k_scale;
            A[col + 1] = u0 * rsq * rsq / cfg->k_scale; A[cnp + col + 1] = u1 * rsq * rsq / cfg->k_scale;
        }
    }
}

/* Forward-difference Jacobian exactly as lib/sba-1.5/sba_levmar_wrap.c:203-256 does it:
 * d = max(|1e-4 * p|, 1e-6); column = (proj(p + d e) - proj(p)) * (1/d). */
static void sm_fd_jacobian(const sm_config *cfg, const double *Rinit, double f_init,
                           const double *a, const double *b, double *A, double *B)
{
    const int cnp = cfg->cnp;
    double aa[9], bb[3], hx[2], hxx[2], d, d1;
    int jj;
    memcpy(aa, a, cnp * sizeof(double));
    memcpy(bb, b, 3 * sizeof(double));
    sm_project(cfg, Rinit, f_init, a, b, hx);
    for (jj = 0; jj < cnp; jj++) {
        d = 1E-04 * aa[jj]; d = fabs(d); if (d < 1E-06) d = 1E-06; d1 = 1.0 / d;
        aa[jj] += d;
        sm_project(cfg, Rinit, f_init, aa, b, hxx);
        aa[jj] = a[jj];
        A[jj] = (hxx[0] - hx[0]) * d1;
        A[cnp + jj] = (hxx[1] - hx[1]) * d1;
    }
    for (jj = 0; jj < 3; jj++) {
        d = 1E-04 * bb[jj]; d = fabs(d); if (d < 1E-06) d = 1E-06; d1 = 1.0 / d;
        bb[jj] += d;
        sm_project(cfg, Rinit, f_init, a, bb, hxx);
        bb[jj] = b[jj];
        B[jj] = (hxx[0] - hx[0]) * d1;
        B[3 + jj] = (hxx[1] - hx[1]) * d1;
    }
}

#endif /* BSFM_ORACLE_SNAVELY_MODEL_H */
