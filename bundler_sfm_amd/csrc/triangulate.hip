// triangulate.hip -- batched multi-view triangulation, gfx950.
//
// Replaces, over a ragged batch of independent points, the reference's
//   triangulate_n          lib/imagelib/triangulate.c:181-272   linear least squares (dgelsy) + lmdif polish, tol 1e-5
//   triangulate_n_refine   lib/imagelib/triangulate.c:133-178   lmdif polish from a given point
//   triangulate            lib/imagelib/triangulate.c:281-338   two views, tol 1e-10, error = sum of squares
// which BundlerApp::TriangulateNViews (src/BundleAdd.cpp:47-123), Bundle.cpp:2240,2758 and BundleFast.cpp:193 call one
// point at a time.  The polish is MINPACK's lmdif (lib/cminpack/lmdif.c through lmdif1.c: ftol = xtol = tol, gtol = 0,
// maxfev = 200 (n+1), epsfcn = 0, mode 1, factor 100) with its forward-difference Jacobian (fdjac2.c), pivoted QR
// (qrfac.c), Levenberg-Marquardt parameter (lmpar.c) and qrsolv.c -- restated here for n = 3 unknowns.
//
// MI355X design: one thread per point, nothing stored per observation.  MINPACK keeps the 2d x 3 Jacobian and
// Householder-factors it in place; with n = 3 everything it uses afterwards is the 3 x 3 triangle R, the first three
// entries of Q^T f and the column norms.  Those are accumulated while STREAMING over the point's views with Givens
// rotations (a row-wise QR update, as backward stable as Householder), and MINPACK's pivoted qrfac then runs on that
// 3 x 3 triangle -- J P = Q0 (R0 P) = Q0 Q1 R, so pivot order, R, Q^T f and the norms are the ones MINPACK gets, up to
// rounding and the signs of R's rows (which cancel in every quantity lmpar / lmdif form).  The linear start of
// triangulate_n is the same streaming QR on the rows [A | b] (dgelsy with rcond = -1 never truncates the rank).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../include/bsfm.h"

namespace {

constexpr double TRI_EPSMCH = 2.22044604926e-16;     // dpmpar(1), lib/cminpack/dpmpar.c
constexpr double TRI_DWARF = 2.22507385852e-308;     // dpmpar(2)

struct TriViews {
    const double* p;        // 2 per view
    const double* R;        // 9 per camera (or per view when cam == nullptr)
    const double* t;        // 3 per camera (or per view)
    const int* cam;         // view -> camera, or nullptr
    int v0, v1;             // this point's views
};

// project() of triangulate.c:81-95: rigid transform, perspective division
__device__ __forceinline__ void tri_project(const TriViews& V, int v, double x0, double x1, double x2, double& px, double& py)
{
#pragma clang fp contract(off)
    const size_t c = V.cam ? (size_t)V.cam[v] : (size_t)v;
    const double* R = V.R + 9 * c;
    const double* t = V.t + 3 * c;
    const double a = R[0] * x0 + R[1] * x1 + R[2] * x2 + t[0];
    const double b = R[3] * x0 + R[4] * x1 + R[5] * x2 + t[1];
    const double d = R[6] * x0 + R[7] * x1 + R[8] * x2 + t[2];
    px = a / d; py = b / d;
}

// triangular factor and rotated right-hand side of a row stream (upper 3 x 3, q = first three entries of Q^T f)
struct Qr3 {
    double r[3][3];
    double q[3];
    __device__ void clear() { for (int i = 0; i < 3; ++i) { q[i] = 0.0; for (int j = 0; j < 3; ++j) r[i][j] = 0.0; } }
    __device__ void add_row(double a0, double a1, double a2, double f)
    {
        double a[3] = { a0, a1, a2 };
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (a[j] == 0.0) continue;
            const double rr = sqrt(r[j][j] * r[j][j] + a[j] * a[j]);
            const double c = r[j][j] / rr, s = a[j] / rr;
            r[j][j] = rr;
#pragma unroll
            for (int k = j + 1; k < 3; ++k) {
                const double tmp = c * r[j][k] + s * a[k];
                a[k] = -s * r[j][k] + c * a[k];
                r[j][k] = tmp;
            }
            const double tq = c * q[j] + s * f;
            f = -s * q[j] + c * f;
            q[j] = tq;
        }
    }
};

__device__ __forceinline__ double norm3(const double* v) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

// fvec = observed - projected (triangulate_n_residual, triangulate.c:115-129); returns enorm(fvec)
__device__ double tri_fnorm(const TriViews& V, const double* x)
{
    double s = 0.0;
    for (int v = V.v0; v < V.v1; ++v) {
        double px, py;
        tri_project(V, v, x[0], x[1], x[2], px, py);
        const double fx = V.p[2 * (size_t)v] - px, fy = V.p[2 * (size_t)v + 1] - py;
        s += fx * fx; s += fy * fy;
    }
    return sqrt(s);
}

// fdjac2 + the Jacobian's triangular factor in one pass over the views
__device__ void tri_jacobian_qr(const TriViews& V, const double* x, Qr3& F)
{
    const double eps = sqrt(TRI_EPSMCH);
    double h[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { h[j] = eps * fabs(x[j]); if (h[j] == 0.0) h[j] = eps; }
    const double xp0 = x[0] + h[0], xp1 = x[1] + h[1], xp2 = x[2] + h[2];
    F.clear();
    for (int v = V.v0; v < V.v1; ++v) {
        double px, py, ax, ay, bx, by, cx, cy;
        tri_project(V, v, x[0], x[1], x[2], px, py);
        tri_project(V, v, xp0, x[1], x[2], ax, ay);
        tri_project(V, v, x[0], xp1, x[2], bx, by);
        tri_project(V, v, x[0], x[1], xp2, cx, cy);
        const double ox = V.p[2 * (size_t)v], oy = V.p[2 * (size_t)v + 1];
        const double fx = ox - px, fy = oy - py;
        F.add_row(((ox - ax) - fx) / h[0], ((ox - bx) - fx) / h[1], ((ox - cx) - fx) / h[2], fx);
        F.add_row(((oy - ay) - fy) / h[0], ((oy - by) - fy) / h[1], ((oy - cy) - fy) / h[2], fy);
    }
}

// qrfac.c with pivoting on the 3 x 3 triangle, then lmdif.c's "form (q transpose)*fvec" loop on the rotated residual.
// Out: a = R (upper, diagonal = rdiag) with the Householder vectors gone, ipvt, acnorm, qtf.
__device__ void tri_qrfac(double a[3][3], double qtf[3], int ipvt[3], double acnorm[3])
{
    double rdiag[3], wa[3];
    for (int j = 0; j < 3; ++j) {
        acnorm[j] = sqrt(a[0][j] * a[0][j] + a[1][j] * a[1][j] + a[2][j] * a[2][j]);
        rdiag[j] = acnorm[j]; wa[j] = rdiag[j]; ipvt[j] = j;
    }
    for (int j = 0; j < 3; ++j) {
        int kmax = j;
        for (int k = j; k < 3; ++k) if (rdiag[k] > rdiag[kmax]) kmax = k;
        if (kmax != j) {
            for (int i = 0; i < 3; ++i) { const double tmp = a[i][j]; a[i][j] = a[i][kmax]; a[i][kmax] = tmp; }
            rdiag[kmax] = rdiag[j]; wa[kmax] = wa[j];
            const int k = ipvt[j]; ipvt[j] = ipvt[kmax]; ipvt[kmax] = k;
        }
        double ss = 0.0;
        for (int i = j; i < 3; ++i) ss += a[i][j] * a[i][j];
        double ajnorm = sqrt(ss);
        if (ajnorm != 0.0) {
            if (a[j][j] < 0.0) ajnorm = -ajnorm;
            for (int i = j; i < 3; ++i) a[i][j] /= ajnorm;
            a[j][j] += 1.0;
            for (int k = j + 1; k < 3; ++k) {
                double sum = 0.0;
                for (int i = j; i < 3; ++i) sum += a[i][j] * a[i][k];
                const double temp = sum / a[j][j];
                for (int i = j; i < 3; ++i) a[i][k] -= temp * a[i][j];
                if (rdiag[k] != 0.0) {
                    const double tk = a[j][k] / rdiag[k];
                    rdiag[k] *= sqrt(fmax(0.0, 1.0 - tk * tk));
                    const double rt = rdiag[k] / wa[k];
                    if (0.05 * (rt * rt) <= TRI_EPSMCH) {
                        double s2 = 0.0;
                        for (int i = j + 1; i < 3; ++i) s2 += a[i][k] * a[i][k];
                        rdiag[k] = sqrt(s2); wa[k] = rdiag[k];
                    }
                }
            }
        }
        rdiag[j] = -ajnorm;
    }
    for (int j = 0; j < 3; ++j) {                     // lmdif.c: qtf from the stored Householder vectors
        if (a[j][j] != 0.0) {
            double sum = 0.0;
            for (int i = j; i < 3; ++i) sum += a[i][j] * qtf[i];
            const double temp = -sum / a[j][j];
            for (int i = j; i < 3; ++i) qtf[i] += a[i][j] * temp;
        }
        a[j][j] = rdiag[j];
    }
}

// qrsolv.c: least squares of [R P^T; D] x ~ [qtb; 0]; strict lower part of r receives S^T, the diagonal is restored
__device__ void tri_qrsolv(double r[3][3], const int ipvt[3], const double diag[3], const double qtb[3], double x[3], double sdiag[3])
{
    double wa[3];
    for (int j = 0; j < 3; ++j) {
        for (int i = j; i < 3; ++i) r[i][j] = r[j][i];
        x[j] = r[j][j]; wa[j] = qtb[j];
    }
    for (int j = 0; j < 3; ++j) {
        const int l = ipvt[j];
        if (diag[l] != 0.0) {
            for (int k = j; k < 3; ++k) sdiag[k] = 0.0;
            sdiag[j] = diag[l];
            double qtbpj = 0.0;
            for (int k = j; k < 3; ++k) {
                if (sdiag[k] == 0.0) continue;
                double c, s;
                if (fabs(r[k][k]) >= fabs(sdiag[k])) {
                    const double tn = sdiag[k] / r[k][k];
                    c = 0.5 / sqrt(0.25 + 0.25 * (tn * tn)); s = c * tn;
                } else {
                    const double ct = r[k][k] / sdiag[k];
                    s = 0.5 / sqrt(0.25 + 0.25 * (ct * ct)); c = s * ct;
                }
                r[k][k] = c * r[k][k] + s * sdiag[k];
                const double temp = c * wa[k] + s * qtbpj;
                qtbpj = -s * wa[k] + c * qtbpj;
                wa[k] = temp;
                for (int i = k + 1; i < 3; ++i) {
                    const double t2 = c * r[i][k] + s * sdiag[i];
                    sdiag[i] = -s * r[i][k] + c * sdiag[i];
                    r[i][k] = t2;
                }
            }
        }
        sdiag[j] = r[j][j];
        r[j][j] = x[j];
    }
    int nsing = 3;
    for (int j = 0; j < 3; ++j) {
        if (sdiag[j] == 0.0 && nsing == 3) nsing = j;
        if (nsing < 3) wa[j] = 0.0;
    }
    for (int j = nsing - 1; j >= 0; --j) {
        double sum = 0.0;
        for (int i = j + 1; i < nsing; ++i) sum += r[i][j] * wa[i];
        wa[j] = (wa[j] - sum) / sdiag[j];
    }
    for (int j = 0; j < 3; ++j) x[ipvt[j]] = wa[j];
}

// lmpar.c
__device__ void tri_lmpar(double r[3][3], const int ipvt[3], const double diag[3], const double qtb[3], double delta,
                          double& par, double x[3], double sdiag[3])
{
    double wa1[3], wa2[3];
    int nsing = 3;
    for (int j = 0; j < 3; ++j) {
        wa1[j] = qtb[j];
        if (r[j][j] == 0.0 && nsing == 3) nsing = j;
        if (nsing < 3) wa1[j] = 0.0;
    }
    for (int j = nsing - 1; j >= 0; --j) {
        wa1[j] /= r[j][j];
        const double temp = wa1[j];
        for (int i = 0; i < j; ++i) wa1[i] -= r[i][j] * temp;
    }
    for (int j = 0; j < 3; ++j) x[ipvt[j]] = wa1[j];
    int iter = 0;
    for (int j = 0; j < 3; ++j) wa2[j] = diag[j] * x[j];
    double dxnorm = norm3(wa2);
    double fp = dxnorm - delta;
    if (fp <= 0.1 * delta) { par = 0.0; return; }
    double parl = 0.0;
    if (nsing >= 3) {
        for (int j = 0; j < 3; ++j) { const int l = ipvt[j]; wa1[j] = diag[l] * (wa2[l] / dxnorm); }
        for (int j = 0; j < 3; ++j) {
            double sum = 0.0;
            for (int i = 0; i < j; ++i) sum += r[i][j] * wa1[i];
            wa1[j] = (wa1[j] - sum) / r[j][j];
        }
        const double temp = norm3(wa1);
        parl = fp / delta / temp / temp;
    }
    for (int j = 0; j < 3; ++j) {
        double sum = 0.0;
        for (int i = 0; i <= j; ++i) sum += r[i][j] * qtb[i];
        wa1[j] = sum / diag[ipvt[j]];
    }
    const double gnorm = norm3(wa1);
    double paru = gnorm / delta;
    if (paru == 0.0) paru = TRI_DWARF / fmin(delta, 0.1);
    par = fmax(par, parl);
    par = fmin(par, paru);
    if (par == 0.0) par = gnorm / dxnorm;
    for (;;) {
        ++iter;
        if (par == 0.0) par = fmax(TRI_DWARF, 0.001 * paru);
        double temp = sqrt(par);
        for (int j = 0; j < 3; ++j) wa1[j] = temp * diag[j];
        tri_qrsolv(r, ipvt, wa1, qtb, x, sdiag);
        for (int j = 0; j < 3; ++j) wa2[j] = diag[j] * x[j];
        dxnorm = norm3(wa2);
        temp = fp;
        fp = dxnorm - delta;
        if (fabs(fp) <= 0.1 * delta || (parl == 0.0 && fp <= temp && temp < 0.0) || iter == 10) break;
        for (int j = 0; j < 3; ++j) { const int l = ipvt[j]; wa1[j] = diag[l] * (wa2[l] / dxnorm); }
        for (int j = 0; j < 3; ++j) {
            wa1[j] /= sdiag[j];
            const double t2 = wa1[j];
            for (int i = j + 1; i < 3; ++i) wa1[i] -= r[i][j] * t2;
        }
        temp = norm3(wa1);
        const double parc = fp / delta / temp / temp;
        if (fp > 0.0) parl = fmax(parl, par);
        if (fp < 0.0) paru = fmin(paru, par);
        par = fmax(parl, par + parc);
    }
}

// lmdif.c for n = 3 (through lmdif1.c's settings); x is refined in place, returns MINPACK's info
__device__ int tri_lmdif(const TriViews& V, double* x, double tol)
{
    const double ftol = tol, xtol = tol, gtol = 0.0, factor = 100.0;
    const int m = 2 * (V.v1 - V.v0), maxfev = 200 * 4;
    if (m < 3) return 0;                              // lmdif_driver: "lmdif called with n > m", x untouched
    int info = 0, nfev = 1, iter = 1;
    double fnorm = tri_fnorm(V, x);
    double par = 0.0, delta = 0.0, xnorm = 0.0;
    double diag[3];
    for (;;) {
        Qr3 F;
        tri_jacobian_qr(V, x, F);
        nfev += 3;
        double qtf[3] = { F.q[0], F.q[1], F.q[2] };
        int ipvt[3];
        double acn[3];
        tri_qrfac(F.r, qtf, ipvt, acn);
        if (iter == 1) {
            double w3[3];
            for (int j = 0; j < 3; ++j) { diag[j] = acn[j]; if (acn[j] == 0.0) diag[j] = 1.0; w3[j] = diag[j] * x[j]; }
            xnorm = norm3(w3);
            delta = factor * xnorm;
            if (delta == 0.0) delta = factor;
        }
        double gnorm = 0.0;
        if (fnorm != 0.0) {
            for (int j = 0; j < 3; ++j) {
                const int l = ipvt[j];
                if (acn[l] == 0.0) continue;
                double sum = 0.0;
                for (int i = 0; i <= j; ++i) sum += F.r[i][j] * (qtf[i] / fnorm);
                gnorm = fmax(gnorm, fabs(sum / acn[l]));
            }
        }
        if (gnorm <= gtol) { info = 4; break; }
        for (int j = 0; j < 3; ++j) diag[j] = fmax(diag[j], acn[j]);
        double ratio = 0.0;
        do {
            double wa1[3], wa2[3], wa3[3], sdiag[3];
            tri_lmpar(F.r, ipvt, diag, qtf, delta, par, wa1, sdiag);
            for (int j = 0; j < 3; ++j) { wa1[j] = -wa1[j]; wa2[j] = x[j] + wa1[j]; wa3[j] = diag[j] * wa1[j]; }
            const double pnorm = norm3(wa3);
            if (iter == 1) delta = fmin(delta, pnorm);
            const double fnorm1 = tri_fnorm(V, wa2);
            ++nfev;
            double actred = -1.0;
            if (0.1 * fnorm1 < fnorm) { const double q = fnorm1 / fnorm; actred = 1.0 - q * q; }
            for (int j = 0; j < 3; ++j) {
                wa3[j] = 0.0;
                const double temp = wa1[ipvt[j]];
                for (int i = 0; i <= j; ++i) wa3[i] += F.r[i][j] * temp;
            }
            const double temp1 = norm3(wa3) / fnorm;
            const double temp2 = sqrt(par) * pnorm / fnorm;
            const double prered = temp1 * temp1 + temp2 * temp2 / 0.5;
            const double dirder = -(temp1 * temp1 + temp2 * temp2);
            ratio = 0.0;
            if (prered != 0.0) ratio = actred / prered;
            if (ratio <= 0.25) {
                double temp = 0.5;
                if (actred < 0.0) temp = 0.5 * dirder / (dirder + 0.5 * actred);
                if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
                delta = temp * fmin(delta, pnorm / 0.1);
                par /= temp;
            } else if (par == 0.0 || ratio >= 0.75) {
                delta = pnorm / 0.5;
                par = 0.5 * par;
            }
            if (ratio >= 1.0e-4) {
                for (int j = 0; j < 3; ++j) { x[j] = wa2[j]; wa2[j] = diag[j] * x[j]; }
                xnorm = norm3(wa2);
                fnorm = fnorm1;
                ++iter;
            }
            const bool small = fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0;
            if (small) info = 1;
            if (delta <= xtol * xnorm) info = 2;
            if (small && info == 2) info = 3;
            if (info != 0) break;
            if (nfev >= maxfev) info = 5;
            if (fabs(actred) <= TRI_EPSMCH && prered <= TRI_EPSMCH && 0.5 * ratio <= 1.0) info = 6;
            if (delta <= TRI_EPSMCH * xnorm) info = 7;
            if (gnorm <= TRI_EPSMCH) info = 8;
            if (info != 0) break;
        } while (ratio < 1.0e-4);
        if (info != 0) break;
    }
    return info == 8 ? 4 : info;                      // lmdif1.c
}

__global__ __launch_bounds__(64) void k_triangulate(int mode, int npoints, const int* __restrict__ view_ptr,
        const double* __restrict__ p, const double* __restrict__ R, const double* __restrict__ t, const int* __restrict__ cam,
        double* __restrict__ X, double* __restrict__ err, int* __restrict__ info_out)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= npoints) return;
    TriViews V { p, R, t, cam, view_ptr[i], view_ptr[i + 1] };
    const int nv = V.v1 - V.v0;
    double x[3] = { X[3 * (size_t)i], X[3 * (size_t)i + 1], X[3 * (size_t)i + 2] };
    if (mode != BSFM_TRI_N_REFINE) {
        // linear start (triangulate.c:196-215): rows R_0 - u R_2, R_1 - v R_2 against t_2 u - t_0, t_2 v - t_1
#pragma clang fp contract(off)
        Qr3 F; F.clear();
        for (int v = V.v0; v < V.v1; ++v) {
            const size_t c = cam ? (size_t)cam[v] : (size_t)v;
            const double* Rv = R + 9 * c; const double* tv = t + 3 * c;
            const double u = p[2 * (size_t)v], w = p[2 * (size_t)v + 1];
            F.add_row(Rv[0] - u * Rv[6], Rv[1] - u * Rv[7], Rv[2] - u * Rv[8], tv[2] * u - tv[0]);
            F.add_row(Rv[3] - w * Rv[6], Rv[4] - w * Rv[7], Rv[5] - w * Rv[8], tv[2] * w - tv[1]);
        }
        x[2] = F.q[2] / F.r[2][2];
        x[1] = (F.q[1] - F.r[1][2] * x[2]) / F.r[1][1];
        x[0] = (F.q[0] - F.r[0][1] * x[1] - F.r[0][2] * x[2]) / F.r[0][0];
    }
    const int info = tri_lmdif(V, x, mode == BSFM_TRI_PAIR ? 1.0e-10 : 1.0e-5);
    X[3 * (size_t)i] = x[0]; X[3 * (size_t)i + 1] = x[1]; X[3 * (size_t)i + 2] = x[2];
    if (info_out) info_out[i] = info;
    if (err) {
#pragma clang fp contract(off)
        double e = 0.0;
        for (int v = V.v0; v < V.v1; ++v) {
            double px, py;
            tri_project(V, v, x[0], x[1], x[2], px, py);
            const double dx = px - p[2 * (size_t)v], dy = py - p[2 * (size_t)v + 1];
            e += dx * dx + dy * dy;
        }
        err[i] = mode == BSFM_TRI_PAIR ? e : sqrt(e / nv);
    }
}

template <typename T> struct DevBuf {
    T* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    bool alloc(size_t n) { return hipMalloc(&p, (n ? n : 1) * sizeof(T)) == hipSuccess; }
    bool up(const T* src, size_t n) { return n == 0 || hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice) == hipSuccess; }
    bool down(T* dst, size_t n) { return n == 0 || hipMemcpy(dst, p, n * sizeof(T), hipMemcpyDeviceToHost) == hipSuccess; }
};

}  // namespace

extern "C" int bsfm_triangulate_batch(int mode, int npoints, const int* view_ptr, const double* p, const int* view_cam,
                                      int ncams, const double* R, const double* t, double* X, double* error, int* info)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fprintf(stderr, "[bsfm] FATAL: no usable HIP device; the MI355X path has no CPU fallback\n");
        return BSFM_ERROR;
    }
    if (mode < BSFM_TRI_N || mode > BSFM_TRI_PAIR || npoints < 0 || !view_ptr || (npoints > 0 && (!p || !R || !t || !X))) {
        fprintf(stderr, "[bsfm] triangulate: bad arguments\n");
        return BSFM_ERROR;
    }
    if (npoints == 0) return 0;
    const int nviews = view_ptr[npoints];
    for (int i = 0; i < npoints; ++i) {
        const int d = view_ptr[i + 1] - view_ptr[i];
        if (d < 0 || (mode == BSFM_TRI_PAIR && d != 2) || (mode != BSFM_TRI_PAIR && d < 2)) {
            // dgelsy_driver / lmdif_driver refuse fewer equations than unknowns (lib/matrix/matrix.c:463-466,790-793)
            fprintf(stderr, "[bsfm] triangulate: point %d has %d views (need %s)\n", i, d, mode == BSFM_TRI_PAIR ? "exactly 2" : ">= 2");
            return BSFM_ERROR;
        }
    }
    const size_t nrt = view_cam ? (size_t)ncams : (size_t)nviews;
    if (view_cam) for (int v = 0; v < nviews; ++v) if (view_cam[v] < 0 || view_cam[v] >= ncams) { fprintf(stderr, "[bsfm] triangulate: camera index out of range\n"); return BSFM_ERROR; }
    DevBuf<int> dptr, dcam, dinfo; DevBuf<double> dp, dR, dt, dX, derr;
    bool ok = dptr.alloc(npoints + 1) && dp.alloc(2 * (size_t)nviews) && dR.alloc(9 * nrt) && dt.alloc(3 * nrt) &&
              dX.alloc(3 * (size_t)npoints) && derr.alloc(npoints) && dinfo.alloc(npoints) && (!view_cam || dcam.alloc(nviews));
    ok = ok && dptr.up(view_ptr, npoints + 1) && dp.up(p, 2 * (size_t)nviews) && dR.up(R, 9 * nrt) && dt.up(t, 3 * nrt) &&
         dX.up(X, 3 * (size_t)npoints) && (!view_cam || dcam.up(view_cam, nviews));
    if (!ok) { fprintf(stderr, "[bsfm] triangulate: device allocation / upload failed\n"); return BSFM_ERROR; }
    hipLaunchKernelGGL(k_triangulate, dim3((npoints + 63) / 64), dim3(64), 0, 0, mode, npoints, dptr.p, dp.p, dR.p, dt.p,
                       view_cam ? dcam.p : (const int*)nullptr, dX.p, derr.p, dinfo.p);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "[bsfm] triangulate: kernel failed\n"); return BSFM_ERROR; }
    ok = dX.down(X, 3 * (size_t)npoints) && (!error || derr.down(error, npoints)) && (!info || dinfo.down(info, npoints));
    return ok ? 0 : BSFM_ERROR;
}
