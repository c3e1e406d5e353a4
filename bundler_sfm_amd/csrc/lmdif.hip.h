// lmdif.hip.h -- MINPACK's lmdif (lib/cminpack/lmdif.c, with fdjac2.c, qrfac.c, lmpar.c, qrsolv.c) restated as device code for a
// small, compile-time number of unknowns N, one problem per thread.
//
// MINPACK keeps the m x N Jacobian and Householder-factors it in place; everything it uses afterwards is the N x N triangle
// R, the first N entries of Q^T f and the column norms.  Here those are accumulated while STREAMING over the residual rows
// with Givens rotations (a row-wise QR update, as backward stable as Householder), and MINPACK's pivoted qrfac then runs on
// that N x N triangle -- J P = Q0 (R0 P) = Q0 Q1 R, so pivot order, R, Q^T f and the norms are the ones MINPACK gets, up to
// rounding and the signs of R's rows (which cancel in every quantity lmpar / lmdif form).  Nothing is stored per residual.
// Users: triangulate.hip (N = 3), fmatrix.hip (N = 8).
#pragma once
#include <hip/hip_runtime.h>

namespace bsfm_lm {

constexpr double LM_EPSMCH = 2.22044604926e-16;      // dpmpar(1), lib/cminpack/dpmpar.c
constexpr double LM_DWARF = 2.22507385852e-308;      // dpmpar(2)

// triangular factor and rotated right-hand side of a row stream (upper N x N, q = first N entries of Q^T f)
template <int N> struct QrN {
    double r[N][N];
    double q[N];
    __device__ void clear() { for (int i = 0; i < N; ++i) { q[i] = 0.0; for (int j = 0; j < N; ++j) r[i][j] = 0.0; } }
    // a[] is destroyed
    __device__ void add_row(double* a, double f)
    {
        for (int j = 0; j < N; ++j) {
            if (a[j] == 0.0) continue;
            const double rr = sqrt(r[j][j] * r[j][j] + a[j] * a[j]);
            const double c = r[j][j] / rr, s = a[j] / rr;
            r[j][j] = rr;
            for (int k = j + 1; k < N; ++k) {
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

template <int N> __device__ __forceinline__ double normN(const double* v)
{
    double s = 0.0;
    for (int i = 0; i < N; ++i) s += v[i] * v[i];
    return sqrt(s);
}

// qrfac.c with pivoting on the N x N triangle, then lmdif.c's "form (q transpose)*fvec" loop on the rotated residual.
// Out: a = R (upper, diagonal = rdiag) with the Householder vectors gone, ipvt, acnorm, qtf.
template <int N> __device__ void lm_qrfac(double a[N][N], double qtf[N], int ipvt[N], double acnorm[N])
{
    double rdiag[N], wa[N];
    for (int j = 0; j < N; ++j) {
        double cs = 0.0;
        for (int i = 0; i < N; ++i) cs += a[i][j] * a[i][j];
        acnorm[j] = sqrt(cs);
        rdiag[j] = acnorm[j]; wa[j] = rdiag[j]; ipvt[j] = j;
    }
    for (int j = 0; j < N; ++j) {
        int kmax = j;
        for (int k = j; k < N; ++k) if (rdiag[k] > rdiag[kmax]) kmax = k;
        if (kmax != j) {
            for (int i = 0; i < N; ++i) { const double tmp = a[i][j]; a[i][j] = a[i][kmax]; a[i][kmax] = tmp; }
            rdiag[kmax] = rdiag[j]; wa[kmax] = wa[j];
            const int k = ipvt[j]; ipvt[j] = ipvt[kmax]; ipvt[kmax] = k;
        }
        double ss = 0.0;
        for (int i = j; i < N; ++i) ss += a[i][j] * a[i][j];
        double ajnorm = sqrt(ss);
        if (ajnorm != 0.0) {
            if (a[j][j] < 0.0) ajnorm = -ajnorm;
            for (int i = j; i < N; ++i) a[i][j] /= ajnorm;
            a[j][j] += 1.0;
            for (int k = j + 1; k < N; ++k) {
                double sum = 0.0;
                for (int i = j; i < N; ++i) sum += a[i][j] * a[i][k];
                const double temp = sum / a[j][j];
                for (int i = j; i < N; ++i) a[i][k] -= temp * a[i][j];
                if (rdiag[k] != 0.0) {
                    const double tk = a[j][k] / rdiag[k];
                    rdiag[k] *= sqrt(fmax(0.0, 1.0 - tk * tk));
                    const double rt = rdiag[k] / wa[k];
                    if (0.05 * (rt * rt) <= LM_EPSMCH) {
                        double s2 = 0.0;
                        for (int i = j + 1; i < N; ++i) s2 += a[i][k] * a[i][k];
                        rdiag[k] = sqrt(s2); wa[k] = rdiag[k];
                    }
                }
            }
        }
        rdiag[j] = -ajnorm;
    }
    for (int j = 0; j < N; ++j) {                     // lmdif.c: qtf from the stored Householder vectors
        if (a[j][j] != 0.0) {
            double sum = 0.0;
            for (int i = j; i < N; ++i) sum += a[i][j] * qtf[i];
            const double temp = -sum / a[j][j];
            for (int i = j; i < N; ++i) qtf[i] += a[i][j] * temp;
        }
        a[j][j] = rdiag[j];
    }
}

// qrsolv.c: least squares of [R P^T; D] x ~ [qtb; 0]; strict lower part of r receives S^T, the diagonal is restored
template <int N> __device__ void lm_qrsolv(double r[N][N], const int ipvt[N], const double diag[N], const double qtb[N], double x[N], double sdiag[N])
{
    double wa[N];
    for (int j = 0; j < N; ++j) {
        for (int i = j; i < N; ++i) r[i][j] = r[j][i];
        x[j] = r[j][j]; wa[j] = qtb[j];
    }
    for (int j = 0; j < N; ++j) {
        const int l = ipvt[j];
        if (diag[l] != 0.0) {
            for (int k = j; k < N; ++k) sdiag[k] = 0.0;
            sdiag[j] = diag[l];
            double qtbpj = 0.0;
            for (int k = j; k < N; ++k) {
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
                for (int i = k + 1; i < N; ++i) {
                    const double t2 = c * r[i][k] + s * sdiag[i];
                    sdiag[i] = -s * r[i][k] + c * sdiag[i];
                    r[i][k] = t2;
                }
            }
        }
        sdiag[j] = r[j][j];
        r[j][j] = x[j];
    }
    int nsing = N;
    for (int j = 0; j < N; ++j) {
        if (sdiag[j] == 0.0 && nsing == N) nsing = j;
        if (nsing < N) wa[j] = 0.0;
    }
    for (int j = nsing - 1; j >= 0; --j) {
        double sum = 0.0;
        for (int i = j + 1; i < nsing; ++i) sum += r[i][j] * wa[i];
        wa[j] = (wa[j] - sum) / sdiag[j];
    }
    for (int j = 0; j < N; ++j) x[ipvt[j]] = wa[j];
}

// lmpar.c
template <int N> __device__ void lm_lmpar(double r[N][N], const int ipvt[N], const double diag[N], const double qtb[N], double delta,
                          double& par, double x[N], double sdiag[N])
{
    double wa1[N], wa2[N];
    int nsing = N;
    for (int j = 0; j < N; ++j) {
        wa1[j] = qtb[j];
        if (r[j][j] == 0.0 && nsing == N) nsing = j;
        if (nsing < N) wa1[j] = 0.0;
    }
    for (int j = nsing - 1; j >= 0; --j) {
        wa1[j] /= r[j][j];
        const double temp = wa1[j];
        for (int i = 0; i < j; ++i) wa1[i] -= r[i][j] * temp;
    }
    for (int j = 0; j < N; ++j) x[ipvt[j]] = wa1[j];
    int iter = 0;
    for (int j = 0; j < N; ++j) wa2[j] = diag[j] * x[j];
    double dxnorm = normN<N>(wa2);
    double fp = dxnorm - delta;
    if (fp <= 0.1 * delta) { par = 0.0; return; }
    double parl = 0.0;
    if (nsing >= N) {
        for (int j = 0; j < N; ++j) { const int l = ipvt[j]; wa1[j] = diag[l] * (wa2[l] / dxnorm); }
        for (int j = 0; j < N; ++j) {
            double sum = 0.0;
            for (int i = 0; i < j; ++i) sum += r[i][j] * wa1[i];
            wa1[j] = (wa1[j] - sum) / r[j][j];
        }
        const double temp = normN<N>(wa1);
        parl = fp / delta / temp / temp;
    }
    for (int j = 0; j < N; ++j) {
        double sum = 0.0;
        for (int i = 0; i <= j; ++i) sum += r[i][j] * qtb[i];
        wa1[j] = sum / diag[ipvt[j]];
    }
    const double gnorm = normN<N>(wa1);
    double paru = gnorm / delta;
    if (paru == 0.0) paru = LM_DWARF / fmin(delta, 0.1);
    par = fmax(par, parl);
    par = fmin(par, paru);
    if (par == 0.0) par = gnorm / dxnorm;
    for (;;) {
        ++iter;
        if (par == 0.0) par = fmax(LM_DWARF, 0.001 * paru);
        double temp = sqrt(par);
        for (int j = 0; j < N; ++j) wa1[j] = temp * diag[j];
        lm_qrsolv<N>(r, ipvt, wa1, qtb, x, sdiag);
        for (int j = 0; j < N; ++j) wa2[j] = diag[j] * x[j];
        dxnorm = normN<N>(wa2);
        temp = fp;
        fp = dxnorm - delta;
        if (fabs(fp) <= 0.1 * delta || (parl == 0.0 && fp <= temp && temp < 0.0) || iter == 10) break;
        for (int j = 0; j < N; ++j) { const int l = ipvt[j]; wa1[j] = diag[l] * (wa2[l] / dxnorm); }
        for (int j = 0; j < N; ++j) {
            wa1[j] /= sdiag[j];
            const double t2 = wa1[j];
            for (int i = j + 1; i < N; ++i) wa1[i] -= r[i][j] * t2;
        }
        temp = normN<N>(wa1);
        const double parc = fp / delta / temp / temp;
        if (fp > 0.0) parl = fmax(parl, par);
        if (fp < 0.0) paru = fmin(paru, par);
        par = fmax(parl, par + parc);
    }
}

// lmdif.c for N unknowns with lmdif1.c's / lmdif_driver2's settings (ftol = xtol = tol, gtol = 0, maxfev = 200 (N + 1),
// epsfcn = 0, mode 1, factor 100); x is refined in place, returns MINPACK's info.  Fcn supplies
//   int rows()                                   number of residuals m
//   double fnorm(const double* x)                enorm(fvec(x))
//   void jac_qr(const double* x, QrN<N>& F)      fdjac2's forward-difference Jacobian, streamed row by row into F together
//                                                with fvec(x) (steps h_j = sqrt(epsmch) |x_j|, or sqrt(epsmch) when x_j = 0)
template <int N, class Fcn> __device__ int lm_lmdif(Fcn& fcn, double* x, double tol)
{
    const double ftol = tol, xtol = tol, gtol = 0.0, factor = 100.0;
    const int m = fcn.rows(), maxfev = 200 * (N + 1);
    if (m < N) return 0;                              // lmdif_driver: "lmdif called with n > m", x untouched
    int info = 0, nfev = 1, iter = 1;
    double fnorm = fcn.fnorm(x);
    double par = 0.0, delta = 0.0, xnorm = 0.0;
    double diag[N];
    for (;;) {
        QrN<N> F;
        fcn.jac_qr(x, F);
        nfev += N;
        double qtf[N];
        for (int j = 0; j < N; ++j) qtf[j] = F.q[j];
        int ipvt[N];
        double acn[N];
        lm_qrfac<N>(F.r, qtf, ipvt, acn);
        if (iter == 1) {
            double w3[N];
            for (int j = 0; j < N; ++j) { diag[j] = acn[j]; if (acn[j] == 0.0) diag[j] = 1.0; w3[j] = diag[j] * x[j]; }
            xnorm = normN<N>(w3);
            delta = factor * xnorm;
            if (delta == 0.0) delta = factor;
        }
        double gnorm = 0.0;
        if (fnorm != 0.0) {
            for (int j = 0; j < N; ++j) {
                const int l = ipvt[j];
                if (acn[l] == 0.0) continue;
                double sum = 0.0;
                for (int i = 0; i <= j; ++i) sum += F.r[i][j] * (qtf[i] / fnorm);
                gnorm = fmax(gnorm, fabs(sum / acn[l]));
            }
        }
        if (gnorm <= gtol) { info = 4; break; }
        for (int j = 0; j < N; ++j) diag[j] = fmax(diag[j], acn[j]);
        double ratio = 0.0;
        do {
            double wa1[N], wa2[N], wa3[N], sdiag[N];
            lm_lmpar<N>(F.r, ipvt, diag, qtf, delta, par, wa1, sdiag);
            for (int j = 0; j < N; ++j) { wa1[j] = -wa1[j]; wa2[j] = x[j] + wa1[j]; wa3[j] = diag[j] * wa1[j]; }
            const double pnorm = normN<N>(wa3);
            if (iter == 1) delta = fmin(delta, pnorm);
            const double fnorm1 = fcn.fnorm(wa2);
            ++nfev;
            double actred = -1.0;
            if (0.1 * fnorm1 < fnorm) { const double q = fnorm1 / fnorm; actred = 1.0 - q * q; }
            for (int j = 0; j < N; ++j) {
                wa3[j] = 0.0;
                const double temp = wa1[ipvt[j]];
                for (int i = 0; i <= j; ++i) wa3[i] += F.r[i][j] * temp;
            }
            const double temp1 = normN<N>(wa3) / fnorm;
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
                for (int j = 0; j < N; ++j) { x[j] = wa2[j]; wa2[j] = diag[j] * x[j]; }
                xnorm = normN<N>(wa2);
                fnorm = fnorm1;
                ++iter;
            }
            const bool small = fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0;
            if (small) info = 1;
            if (delta <= xtol * xnorm) info = 2;
            if (small && info == 2) info = 3;
            if (info != 0) break;
            if (nfev >= maxfev) info = 5;
            if (fabs(actred) <= LM_EPSMCH && prered <= LM_EPSMCH && 0.5 * ratio <= 1.0) info = 6;
            if (delta <= LM_EPSMCH * xnorm) info = 7;
            if (gnorm <= LM_EPSMCH) info = 8;
            if (info != 0) break;
        } while (ratio < 1.0e-4);
        if (info != 0) break;
    }
    return info;
}

}  // namespace bsfm_lm
