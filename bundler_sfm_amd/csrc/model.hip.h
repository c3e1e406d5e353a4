// model.hip.h -- device-side Snavely camera model for gfx950 (wave64, FP64 VALU).
//
// Restates, per observation, what the reference evaluates through host callbacks:
//   rot_update            lib/sfm-driver/sfm.c:77-116
//   sfm_project_point3    lib/sfm-driver/sfm.c:503-552   (a[6]/f_scale, k/k_scale)
//   sfm_project_rd        lib/sfm-driver/sfm.c:302-380   (P = R(b-c), p = -f P.xy/P.z, radial factor)
//   forward differences   lib/sba-1.5/sba_levmar_wrap.c:203-256
// Same model as include/snavely_reprojection_error.h:57-92.
//
// MI355X design: everything that depends only on the camera (Rodrigues update, the derivative factor
// Q = dR*M of the exponential map, the three w-perturbed rotations and the nine finite-difference steps)
// is computed ONCE per camera per parameter vector by k_cam_table (m threads) into a 72-double row, so
// that the per-observation kernels do no transcendental math at all: they gather one 576-byte row that
// stays resident in L2 (1000 cameras = 576 KB) and stream the observation records coalesced.
#pragma once
#include <hip/hip_runtime.h>

namespace bsfm {

struct ModelCfg {
    int cnp;               // 6..9
    int est_focal;
    int undistort;
    int explicit_centers;
    double f_scale, k_scale;
};

// camera-table row layout (doubles)
constexpr int CT_R = 0;      // 9  current rotation R = dR(w) Rinit
constexpr int CT_Q = 9;      // 9  Q = dR(w) M(w)   (dP/dw = -[P_r]x Q)
constexpr int CT_A = 18;     // 9  packed parameters a_j (cnp used)
constexpr int CT_F = 27;     // focal actually used (a6/f_scale or f_init)
constexpr int CT_K1 = 28;
constexpr int CT_K2 = 29;
constexpr int CT_D = 32;     // 9  FD steps d = max(|1e-4 a|, 1e-6)
constexpr int CT_RP = 41;    // 27 rotations for w + d e_k, k = 0..2
constexpr int CT_KN = 72;    // known-intrinsics flag (0/1), then k_known[5] (CT_KN+1..5) and K_known[0,1,2,4,5] (CT_KN+6..10);
                             // CT_KN+11: 1 when the problem runs the fisheye projection (sfm.c:448-492), CT_KN+12: this camera's
                             // fisheye flag, CT_KN+13..17: f_cx, f_cy, f_rad, f_angle, f_focal
constexpr int CT_EXT = 18;   // doubles in the extended-model block at CT_KN
constexpr int CT_STRIDE = 90;

__device__ __forceinline__ void rot_update(const double* __restrict__ Rinit, double w0, double w1, double w2,
                                           double* __restrict__ R)
{
#pragma clang fp contract(off)
    const double th = sqrt(w0 * w0 + w1 * w1 + w2 * w2);
    if (th == 0.0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = Rinit[k];
        return;
    }
    const double n0 = w0 / th, n1 = w1 / th, n2 = w2 / th;
    const double nx[9] = { 0.0, -n2, n1, n2, 0.0, -n0, -n1, n0, 0.0 };
    double nxsq[9], dR[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            nxsq[3 * r + c] = nx[3 * r] * nx[c] + nx[3 * r + 1] * nx[3 + c] + nx[3 * r + 2] * nx[6 + c];
    const double s = sin(th), c1 = 1.0 - cos(th);
#pragma unroll
    for (int k = 0; k < 9; ++k) dR[k] = ((k % 4 == 0) ? 1.0 : 0.0) + nx[k] * s + nxsq[k] * c1;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            R[3 * r + c] = dR[3 * r] * Rinit[c] + dR[3 * r + 1] * Rinit[3 + c] + dR[3 * r + 2] * Rinit[6 + c];
}

// Q = dR(w) * M(w), M = a I + (1-a) n n^T - b [n]x, a = sin(th)/th, b = (1-cos th)/th; Q = I at th == 0.
__device__ __forceinline__ void rot_deriv_factor(const double* __restrict__ Rinit, const double* __restrict__ R,
                                                 double w0, double w1, double w2, double* __restrict__ Q)
{
    const double th = sqrt(w0 * w0 + w1 * w1 + w2 * w2);
    if (th == 0.0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) Q[k] = (k % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    const double n0 = w0 / th, n1 = w1 / th, n2 = w2 / th;
    const double sa = sin(th) / th, hs = sin(0.5 * th), sb = 2.0 * hs * hs / th, ia = 1.0 - sa;
    const double M[9] = { sa + ia * n0 * n0, ia * n0 * n1 + sb * n2, ia * n0 * n2 - sb * n1,
                          ia * n0 * n1 - sb * n2, sa + ia * n1 * n1, ia * n1 * n2 + sb * n0,
                          ia * n0 * n2 + sb * n1, ia * n1 * n2 - sb * n0, sa + ia * n2 * n2 };
    double dR[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)   // dR = R Rinit^T
            dR[3 * r + c] = R[3 * r] * Rinit[3 * c] + R[3 * r + 1] * Rinit[3 * c + 1] + R[3 * r + 2] * Rinit[3 * c + 2];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            Q[3 * r + c] = dR[3 * r] * M[c] + dR[3 * r + 1] * M[3 + c] + dR[3 * r + 2] * M[6 + c];
}

// Projection with an explicit rotation / centre / focal / distortion (operation order of sfm.c:316-377).
// kn = camera-table slot CT_KN: cameras with known intrinsics go through the reference's 5-parameter Brown model and
// their own K before the (optional) radial term (sfm_project_rd, lib/sfm-driver/sfm.c:339-358).  In fisheye mode
// (kn[11] != 0; sfm_project_point2_fisheye, sfm.c:448-492) the radial term is NOT applied (sfm_project2, sfm.c:231-299) and
// cameras flagged fisheye map the pinhole point through the equidistant model (sfm_fisheye_distort, sfm.c:426-446).
__device__ __forceinline__ void project_core(int explicit_centers, int undistort, const double* __restrict__ R,
                                             double c0, double c1, double c2, double f, double k1, double k2,
                                             double b0, double b1, double b2, double& x0, double& x1,
                                             const double* __restrict__ kn = nullptr)
{
#pragma clang fp contract(off)
    double P0, P1, P2;
    if (explicit_centers) {
        const double d0 = b0 - c0, d1 = b1 - c1, d2 = b2 - c2;
        P0 = R[0] * d0 + R[1] * d1 + R[2] * d2;
        P1 = R[3] * d0 + R[4] * d1 + R[5] * d2;
        P2 = R[6] * d0 + R[7] * d1 + R[8] * d2;
    } else {
        P0 = R[0] * b0 + R[1] * b1 + R[2] * b2; P0 += c0;
        P1 = R[3] * b0 + R[4] * b1 + R[5] * b2; P1 += c1;
        P2 = R[6] * b0 + R[7] * b1 + R[8] * b2; P2 += c2;
    }
    double p0, p1;
    if (kn && kn[0] != 0.0) {
        const double xn = -P0 / P2, yn = -P1 / P2;
        const double r2 = xn * xn + yn * yn;
        const double fac = 1.0 + kn[1] * r2 + kn[2] * r2 * r2 + kn[5] * r2 * r2 * r2;
        const double dxx = 2 * kn[3] * xn * yn + kn[4] * (r2 + 2 * xn * xn);
        const double dxy = kn[3] * (r2 + 2 * yn * yn) + 2 * kn[4] * xn * yn;
        const double xd = xn * fac + dxx, yd = yn * fac + dxy;
        p0 = kn[6] * xd + kn[7] * yd + kn[8];
        p1 = kn[9] * yd + kn[10];
    } else {
        p0 = -P0 * f / P2;
        p1 = -P1 * f / P2;
    }
    if (kn && kn[11] != 0.0) {
        if (kn[12] != 0.0) {
            const double r = sqrt(p0 * p0 + p1 * p1);
            const double angle = 180.0 * atan(r / kn[17]) / 3.14159265358979323846;
            const double rnew = kn[15] * angle / (0.5 * kn[16]);
            p0 = p0 * (rnew / r) + kn[13];
            p1 = p1 * (rnew / r) + kn[14];
        }
    } else if (undistort) {
        const double rsq = (p0 * p0 + p1 * p1) / (f * f);
        const double factor = 1.0 + k1 * rsq + k2 * rsq * rsq;
        p0 *= factor; p1 *= factor;
    }
    x0 = p0; x1 = p1;
}

// Projection from a camera-table row.
template <bool KNOWN = false>
__device__ __forceinline__ void project_row(const ModelCfg& cfg, const double* __restrict__ ct,
                                            double b0, double b1, double b2, double& x0, double& x1)
{
    project_core(cfg.explicit_centers, cfg.undistort, ct + CT_R, ct[CT_A], ct[CT_A + 1], ct[CT_A + 2],
                 ct[CT_F], ct[CT_K1], ct[CT_K2], b0, b1, b2, x0, x1, KNOWN ? ct + CT_KN : nullptr);
}

// Analytic A (2 x cnp row-major, A[r*cnp+c]) and B (2 x 3) plus the projection itself.
template <int CNP>
__device__ __forceinline__ void jac_analytic(const ModelCfg& cfg, const double* __restrict__ ct,
                                             double b0, double b1, double b2,
                                             double* __restrict__ A, double* __restrict__ B, double& x0, double& x1)
{
    const double* R = ct + CT_R;
    const double* Q = ct + CT_Q;
    const double c0 = ct[CT_A], c1 = ct[CT_A + 1], c2 = ct[CT_A + 2];
    const double f = ct[CT_F], k1 = cfg.undistort ? ct[CT_K1] : 0.0, k2 = cfg.undistort ? ct[CT_K2] : 0.0;
    double Pr0, Pr1, Pr2, P0, P1, P2;
    if (cfg.explicit_centers) {
        const double d0 = b0 - c0, d1 = b1 - c1, d2 = b2 - c2;
        Pr0 = R[0] * d0 + R[1] * d1 + R[2] * d2;
        Pr1 = R[3] * d0 + R[4] * d1 + R[5] * d2;
        Pr2 = R[6] * d0 + R[7] * d1 + R[8] * d2;
        P0 = Pr0; P1 = Pr1; P2 = Pr2;
    } else {
        Pr0 = R[0] * b0 + R[1] * b1 + R[2] * b2;
        Pr1 = R[3] * b0 + R[4] * b1 + R[5] * b2;
        Pr2 = R[6] * b0 + R[7] * b1 + R[8] * b2;
        P0 = Pr0 + c0; P1 = Pr1 + c1; P2 = Pr2 + c2;
    }
    const double iz = 1.0 / P2;
    const double u0 = -P0 * f * iz, u1 = -P1 * f * iz;
    const double rsq = (P0 * P0 + P1 * P1) * iz * iz;
    const double g = 1.0 + k1 * rsq + k2 * rsq * rsq;
    const double dg = k1 + 2.0 * k2 * rsq;
    x0 = g * u0; x1 = g * u1;
    // D = dx/dP (2x3)
    const double dr0 = 2.0 * P0 * iz * iz, dr1 = 2.0 * P1 * iz * iz, dr2 = -2.0 * rsq * iz;
    const double fiz = f * iz;
    double D[6];
    D[0] = -g * fiz + u0 * dg * dr0;  D[1] = u0 * dg * dr1;            D[2] = g * fiz * P0 * iz + u0 * dg * dr2;
    D[3] = u1 * dg * dr0;             D[4] = -g * fiz + u1 * dg * dr1; D[5] = g * fiz * P1 * iz + u1 * dg * dr2;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double v = D[3 * r] * R[c] + D[3 * r + 1] * R[3 + c] + D[3 * r + 2] * R[6 + c];
            B[3 * r + c] = v;
            A[CNP * r + c] = cfg.explicit_centers ? -v : D[3 * r + c];
        }
    // dx/dw = D * (-[Pr]x) * Q ; G = -D [Pr]x  (2x3)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const double a = D[3 * r], b = D[3 * r + 1], c = D[3 * r + 2];
        // -[Pr]x = [[0, Pr2, -Pr1], [-Pr2, 0, Pr0], [Pr1, -Pr0, 0]]
        const double G0 = -b * Pr2 + c * Pr1;
        const double G1 = a * Pr2 - c * Pr0;
        const double G2 = -a * Pr1 + b * Pr0;
#pragma unroll
        for (int cc = 0; cc < 3; ++cc)
            A[CNP * r + 3 + cc] = G0 * Q[cc] + G1 * Q[3 + cc] + G2 * Q[6 + cc];
    }
    // cnp fixes the layout statically (sfm.c:637-643): 7/9 carry the focal at 6, 8/9 carry k1,k2 last
    constexpr bool EST = (CNP == 7 || CNP == 9), UND = (CNP >= 8);
    constexpr int KC = EST ? 7 : 6;
    if constexpr (EST) {
        const double s = 1.0 / (f * cfg.f_scale);
        A[6] = x0 * s; A[CNP + 6] = x1 * s;
    }
    if constexpr (UND) {
        const double s = 1.0 / cfg.k_scale;
        A[KC] = u0 * rsq * s;           A[CNP + KC] = u1 * rsq * s;
        A[KC + 1] = u0 * rsq * rsq * s; A[CNP + KC + 1] = u1 * rsq * rsq * s;
    }
}

// Forward-difference A, B with the reference's steps; hx = base projection.
// KNOWN: the scene has cameras with known intrinsics (their table block is consulted); false removes that branch at
// compile time from the hot kernels of ordinary scenes.
template <int CNP, bool KNOWN = false>
__device__ __forceinline__ void jac_fd(const ModelCfg& cfg, const double* __restrict__ ct,
                                       double b0, double b1, double b2,
                                       double* __restrict__ A, double* __restrict__ B, double& x0, double& x1)
{
#pragma clang fp contract(off)
    const double* R = ct + CT_R;
    const double c0 = ct[CT_A], c1 = ct[CT_A + 1], c2 = ct[CT_A + 2];
    const double f = ct[CT_F], k1 = ct[CT_K1], k2 = ct[CT_K2];
    const double* d = ct + CT_D;
    double h0, h1, q0, q1;
    project_core(cfg.explicit_centers, cfg.undistort, R, c0, c1, c2, f, k1, k2, b0, b1, b2, h0, h1, KNOWN ? ct + CT_KN : nullptr);
    x0 = h0; x1 = h1;
    // centre / translation
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
        const double dj = d[jj], d1 = 1.0 / dj;
        project_core(cfg.explicit_centers, cfg.undistort, R,
                     jj == 0 ? c0 + dj : c0, jj == 1 ? c1 + dj : c1, jj == 2 ? c2 + dj : c2,
                     f, k1, k2, b0, b1, b2, q0, q1, KNOWN ? ct + CT_KN : nullptr);
        A[jj] = (q0 - h0) * d1; A[CNP + jj] = (q1 - h1) * d1;
    }
    // rotation increments: perturbed rotations come from the camera table
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
        const double d1 = 1.0 / d[3 + jj];
        project_core(cfg.explicit_centers, cfg.undistort, ct + CT_RP + 9 * jj, c0, c1, c2, f, k1, k2, b0, b1, b2, q0, q1, KNOWN ? ct + CT_KN : nullptr);
        A[3 + jj] = (q0 - h0) * d1; A[CNP + 3 + jj] = (q1 - h1) * d1;
    }
    constexpr bool EST = (CNP == 7 || CNP == 9), UND = (CNP >= 8);
    constexpr int KC = EST ? 7 : 6;
    if constexpr (EST) {
        const double dj = d[6], d1 = 1.0 / dj;
        const double fp = (ct[CT_A + 6] + dj) / cfg.f_scale;
        project_core(cfg.explicit_centers, cfg.undistort, R, c0, c1, c2, fp, k1, k2, b0, b1, b2, q0, q1, KNOWN ? ct + CT_KN : nullptr);
        A[6] = (q0 - h0) * d1; A[CNP + 6] = (q1 - h1) * d1;
    }
    if constexpr (UND) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const double dj = d[KC + t], d1 = 1.0 / dj;
            const double kp = (ct[CT_A + KC + t] + dj) / cfg.k_scale;
            project_core(cfg.explicit_centers, 1, R, c0, c1, c2, f, t == 0 ? kp : k1, t == 1 ? kp : k2,
                         b0, b1, b2, q0, q1, KNOWN ? ct + CT_KN : nullptr);
            A[KC + t] = (q0 - h0) * d1; A[CNP + KC + t] = (q1 - h1) * d1;
        }
    }
    // point coordinates
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
        const double bj = jj == 0 ? b0 : (jj == 1 ? b1 : b2);
        double dj = 1E-04 * bj; dj = fabs(dj); if (dj < 1E-06) dj = 1E-06;
        const double d1 = 1.0 / dj;
        project_core(cfg.explicit_centers, cfg.undistort, R, c0, c1, c2, f, k1, k2,
                     jj == 0 ? b0 + dj : b0, jj == 1 ? b1 + dj : b1, jj == 2 ? b2 + dj : b2, q0, q1, KNOWN ? ct + CT_KN : nullptr);
        B[jj] = (q0 - h0) * d1; B[3 + jj] = (q1 - h1) * d1;
    }
}

}  // namespace bsfm
