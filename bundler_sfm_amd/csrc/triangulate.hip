// triangulate.hip -- batched multi-view triangulation, gfx950.
//
// Replaces, over a ragged batch of independent points, the reference's
//   triangulate_n          lib/imagelib/triangulate.c:181-272   linear least squares (dgelsy) + lmdif polish, tol 1e-5
//   triangulate_n_refine   lib/imagelib/triangulate.c:133-178   lmdif polish from a given point
//   triangulate            lib/imagelib/triangulate.c:281-338   two views, tol 1e-10, error = sum of squares
// which BundlerApp::TriangulateNViews (src/BundleAdd.cpp:47-123), Bundle.cpp:2240,2758 and BundleFast.cpp:193 call one
// point at a time.  The polish is MINPACK's lmdif (lib/cminpack/lmdif.c through lmdif1.c: ftol = xtol = tol, gtol = 0,
// maxfev = 200 (n+1), epsfcn = 0, mode 1, factor 100) with its forward-difference Jacobian (fdjac2.c), pivoted QR
// (qrfac.c), Levenberg-Marquardt parameter (lmpar.c) and qrsolv.c -- restated in lmdif.hip.h, used here with 3 unknowns.
//
// MI355X design: one thread per point, nothing stored per observation: the Jacobian's triangular factor is accumulated
// while STREAMING over the point's views with Givens rotations (lmdif.hip.h).  The linear start of triangulate_n is the
// same streaming QR on the rows [A | b] (dgelsy with rcond = -1 never truncates the rank).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../include/bsfm.h"
#include "lmdif.hip.h"

namespace {

using namespace bsfm_lm;

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

// residual functor of triangulate_n_residual (triangulate.c:115-129) for lm_lmdif<3>
struct TriFcn {
    TriViews V;
    __device__ int rows() const { return 2 * (V.v1 - V.v0); }
    // fvec = observed - projected; returns enorm(fvec)
    __device__ double fnorm(const double* x) const
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
    __device__ void jac_qr(const double* x, QrN<3>& F) const
    {
        const double eps = sqrt(LM_EPSMCH);
        double h[3];
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
            double r0[3] = { ((ox - ax) - fx) / h[0], ((ox - bx) - fx) / h[1], ((ox - cx) - fx) / h[2] };
            F.add_row(r0, fx);
            double r1[3] = { ((oy - ay) - fy) / h[0], ((oy - by) - fy) / h[1], ((oy - cy) - fy) / h[2] };
            F.add_row(r1, fy);
        }
    }
};

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
        QrN<3> F; F.clear();
        for (int v = V.v0; v < V.v1; ++v) {
            const size_t c = cam ? (size_t)cam[v] : (size_t)v;
            const double* Rv = R + 9 * c; const double* tv = t + 3 * c;
            const double u = p[2 * (size_t)v], w = p[2 * (size_t)v + 1];
            double r0[3] = { Rv[0] - u * Rv[6], Rv[1] - u * Rv[7], Rv[2] - u * Rv[8] };
            F.add_row(r0, tv[2] * u - tv[0]);
            double r1[3] = { Rv[3] - w * Rv[6], Rv[4] - w * Rv[7], Rv[5] - w * Rv[8] };
            F.add_row(r1, tv[2] * w - tv[1]);
        }
        x[2] = F.q[2] / F.r[2][2];
        x[1] = (F.q[1] - F.r[1][2] * x[2]) / F.r[1][1];
        x[0] = (F.q[0] - F.r[0][1] * x[1] - F.r[0][2] * x[2]) / F.r[0][0];
    }
    TriFcn fcn { V };
    int info = lm_lmdif<3>(fcn, x, mode == BSFM_TRI_PAIR ? 1.0e-10 : 1.0e-5);
    if (info == 8) info = 4;                          // lmdif1.c
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
