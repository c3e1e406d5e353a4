// EXPERIMENT (round 6): A1 = the 16 x 16 block factorisation + inverse of the dataflow POTRF with the rank-1 updates on the matrix pipe.
// The block stays in the ACCUMULATOR layout of v_mfma_f64_16x16x4 (register q of lane l = row 4 q + (l >> 4), column l & 15), symmetric (both
// triangles).  Column j (q = j >> 2, g = j & 3): row j of C is register q of the 16 lanes of lane-row g -- which is at once the A operand
// (A[m][k = g]) and the B operand (B[k = g][n]) of a product whose other k are zero.  So
//     pivot = readlane(C[q], 16 g + j);  rs = 1 / sqrt(pivot);  a = lanes (g, m >= j) ? C[q] * rs : 0;   C = mfma(-a, a, C)
// is the whole elimination step: one matrix instruction instead of (15 - j) DPP broadcasts + FMAs on a quarter of the lanes.  The inverse
// X = inv(L) rides along: row j of X scaled by rs (VALU, 16 lanes), rows m > j: X[m] -= l_mj X[j] = mfma(w, X[q], X), w = -a without lane j.
// Variants: 0 = shipped flow_factor16_body (through LDS, one lane per row); 1 = MFMA, factor + inverse; 2 = MFMA, factor only;
//           3 = MFMA, factor + inverse, next pivot computed ahead on the VALU (off the matrix instruction's latency)
#include "../../bundler_sfm_amd/csrc/chol_flow.hip.h"
#include <cstring>
#include <cmath>
namespace bsfm {

template <int NEWTON> __device__ __forceinline__ double rsq_n(double v)
{
    double s = __builtin_amdgcn_rsq(v);
#pragma unroll
    for (int i = 0; i < NEWTON; ++i) {
        const double g = v * s;                 // ~ sqrt(v)
        const double e = fma(-g, s, 1.0);       // 1 - v s^2
        s = fma(s * 0.5, e, s);
    }
    return s;
}

__device__ __forceinline__ double rdlane(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

template <int J, int MODE, int NEWTON> struct MfCol {
    static __device__ __forceinline__ void run(v4d& C, v4d& X, int& bad, int lane, double pv_ahead)
    {
        constexpr int q = J >> 2, g = J & 3;
        double piv = (MODE == 3 && J > 0) ? pv_ahead : rdlane(C[q], 16 * g + J);
        bad = (bad < 0 && !(piv > 0.0)) ? J : bad;
        const double rs = rsq_n<NEWTON>(piv);
        const bool inrow = (lane >> 4) == g;
        const bool ge = inrow && (lane & 15) >= J, gt = inrow && (lane & 15) > J;
        const double cr = C[q] * rs;
        const double a = ge ? cr : 0.0;
        const double na = ge ? -cr : 0.0;
        double next = 0.0;
        if (MODE == 3 && J < 15) {
            constexpr int q1 = (J + 1) >> 2, g1 = (J + 1) & 3;
            const double an = rdlane(a, 16 * g + J + 1), cn = rdlane(C[q1], 16 * g1 + J + 1);
            next = fma(-an, an, cn);
        }
        C = __builtin_amdgcn_mfma_f64_16x16x4f64(na, a, C, 0, 0, 0);
        if (MODE != 2) {
            const double xs = X[q] * rs;
            X[q] = inrow ? xs : X[q];
            const double w = gt ? na : 0.0;
            X = __builtin_amdgcn_mfma_f64_16x16x4f64(w, X[q], X, 0, 0, 0);
        }
        MfCol<J + 1, MODE, NEWTON>::run(C, X, bad, lane, next);
    }
};
template <int MODE, int NEWTON> struct MfCol<16, MODE, NEWTON> { static __device__ __forceinline__ void run(v4d&, v4d&, int&, int, double) {} };

// cur: the block in accumulator layout, SYMMETRIC.  Leaves inv(L) in blk (swizzled, as flow_factor16_body does).
template <int MODE, int NEWTON>
__device__ __forceinline__ int factor16_mfma(double* blk, v4d C, int lane_in)
{
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int lr = lane >> 4, lc = lane & 15;
    v4d X;
#pragma unroll
    for (int q = 0; q < 4; ++q) X[q] = (4 * q + lr == lc) ? 1.0 : 0.0;
    int bad = -1;
    MfCol<0, MODE, NEWTON>::run(C, X, bad, lane, 0.0);
    if (MODE == 2) X = C;
#pragma unroll
    for (int q = 0; q < 4; ++q) blk[swz16(4 * q + lr, lc)] = (lc <= 4 * q + lr) ? X[q] : 0.0;
    return bad;
}

template <int NEWTON>
__global__ __launch_bounds__(64) void k_bench(const double* __restrict__ A, double* __restrict__ out, long long* __restrict__ cyc, int variant, int reps)
{
    __shared__ double blk[256];
    const int lane = threadIdx.x;
    long long best = 1ll << 60;
    for (int rep = 0; rep < reps; ++rep) {
        v4d C;
        for (int q = 0; q < 4; ++q) {
            const int r = 4 * q + (lane >> 4), c = lane & 15;
            C[q] = A[r * 16 + c];
            blk[swz16(r, c)] = c <= r ? C[q] : 0.0;
        }
        __syncthreads();
        const long long t0 = __builtin_readcyclecounter();
        int bad;
        if (variant == 0) bad = flow_factor16_body(blk, lane);
        else if (variant == 4) { bad = -1; if (lane < 16) bad = flow_factor16_body(blk, lane); }
        else if (variant == 5) { bad = -1; if (lane < 32) bad = flow_factor16_body(blk, lane); }
        else if (variant == 1) bad = factor16_mfma<1, NEWTON>(blk, C, lane);
        else if (variant == 2) bad = factor16_mfma<2, NEWTON>(blk, C, lane);
        else bad = factor16_mfma<3, NEWTON>(blk, C, lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const long long t1 = __builtin_readcyclecounter();
        if (t1 - t0 < best) best = t1 - t0;
        if (bad >= 0) out[300] = bad;
        __syncthreads();
    }
    for (int q = 0; q < 4; ++q) out[(4 * q + (lane >> 4)) * 16 + (lane & 15)] = blk[swz16(4 * q + (lane >> 4), lane & 15)];
    if (lane == 0) cyc[0] = best;
}
}  // namespace bsfm

int main()
{
    using namespace bsfm;
    double hA[256], hO[512], L[256] = { 0 }, Xi[256] = { 0 };
    for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) hA[r * 16 + c] = (r == c ? 2000.0 * (1 + r) : 0.0) + 300.0 / (1 + r + c) + ((r * 7 + c * 7) % 5);
    // reference: long double Cholesky + inverse
    {
        long double l[16][16] = { { 0 } }, x[16][16] = { { 0 } };
        for (int j = 0; j < 16; ++j) {
            long double s = hA[j * 16 + j];
            for (int p = 0; p < j; ++p) s -= l[j][p] * l[j][p];
            l[j][j] = sqrtl(s);
            for (int i = j + 1; i < 16; ++i) { long double t = hA[i * 16 + j]; for (int p = 0; p < j; ++p) t -= l[i][p] * l[j][p]; l[i][j] = t / l[j][j]; }
        }
        for (int c = 0; c < 16; ++c)
            for (int i = c; i < 16; ++i) { long double t = i == c ? 1.0L : 0.0L; for (int p = c; p < i; ++p) t -= l[i][p] * x[p][c]; x[i][c] = t / l[i][i]; }
        for (int i = 0; i < 256; ++i) { L[i] = (double)l[i / 16][i % 16]; Xi[i] = (double)x[i / 16][i % 16]; }
    }
    double* dA; double* dO; long long* dC;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dO, sizeof hO); hipMalloc(&dC, 128);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
    for (int newton = 1; newton <= 2; ++newton)
        for (int v = 0; v < 6; ++v) {
            hipMemset(dO, 0, sizeof hO);
            if (newton == 1) hipLaunchKernelGGL(k_bench<1>, dim3(1), dim3(64), 0, 0, dA, dO, dC, v, 20);
            else hipLaunchKernelGGL(k_bench<2>, dim3(1), dim3(64), 0, 0, dA, dO, dC, v, 20);
            hipDeviceSynchronize();
            long long c; hipMemcpy(&c, dC, 8, hipMemcpyDeviceToHost); hipMemcpy(hO, dO, sizeof hO, hipMemcpyDeviceToHost);
            double err = 0.0;
            const double* ref = v == 2 ? nullptr : Xi;
            if (ref) for (int i = 0; i < 256; ++i) { const double d = fabs(hO[i] - ref[i]) / (fabs(ref[i]) + 1e-300); if (ref[i] != 0.0 && d > err) err = d; if (ref[i] == 0.0 && hO[i] != 0.0) err = 1.0; }
            printf("newton %d variant %d: %lld cycles; max relative error of inv(L) against long double: %.3g   out[0] %g out[16] %g out[17] %g bad %g\n", newton, v, c, err, hO[0], hO[16], hO[17], hO[300]);
        }
    return 0;
}
