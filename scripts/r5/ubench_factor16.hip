// EXPERIMENT (round 5, not in the library): A1 = the 16 x 16 block factorisation + inverse of the dataflow POTRF, split over two waves.
// Measured on MI355X, cycles of s_memtime, one workgroup alone on a CU:
//   one wave, factor + inverse (flow_factor16_body as shipped, v_mov_b64_dpp broadcasts)         3 950 - 4 020   (32-bit DPP pairs: 4 524)
//   factor only, compiler-scheduled                                                             2 500
//   lead wave of the split (factor + one ds_write_b128 per column, asm blocks of 4 DPP + 4 FMA) 2 670
//   follower alone with every column already published                                          4 900 - 5 700
//   lead + follower live                                                                        6 100
// The follower cannot keep up: one LDS round trip per column (poll + operands) is ~240 cycles of fixed cost against the lead's ~170 per
// column, so the pair is SLOWER than one wave doing both chains.  v_readlane + SGPR operand instead of DPP: lead 3 070 (worse); both chains
// through asm blocks of 4 DPP + 8 FMA in one wave: 4 290 (worse than the compiler's interleaving).  What stayed in the library: the 64-bit
// DPP move (-13 %) and s_setprio around A1.
// How long does the 16 x 16 block factorisation + inverse of the dataflow POTRF take on ONE wave (cycles of s_memtime), and what do variants buy?
//   variant 0: flow_factor16_body of chol_flow.hip.h as shipped
//   variant 1: factor only (no inverse chain): lower bound of the critical chain
//   variant 2: see V2 below (experiments)
#include "../../bundler_sfm_amd/csrc/chol_flow.hip.h"
#include <cstring>
namespace bsfm {
// The column updates a[C] -= m * R[C][J] in blocks of FOUR: four broadcasts, then the four FMAs that use them, as ONE asm statement.
// Left to the compiler the pairs come out as broadcast / FMA / broadcast / FMA through one temporary -- every FMA waits for the DPP move
// in front of it (3 450 cycles for the factor against 2 350) -- or, without order fences, all broadcasts of a column hoisted and the
// updates deferred (4 000).
template <int C> struct FlowBcastFma4;
#ifndef BSFM_FLOW_BF4_READLANE
#define BSFM_DPP_BCAST(C) "row_newbcast:" #C " row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define BSFM_FLOW_BF4(C0, C1, C2, C3)                                                                                           \
    template <> struct FlowBcastFma4<C0> {                                                                                      \
        static __device__ __forceinline__ void run(double src, double nmul, double& a0, double& a1, double& a2, double& a3)     \
        {                                                                                                                       \
            double t0, t1, t2, t3;                                                                                              \
            asm volatile("v_mov_b64_dpp %0, %8 " BSFM_DPP_BCAST(C0) "\n\t"                                                      \
                         "v_mov_b64_dpp %1, %8 " BSFM_DPP_BCAST(C1) "\n\t"                                                      \
                         "v_mov_b64_dpp %2, %8 " BSFM_DPP_BCAST(C2) "\n\t"                                                      \
                         "v_mov_b64_dpp %3, %8 " BSFM_DPP_BCAST(C3) "\n\t"                                                      \
                         "v_fma_f64 %4, %9, %0, %4\n\t"                                                                         \
                         "v_fma_f64 %5, %9, %1, %5\n\t"                                                                         \
                         "v_fma_f64 %6, %9, %2, %6\n\t"                                                                         \
                         "v_fma_f64 %7, %9, %3, %7"                                                                             \
                         : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)                   \
                         : "v"(src), "v"(nmul));                                                                                \
        }                                                                                                                       \
    };
#else
// (the value of lane C is the same for the whole wave -- the four 16-lane rows mirror each other: v_readlane into an SGPR pair, which the
//  FMA takes as its constant operand; 2 x 4 + 8 cycles per update against 16 + 8 with v_mov_b64_dpp)
#define BSFM_FLOW_BF4(C0, C1, C2, C3)                                                                                           \
    template <> struct FlowBcastFma4<C0> {                                                                                      \
        static __device__ __forceinline__ void run(double src, double nmul, double& a0, double& a1, double& a2, double& a3)     \
        {                                                                                                                       \
            const int lo = __double2loint(src), hi = __double2hiint(src);                                                       \
            asm volatile("v_readlane_b32 s84, %4, " #C0 "\n\t" "v_readlane_b32 s85, %5, " #C0 "\n\t"                           \
                         "v_readlane_b32 s86, %4, " #C1 "\n\t" "v_readlane_b32 s87, %5, " #C1 "\n\t"                           \
                         "v_readlane_b32 s88, %4, " #C2 "\n\t" "v_readlane_b32 s89, %5, " #C2 "\n\t"                           \
                         "v_readlane_b32 s90, %4, " #C3 "\n\t" "v_readlane_b32 s91, %5, " #C3 "\n\t"                           \
                         "v_fma_f64 %0, %6, s[84:85], %0\n\t"                                                                   \
                         "v_fma_f64 %1, %6, s[86:87], %1\n\t"                                                                   \
                         "v_fma_f64 %2, %6, s[88:89], %2\n\t"                                                                   \
                         "v_fma_f64 %3, %6, s[90:91], %3"                                                                       \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)                                                               \
                         : "v"(lo), "v"(hi), "v"(nmul)                                                                          \
                         : "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91");                                             \
        }                                                                                                                       \
    };
#endif
BSFM_FLOW_BF4(1, 2, 3, 4) BSFM_FLOW_BF4(2, 3, 4, 5) BSFM_FLOW_BF4(3, 4, 5, 6) BSFM_FLOW_BF4(4, 5, 6, 7) BSFM_FLOW_BF4(5, 6, 7, 8)
BSFM_FLOW_BF4(6, 7, 8, 9) BSFM_FLOW_BF4(7, 8, 9, 10) BSFM_FLOW_BF4(8, 9, 10, 11) BSFM_FLOW_BF4(9, 10, 11, 12) BSFM_FLOW_BF4(10, 11, 12, 13)
BSFM_FLOW_BF4(11, 12, 13, 14) BSFM_FLOW_BF4(12, 13, 14, 15)
#undef BSFM_FLOW_BF4
// the same for TWO chains through one set of broadcasts (the one-wave A1: factor and inverse)
template <int C> struct FlowBcastFma4x2;
#define BSFM_FLOW_BF4X2(C0, C1, C2, C3)                                                                                         \
    template <> struct FlowBcastFma4x2<C0> {                                                                                    \
        static __device__ __forceinline__ void run(double src, double nm1, double nm2, double& a0, double& a1, double& a2, double& a3,   \
                                                   double& b0, double& b1, double& b2, double& b3)                              \
        {                                                                                                                       \
            double t0, t1, t2, t3;                                                                                              \
            asm volatile("v_mov_b64_dpp %0, %12 " BSFM_DPP_BCAST(C0) "\n\t"                                                     \
                         "v_mov_b64_dpp %1, %12 " BSFM_DPP_BCAST(C1) "\n\t"                                                     \
                         "v_mov_b64_dpp %2, %12 " BSFM_DPP_BCAST(C2) "\n\t"                                                     \
                         "v_mov_b64_dpp %3, %12 " BSFM_DPP_BCAST(C3) "\n\t"                                                     \
                         "v_fma_f64 %4, %13, %0, %4\n\t"                                                                        \
                         "v_fma_f64 %5, %13, %1, %5\n\t"                                                                        \
                         "v_fma_f64 %6, %13, %2, %6\n\t"                                                                        \
                         "v_fma_f64 %7, %13, %3, %7\n\t"                                                                        \
                         "v_fma_f64 %8, %14, %0, %8\n\t"                                                                        \
                         "v_fma_f64 %9, %14, %1, %9\n\t"                                                                        \
                         "v_fma_f64 %10, %14, %2, %10\n\t"                                                                      \
                         "v_fma_f64 %11, %14, %3, %11"                                                                          \
                         : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3),                  \
                           "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3)                                                               \
                         : "v"(src), "v"(nm1), "v"(nm2));                                                                       \
        }                                                                                                                       \
    };
BSFM_FLOW_BF4X2(1, 2, 3, 4) BSFM_FLOW_BF4X2(2, 3, 4, 5) BSFM_FLOW_BF4X2(3, 4, 5, 6) BSFM_FLOW_BF4X2(4, 5, 6, 7) BSFM_FLOW_BF4X2(5, 6, 7, 8)
BSFM_FLOW_BF4X2(6, 7, 8, 9) BSFM_FLOW_BF4X2(7, 8, 9, 10) BSFM_FLOW_BF4X2(8, 9, 10, 11) BSFM_FLOW_BF4X2(9, 10, 11, 12) BSFM_FLOW_BF4X2(10, 11, 12, 13)
BSFM_FLOW_BF4X2(11, 12, 13, 14) BSFM_FLOW_BF4X2(12, 13, 14, 15)
#undef BSFM_FLOW_BF4X2
template <int C, int N = 16 - C> struct FlowBcastFmaRange2 {
    static __device__ __forceinline__ void run(double (&a)[16], double (&b)[16], double src, double nm1, double nm2)
    {
        if constexpr (N >= 4) {
            FlowBcastFma4x2<C>::run(src, nm1, nm2, a[C], a[C + 1], a[C + 2], a[C + 3], b[C], b[C + 1], b[C + 2], b[C + 3]);
            FlowBcastFmaRange2<C + 4, N - 4>::run(a, b, src, nm1, nm2);
        } else if constexpr (N >= 1) {
            const double t = flow_bcast16<C>(src);
            a[C] = fma(nm1, t, a[C]);
            b[C] = fma(nm2, t, b[C]);
            asm volatile("" : "+v"(a[C]), "+v"(b[C]));
            FlowBcastFmaRange2<C + 1, N - 1>::run(a, b, src, nm1, nm2);
        }
    }
};
template <int N> struct FlowBcastFmaRange2<16, N> { static __device__ __forceinline__ void run(double (&)[16], double (&)[16], double, double, double) {} };
// a[C] += nmul * (lane C's src) for C = C0 .. 15
template <int C, int N = 16 - C> struct FlowBcastFmaRange {
    static __device__ __forceinline__ void run(double (&a)[16], double src, double nmul)
    {
        if constexpr (N >= 4) {
            FlowBcastFma4<C>::run(src, nmul, a[C], a[C + 1], a[C + 2], a[C + 3]);
            FlowBcastFmaRange<C + 4, N - 4>::run(a, src, nmul);
        } else if constexpr (N >= 1) {
            a[C] = fma(nmul, flow_bcast16<C>(src), a[C]);
            FlowBcastFmaRange<C + 1, N - 1>::run(a, src, nmul);
        }
    }
};
template <int N> struct FlowBcastFmaRange<16, N> { static __device__ __forceinline__ void run(double (&)[16], double, double) {} };
// ---- A1 on TWO waves (round 5).  In one wave the factorisation and the inverse are two chains through the same broadcasts: 624
// instructions, 3 950 cycles alone on a SIMD (scripts/r5/ubench_factor16.hip) of which the factor alone takes 2 350 -- and A1 is on the
// critical path of the whole Cholesky eight times per tile column.  The LEAD wave (the owner of the diagonal block) now only factors, and
// publishes column J of R (the raw values) together with 1 / pivot_J in ONE ds_write2; a FOLLOWER wave on another SIMD builds inv(R)
// behind it from those columns, finds the first non-positive pivot, and writes inv(L) = diag(sqrt(pivot)) inv(R) where A2 expects it,
// row by row as the rows become final.  No barrier between the two: LDS executes the DS instructions of a CU in order, the follower
// reads 1 / pivot_J first and column J behind it, so a non-zero 1 / pivot_J means column J is the published one.  The follower puts
// the zeros back when it is done (the next A1 is at least one barrier away).
constexpr int FLOW_HC = 36 * 256;                 // hand-off area behind the 36 blocks
constexpr int FLOW_HC_DOUBLES = 2 * 16 * 16;      // 16 columns of R, 16 x (1 / pivot) with one copy per row lane: every store is a full-wave, branch-free ds_write
// (volatile accesses through a generic pointer become flat instructions with system scope and a vmcnt(0) behind each: the hand-off words are
//  addressed as what they are, LDS)
typedef double FlowD2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) volatile FlowD2 FlowLdsDouble;      // (R[r][J], 1 / pivot_J): ONE ds_write_b128 / ds_read_b128
template <int J> struct FlowLeadCol {
    static __device__ __forceinline__ void run(double (&d)[16], int r, FlowLdsDouble* hc)
    {
        const double piv = flow_bcast16<J>(d[J]);
        double inv = __builtin_amdgcn_rcp(piv);
        inv = fma(fma(-piv, inv, 1.0), inv, inv);
        inv = fma(fma(-piv, inv, 1.0), inv, inv);
        hc[J * 16 + r] = FlowD2{ d[J], inv };            // column J of R, rows J.. (row J = the pivot) and the word the follower waits for; the four 16-lane rows of the wave mirror each other
        FlowBcastFmaRange<J + 1>::run(d, d[J], -(d[J] * inv));      // d[C] -= (R[r][j] / pivot_j) R[C][j]
        FlowLeadCol<J + 1>::run(d, r, hc);
    }
};
template <> struct FlowLeadCol<16> { static __device__ __forceinline__ void run(double (&)[16], int, FlowLdsDouble*) {} };
__device__ __forceinline__ void flow_factor16_lead(const double* blk, double* hc, int lane_in)
{
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int r = lane & 15;
    double d[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) d[c] = blk[swz16(r, c)];
    FlowLeadCol<0>::run(d, r, (FlowLdsDouble*)hc);
}
// rows LO .. LO + 3 of inv(L) = diag(sqrt(pivot)) inv(R): final once column LO + 3 is done.  The square roots of four pivots are ONE chain
// (lane r: its own pivot); taken per column, sixteen such chains of twelve dependent operations were most of the follower's time.
template <int I, int N> struct FlowFollowOut {
    static __device__ __forceinline__ void run(double* blk, const double (&x)[16], double mysq, int r, bool w)
    {
        const double sq_i = flow_bcast16<I>(mysq);
        if (w) blk[swz16(I, r)] = x[I] * sq_i;          // inv(L)[i][r] = sqrt(pivot_i) inv(R)[i][r]
        FlowFollowOut<I + 1, N - 1>::run(blk, x, mysq, r, w);
    }
};
template <int I> struct FlowFollowOut<I, 0> { static __device__ __forceinline__ void run(double*, const double (&)[16], double, int, bool) {} };
template <int J, int EXP = 7> struct FlowFollowCol {
    // inv, col: what a read of column J returned (0: not published at that time)
    static __device__ __forceinline__ bool run(double (&x)[16], FlowLdsDouble* hc, double* blk, int r, bool w, double inv, double col, double& myp)
    {
        unsigned spins = 0;
        // (wave-uniform test -- every lane reads a copy written by the same ds_write: a valid 1 / pivot has a non-zero high word)
        while (__builtin_amdgcn_readfirstlane(__double2hiint(inv)) == 0) {
            if (++spins > (1u << 22)) return false;
            const FlowD2 t = hc[J * 16 + r];
            col = t.x; inv = t.y;
        }
        double inv_n = 0.0, col_n = 0.0;
        if (J + 1 < 16 && (EXP & 4)) { const FlowD2 t = hc[(J + 1) * 16 + r]; col_n = t.x; inv_n = t.y; }      // (in flight while this column is worked on)
        const double xj = x[J] * inv;                   // entry (j, r) of inv(R): final
        x[J] = xj;
        if (!(EXP & 8)) FlowBcastFmaRange<J + 1>::run(x, col, -xj);
        if (EXP & 2) myp = (r == J) ? col : myp;                     // R[j][j], kept by lane j
        if ((J & 3) == 3 && (EXP & 1)) {
            const double mysq = myp * rsqrt_f64(myp);   // L[r][r] = sqrt(pivot_r) (lanes r <= J)
            FlowFollowOut<J - 3, 4>::run(blk, x, mysq, r, w);
        }
        return FlowFollowCol<J + 1, EXP>::run(x, hc, blk, r, w, inv_n, col_n, myp);
    }
};
template <int EXP> struct FlowFollowCol<16, EXP> { static __device__ __forceinline__ bool run(double (&)[16], FlowLdsDouble*, double*, int, bool, double, double, double&) { return true; } };
// the follower: lane c builds column c of inv(R) (all four 16-lane rows mirror each other) and leaves inv(L) in blk; returns the index of
// the first non-positive pivot, -1 (none), or -2: the lead wave never arrived (cannot happen inside one workgroup; the caller raises the
// launch's time-out word rather than hang)
template <int EXP = 7>
__device__ __forceinline__ int flow_factor16_follow(double* blk, double* hc_in, int lane_in)
{
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int r = lane & 15;
    FlowLdsDouble* hc = (FlowLdsDouble*)hc_in;
    double x[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) x[c] = (c == r) ? 1.0 : 0.0;
    double myp = 1.0;
    const FlowD2 t0 = hc[r];
    const double inv0 = t0.y, col0 = t0.x;
    const bool ok = FlowFollowCol<0, EXP>::run(x, hc, blk, r, lane < 16, inv0, col0, myp);
    if (!(EXP & 1)) { for (int c = 0; c < 16; ++c) blk[swz16(c, r)] = x[c]; }
#pragma unroll
    for (int j = 0; j < 16; ++j) hc[j * 16 + r] = FlowD2{ 0.0, 0.0 };
    // first non-positive pivot = dpotrf's info
    const unsigned long long notpos = __ballot(!(myp > 0.0)) & 0xffffull;
    const int bad = notpos ? (int)__builtin_ctzll(notpos) : -1;
    return ok ? bad : -2;
}

}
using namespace bsfm;

template <int J, int C> struct F1Upd {
    static __device__ __forceinline__ void run(double (&d)[16], double lrj)
    {
        const double b = flow_bcast16<C>(d[J]);
        d[C] -= lrj * b;
        F1Upd<J, C + 1>::run(d, lrj);
    }
};
template <int J> struct F1Upd<J, 16> { static __device__ __forceinline__ void run(double (&)[16], double) {} };
template <int J> struct F1Col {
    static __device__ __forceinline__ void run(double (&d)[16], double& myp, int r)
    {
        const double piv = flow_bcast16<J>(d[J]);
        myp = (r == J) ? piv : myp;
        double inv = __builtin_amdgcn_rcp(piv);
        inv = fma(fma(-piv, inv, 1.0), inv, inv);
        inv = fma(fma(-piv, inv, 1.0), inv, inv);
        const double lrj = d[J] * inv;
        F1Upd<J, J + 1>::run(d, lrj);
        F1Col<J + 1>::run(d, myp, r);
    }
};
template <> struct F1Col<16> { static __device__ __forceinline__ void run(double (&)[16], double&, int) {} };

__device__ __forceinline__ int factor_only(double* blk, int lane)
{
    const int r = lane & 15;
    double d[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) d[c] = blk[swz16(r, c)];
    double myp = 1.0;
    F1Col<0>::run(d, myp, r);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < 16; ++c) if (lane < 16) blk[swz16(c, r)] = d[c] * myp;
    return -1;
}

template <int EXP>
__global__ __launch_bounds__(512) void k_bench2(const double* __restrict__ A, double* __restrict__ out, long long* __restrict__ cyc, int reps, int fw)
{
    __shared__ double blk[256];
    __shared__ double hc[FLOW_HC_DOUBLES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long best = 1ll << 60, bestl = 1ll << 60;
    for (int q = threadIdx.x; q < FLOW_HC_DOUBLES; q += blockDim.x) hc[q] = 0.0;
    for (int rep = 0; rep < reps; ++rep) {
        if (wave == 0) for (int q = 0; q < 4; ++q) blk[swz16(4 * q + (lane >> 4), lane & 15)] = A[(4 * q + (lane >> 4)) * 16 + (lane & 15)];
        __syncthreads();
        const long long t0 = __builtin_readcyclecounter();
        if (fw >= 8) {      // serial: the follower starts when everything is published
            if (wave == 0) flow_factor16_lead(blk, hc, lane);
            __syncthreads();
            const long long t3 = __builtin_readcyclecounter();
            if (wave == fw - 8) { const int bad = flow_factor16_follow<EXP>(blk, hc, lane); if (bad != -1) out[300] = bad; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
            const long long t4 = __builtin_readcyclecounter();
            if (wave == fw - 8 && lane == 0) cyc[10] = t4 - t3;
        } else
        if (wave == 0) flow_factor16_lead(blk, hc, lane);
        else if (wave == fw) { const int bad = flow_factor16_follow(blk, hc, lane); if (bad != -1) out[300] = bad; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const long long t1 = __builtin_readcyclecounter();
        __syncthreads();
        const long long t2 = __builtin_readcyclecounter();
        if (wave == 0) { best = t2 - t0 < best ? t2 - t0 : best; bestl = t1 - t0 < bestl ? t1 - t0 : bestl; }
    }
    if (wave == 0) for (int q = 0; q < 4; ++q) out[(4 * q + (lane >> 4)) * 16 + (lane & 15)] = blk[swz16(4 * q + (lane >> 4), lane & 15)];
    if (threadIdx.x == 0) { cyc[0] = best; cyc[1] = bestl; }
    if (lane == 0) cyc[2 + wave] = __builtin_amdgcn_s_getreg((3 << 11) | (4 << 6) | 4) ;      // HW_ID bits 4..7: SIMD_ID (2 bits) + ...
}

__global__ __launch_bounds__(64) void k_bench(const double* __restrict__ A, double* __restrict__ out, long long* __restrict__ cyc, int variant, int reps)
{
    __shared__ double blk[256];
    __shared__ double hcs[FLOW_HC_DOUBLES];
    const int lane = threadIdx.x;
    long long best = 1ll << 60;
    for (int rep = 0; rep < reps; ++rep) {
        for (int q = 0; q < 4; ++q) blk[swz16(4 * q + (lane >> 4), lane & 15)] = A[(4 * q + (lane >> 4)) * 16 + (lane & 15)];
        __syncthreads();
        const long long w0 = wall_clock64();
        const long long t0 = __builtin_readcyclecounter();
        int bad;
        if (variant == 0) bad = flow_factor16_body(blk, lane);
        else if (variant == 2) { flow_factor16_lead(blk, hcs, lane); bad = -1; }
        else bad = factor_only(blk, lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const long long t1 = __builtin_readcyclecounter();
        const long long w1 = wall_clock64();
        if (t1 - t0 < best) { best = t1 - t0; cyc[1] = w1 - w0; }
        if (bad >= 0) out[300] = bad;
        __syncthreads();
    }
    for (int q = 0; q < 4; ++q) out[(4 * q + (lane >> 4)) * 16 + (lane & 15)] = blk[swz16(4 * q + (lane >> 4), lane & 15)];
    if (lane == 0) cyc[0] = best;
}

int main()
{
    double hA[256], hO[512];
    for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) hA[r * 16 + c] = (r == c ? 20.0 : 0.0) + 1.0 / (1 + r + c);
    double* dA; double* dO; long long* dC;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dO, sizeof hO); hipMalloc(&dC, 128);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
    for (int v = 2; v >= 0; --v) {
        hipLaunchKernelGGL(k_bench, dim3(1), dim3(64), 0, 0, dA, dO, dC, v, 20);
        hipDeviceSynchronize();
        long long c, wc; hipMemcpy(&c, dC, 8, hipMemcpyDeviceToHost); hipMemcpy(&wc, dC + 1, 8, hipMemcpyDeviceToHost); printf("[wall clock of that repetition: %lld ticks of 100 MHz = %.2f us -> s_memtime runs at %.0f MHz] ", wc, wc / 100.0, c / (wc / 100.0)); hipMemcpy(hO, dO, 256 * 8, hipMemcpyDeviceToHost);
        printf("variant %d: %lld ticks (s_memrealtime 100 MHz -> %.2f us; or cycles)  out[0..2] %g %g %g\n", v, c, c / 100.0, hO[0], hO[16], hO[17]);
    }
    {
        double ref[256]; memcpy(ref, hO, sizeof ref);
      for (int fw = 1; fw < 10; fw += 4) {
        hipLaunchKernelGGL(k_bench2<7>, dim3(1), dim3(512), 0, 0, dA, dO, dC, 20, fw);
        hipDeviceSynchronize();
        long long c[11]; hipMemcpy(c, dC, 88, hipMemcpyDeviceToHost); if (fw >= 8) printf("follower alone, everything published: %lld\n", c[10]); hipMemcpy(hO, dO, 256 * 8, hipMemcpyDeviceToHost);
        printf("follower = wave %d; HW_ID[7:4] of waves 0..7: %lld %lld %lld %lld %lld %lld %lld %lld\n", fw, c[2], c[3], c[4], c[5], c[6], c[7], c[8], c[9]);
        int diff = 0; for (int i = 0; i < 256; ++i) diff += memcmp(&ref[i], &hO[i], 8) != 0;
        printf("entries that differ from the one-wave result: %d\n", diff);
        printf("two waves: lead alone %lld, both (to the barrier) %lld ticks; out[0..2] %g %g %g\n", c[1], c[0], hO[0], hO[16], hO[17]);
      }
    }
    {
        long long c[11];
#define RUNX(E) hipLaunchKernelGGL(k_bench2<E>, dim3(1), dim3(512), 0, 0, dA, dO, dC, 20, 9); hipDeviceSynchronize(); hipMemcpy(c, dC, 88, hipMemcpyDeviceToHost); printf("follower alone EXP=%d: %lld\n", E, c[10]);
        RUNX(4) RUNX(12) RUNX(8) RUNX(7)
    }
    return 0;
}
