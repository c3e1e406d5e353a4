// schur.hip.h -- Schur-complement task kernel (coalesced record staging through LDS).
//
// S_jk (j <= k) receives sum_i Y_ij W_ik^T = sum_i A_ij^T (B_ij V*_i^-1 B_ik^T) A_ik over the points seen by both cameras
// (lib/sba-1.5/sba_levmar.c:1182-1302; W is never materialised).  The co-visibility triples (record of (i,j), record of
// (i,k), point i) were bucketed and ordered once per problem by block; a task = <= 168 triples of ONE block = one wave,
// 3 lanes per triple (lane r owns output columns r, r+3, r+6), 21 triples per pass, the 21 lane groups are folded with
// shuffles.  No atomics anywhere: partials are summed in task order by k_schur_assemble => run-to-run deterministic.
// Tasks of DIAGONAL blocks (j == k) also accumulate this task's part of the reduced right-hand side
// e_j = ea_j - sum_i A_ij^T (B_ij V*_i^-1 eb_i) (sba_levmar.c:1320-1339): B_ij V*_i^-1 is already there, so E costs one
// 24-byte gather of eb_i and 8 FMAs per triple instead of a second pass over all Jacobian records (that pass took 0.55 ms).
// Data path:
//   * (a first version let every lane gather its 42 doubles itself: each wave-level load touched ~20-30 different lines;)
//   * this one reads the two 192-byte Jacobian records of each triple (camera-major copy Jc: consecutive triples of a block
//     are monotone, mostly consecutive records) and the 48-byte V*^-1 with 16-byte-per-lane coalesced loads
//     (12 lanes per record), parks them in a wave-private LDS slab (record stride 26 doubles: 16-byte aligned, at
//     most 2-way bank conflicts), prefetches the next pass into registers while the current pass computes, and the
//     3 lanes of a triple read their operands from LDS (broadcast where they coincide).
// (Tried and dropped: dealing whole block rows S_j* to one XCD each (workgroups with blockIdx % 8 == x) so that camera j's
// records stay in that XCD's L2 across the row's blocks: 2.03 ms either way -- an XCD runs ~6 rows at a time, 12 MB of
// records against 4 MB of L2.  Also: fetching the record of a diagonal block only once: no gain.)
// Launch order (solver.hip:build_schur_structure): the counters showed 10.6 GB fetched past L2 per launch for 12 GB of gathers
// (profiles/r01_cfg3_fd_v6_pmc_traffic.json) -- in block order the ~11 tasks that read one record run far apart and on
// different XCDs.  Tasks are therefore launched sorted by the first point they touch (tasks over the same points become
// neighbours) and each XCD (blockIdx % 8) is handed one contiguous stretch of that order, so a record is fetched into ONE L2
// and found there by the other tasks that need it.  Output slots stay in block order: sums are bit-identical.
// Then the instruction stream: the disassembly showed ~50 exec-mask blocks per pass (predicated staging, lane-dependent row tests
// the compiler could not fold) and mul + fma + add per accumulator; clamped staging slots, __builtin_assume on the lane's
// column index and explicit FMAs brought the pass to ~235 instructions (96 FP64) and the kernel from 1.86 to 1.67 ms.  L2 read
// latency (268 cycles average) and the TLB (0.05 % misses) are not in the way; what remains is LDS traffic (every lane of a
// triple reads the shared B / V^-1 operands) against two waves per SIMD.
// Split timing (kernel variants, not kept): staging alone (loads + parking, no arithmetic) 1.3 ms, arithmetic alone (no global
// loads after the first pass) 1.2 ms, together 1.55 ms -- the two halves already overlap well and are of equal weight: 12 GB
// of 16-byte gathers per launch through L1 / L2 (9 TB/s) on one side, LDS reads + FMAs on the other.  Going faster needs
// fewer gathers AND fewer LDS reads at once, i.e. records staged once per point chunk and shared by all blocks they feed.
// NOTE (gfx950 / hipcc 7.2): the prefetch registers are arrays of plain double -- arrays of the double2 vector
// struct are not promoted to registers and end up in scratch, which serialises the whole pipeline.
#pragma once
#include "kernels.hip.h"

#ifndef BSFM_SCHUR_MFMA16
#define BSFM_SCHUR_MFMA16 1
#endif

namespace bsfm {

// value of lane Q of every quad (DPP quad_perm [Q, Q, Q, Q]): two 32-bit moves, no LDS
template <int Q>
__device__ __forceinline__ double quad_bcast(double v)
{
    constexpr int ctrl = Q | (Q << 2) | (Q << 4) | (Q << 6);
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

constexpr int SCH_MAXT = 168;         // triples per task (SCHUR_CHUNK in index_build.h)

// (k_schur_tasks_v2, the round-1 VALU kernel -- 3 lanes per triple, 224 VGPRs, 1.56 ms at config 3 -- was removed from the library in
// round 3; its description above is kept because the staging scheme is shared.)

// ------------------------------------------------------------------------------------------------------------------------------
// Round 2: the same tasks with the contraction on the FP64 matrix cores (v_mfma_f64_4x4x4_4b).
// Per S block the sum over the task's co-visibility triples is a skinny GEMM with a long reduction dimension,
//     D[a][b] = sum_rows X[row][a] * Yh[row][b],   rows = (triple p, image row h):  X = A_ij (2 x cnp per triple, straight from the
//     staged record),  Yh[2p+h] = ( (B_ij V*^-1 B_ik^T) A_ik )[h]  ||  (B_ij V*^-1 eb_i)[h]  -- the 2 x 2 core times A_ik, plus the
//     right-hand-side column for diagonal blocks,
// so D[0..cnp)[0..cnp) is the task's part of sum_i Y_ij W_ik^T and D[.][cnp] its part of sum_i Y_ij eb_i (sba_levmar.c:1182-1339).
// v2 did all of it on the VALU with 3 lanes per triple, each re-reading the shared B / B' / V^-1 operands from LDS (LDS pipe 53 %
// busy, 208 VGPRs, 2 waves per SIMD).  Here 4 lanes per triple form the core once and write 2 x (cnp + 1) doubles of Yh to LDS; the
// reduction over the pass's 32 rows is 8 x NI matrix instructions whose operands are single 8-byte LDS reads (X directly from the
// staged record, Yh from its own area); the accumulators are NI doubles per lane, so the kernel needs ~1/3 of v2's registers and
// three workgroups fit a CU (LDS-bound: 13 KB per wave).
// Operand layout of the instruction (probed, potrf.hip.h): lane l = 16 kk + 4 g + r supplies A[g][i = r][kk] and B[g][kk][j = r] of the
// four independent 4 x 4 x 4 products g; D[g][i][j] comes back at lane 16 i + 4 g + j.  The (cnp + 1)-column output is cut into 4 x 4
// sub-blocks (br, bc); product slot g of instruction q takes sub-block 4 q + g of the row-major list.
constexpr int SCM_PASS = 16;          // triples per pass (4 lanes each)

template <int CNP>
__global__ __launch_bounds__(256, 3) void k_schur_tasks_mfma(DevProblem P, const SchurTask* __restrict__ tasks, int ntasks,
        const int2* __restrict__ triples, const int* __restrict__ tri_pt, double* __restrict__ partials,
        double* __restrict__ epart)
{
    constexpr int JS = 2 * CNP + 6;            // doubles per Jacobian record
    constexpr int RS = JS + 2;                 // LDS record stride (doubles), 16-byte aligned
    constexpr int CH = JS / 2;                 // 16-byte chunks per record
    constexpr int NA = (SCM_PASS * CH + 63) / 64;      // staging rounds for one record stream
    constexpr int YS = 12;                     // row stride of Yh (doubles)
    constexpr int RB = (CNP + 3) / 4;          // 4-row sub-blocks of the output
    constexpr int CB = (CNP + 1 + 3) / 4;      // 4-column sub-blocks incl. the right-hand-side column
    constexpr int NSB = RB * CB;
    constexpr int NI = (NSB + 3) / 4;          // matrix instructions per 4 reduction rows
    constexpr int SLAB = 2 * SCM_PASS * RS + SCM_PASS * 6 + SCM_PASS * 4 + 2 * SCM_PASS * YS + 2;   // (+2: a word that stays zero, see MFMA16)
    __shared__ __attribute__((aligned(16))) double sm[4][SLAB];
    __shared__ int sm_tri[4][3 * SCH_MAXT];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int task = blockIdx.x * 4 + wave;
    if (task >= ntasks) return;
    const SchurTask tk = tasks[task];
    if (tk.out < 0) return;
    double* recA = sm[wave];
    double* recB = recA + SCM_PASS * RS;
    double* vin = recB + SCM_PASS * RS;
    double* ebin = vin + SCM_PASS * 6;          // eb_i of the pass's points (diagonal-block tasks only), stride 4
    double* Yh = ebin + SCM_PASS * 4;           // [2 * SCM_PASS][YS]
    int* tq = sm_tri[wave];
    const bool diag = tk.diag != 0;
    for (int t = lane; t < tk.count; t += 64) {         // all triples of the task -> LDS (qa, qb, pt)
        const int2 tr = triples[tk.start + t];
        tq[3 * t] = tr.x; tq[3 * t + 1] = tr.y; tq[3 * t + 2] = tri_pt[tk.start + t];
    }
    // The whole slab starts as zeros: Yh's padding columns stay zero for the task, and record rows past the end of a short
    // task are multiplied (by zero rows of Yh) without ever having been staged -- uninitialised LDS could hold NaN patterns.
    for (int t = lane; t < SLAB; t += 64) recA[t] = 0.0;

    // ---- roles of this lane
    // (1) staging: chunk c = lane + 64 q -> record c / CH, 16-byte part c % CH
    int srec[NA], spart[NA];
#pragma unroll
    for (int q = 0; q < NA; ++q) { const int c = lane + 64 * q; srec[q] = c / CH; spart[q] = c - srec[q] * CH; }
    const int vrec = lane / 3, vpart = lane - 3 * vrec;           // V^-1 / eb: 3 lanes per triple (lanes 0..47)
    // (2) core: triple cp of the pass, lane cq of its four: output columns cq, cq + 4, cq + 8 of Yh (+ the rhs column on lane CNP & 3)
    const int cp = lane >> 2, cq = lane & 3;
    // (3) matrix instruction operands: reduction row kk, product slot g, index r
    const int kk = lane >> 4, g = (lane >> 2) & 3, r = lane & 3;
#if BSFM_SCHUR_MFMA16
    // ONE v_mfma_f64_16x16x4_f64 per 4 reduction rows covers the whole CNP x (CNP + 1) block (A[i][k] at lane i + 16 k, B[k][j] at lane
    // j + 16 k: scripts/probe_mfma16.hip): 2 operand reads per lane and step instead of 2 NI -- the kernel is bound by the LDS pipeline
    // (SQ_LDS_IDX_ACTIVE = 85 % of its duration with the 4x4x4 form, whose NI = 3 instructions per step re-read the slab three times).
    typedef double v4d_ __attribute__((ext_vector_type(4)));
    // Lanes beyond the block read one zero word (stride 0: a broadcast; letting them read the finite numbers that follow in the slab
    // at the common stride -- which pairs the reads into ds_read2_b64 -- measured slower, 1.48 vs 1.43 ms).
    const int oi = lane & 15;                                     // output row of the A operand / output column of the B operand
    const int xbase = oi < CNP ? (kk >> 1) * RS + (kk & 1) * CNP + oi : SLAB - 1, xstep = oi < CNP ? 2 * RS : 0;
    const int ybase = oi < YS ? (int)(Yh - recA) + kk * YS + oi : SLAB - 1, ystep = oi < YS ? 4 * YS : 0;
    v4d_ acc16 = { 0.0, 0.0, 0.0, 0.0 };
    (void)g; (void)r; (void)NI;
#else
    int xoff[NI], yoff[NI];
#pragma unroll
    for (int q = 0; q < NI; ++q) {
        const int sb = min(4 * q + g, NSB - 1);                   // spare slots of the last instruction repeat the last sub-block
        const int br = sb / CB, bc = sb - br * CB;
        xoff[q] = (kk >> 1) * RS + (kk & 1) * CNP + 4 * br + r;    // X[row 4 K4 + kk][4 br + r] inside the staged record
        yoff[q] = kk * YS + 4 * bc + r;
    }
    double acc[NI];
#pragma unroll
    for (int q = 0; q < NI; ++q) acc[q] = 0.0;
#endif
    // Register sets 0 and 1 alternate between passes (one pass of gathers in flight; two in flight measured no faster: 1.68 vs
    // 1.59 ms at config 3 -- the kernel is not bound by the latency of its gathers).
    double pa[2][NA][2], pb[2][NA][2], pv[2][2], pe[2] = { 0.0, 0.0 };

#define BSFM_SCM_ISSUE(p0_, S_)                                                                                     \
    {                                                                                                               \
        const int last_ = min(SCM_PASS, tk.count - (p0_)) - 1;                                                      \
        _Pragma("unroll") for (int q = 0; q < NA; ++q) {                                                            \
            const int rq_ = min(srec[q], last_);                                                                    \
            const double2 ta = *reinterpret_cast<const double2*>(P.Jc + (size_t)tq[3 * ((p0_) + rq_)] * JS + 2 * spart[q]);     \
            const double2 tb = *reinterpret_cast<const double2*>(P.Jc + (size_t)tq[3 * ((p0_) + rq_) + 1] * JS + 2 * spart[q]); \
            pa[S_][q][0] = ta.x; pa[S_][q][1] = ta.y; pb[S_][q][0] = tb.x; pb[S_][q][1] = tb.y;                     \
        }                                                                                                           \
        {                                                                                                           \
            const int rv_ = min(vrec, last_);                                                                       \
            const double2 tv = *reinterpret_cast<const double2*>(P.Vinv + (size_t)tq[3 * ((p0_) + rv_) + 2] * 6 + 2 * vpart); \
            pv[S_][0] = tv.x; pv[S_][1] = tv.y;                                                                     \
            if (diag) pe[S_] = P.eb[(size_t)tq[3 * ((p0_) + rv_) + 2] * 3 + vpart];                                 \
        }                                                                                                           \
    }
#define BSFM_SCM_PARK(p0_, S_)                                                                                      \
    {                                                                                                               \
        const int last_ = min(SCM_PASS, tk.count - (p0_)) - 1;                                                      \
        _Pragma("unroll") for (int q = 0; q < NA; ++q) {                                                            \
            const int rq_ = min(srec[q], last_);                                                                    \
            *reinterpret_cast<double2*>(recA + rq_ * RS + 2 * spart[q]) = make_double2(pa[S_][q][0], pa[S_][q][1]); \
            *reinterpret_cast<double2*>(recB + rq_ * RS + 2 * spart[q]) = make_double2(pb[S_][q][0], pb[S_][q][1]); \
        }                                                                                                           \
        if (lane < 3 * SCM_PASS) {                                                                                  \
            const int rv_ = min(vrec, last_);                                                                       \
            *reinterpret_cast<double2*>(vin + rv_ * 6 + 2 * vpart) = make_double2(pv[S_][0], pv[S_][1]);            \
            if (diag) ebin[rv_ * 4 + vpart] = pe[S_];                                                               \
        }                                                                                                           \
    }
#if BSFM_SCHUR_MFMA16
#define BSFM_SCM_REDUCE                                                                                             \
        {                                                                                                           \
            double xa[SCM_PASS / 2], yb[SCM_PASS / 2];                                                              \
            _Pragma("unroll") for (int k4 = 0; k4 < SCM_PASS / 2; ++k4) { xa[k4] = recA[xbase + k4 * xstep]; yb[k4] = recA[ybase + k4 * ystep]; } \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
            _Pragma("unroll") for (int k4 = 0; k4 < SCM_PASS / 2; ++k4)                                             \
                acc16 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[k4], yb[k4], acc16, 0, 0, 0);                       \
        }
#else
#define BSFM_SCM_REDUCE                                                                                             \
        _Pragma("unroll") for (int k4 = 0; k4 < SCM_PASS / 2; k4 += 2) {                                            \
            double xa[2][NI], yb[2][NI];                                                                            \
            _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                           \
                _Pragma("unroll") for (int q = 0; q < NI; ++q) {                                                    \
                    xa[u][q] = recA[2 * (k4 + u) * RS + xoff[q]];                                                   \
                    yb[u][q] = Yh[4 * (k4 + u) * YS + yoff[q]];                                                     \
                }                                                                                                   \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
            _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                           \
                _Pragma("unroll") for (int q = 0; q < NI; ++q)                                                      \
                    acc[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(xa[u][q], yb[u][q], acc[q], 0, 0, 0);               \
        }
#endif
// the 2 x 2 core of triple cp and this lane's columns of Yh, then the reduction over the pass's 2 x SCM_PASS rows on the matrix cores
#define BSFM_SCM_COMPUTE(p0_)                                                                                       \
    {                                                                                                               \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      /* records parked (same wave: LDS operations complete in order) */ \
        {                                                                                                           \
            const bool live = (p0_) + cp < tk.count;                                                                \
            const double* Ja = recA + cp * RS;                                                                      \
            const double* Jb = recB + cp * RS;                                                                      \
            /* ONE lane of the triple's four forms the 2 x 2 core (18 LDS reads); its neighbours take it by DPP quad broadcast -- the \
               kernel is LDS-bound, and all four lanes reading the same B blocks and V^-1 was a third of its LDS traffic.  The forming \
               lane is the one that also owns the right-hand-side column (cq == CNP & 3), so c00 .. c12 never leave it. */            \
            double c00 = 0.0, c01 = 0.0, c02 = 0.0, c10 = 0.0, c11 = 0.0, c12 = 0.0, m00 = 0.0, m01 = 0.0, m10 = 0.0, m11 = 0.0;     \
            if (cq == (CNP & 3)) {                                                                                  \
                const double* vi = vin + cp * 6;                                                                    \
                const double i00 = vi[0], i01 = vi[1], i02 = vi[2], i11 = vi[3], i12 = vi[4], i22 = vi[5];          \
                const double* Ba = Ja + 2 * CNP;                                                                    \
                const double* Bb = Jb + 2 * CNP;                                                                    \
                c00 = Ba[0] * i00 + Ba[1] * i01 + Ba[2] * i02;                                                      \
                c01 = Ba[0] * i01 + Ba[1] * i11 + Ba[2] * i12;                                                      \
                c02 = Ba[0] * i02 + Ba[1] * i12 + Ba[2] * i22;                                                      \
                c10 = Ba[3] * i00 + Ba[4] * i01 + Ba[5] * i02;                                                      \
                c11 = Ba[3] * i01 + Ba[4] * i11 + Ba[5] * i12;                                                      \
                c12 = Ba[3] * i02 + Ba[4] * i12 + Ba[5] * i22;                                                      \
                m00 = c00 * Bb[0] + c01 * Bb[1] + c02 * Bb[2];                                                      \
                m01 = c00 * Bb[3] + c01 * Bb[4] + c02 * Bb[5];                                                      \
                m10 = c10 * Bb[0] + c11 * Bb[1] + c12 * Bb[2];                                                      \
                m11 = c10 * Bb[3] + c11 * Bb[4] + c12 * Bb[5];                                                      \
            }                                                                                                       \
            m00 = quad_bcast<CNP & 3>(m00); m01 = quad_bcast<CNP & 3>(m01);                                         \
            m10 = quad_bcast<CNP & 3>(m10); m11 = quad_bcast<CNP & 3>(m11);                                         \
            double* y0 = Yh + (2 * cp) * YS;                                                                        \
            double* y1 = y0 + YS;                                                                                   \
            _Pragma("unroll") for (int a = 0; a < (CNP + 3) / 4; ++a) {                                             \
                const int col = cq + 4 * a;                                                                         \
                if (col < CNP) {                                                                                    \
                    const double b0 = Jb[col], b1 = Jb[CNP + col];                                                  \
                    y0[col] = live ? m00 * b0 + m01 * b1 : 0.0;                                                     \
                    y1[col] = live ? m10 * b0 + m11 * b1 : 0.0;                                                     \
                }                                                                                                   \
            }                                                                                                       \
            if (diag && cq == (CNP & 3)) {                        /* right-hand-side column: B_ij V*^-1 eb_i */     \
                const double e0 = ebin[cp * 4], e1 = ebin[cp * 4 + 1], e2 = ebin[cp * 4 + 2];                       \
                y0[CNP] = live ? c00 * e0 + c01 * e1 + c02 * e2 : 0.0;                                              \
                y1[CNP] = live ? c10 * e0 + c11 * e1 + c12 * e2 : 0.0;                                              \
            }                                                                                                       \
        }                                                                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                          \
        /* operands of two reduction steps are fetched together: one LDS wait per 2 NI matrix instructions, not one per instruction */ \
        BSFM_SCM_REDUCE                                                                                             \
        asm volatile("" ::: "memory");                           /* the next parking must not move above these reads */ \
    }

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the triple list is in LDS, the slab is cleared
    BSFM_SCM_ISSUE(0, 0)
    for (int p0 = 0; p0 < tk.count; p0 += 2 * SCM_PASS) {        // one pass of gathers in flight while the previous one is reduced
        BSFM_SCM_PARK(p0, 0)
        if (p0 + SCM_PASS < tk.count) BSFM_SCM_ISSUE(p0 + SCM_PASS, 1)
        BSFM_SCM_COMPUTE(p0)
        if (p0 + SCM_PASS < tk.count) {
            BSFM_SCM_PARK(p0 + SCM_PASS, 1)
            if (p0 + 2 * SCM_PASS < tk.count) BSFM_SCM_ISSUE(p0 + 2 * SCM_PASS, 0)
            BSFM_SCM_COMPUTE(p0 + SCM_PASS)
        }
    }
#undef BSFM_SCM_ISSUE
#undef BSFM_SCM_PARK
#undef BSFM_SCM_COMPUTE
#undef BSFM_SCM_REDUCE
#if BSFM_SCHUR_MFMA16
    // D[a][b]: register r of lane l holds row a = 4 r + (l >> 4), column b = l & 15
    {
        double* out = partials + (size_t)tk.out * CNP * CNP;
        const int b = lane & 15;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int a = 4 * rr + (lane >> 4);
            if (a < CNP) {
                if (b < CNP) out[a * CNP + b] = acc16[rr];
                else if (b == CNP && diag) epart[(size_t)tk.out * CNP + a] = acc16[rr];
            }
        }
    }
#else
    // D[g][i][j] of instruction q sits at lane 16 i + 4 g + j
    {
        const int i = lane >> 4, j = lane & 3;
        double* out = partials + (size_t)tk.out * CNP * CNP;
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const int sb = 4 * q + g;
            if (sb < NSB) {
                const int br = sb / CB, bc = sb - br * CB;
                const int a = 4 * br + i, b = 4 * bc + j;
                if (a < CNP) {
                    if (b < CNP) out[a * CNP + b] = acc[q];
                    else if (b == CNP && diag) epart[(size_t)tk.out * CNP + a] = acc[q];
                }
            }
        }
    }
#endif
}

}  // namespace bsfm
