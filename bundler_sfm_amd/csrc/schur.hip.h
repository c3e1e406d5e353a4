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
// Round 4: that order only works when the blocks of a camera group cut their point lists at the same places (the generator's
// cliques).  On a connected scene it left 9 % L2 hits and 17 GB fetched per launch; the launch order is now (slice of the point
// range, lower camera, higher camera) with the cameras in breadth-first numbering of the co-visibility graph
// (index_build.hip: SCHUR_ORDER_CLUSTERED; profiles/r04_schur_connected_counters.txt): same on the cliques, 3.2 -> 2.5 ms connected.
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

namespace bsfm {

template <int Q> __device__ __forceinline__ double quad_bcast(double v) { return quad_bcast_d<Q>(v); }
template <int N> __device__ __forceinline__ double row_shr_d(double v)      // value of lane - N inside the 16-lane row, 0 where there is none
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x110 + N, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x110 + N, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

constexpr int SCH_MAXT = SCHUR_CHUNK_MAX;   // triples per task at most (index_build.h)

// (k_schur_tasks_v2, the round-1 VALU kernel -- 3 lanes per triple, 224 VGPRs, 1.56 ms at config 3 -- was removed from the library in
// round 3; its description above is kept because the staging scheme is shared.)

// ------------------------------------------------------------------------------------------------------------------------------
// Round 3: k_schur_tasks -- the same tasks, off the LDS pipeline.
// Round 2's kernel (k_schur_tasks_mfma) parked the raw 192-byte records of both cameras and V*^-1 in a wave-private LDS slab, had
// the four lanes of a triple read them back piecewise (2 x 2 core, Yh), wrote Yh and re-read everything in matrix-operand layout:
// ~31 KB of LDS traffic per wave and pass of 16 triples, SQ_LDS_IDX_ACTIVE 85 % of the kernel, 0.12 of the FP64 peak.
// Here NOTHING raw goes through LDS:
//   * the four lanes (p, q) of a triple fetch the 16-byte chunks they own STRAIGHT INTO REGISTERS: chunks c0 = q + 4 (p & 1),
//     c0 ^ 4 (and 8 on q = 0) of A_ij and of A_ik (a chunk = (A[0][c], A[1][c]), kernels.hip.h), chunk q of C_ij || r_ij and of B_ik;
//   * the 2 x 2 core M = C_ij B_ik^T needs 12 doubles that sit in the quad: 12 DPP quad broadcasts + 12 FMAs (k_schur_prep has
//     already folded V*^-1 into C, so there is no 3 x 3 work and no V*^-1 / eb gather per triple);
//   * each lane turns its A_ik chunks into Yh chunks (Yh[h'][c] = M[h'][0] A[0][c] + M[h'][1] A[1][c], 4 FMAs per chunk) and
//     writes ONLY the two matrix-operand slabs: X[p][c] = its A_ij chunk as it came, Y[p][c] = the Yh chunk (+ the right-hand-side
//     chunk r_ij in column cnp for diagonal-block tasks).  Rows are 256 bytes apart: the operand reads below are conflict-free by
//     construction, and the chunk order per lane (c0 depends on p & 1) keeps the two triples of an 8-lane store group on
//     different banks;
//   * the block sum D = X^T Y over the pass's 32 rows is 8 x v_mfma_f64_16x16x4.  Row (triple 4 u + kk, image row h) goes to
//     k-step 2 u + h, slot kk: lane (a, kk) gets BOTH image rows of its operand with ONE 16-byte read per u -- 8 ds_read_b128 per
//     pass instead of 16 ds_read_b64 (+ 45 reads / writes of the old staging scheme).
// LDS traffic per wave and pass: 16 triples x (9 + 10) chunks written + 8 x 64 x 16 bytes read = 13 KB (was ~31 KB).
// The task list, the block order of the partial sums (deterministic: no atomics), the diagonal-block right-hand side and the
// epilogue are those of round 2.  v_mfma_f64_16x16x4: A[i][k] at lane i + 16 k, B[k][j] at lane j + 16 k, D[i][j]: register r of
// lane l = row 4 r + (l >> 4), column l & 15 (scripts/probe_mfma16.hip).
constexpr int SCM_PASS = 16;          // triples per pass (4 lanes each)
// WPS: waves per SIMD the kernel is compiled for (= workgroups per CU); BSFM_SCHUR_WPS selects 2 / 3 / 4 at run time
// Split timing at config 3 (variants of this kernel, not kept; profiles/r03_schur_split.txt): as shipped 1.02 ms (3 workgroups per CU,
// 160 triples per task; 0.94 with 4 per CU and 192); no gathers after a task's first pass 0.63; no matrix instructions 0.79; gathers +
// 2 x 2 core only (no LDS, no matrix instructions) 0.81 -- the kernel is bound by the L1-miss path of its 16-byte gathers (11.4 GB
// per launch through L1 = 23 bytes / clock / CU, of which 1.7-3.3 GB miss L2), no longer by the LDS pipeline (SQ_LDS_IDX_ACTIVE 30 %
// of the kernel, was 85 %) or the matrix pipe (35 % busy).
// Round 5: M4 = the block sum on v_mfma_f64_4x4x4_4b instead of v_mfma_f64_16x16x4 (see the note at the end of this file: FP64 matrix instructions
// do not overlap with vector work on gfx950, so the 68 % of a 16 x 16 tile that is padding costs real time): the four blocks of an
// instruction take four pairs of triples, a half pass of 8 triples is NAI x NBJ instructions on independent accumulators fed by
// NAI + NBJ operand reads of 8 bytes per lane; the four partial blocks are added across the lanes at the end of the task.
// 288 instead of 512 matrix cycles per pass, 6 instead of 8 KB of operand reads.
template <int CNP, int WPS, bool M4>
__global__ __launch_bounds__(256, WPS) void k_schur_tasks(DevProblem P, const SchurTask* __restrict__ tasks, int ntasks,
        const int2* __restrict__ triples, double* __restrict__ partials, double* __restrict__ epart)
{
    typedef double d2_ __attribute__((ext_vector_type(2)));
    typedef double v4d_ __attribute__((ext_vector_type(4)));
    // chunks per slab row: 256 bytes for the 16 x 16 x 4 operand reads (a row = the whole bank cycle, conflict-free by construction); the
    // 4 x 4 x 4 operand reads touch 64 bytes of EIGHT rows at once -- 256 bytes apart they would all sit on the same banks, 208 spreads them
    constexpr int ROWC = M4 ? 13 : 16;
    constexpr int NAI = (CNP + 3) / 4, NBJ = (CNP + 1 + 3) / 4;      // 4-column blocks of X (output rows) and of Y (output columns incl. column CNP)
    constexpr bool THIRD = CNP > 8;                   // a ninth chunk exists (lane q = 0 takes it)
    constexpr bool ALLC = CNP >= 8;                   // chunks 0 .. 7 all exist: no per-lane column test
    __shared__ __attribute__((aligned(16))) d2_ smx[4][SCM_PASS * ROWC];
    __shared__ __attribute__((aligned(16))) d2_ smy[4][SCM_PASS * ROWC];
    __shared__ int2 sm_tri[4][SCH_MAXT];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int task = blockIdx.x * 4 + wave;
    if (task >= ntasks) return;
    const SchurTask tk = tasks[task];
    if (tk.out < 0) return;
    d2_* X = smx[wave];
    d2_* Y = smy[wave];
    int2* tq = sm_tri[wave];
    const bool diag = tk.diag != 0;
    for (int t = lane; t < tk.count; t += 64) tq[t] = triples[tk.start + t];
    // columns >= cnp of X and > cnp (>= cnp for off-diagonal blocks) of Y are never written: they must read as zeros
    for (int t = lane; t < SCM_PASS * ROWC; t += 64) { const d2_ z = { 0.0, 0.0 }; X[t] = z; Y[t] = z; }

    const int p = lane >> 2, q = lane & 3;            // triple of the pass, lane of its quad
    const int c0 = q + 4 * (p & 1), c1 = c0 ^ 4;      // this lane's chunks (a column c < cnp exists)
    const int l0 = c0 < CNP ? c0 : CNP - 1, l1 = c1 < CNP ? c1 : CNP - 1;      // clamped for the loads
    const int kk = lane >> 4, oi = lane & 15;         // matrix-operand role: reduction slot, output row / column
    const d2_* Ac2 = reinterpret_cast<const d2_*>(P.Ac);
    const d2_* Bc2 = reinterpret_cast<const d2_*>(P.Bc);
    const d2_* Cc2 = reinterpret_cast<const d2_*>(P.Cc);
    v4d_ acc0 = { 0.0, 0.0, 0.0, 0.0 }, acc1 = { 0.0, 0.0, 0.0, 0.0 };
    // M4: lane l = 16 k + 4 g + r feeds triple 2 g + (k >> 1) of a half pass, image row k & 1, column r of a 4-column block
    const int mt = 2 * ((lane >> 2) & 3) + (lane >> 5);
    const unsigned char* Xm = reinterpret_cast<const unsigned char*>(X) + mt * (ROWC * 16) + (lane & 3) * 16 + ((lane >> 4) & 1) * 8;
    const unsigned char* Ym = reinterpret_cast<const unsigned char*>(Y) + mt * (ROWC * 16) + (lane & 3) * 16 + ((lane >> 4) & 1) * 8;
    double acc4[NAI][NBJ];
#pragma unroll
    for (int a = 0; a < NAI; ++a)
#pragma unroll
        for (int b = 0; b < NBJ; ++b) acc4[a][b] = 0.0;

    // register sets 0 / 1 alternate between passes: one pass of gathers is in flight while the previous one is reduced
    d2_ aj[2][3], ak[2][3], cj[2], bk[2];
#define BSFM_SCH_ISSUE(p0_, S_)                                                                                     \
    {                                                                                                               \
        const int tr_ = min((p0_) + p, tk.count - 1);                                                               \
        const int2 ab_ = tq[tr_];                                                                                   \
        const d2_* ra_ = Ac2 + (size_t)ab_.x * CNP;                                                                 \
        const d2_* rb_ = Ac2 + (size_t)ab_.y * CNP;                                                                 \
        aj[S_][0] = ra_[l0]; aj[S_][1] = ra_[l1];                                                                   \
        ak[S_][0] = rb_[l0]; ak[S_][1] = rb_[l1];                                                                   \
        if (THIRD) { aj[S_][2] = ra_[CNP - 1]; ak[S_][2] = rb_[CNP - 1]; }                                          \
        cj[S_] = Cc2[(size_t)ab_.x * 4 + q];                                                                        \
        bk[S_] = Bc2[(size_t)ab_.y * 4 + min(q, 2)];                                                                \
    }
#define BSFM_SCH_COMPUTE(p0_, S_)                                                                                   \
    {                                                                                                               \
        const bool live_ = (p0_) + p < tk.count;                                                                    \
        /* C (2 x 3 row-major over the chunks of lanes 0..2) and B likewise, to every lane of the quad */           \
        const double C00 = quad_bcast<0>(cj[S_].x), C01 = quad_bcast<0>(cj[S_].y), C02 = quad_bcast<1>(cj[S_].x);   \
        const double C10 = quad_bcast<1>(cj[S_].y), C11 = quad_bcast<2>(cj[S_].x), C12 = quad_bcast<2>(cj[S_].y);   \
        const double B00 = quad_bcast<0>(bk[S_].x), B01 = quad_bcast<0>(bk[S_].y), B02 = quad_bcast<1>(bk[S_].x);   \
        const double B10 = quad_bcast<1>(bk[S_].y), B11 = quad_bcast<2>(bk[S_].x), B12 = quad_bcast<2>(bk[S_].y);   \
        double m00 = C00 * B00 + C01 * B01 + C02 * B02, m01 = C00 * B10 + C01 * B11 + C02 * B12;                    \
        double m10 = C10 * B00 + C11 * B01 + C12 * B02, m11 = C10 * B10 + C11 * B11 + C12 * B12;                    \
        if (!live_) { m00 = 0.0; m01 = 0.0; m10 = 0.0; m11 = 0.0; }                                                 \
        d2_* xr_ = X + p * ROWC;                                                                                    \
        d2_* yr_ = Y + p * ROWC;                                                                                    \
        if (ALLC || c0 < CNP) { xr_[c0] = aj[S_][0]; const d2_ y_ = { m00 * ak[S_][0].x + m01 * ak[S_][0].y, m10 * ak[S_][0].x + m11 * ak[S_][0].y }; yr_[c0] = y_; } \
        if (ALLC || c1 < CNP) { xr_[c1] = aj[S_][1]; const d2_ y_ = { m00 * ak[S_][1].x + m01 * ak[S_][1].y, m10 * ak[S_][1].x + m11 * ak[S_][1].y }; yr_[c1] = y_; } \
        if (THIRD && q == 0) { xr_[CNP - 1] = aj[S_][2]; const d2_ y_ = { m00 * ak[S_][2].x + m01 * ak[S_][2].y, m10 * ak[S_][2].x + m11 * ak[S_][2].y }; yr_[CNP - 1] = y_; } \
        if (diag && q == 3) { const d2_ r_ = { live_ ? cj[S_].x : 0.0, live_ ? cj[S_].y : 0.0 }; yr_[CNP] = r_; }   \
        /* same wave: LDS operations complete in order, the reads below see the stores above */                    \
        if (M4) {                                                                                                   \
            double xm_[2][NAI], ym_[2][NBJ];                                                                        \
            _Pragma("unroll") for (int a = 0; a < NAI; ++a) {                                                       \
                xm_[0][a] = *reinterpret_cast<const double*>(Xm + 64 * a);                                          \
                xm_[1][a] = *reinterpret_cast<const double*>(Xm + 8 * (ROWC * 16) + 64 * a); }                      \
            _Pragma("unroll") for (int b = 0; b < NBJ; ++b) {                                                       \
                ym_[0][b] = *reinterpret_cast<const double*>(Ym + 64 * b);                                          \
                ym_[1][b] = *reinterpret_cast<const double*>(Ym + 8 * (ROWC * 16) + 64 * b); }                      \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
            _Pragma("unroll") for (int hp = 0; hp < 2; ++hp)                                                        \
                _Pragma("unroll") for (int a = 0; a < NAI; ++a)                                                     \
                    _Pragma("unroll") for (int b = 0; b < NBJ; ++b)                                                 \
                        acc4[a][b] = __builtin_amdgcn_mfma_f64_4x4x4f64(xm_[hp][a], ym_[hp][b], acc4[a][b], 0, 0, 0); \
        } else {                                                                                                    \
        d2_ xa_[SCM_PASS / 4], yb_[SCM_PASS / 4];                                                                   \
        _Pragma("unroll") for (int u = 0; u < SCM_PASS / 4; ++u) { xa_[u] = X[(4 * u + kk) * ROWC + oi]; yb_[u] = Y[(4 * u + kk) * ROWC + oi]; } \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                          \
        _Pragma("unroll") for (int u = 0; u < SCM_PASS / 4; ++u) {                                                  \
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa_[u].x, yb_[u].x, acc0, 0, 0, 0);                         \
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa_[u].y, yb_[u].y, acc1, 0, 0, 0);                         \
        }                                                                                                           \
        }                                                                                                           \
        asm volatile("" ::: "memory");                           /* the next pass's stores must not move above these reads */ \
    }

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the triple list is in LDS, the slabs are cleared
    // The prefetch of the next pass is issued UNCONDITIONALLY (past the end it repeats the last triple: the index is clamped): behind a
    // branch the compiler cannot count the loads in flight and waits for vmcnt(0) in front of every pass -- prefetch and all -- which
    // is what rounds 3-4 ran with (round 5, found in the disassembly of the row-wise kernel).
    BSFM_SCH_ISSUE(0, 0)
    for (int p0 = 0; p0 < tk.count; p0 += 2 * SCM_PASS) {
        BSFM_SCH_ISSUE(p0 + SCM_PASS, 1)
        BSFM_SCH_COMPUTE(p0, 0)
        BSFM_SCH_ISSUE(p0 + 2 * SCM_PASS, 0)
        if (p0 + SCM_PASS < tk.count) BSFM_SCH_COMPUTE(p0 + SCM_PASS, 1)
    }
#undef BSFM_SCH_ISSUE
#undef BSFM_SCH_COMPUTE
    if (M4) {      // D[g][i][j] at lane 16 i + 4 g + j: the sum over the four pairs g ends up in the lanes of g = 3
        double* out = partials + (size_t)tk.out * CNP * CNP;
        const int i_ = lane >> 4, j_ = lane & 3;
        const bool own = (lane & 12) == 12;
#pragma unroll
        for (int a = 0; a < NAI; ++a)
#pragma unroll
            for (int b = 0; b < NBJ; ++b) {
                double v = acc4[a][b];
                v += row_shr_d<4>(v);
                v += row_shr_d<8>(v);
                const int ra = 4 * a + i_, cb = 4 * b + j_;
                if (own && ra < CNP) {
                    if (cb < CNP) out[ra * CNP + cb] = v;
                    else if (cb == CNP && diag) epart[(size_t)tk.out * CNP + ra] = v;
                }
            }
        return;
    }
    // D[a][b]: register r of lane l holds row a = 4 r + (l >> 4), column b = l & 15
    {
        double* out = partials + (size_t)tk.out * CNP * CNP;
        const int b = lane & 15;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int a = 4 * rr + (lane >> 4);
            const double v = acc0[rr] + acc1[rr];
            if (a < CNP) {
                if (b < CNP) out[a * CNP + b] = v;
                else if (b == CNP && diag) epart[(size_t)tk.out * CNP + a] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Round 5 also built a ROW-WISE kernel (k_schur_rows: a workgroup owns (camera j, a segment of its records), the j side of every triple from an LDS slab,
// only the k side gathered) through five versions: 1.16 -> 0.95 ms against 0.85 ms for the task kernel above, on both scenes
// (profiles/r05_schur_kernels.txt; docs/HISTORY.md section F).  Two measured facts came out of it and stay: (1) on gfx950 an FP64 matrix instruction
// does NOT run beside VALU work of other waves on the same SIMD -- the times add, and the matrix rate equals the vector FMA rate, so what a 16 x 16 x 4
// tile computes on padding is lost: hence v_mfma_f64_4x4x4_4b above; (2) with every gather served from L1 AND the matrix instructions removed the pass loop
// still takes 0.63 ms: the kernel is bound by its VALU / LDS instruction stream.  The kernel itself, its plan (schur_rows.h) and their tests were REMOVED in
// round 6 (decide once): slower on every scene measured, default-off, 600 lines.

}  // namespace bsfm
