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
// Round 5: M4 = the block sum on v_mfma_f64_4x4x4_4b instead of v_mfma_f64_16x16x4 (see the note at k_schur_rows: FP64 matrix instructions
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
    // is what rounds 3-4 ran with (round 5, found in the disassembly of k_schur_rows).
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
// Round 5: k_schur_rows -- the j side of every triple comes out of LDS, only the k side is gathered (plan: schur_rows.h).
// k_schur_tasks fetches 400 bytes per triple in 16-byte chunks (25 chunks: A_ij, A_ik, C_ij || r_ij, B_ik) and every wave starts with a
// chain of dependent index loads (task, triple list, records).  208 of the 400 bytes are the j side, and a point seen by d cameras has
// its record (i, j) gathered again for every partner k >= j.  Here a workgroup owns (camera j, a segment of <= L consecutive records of
// j) and
//   * streams, ONCE and fully coalesced, everything it is going to index with into LDS: the segment's A_ij chunks and C_ij || r_ij
//     records (the slab, row stride CNP + 4 chunks), its own stretch of the row triple array (slab row of the j side, record of the k
//     side; every piece padded to whole passes of 16) and its piece headers -- one barrier, after which no wave waits for an index again;
//   * each of its four waves then runs through its passes -- contiguous runs dealt out by the host so that the waves carry equal
//     numbers -- like a task of k_schur_tasks, except that
//       - only A_ik (CNP chunks) and B_ik (3 chunks) are gathered into registers: 12 chunks per triple instead of 25, prefetched one
//         pass ahead ACROSS piece boundaries (the pass list of a wave is flat);
//       - C_ij comes from the slab (one 16-byte LDS read per lane);
//       - the X operand of the matrix instructions is read straight from the slab row of the triple's j-side record (the slab IS the
//         operand layout: chunk c of a row = (A[0][c], A[1][c])), so a pass writes only the Y slab (16 rows x 160 bytes per wave).
//     Lanes whose output row / column does not exist read a clamped chunk: D[a][b] = sum_k X[k][a] Y[k][b], so a junk operand
//     column only reaches output entries that are never stored.
// Partial sums go to the piece's slot; k_schur_assemble adds a block's slots in slot order (segment after segment): deterministic.
// (First version, profiles/r05_schur_rows_v1.txt: triple lists and piece headers fetched piece by piece -- 300 000 pieces x three
// dependent global round trips at 12 waves per CU: 1.16 ms against 0.91 ms for the task kernel.)
// Matrix instruction (round 5, measured on the box by scripts/r5/ubench_coexec.hip): on gfx950 an FP64 matrix instruction does NOT run
// beside VALU work of other waves -- mfma_f64_16x16x4 (64 cycles) + v_fma_f64 / v_mov_dpp streams of a second wave on the same SIMD take
// the SUM of their times -- and its rate equals the vector FMA rate, so what a matrix instruction computes on padding is simply lost.
// D = X^T Y is 9 x 9 (10 columns with the right-hand side): on 16 x 16 x 4 tiles 32 % of the multiplies are useful, 512 cycles per
// pass of 16 triples.  v_mfma_f64_4x4x4_4b (four independent 4 x 4 x 4 blocks, 16 cycles) covers the 12 x 12 hull with 3 x 3 block
// products: the four blocks of an instruction take four different PAIRS OF TRIPLES (k = 2 t + h: triple t of the pair, image row h), so
// all four multiply the same (row block, column block) and a half pass of 8 triples is 9 instructions on 9 independent accumulators
// from 3 + 3 operand reads of 8 bytes per lane -- 288 cycles per pass (56 % useful), 6 KB of LDS reads instead of 8.  The four partial
// blocks (one per pair) are added across the lanes when a piece ends (two DPP row shifts per accumulator).
//   lane l = 16 k + 4 g + r:  A operand = X[triple 2 g + (k >> 1)][row k & 1][4 ai + r],  B operand = Y[same triple][k & 1][4 bj + r],
//   D[g][i][j] at lane 16 i + 4 g + j  (potrf.hip.h, scripts/probe_mfma4.hip).
// Software pipeline of a wave (round 5, after profiles/r05_schur_rows_diagnosis.txt: with the gathers served from L1 AND the matrix
// instructions removed the kernel still took 0.87 of 0.96 ms -- neither was the bound; a pass was a chain of dependent LDS round trips
// (triple entry -> C_ij -> 2 x 2 core -> Y write -> operand reads -> matrix instructions) at three to four waves per SIMD).  Every
// iteration now works on THREE passes whose stages do not depend on each other, so the LDS and memory latencies of one hide behind the
// arithmetic of the others:
//   consume(n)     operand reads of pass n (X rows through the slab index read one iteration earlier, Y rows written one iteration
//                  earlier) are ISSUED first;
//   issue(n + 2)   triple entries (read one iteration earlier) -> the k-side gathers of pass n + 2;
//   produce(n + 1) C_ij from the slab, the gathers issued one iteration earlier, 2 x 2 core, Y rows written -- into the SAME Y buffer the
//                  operand reads above were issued on: LDS operations of a wave execute in order, so the reads return pass n's rows;
//   consume(n)     the matrix instructions on the operands, and the piece bookkeeping.
// Nothing pins the order with memory clobbers or hand-written waits any more: the compiler counts the LDS and memory operations in
// flight itself (all of them are unconditional inside the loop body).
template <int CNP>
__global__ __launch_bounds__(256, 3) void k_schur_rows(DevProblem P, const RowWG* __restrict__ wgs, const RowPiece* __restrict__ pieces,
        const int2* __restrict__ rtri, double* __restrict__ partials, double* __restrict__ epart, int slab_chunks)
{
    typedef double d2_ __attribute__((ext_vector_type(2)));
    constexpr int RS = CNP + 4;                       // chunks per slab row: A_ij (CNP), C_ij || r_ij (4)
    constexpr int NAI = (CNP + 3) / 4;                // 4-column blocks of X (output rows)
    constexpr int NBJ = (CNP + 1 + 3) / 4;            // ... of Y (output columns incl. the right-hand-side column CNP)
    constexpr int YS = 4 * NBJ;                       // chunks per Y row
    constexpr bool THIRD = CNP > 8;
    constexpr bool ALLC = CNP >= 8;
    static_assert(4 * NAI <= RS, "the operand reads of the last row block stay inside the slab row");
    extern __shared__ __attribute__((aligned(16))) unsigned char row_dyn[];      // the slab (L x RS chunks), then the triple entries
    __shared__ __attribute__((aligned(16))) d2_ smy[ROW_NW][SCM_PASS * YS];
    __shared__ RowPiece sm_pc[ROW_PMAX];
    d2_* Xs = reinterpret_cast<d2_*>(row_dyn);
    int2* tq = reinterpret_cast<int2*>(row_dyn + (size_t)slab_chunks * 16);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int* hdr = reinterpret_cast<const int*>(wgs + blockIdx.x);      // (read field by field: indexing arrays of a by-value copy with the wave number puts it in scratch)
    const int rec0 = hdr[0], nrec = hdr[1], tri0 = hdr[2], npass_wg = hdr[3], piece0 = hdr[4], npieces = hdr[5];
    const d2_* Ac2 = reinterpret_cast<const d2_*>(P.Ac);
    const d2_* Bc2 = reinterpret_cast<const d2_*>(P.Bc);
    {   // the segment's records (both streams contiguous), the workgroup's triple entries, its piece headers: ALL loads first, then
        // the LDS stores (a load-store-load-store loop pays the memory latency once per trip)
        constexpr int NA = (ROW_LMAX * CNP + 255) / 256, NC = (ROW_LMAX * 4 + 255) / 256, NT = 4;      // trips at the largest segment
        const d2_* srcA = Ac2 + (size_t)rec0 * CNP;
        const d2_* srcC = reinterpret_cast<const d2_*>(P.Cc) + (size_t)rec0 * 4;
        const int4* srcT = reinterpret_cast<const int4*>(rtri + tri0);          // entries in pairs (16 bytes; tri0 is a multiple of 16)
        const int nA = nrec * CNP, nC = nrec * 4, nT = npass_wg * (SCM_PASS / 2);
        d2_ va[NA], vc[NC]; int4 vt[NT];
#pragma unroll
        for (int u = 0; u < NA; ++u) { const int c = threadIdx.x + 256 * u; va[u] = srcA[min(c, nA - 1)]; }
#pragma unroll
        for (int u = 0; u < NC; ++u) { const int c = threadIdx.x + 256 * u; vc[u] = srcC[min(c, nC - 1)]; }
#pragma unroll
        for (int u = 0; u < NT; ++u) { const int c = threadIdx.x + 256 * u; vt[u] = srcT[min(c, nT - 1)]; }
        RowPiece pc = pieces[piece0 + min((int)threadIdx.x, npieces - 1)];
#pragma unroll
        for (int u = 0; u < NA; ++u) { const int c = threadIdx.x + 256 * u; if (c < nA) { const int r = c / CNP, part = c - r * CNP; Xs[r * RS + part] = va[u]; } }
#pragma unroll
        for (int u = 0; u < NC; ++u) { const int c = threadIdx.x + 256 * u; if (c < nC) Xs[(c >> 2) * RS + CNP + (c & 3)] = vc[u]; }
#pragma unroll
        for (int u = 0; u < NT; ++u) { const int c = threadIdx.x + 256 * u; if (c < nT) reinterpret_cast<int4*>(tq)[c] = vt[u]; }
        for (int c = threadIdx.x + 256 * NT; c < nT; c += 256) reinterpret_cast<int4*>(tq)[c] = srcT[c];        // (a triple budget above 2 048 entries)
        if ((int)threadIdx.x < npieces) sm_pc[threadIdx.x] = pc;
    }
    d2_* Y = smy[wave];
    for (int t = lane; t < SCM_PASS * YS; t += 64) { const d2_ z = { 0.0, 0.0 }; Y[t] = z; }
    __syncthreads();
    // everything that steers the pass loop is wave-uniform: kept in SGPRs (v_readfirstlane) so that the loop and the piece boundaries are
    // scalar branches
    const int g0 = __builtin_amdgcn_readfirstlane(hdr[6 + wave]);
    const int g1 = __builtin_amdgcn_readfirstlane(wave + 1 < ROW_NW ? hdr[7 + wave] : npass_wg);
    if (g0 >= g1) return;
    // two cursors over the wave's pieces: the producer (pass n + 1) needs to know whether its pass belongs to a diagonal block, the
    // consumer (pass n) where the partial sum goes
    int pi = __builtin_amdgcn_readfirstlane(hdr[6 + ROW_NW + wave]), ppi = pi;
    int left, pleft, cdiag, cout_, pdiag;
    { const RowPiece t_ = sm_pc[pi]; left = __builtin_amdgcn_readfirstlane(t_.npass); cdiag = __builtin_amdgcn_readfirstlane(t_.diag);
      cout_ = __builtin_amdgcn_readfirstlane(t_.out); pleft = left; pdiag = cdiag; }

    const int p = lane >> 2, q = lane & 3;            // producer role: triple of the pass, lane of its quad
    const int c0 = q + 4 * (p & 1), c1 = c0 ^ 4;      // this lane's chunks of A_ik
    const int l0 = c0 < CNP ? c0 : CNP - 1, l1 = c1 < CNP ? c1 : CNP - 1;
    // the twelve 16-byte chunks of the k side (CNP of A_ik, 3 of B_ik) are THREE loads per lane: lane q takes A chunks c0 and c1 and, as
    // its third, A chunk 8 (q = 0) or B chunk q - 1 (q = 1 .. 3)
    const bool third_is_a = q == 0;
    const int l2 = third_is_a ? (CNP - 1) : (q - 1);
    // matrix-operand role: k = lane >> 4 (triple of the pair, image row), g = pair, r = column inside the 4-column block
    const int mt = 2 * ((lane >> 2) & 3) + (lane >> 5);                    // triple of the half pass this lane feeds: 2 g + (k >> 1)
    const int mofs = (lane & 3) * 16 + ((lane >> 4) & 1) * 8;              // byte offset inside a row: chunk r, image row k & 1
    const unsigned char* Yb = reinterpret_cast<const unsigned char*>(Y) + mt * (YS * 16) + mofs;
    const unsigned char* Xb = row_dyn + mofs;
    double acc[NAI][NBJ];
#pragma unroll
    for (int a = 0; a < NAI; ++a)
#pragma unroll
        for (int b = 0; b < NBJ; ++b) acc[a][b] = 0.0;
    // THREE register sets of gathers: pass m sits in set m % 3; two passes are in flight while a third is turned into Y rows
    d2_ ak[3][3];
    int lj[3];
    int2 ent;                                         // triple entry of the NEXT pass to issue (read one stage early)
    int lx0, lx1;                                     // slab rows of the operand reads of the pass to consume
#define BSFM_ROW_ISSUE(S_, gnext_)                    /* gathers of the pass whose entry is in `ent`; then the entry of pass gnext_ */ \
    {                                                                                                               \
        const d2_* rb_ = Ac2 + (size_t)ent.y * CNP;                                                                 \
        const d2_* r3_ = third_is_a ? rb_ : Bc2 + (size_t)ent.y * 4;                                                \
        ak[S_][0] = rb_[l0]; ak[S_][1] = rb_[l1]; ak[S_][2] = r3_[l2];                                              \
        lj[S_] = ent.x;                                                                                             \
        ent = tq[(gnext_) * SCM_PASS + p];                                                                          \
    }
#define BSFM_ROW_PRODUCE(S_)                          /* Y rows of the pass whose gathers sit in set S_ */            \
    {                                                                                                               \
        const bool live_ = lj[S_] < ROW_DEAD;                                                                       \
        const d2_ cj_ = Xs[(lj[S_] & (ROW_DEAD - 1)) * RS + CNP + q];           /* chunk q of C_ij || r_ij */          \
        const double C00 = quad_bcast<0>(cj_.x), C01 = quad_bcast<0>(cj_.y), C02 = quad_bcast<1>(cj_.x);            \
        const double C10 = quad_bcast<1>(cj_.y), C11 = quad_bcast<2>(cj_.x), C12 = quad_bcast<2>(cj_.y);            \
        const double B00 = quad_bcast<1>(ak[S_][2].x), B01 = quad_bcast<1>(ak[S_][2].y), B02 = quad_bcast<2>(ak[S_][2].x); \
        const double B10 = quad_bcast<2>(ak[S_][2].y), B11 = quad_bcast<3>(ak[S_][2].x), B12 = quad_bcast<3>(ak[S_][2].y); \
        double m00 = C00 * B00 + C01 * B01 + C02 * B02, m01 = C00 * B10 + C01 * B11 + C02 * B12;                    \
        double m10 = C10 * B00 + C11 * B01 + C12 * B02, m11 = C10 * B10 + C11 * B11 + C12 * B12;                    \
        if (!live_) { m00 = 0.0; m01 = 0.0; m10 = 0.0; m11 = 0.0; }                                                 \
        d2_* yr_ = Y + p * YS;                                                                                      \
        if (ALLC || c0 < CNP) { const d2_ y_ = { m00 * ak[S_][0].x + m01 * ak[S_][0].y, m10 * ak[S_][0].x + m11 * ak[S_][0].y }; yr_[c0] = y_; } \
        if (ALLC || c1 < CNP) { const d2_ y_ = { m00 * ak[S_][1].x + m01 * ak[S_][1].y, m10 * ak[S_][1].x + m11 * ak[S_][1].y }; yr_[c1] = y_; } \
        if (THIRD && q == 0) { const d2_ y_ = { m00 * ak[S_][2].x + m01 * ak[S_][2].y, m10 * ak[S_][2].x + m11 * ak[S_][2].y }; yr_[CNP - 1] = y_; } \
        if (pdiag && q == 3) { const d2_ r_ = { live_ ? cj_.x : 0.0, live_ ? cj_.y : 0.0 }; yr_[CNP] = r_; }        \
        if (--pleft == 0 && ppi + 1 < npieces) { const RowPiece t_ = sm_pc[++ppi]; pleft = __builtin_amdgcn_readfirstlane(t_.npass); pdiag = __builtin_amdgcn_readfirstlane(t_.diag); } \
    }
#define BSFM_ROW_OPERANDS()                           /* operand reads of the pass to consume: slab rows lx0 / lx1, the Y rows as they are NOW */ \
        double xa_[2][NAI], yb_[2][NBJ];                                                                            \
        _Pragma("unroll") for (int a = 0; a < NAI; ++a) {                                                           \
            xa_[0][a] = *reinterpret_cast<const double*>(Xb + lx0 * (RS * 16) + 64 * a);                            \
            xa_[1][a] = *reinterpret_cast<const double*>(Xb + lx1 * (RS * 16) + 64 * a); }                          \
        _Pragma("unroll") for (int b = 0; b < NBJ; ++b) {                                                           \
            yb_[0][b] = *reinterpret_cast<const double*>(Yb + 64 * b);                                              \
            yb_[1][b] = *reinterpret_cast<const double*>(Yb + 8 * (YS * 16) + 64 * b); }
#define BSFM_ROW_CONSUME(g_)                          /* matrix instructions of pass g_ on xa_ / yb_, piece bookkeeping */ \
    {                                                                                                               \
        _Pragma("unroll") for (int hp = 0; hp < 2; ++hp)                                                            \
            _Pragma("unroll") for (int a = 0; a < NAI; ++a)                                                         \
                _Pragma("unroll") for (int b = 0; b < NBJ; ++b)                                                     \
                    acc[a][b] = __builtin_amdgcn_mfma_f64_4x4x4f64(xa_[hp][a], yb_[hp][b], acc[a][b], 0, 0, 0);     \
        if (--left == 0) {                                       /* the piece is complete: its partial sum goes to its slot */ \
            double* out_ = partials + (size_t)cout_ * CNP * CNP;                                                    \
            const int i_ = lane >> 4, j_ = lane & 3;                                                                \
            const bool own_ = (lane & 12) == 12;                 /* the lanes of pair g = 3 end up with the sum over the four pairs */ \
            _Pragma("unroll") for (int a = 0; a < NAI; ++a)                                                         \
                _Pragma("unroll") for (int b = 0; b < NBJ; ++b) {                                                   \
                    double v_ = acc[a][b];                                                                          \
                    v_ += row_shr_d<4>(v_);                                                                         \
                    v_ += row_shr_d<8>(v_);                                                                         \
                    const int ra_ = 4 * a + i_, cb_ = 4 * b + j_;                                                   \
                    if (own_ && ra_ < CNP) {                                                                        \
                        if (cb_ < CNP) out_[ra_ * CNP + cb_] = v_;                                                  \
                        else if (cb_ == CNP && cdiag) epart[(size_t)cout_ * CNP + ra_] = v_;                        \
                    }                                                                                               \
                    acc[a][b] = 0.0;                                                                                \
                }                                                                                                   \
            if ((g_) + 1 < g1) { const RowPiece t_ = sm_pc[++pi]; left = __builtin_amdgcn_readfirstlane(t_.npass);  \
                                 cdiag = __builtin_amdgcn_readfirstlane(t_.diag); cout_ = __builtin_amdgcn_readfirstlane(t_.out); } \
        }                                                                                                           \
    }
#define BSFM_ROW_LX(g_) { lx0 = tq[(g_) * SCM_PASS + mt].x & (ROW_DEAD - 1); lx1 = tq[(g_) * SCM_PASS + 8 + mt].x & (ROW_DEAD - 1); }
#define BSFM_ROW_STEP(g_, SP_, SI_)                   /* consume pass g_, produce pass g_ + 1 (set SP_), issue pass g_ + 3 (set SI_) */ \
    {                                                                                                               \
        BSFM_ROW_OPERANDS()                                                                                         \
        BSFM_ROW_ISSUE(SI_, min((g_) + 4, g1 - 1))                                                                  \
        BSFM_ROW_LX(min((g_) + 1, g1 - 1))                                                                          \
        BSFM_ROW_PRODUCE(SP_)                                                                                       \
        BSFM_ROW_CONSUME(g_)                                                                                        \
    }

    // prologue: gathers of the first three passes in flight, Y rows of the first pass written.  Past the wave's last pass the pipeline
    // repeats it (clamped indices) and never consumes the result.  All loads of the loop body are unconditional: behind a branch the
    // compiler cannot count the loads in flight and waits for all of them.
    ent = tq[g0 * SCM_PASS + p];
    BSFM_ROW_ISSUE(0, min(g0 + 1, g1 - 1))
    BSFM_ROW_ISSUE(1, min(g0 + 2, g1 - 1))
    BSFM_ROW_ISSUE(2, min(g0 + 3, g1 - 1))
    BSFM_ROW_LX(g0)
    BSFM_ROW_PRODUCE(0)
    for (int g = g0; g < g1; g += 3) {
        BSFM_ROW_STEP(g, 1, 0)                                   // (set 0 held pass g: produced in the previous step)
        if (g + 1 < g1) BSFM_ROW_STEP(g + 1, 2, 1)
        if (g + 2 < g1) BSFM_ROW_STEP(g + 2, 0, 2)
    }
#undef BSFM_ROW_STEP
#undef BSFM_ROW_ISSUE
#undef BSFM_ROW_PRODUCE
#undef BSFM_ROW_OPERANDS
#undef BSFM_ROW_CONSUME
#undef BSFM_ROW_LX
}

}  // namespace bsfm
