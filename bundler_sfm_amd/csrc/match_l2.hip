// match_l2.hip -- brute-force SIFT descriptor matcher for gfx950 (replaces the ANN kd-tree of KeyMatchFull).
//
// Reference behaviour restated (paths relative to the reference tree):
//   MatchKeys            src/keys2a.cpp:347-372      2-NN of every key of image j among the keys of image i,
//                                                    keep (q, nn0) iff (double)d0 < ratio*ratio*(double)d1
//   ANN distance         lib/ann_1.1_char/include/ANN/ANN.h:161-162   squared L2 over 128 uchar in int32
//   KeyMatchFull main    src/KeyMatchFull.cpp:105-151                  pair loop + "j i / N / idx idx" text format
// The reference searches approximately (priority search capped at 200 visited points); this kernel is the exact
// search that approximation converges to (annMaxPtsVisit(0), eps = 0): distances are exact integers, so the match
// list is bit-identical to the exact reference search (ties between the two nearest can never pass the strict test).
//
// MI355X design: a dense int8 contraction on v_mfma_i32_16x16x64_i8 with exact int32 accumulation.
//   * uchar -> int8 by XOR 0x80 (x - 128) in registers; d = qa + qb - 2 * dot(a', b') with the per-key terms
//     qa = |a|^2 - 256 sum(a') - 2*128^3, qb = |b|^2 - 256 sum(b') precomputed once per key (k_key_stats);
//   * a workgroup owns 128 queries (8 row-groups x 2 k-steps of A fragments live in registers for the whole scan);
//     each of its 4 waves streams a different 16-key slice of every 64-key database tile straight from global
//     memory into B fragments (a 640 KB image stays L2-resident; no LDS staging needed: a B fragment is consumed
//     by exactly one wave, 16 MFMAs per fragment pair);
//   * every lane keeps a branch-free running top-2 on packed (distance, tile) keys for its 32 (row, column-class)
//     slots; the 16 column classes of a row are merged with wave shuffles, the four waves through LDS, then the ratio
//     test runs in FP64 exactly as the reference writes it.
//   * KeyMatchFull: all pairs (j < i) of one database image i go into ONE launch (grid = sum of query blocks).
// Two scan kernels give that same table: k_match_l2 (above: exact running top-2, 3 VALU per distance, 128 queries per workgroup)
// and k_match_bound (round 3: 1 VALU per distance, 256 queries per workgroup, bounds from the scan + exact rescan of the winning
// slot); match_set_run picks one per launch, k_pair_counts / k_pair_write hand the accepted matches to the host compacted.
#include <hip/hip_runtime.h>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../../include/bsfm.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int QB = 128;            // queries per workgroup
constexpr int BIG = 0x3fffffff;
constexpr size_t KEY_SLACK = 64;        // keys of slack behind the last image: the scan prefetches whole tiles (k_match_l2)

#define HIPM(call)                                                                                   \
    do { hipError_t _e = (call); if (_e != hipSuccess) {                                             \
        fprintf(stderr, "[bsfm] HIP error %s at %s:%d\n", hipGetErrorName(_e), __FILE__, __LINE__); \
        return BSFM_ERROR; } } while (0)

// per key: q = |x|^2 - 256 * sum(x - 128)   (the query side subtracts the constant 2*128^3 later).  The descriptor is rewritten
// IN PLACE as signed bytes x - 128 (one xor per word), the operand format of the matrix instruction: the scan kernel then
// feeds what it loads straight into the MFMA (the flip was 8 VALU instructions per wave and tile of the VALU-bound scan).
__global__ void k_key_stats(unsigned char* __restrict__ keys, int n, int* __restrict__ q)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint4* p = reinterpret_cast<uint4*>(keys + (size_t)i * 128);
    int sq = 0, s = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const uint4 v = p[w];
        const unsigned u[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int b = 0; b < 4; ++b) { const int x = (u[c] >> (8 * b)) & 255; sq += x * x; s += x - 128; }
        p[w] = make_uint4(v.x ^ 0x80808080u, v.y ^ 0x80808080u, v.z ^ 0x80808080u, v.w ^ 0x80808080u);
    }
    q[i] = sq - 256 * s;
}

struct PairDesc { int q_off, q_n, out_off, blk0; };   // query image: key offset / count; output offset; first block

__device__ __forceinline__ v4i load_frag(const unsigned char* __restrict__ keys, int key, int lane, int kstep)
{
    const uint4 v = *reinterpret_cast<const uint4*>(keys + (size_t)key * 128 + 64 * kstep + 16 * (lane >> 4));
    v4i r;
    r.x = (int)v.x; r.y = (int)v.y; r.z = (int)v.z; r.w = (int)v.w;      // already signed bytes (k_key_stats)
    return r;
}

__device__ __forceinline__ uint4 load_raw(const unsigned char* __restrict__ keys, int key, int lane, int kstep)
{
    return *reinterpret_cast<const uint4*>(keys + (size_t)key * 128 + 64 * kstep + 16 * (lane >> 4));
}
__device__ __forceinline__ v4i flip(const uint4 v)          // the loaded words as the MFMA operand (signed bytes already)
{
    v4i r;
    r.x = (int)v.x; r.y = (int)v.y; r.z = (int)v.z; r.w = (int)v.w;
    return r;
}

// nn_out[out_off + query] = index of the accepted nearest neighbour in the database image, or -1.
__global__ __launch_bounds__(256, 2) void k_match_l2(const unsigned char* __restrict__ keys, const int* __restrict__ qstat,
        const PairDesc* __restrict__ pairs, int npairs, int db_off, int db_n, double ratio_sq, int* __restrict__ nn_out, int one)
{
    constexpr int NG = QB / 16;                       // row groups of 16 queries held by every wave
    constexpr int XS = 65;                            // row stride (uint2) of the exchange buffer: 64 column classes + 1 pad
    __shared__ uint2 xch[QB * XS];                    // per segment: every (query row, column class) slot's packed (best, second)
    // Locate the pair this block belongs to: a binary search on the pairs' first-block numbers that every thread runs on
    // uniform addresses (scalar loads, no barrier; one thread searching + a barrier kept 255 threads idle for 9 round trips).
    int s_pair = 0;
    {
        int lo = 0, hi = npairs - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (pairs[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
        s_pair = lo;
    }
    const PairDesc pd = pairs[s_pair];
    const int qbase = (blockIdx.x - pd.blk0) * QB;
    // running (nearest, second nearest, column) of query row threadIdx.x / 2, kept by the two threads that merge that row
    const int row_qa = qstat[pd.q_off + min(qbase + (int)(threadIdx.x >> 1), pd.q_n - 1)] - 2 * 128 * 128 * 128;
    int row_d0 = BIG, row_d1 = BIG, row_idx = -1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char* qkeys = keys + (size_t)pd.q_off * 128;
    const unsigned char* dkeys = keys + (size_t)db_off * 128;

    // A fragments: NG row groups x 2 k-steps stay in registers for the whole scan.  QB = 128 queries per workgroup: every
    // database fragment a wave fetches from L2 is used for 8 MFMA pairs -- the scan is bound by that L2 -> CU traffic
    // (128 MAC per byte at QB = 128), it was 8.0 us per 5000 x 5000 pair with 64 queries per workgroup.
    v4i afrag[NG][2];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int row = min(qbase + 16 * g + (lane & 15), pd.q_n - 1);
        afrag[g][0] = load_frag(qkeys, row, lane, 0);
        afrag[g][1] = load_frag(qkeys, row, lane, 1);
    }
    // Running top-2 per slot on PACKED keys.  Ranking needs only e = qb - 2 dot (the query term qa is the same for every
    // candidate of a row), so the per-distance work is ONE 24-bit multiply-add that builds the key (below), one v_min_u32
    // for the best and one v_med3_u32 for the second (for b0 <= b1 the new second is the median of b0, b1, key).
    // BIAS makes e non-negative for every possible key pair: qa = |a|^2 - 256 sum(a - 128) - 2*128^3 lies in
    // [-8 355 840, 8 323 200], d = qa + e in [0, 8 323 200], so e + BIAS < 2^25 with BIAS = 8 388 608, which leaves
    // TB = 7 bits for the tile number inside a segment of 128 tiles (8 192 database keys; larger images are scanned
    // segment by segment).  Equal distances order by tile = by column like the reference's first-found rule, and two equal
    // nearest distances can never pass the strict ratio test anyway.
    constexpr int TB = 7;
    constexpr int SEG = 64 << TB;
    // -(2 << TB) reaches the kernel as data (`one` == 1) so that the compiler keeps the 24-bit multiply-add: with a literal
    // power of two it strength-reduces it to shift + subtract, one VALU instruction more per distance
    const int mscale = -(2 << TB) * one;
    constexpr int EBIAS = 1 << 23;
    // padding columns (past the end of the database) carry qb + BIAS = DEADQ: with |2 dot| <= 4 194 304 their e + BIAS stays
    // above the largest real value (8 323 200 + 8 355 840 + 8 388 608 = 25 067 648) and below 2^25 -- no select per distance
    constexpr int DEADQ = 29300000;
    constexpr unsigned DEADTHR = 25100000u;
    for (int seg = 0; seg < db_n; seg += SEG) {
        unsigned b0[NG][4], b1[NG][4];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) { b0[g][r] = 0xffffffffu; b1[g][r] = 0xffffffffu; }
        const int seg_end = min(db_n, seg + SEG);
        // software pipeline: the B fragments of tile t+1 are in flight while tile t is multiplied and ranked.  The loads stay RAW
        // (no sign flip, no select on the loaded statistic) until the top of the next trip: anything computed from them here
        // makes the compiler wait for the data before the current tile's MFMAs, which serialises L2 latency and arithmetic
        // (round 2: that was half of the kernel's time)
        // (running pointers, no clamp: the key and statistic arrays carry one tile of slack behind the last image, and what
        //  is loaded past the database image only ever meets DEADQ)
        const int colf = seg + 16 * wave + (lane & 15);
        const unsigned char* kp = dkeys + (size_t)colf * 128 + 16 * (lane >> 4);
        const int* qp = qstat + db_off + colf;
        uint4 rf0 = *reinterpret_cast<const uint4*>(kp), rf1 = *reinterpret_cast<const uint4*>(kp + 64);
        int rqb = *qp;
        int coln = colf;
        for (int tile = seg; tile < seg_end; tile += 64) {
            const v4i bf0 = flip(rf0), bf1 = flip(rf1);
            const int qbb = coln < db_n ? rqb + EBIAS : DEADQ;
            if (tile + 64 < seg_end) {
                kp += 64 * 128; qp += 64; coln += 64;
                rf0 = *reinterpret_cast<const uint4*>(kp); rf1 = *reinterpret_cast<const uint4*>(kp + 64);
                rqb = *qp;
            }
            // key = ((qb + BIAS - 2 dot) << TB) | tile = K - (dot << (TB + 1)) with K = ((qb + BIAS) << TB) | tile: the low TB
            // bits are untouched by the subtraction, so ONE 24-bit multiply-add per distance builds the packed key
            const int K = (int)(((unsigned)qbb << TB) | (unsigned)((tile - seg) >> 6));
            // Group g + 1 is multiplied while group g is ranked: each matrix instruction is followed by six independent VALU
            // instructions (the order is pinned with sched_barrier), enough to cover its 4 passes, so neither the dependent
            // second k-step nor the ranking ever waits for the matrix pipe.
            v4i acc = { 0, 0, 0, 0 };
            acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(afrag[0][0], bf0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(afrag[0][1], bf1, acc, 0, 0, 0);
#define BSFM_RANK(R)                                                                                                      \
                {   /* (no inline asm on the accumulator itself: the compiler must see the MFMA -> VALU dependency) */     \
                    const unsigned key = (unsigned)(__mul24(acc[R], mscale) + K);        /* v_mad_i32_i24 */              \
                    unsigned m;                                                                                           \
                    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(m) : "v"(b0[g][R]), "v"(b1[g][R]), "v"(key));                  \
                    b1[g][R] = m;                                                                                         \
                    b0[g][R] = min(b0[g][R], key);                                                                        \
                }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                v4i nacc = { 0, 0, 0, 0 };
                if (g + 1 < NG) nacc = __builtin_amdgcn_mfma_i32_16x16x64_i8(afrag[g + 1][0], bf0, nacc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                BSFM_RANK(0) BSFM_RANK(1)
                __builtin_amdgcn_sched_barrier(0);
                if (g + 1 < NG) nacc = __builtin_amdgcn_mfma_i32_16x16x64_i8(afrag[g + 1][1], bf1, nacc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                BSFM_RANK(2) BSFM_RANK(3)
                __builtin_amdgcn_sched_barrier(0);
                acc = nacc;
            }
#undef BSFM_RANK
        }
        // Segment merge through LDS: slot (row, column class 16 wave + lane % 16) goes to xch[row][class]; then two threads per
        // query row scan 32 classes each (6 VALU per class), combine with one DPP swap, and the even thread folds the segment's
        // (best, second, column) into the row's running state, which lives in ITS registers.  (Round 2: the 16-lane butterfly
        // on all 32 slots of every lane cost 1 400 VALU instructions per wave, a tenth of the kernel.)
        __syncthreads();                               // the previous segment's readers are done
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                xch[(16 * g + 4 * (lane >> 4) + r) * XS + 16 * wave + (lane & 15)] = make_uint2(b0[g][r], b1[g][r]);
        __syncthreads();
        {
            const int half = threadIdx.x & 1;
            const uint2* src = xch + (threadIdx.x >> 1) * XS + 32 * half;
            unsigned k0 = 0xffffffffu, k1 = 0xffffffffu; int cls = 0;
#pragma unroll 8
            for (int e = 0; e < 32; ++e) {
                const uint2 o = src[e];
                k1 = min(max(k0, o.x), min(k1, o.y));
                cls = o.x < k0 ? 32 * half + e : cls;
                k0 = min(k0, o.x);
            }
            {   // the other half of the row sits in the neighbouring lane
                const unsigned o0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)k0, 0xB1, 0xf, 0xf, false);
                const unsigned o1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)k1, 0xB1, 0xf, 0xf, false);
                const int oc = __builtin_amdgcn_update_dpp(0, cls, 0xB1, 0xf, 0xf, false);
                k1 = min(max(k0, o0), min(k1, o1));
                cls = o0 < k0 ? oc : cls;
                k0 = min(k0, o0);
            }
            const unsigned e0 = k0 >> TB, e1 = k1 >> TB;
            const int c0 = e0 >= DEADTHR ? BIG : (int)e0 - EBIAS + row_qa;
            const int c1 = e1 >= DEADTHR ? BIG : (int)e1 - EBIAS + row_qa;
            const int ci = seg + (int)((k0 & ((1u << TB) - 1)) << 6) + cls;
            if (c0 < row_d0) { row_d1 = min(row_d0, c1); row_d0 = c0; row_idx = ci; }
            else row_d1 = min(row_d1, c0);
        }
    }
    if ((threadIdx.x & 1) == 0) {
        const int q = qbase + (threadIdx.x >> 1);
        if (q < pd.q_n) {
            bool ok;
            {
#pragma clang fp contract(off)
                ok = ((double)row_d0) < ratio_sq * ((double)row_d1);      // src/keys2a.cpp:362
            }
            nn_out[pd.out_off + q] = ok ? row_idx : -1;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_match_bound (round 3): the same exact search at ONE VALU instruction per distance and 256 query rows per workgroup.
//
// d = qa + qb - 2 dot.  With c = ceil(qb / 2) and p = qb & 1:  -(qb - 2 dot) = 2 (dot - c) + p, so t = dot - c is the distance up
// to the parity bit: d = qa - 2 t - p.  -c rides in as the C operand of the first matrix instruction (one register quad per tile),
// the accumulator IS t, and a slot (query row, column class mod 64, 8 192-key segment) keeps max t with one v_max_i32: no key
// packing, no tile number, no second best -- which also frees the registers for 16 row groups of A fragments per wave (half the
// database fragments per MAC, what the scan is bound by).  What the scan leaves per row is therefore a BOUND, made exact afterwards:
//   * T0 = the largest slot maximum, T1 = the second largest (equal to T0 when two slots tie).  The nearest distance lies in
//     [qa - 2 T0 - 1, qa - 2 T0] and the second nearest is at most qa - 2 T1.  A row whose lower bound fails the ratio test against
//     that upper bound fails it for good (most rows): -1.
//   * The others get the <= 128 columns of the winning slot measured exactly (v_dot4_i32_i8): nearest E0, its column, second E1 in
//     the slot.  With T1 < T0 every other slot is at least qa - 2 T1 - 1 > E0, so E0 is THE nearest, and the second nearest is
//     min(E1, best of the second slot), the latter known to within one unit: if the ratio test gives the same answer at both ends
//     of that unit it is decided;
//   * the rest (T1 == T0, or a test that hinges on the parity bit of the second slot: ~1e-5 of the rows) are measured against the
//     whole database image exactly, one row at a time by the whole workgroup.
// The fragment loads are written by hand two tiles ahead (three rotating register sets, loop unrolled three times, hand-counted
// vmcnt): the compiler hoists, clusters or copies loads it manages itself, and every such attempt ended in a vmcnt(0) or in a copy
// of a register that was still being loaded.
constexpr int QBB = 256;                 // query rows per workgroup
constexpr int BSEG = 8192;               // database keys per segment = 128 tiles: bounds a slot (and a rescan) at 128 columns
constexpr int DEADC = -(1 << 29);        // C operand of padding columns: below every real t (|dot| <= 2^21, |c| <= 2^23)
constexpr int TMIN = -(1 << 30);

__global__ __launch_bounds__(256, 2) void k_match_bound(const unsigned char* __restrict__ keys, const int* __restrict__ qstat,
        const PairDesc* __restrict__ pairs, int npairs, int db_off, int db_n, double ratio_sq, int* __restrict__ nn_out)
{
    constexpr int NG = QBB / 16;                      // row groups of 16 queries held by every wave
    constexpr int XS = 65;                            // row stride of the exchange buffer: 64 column classes + 1 pad
    constexpr int ITEM_CAP = 128;                     // rescans per pass: 128 x 128 distances fit the exchange buffer
    __shared__ int xch[QBB * XS];                     // per segment: every (query row, column class) slot's max t; later the rescan's distances
    __shared__ int s_cnt, s_full, s_row[QBB], s_slot[QBB], s_res[QBB][3], s_frow[QBB];
    __shared__ unsigned long long s_best;
    __shared__ int s_second;
    int s_pair = 0;
    {   // block -> pair: binary search on uniform addresses (scalar loads)
        int lo = 0, hi = npairs - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (pairs[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
        s_pair = lo;
    }
    const PairDesc pd = pairs[s_pair];
    const int qbase = (blockIdx.x - pd.blk0) * QBB;
    const int my_row = (int)threadIdx.x;              // the query row whose slots this thread merges after every segment
    const bool my_valid = qbase + my_row < pd.q_n;
    const int row_qa = qstat[pd.q_off + min(qbase + my_row, pd.q_n - 1)] - 2 * 128 * 128 * 128;
    int R0 = TMIN, R1 = TMIN, Rslot = 0;              // largest / second largest slot maximum of the row, first column of the winning slot
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char* qkeys = keys + (size_t)pd.q_off * 128;
    const unsigned char* dkeys = keys + (size_t)db_off * 128;

    v4i afrag[NG][2];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int row = min(qbase + 16 * g + (lane & 15), pd.q_n - 1);
        afrag[g][0] = load_frag(qkeys, row, lane, 0);
        afrag[g][1] = load_frag(qkeys, row, lane, 1);
    }
    // the hand-counted waits below assume that nothing of the compiler's own is in flight: make it wait for the A fragments HERE
    // (its bookkeeping does not see the inline-asm loads, and a late compiler wait inside the loop would wait for the prefetches too)
#pragma unroll
    for (int g = 0; g < NG; ++g) asm volatile("" : "+v"(afrag[g][0]), "+v"(afrag[g][1]));

    const unsigned koff = (unsigned)((16 * wave + (lane & 15)) * 128 + 16 * (lane >> 4));
    const unsigned qoff = (unsigned)(4 * (16 * wave + (lane & 15)));
    for (int seg = 0; seg < db_n; seg += BSEG) {
        int b0[NG][4];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) b0[g][r] = TMIN;
        const int seg_end = min(db_n, seg + BSEG);
        const int seg_last = seg + ((seg_end - seg - 1) & ~63);          // first column of the segment's last tile
        const int colf = seg + 16 * wave + (lane & 15);
        // one tile: groups g + 1 and g + 2 are in the matrix pipe while group g is ranked (three accumulators; order pinned with
        // sched_barrier).  With two, the two VALU instructions between a group's second matrix instruction and its ranking did not
        // cover the instruction's latency any more (s_nop 4-5 after every one of them).
        auto rank_tile = [&](const v4i bf0, const v4i bf1, const int cq) __attribute__((always_inline)) {
            const v4i cvec = { cq, cq, cq, cq };
            v4i acc[3];
            acc[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(afrag[0][0], bf0, cvec, 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(afrag[0][1], bf1, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(afrag[1][0], bf0, cvec, 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(afrag[1][1], bf1, acc[1], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int cur = g % 3, nxt = (g + 2) % 3;
                if (g + 2 < NG) acc[nxt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(afrag[g + 2][0], bf0, cvec, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                b0[g][0] = max(b0[g][0], acc[cur][0]); b0[g][1] = max(b0[g][1], acc[cur][1]);
                __builtin_amdgcn_sched_barrier(0);
                if (g + 2 < NG) acc[nxt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(afrag[g + 2][1], bf1, acc[nxt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                b0[g][2] = max(b0[g][2], acc[cur][2]); b0[g][3] = max(b0[g][3], acc[cur][3]);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        v4i x0, x1, y0, y1, z0, z1; int xq, yq, zq;
        // every tile issues exactly three loads (the last ones reload the segment's last tile), so "all but the newest three" is the
        // wait for the set about to be used; the counter is drained before the registers are reused
#define BSFM_ISSUE(S0, S1, SQ, T)                                                                                          \
        {   const int tn = min((T), seg_last);                                                                            \
            const unsigned char* sp = dkeys + (size_t)tn * 128;                                                           \
            const int* sq = qstat + db_off + tn;                                                                          \
            asm volatile("global_load_dwordx4 %0, %3, %4\n\tglobal_load_dwordx4 %1, %3, %4 offset:64\n\t"                \
                         "global_load_dword %2, %5, %6"                                                                   \
                         : "=&v"(S0), "=&v"(S1), "=&v"(SQ) : "v"(koff), "s"(sp), "v"(qoff), "s"(sq) : "memory"); }
#define BSFM_USE(S0, S1, SQ, L0, L1, LQ, T)                                                                                \
        {   asm volatile("s_waitcnt vmcnt(3)" : "+v"(S0), "+v"(S1), "+v"(SQ));                                            \
            const v4i bf0 = S0, bf1 = S1;                                                                                 \
            const int cq = colf + ((T) - seg) < db_n ? -((SQ + 1) >> 1) : DEADC;                                          \
            BSFM_ISSUE(L0, L1, LQ, (T) + 128)                                                                             \
            rank_tile(bf0, bf1, cq); }
        BSFM_ISSUE(x0, x1, xq, seg)
        BSFM_ISSUE(y0, y1, yq, seg + 64)
        for (int tile = seg;;) {
            BSFM_USE(x0, x1, xq, z0, z1, zq, tile) tile += 64; if (tile >= seg_end) break;
            BSFM_USE(y0, y1, yq, x0, x1, xq, tile) tile += 64; if (tile >= seg_end) break;
            BSFM_USE(z0, z1, zq, y0, y1, yq, tile) tile += 64; if (tile >= seg_end) break;
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(x0), "+v"(x1), "+v"(xq), "+v"(y0), "+v"(y1), "+v"(yq), "+v"(z0), "+v"(z1), "+v"(zq));
#undef BSFM_USE
#undef BSFM_ISSUE
        // segment merge through LDS: slot (row, class 16 wave + lane % 16) -> xch[row][class]; one thread per query row then scans
        // the 64 classes and folds (largest, second largest, class of the largest) into the row's running state
        __syncthreads();                               // the previous segment's readers are done
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                xch[(16 * g + 4 * (lane >> 4) + r) * XS + 16 * wave + (lane & 15)] = b0[g][r];
        __syncthreads();
        {
            const int* src = xch + my_row * XS;
            int k0 = TMIN, k1 = TMIN, cls = 0;
#pragma unroll 8
            for (int e = 0; e < 64; ++e) {
                const int o = src[e];
                k1 = max(min(k0, o), k1);
                cls = o > k0 ? e : cls;
                k0 = max(k0, o);
            }
            if (k0 > R0) { R1 = max(R0, k1); R0 = k0; Rslot = seg + cls; }
            else R1 = max(R1, k0);
        }
    }

    // ---- from the bounds to the exact answer
    auto exact_dist = [&](const int q, const int col) {
        const uint4* a = reinterpret_cast<const uint4*>(qkeys + (size_t)q * 128);
        const uint4* b = reinterpret_cast<const uint4*>(dkeys + (size_t)col * 128);
        int dot = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const uint4 x = a[w], y = b[w];
            dot = __builtin_amdgcn_sdot4((int)x.x, (int)y.x, dot, false); dot = __builtin_amdgcn_sdot4((int)x.y, (int)y.y, dot, false);
            dot = __builtin_amdgcn_sdot4((int)x.z, (int)y.z, dot, false); dot = __builtin_amdgcn_sdot4((int)x.w, (int)y.w, dot, false);
        }
        return qstat[pd.q_off + q] - 2 * 128 * 128 * 128 + qstat[db_off + col] - 2 * dot;
    };
    auto passes = [&](const int d0, const int d1) {
        bool ok;
        {
#pragma clang fp contract(off)
            ok = ((double)d0) < ratio_sq * ((double)d1);          // src/keys2a.cpp:362
        }
        return ok;
    };
    if (threadIdx.x == 0) { s_cnt = 0; s_full = 0; }
    __syncthreads();                                   // (also: the last segment's readers of xch are done)
    int result = -1;
    // 1 = may pass: the nearest's lower bound against the second nearest's upper bound (a true pass always lands here)
    const bool maybe = my_valid && passes(row_qa - 2 * R0 - 1, row_qa - 2 * R1);
    int slot = -1;
    if (maybe) { slot = atomicAdd(&s_cnt, 1); s_row[slot] = my_row; s_slot[slot] = Rslot; }
    __syncthreads();
    const int cnt = s_cnt;
    bool need_full = false;
    for (int p0 = 0; p0 < cnt; p0 += ITEM_CAP) {
        const int np = min(cnt - p0, ITEM_CAP);
        // every (item, column of its slot) pair: the exact distance into xch[item][column index]
        for (int it = threadIdx.x; it < np * 128; it += 256) {
            const int p = p0 + (it >> 7), t = it & 127;
            const int s0 = s_slot[p];
            const int col = s0 + 64 * t;
            const int send = min(db_n, (s0 / BSEG) * BSEG + BSEG);
            xch[it] = col < send ? exact_dist(qbase + s_row[p], col) : BIG;
        }
        __syncthreads();
        if ((int)threadIdx.x < np) {                   // one thread per item: nearest, its column index, second nearest of the slot
            const int* dv = xch + 128 * threadIdx.x;
            int e0 = BIG, e1 = BIG, c0 = 0;
            for (int t = 0; t < 128; ++t) {
                const int d = dv[t];
                e1 = min(max(e0, d), e1);
                c0 = d < e0 ? t : c0;
                e0 = min(e0, d);
            }
            s_res[p0 + threadIdx.x][0] = e0; s_res[p0 + threadIdx.x][1] = c0; s_res[p0 + threadIdx.x][2] = e1;
        }
        __syncthreads();
    }
    if (maybe) {
        const int e0 = s_res[slot][0], e1 = s_res[slot][2];
        const int others_hi = row_qa - 2 * R1;         // the best of every other slot lies in [others_hi - 1, others_hi]
        const bool hi = passes(e0, min(e1, others_hi)), lo = passes(e0, min(e1, others_hi - 1));
        if (R1 < R0 && hi == lo) result = hi ? Rslot + 64 * s_res[slot][1] : -1;
        else need_full = true;                         // two slots tie for the nearest, or the test hinges on a parity bit
    }
    if (need_full) s_frow[atomicAdd(&s_full, 1)] = my_row;
    __syncthreads();
    const int nfull = s_full;
    for (int f = 0; f < nfull; ++f) {                  // rare: the whole database image against one row, exactly
        const int row = s_frow[f];
        if (threadIdx.x == 0) { s_best = ~0ull; s_second = BIG; }
        __syncthreads();
        int m0 = BIG, m1 = BIG, c0 = 0;
        for (int col = threadIdx.x; col < db_n; col += 256) {
            const int d = exact_dist(qbase + row, col);
            m1 = min(max(m0, d), m1);
            c0 = d < m0 ? col : c0;
            m0 = min(m0, d);
        }
        atomicMin(&s_best, ((unsigned long long)(unsigned)m0 << 32) | (unsigned)c0);
        __syncthreads();
        const int bcol = (int)(unsigned)(s_best & 0xffffffffull);
        atomicMin(&s_second, m0 < BIG && c0 == bcol ? m1 : m0);
        __syncthreads();
        if (my_row == row) result = passes((int)(s_best >> 32), s_second) ? bcol : -1;
        __syncthreads();
    }
    if (my_valid) nn_out[pd.out_off + qbase + my_row] = result;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The accepted matches of a launch, compacted on the device (round 3): the host used to copy the whole nearest-neighbour table
// back (one int per query and pair: 2.5 GB per pass at config 5, serialised with the kernels on the one stream) and scan it.
// k_pair_counts: accepted matches per pair.  k_pair_write: every pair's (query, neighbour) list in query order -- the order of
// the reference's loop, keys2a.cpp:356-369 -- at the exclusive prefix of the counts, written straight into pinned host memory
// together with the counts (a few hundred bytes per pair instead of 4 bytes per query).
__global__ __launch_bounds__(256) void k_pair_counts(const PairDesc* __restrict__ pairs, const int* __restrict__ nn, int* __restrict__ cnt)
{
    __shared__ int sm[4];
    const PairDesc pd = pairs[blockIdx.x];
    int c = 0;
    for (int q = threadIdx.x; q < pd.q_n; q += 256) c += nn[pd.out_off + q] >= 0;
    for (int o = 32; o; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) cnt[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

__global__ __launch_bounds__(256) void k_pair_write(const PairDesc* __restrict__ pairs, const int* __restrict__ nn, const int* __restrict__ cnt,
                                                    int* __restrict__ h_cnt, int2* __restrict__ h_matches)
{
    __shared__ int sm[4], s_base;
    const int p = blockIdx.x;
    const PairDesc pd = pairs[p];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int before = 0;
    for (int i = threadIdx.x; i < p; i += 256) before += cnt[i];
    for (int o = 32; o; o >>= 1) before += __shfl_xor(before, o);
    if (lane == 0) sm[wave] = before;
    __syncthreads();
    if (threadIdx.x == 0) { s_base = sm[0] + sm[1] + sm[2] + sm[3]; h_cnt[p] = cnt[p]; }
    __syncthreads();
    int base = s_base;
    for (int q0 = 0; q0 < pd.q_n; q0 += 256) {
        const int q = q0 + (int)threadIdx.x;
        const int v = q < pd.q_n ? nn[pd.out_off + q] : -1;
        const unsigned long long m = __ballot(v >= 0);
        __syncthreads();                              // sm is reused
        if (lane == 0) sm[wave] = __popcll(m);
        __syncthreads();
        int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; ++w) pos += sm[w];
        if (v >= 0) h_matches[pos] = make_int2(q, v);
        base += sm[0] + sm[1] + sm[2] + sm[3];
    }
}

// Which scan kernel a launch uses.  Both are exact and give the same table; they differ in what they cost:
//   top-2   (k_match_l2):    3 VALU per distance, 128 queries per workgroup, indifferent to the data;
//   rescan  (k_match_bound): 1 VALU per distance, 256 queries per workgroup (half the L2 -> CU fragment traffic per MAC), + 128 exact
//           distances for every query row that may pass the ratio test: faster while few rows pass (unordered photo collections),
//           slower when a quarter of them do (video-like sets), see DESIGN section 10 / profiles/r03_match_kernels_ab.txt.
// auto (the default) starts with rescan and switches per launch on the share of accepted matches in the last launch whose
// results the host has seen.  BSFM_MATCH_KERNEL=top2|rescan|auto or bsfm_match_kernel() pin it.
enum { MATCH_AUTO = 0, MATCH_TOP2 = 1, MATCH_RESCAN = 2 };
constexpr double MATCH_RESCAN_MAX_ACCEPT = 0.05;      // accepted matches per query above which the top-2 kernel is the cheaper one
int& match_mode()
{
    static int mode = [] {
        const char* e = getenv("BSFM_MATCH_KERNEL");
        if (e && !strcmp(e, "top2")) return (int)MATCH_TOP2;
        if (e && !strcmp(e, "rescan")) return (int)MATCH_RESCAN;
        return (int)MATCH_AUTO;
    }();
    return mode;
}
constexpr int match_qb(bool rescan) { return rescan ? QBB : QB; }       // queries per workgroup of the two kernels

void launch_match(bool rescan, int blocks, hipStream_t st, const unsigned char* keys, const int* qstat, const PairDesc* pairs, int npairs,
                  int db_off, int db_n, double ratio_sq, int* nn_out)
{
    if (rescan) hipLaunchKernelGGL(k_match_bound, dim3(blocks), dim3(256), 0, st, keys, qstat, pairs, npairs, db_off, db_n, ratio_sq, nn_out);
    else hipLaunchKernelGGL(k_match_l2, dim3(blocks), dim3(256), 0, st, keys, qstat, pairs, npairs, db_off, db_n, ratio_sq, nn_out, 1);
}

struct DevKeys {
    unsigned char* keys = nullptr; int* qstat = nullptr; PairDesc* pairs = nullptr; int* nn = nullptr;
    ~DevKeys() { if (keys) (void)hipFree(keys); if (qstat) (void)hipFree(qstat); if (pairs) (void)hipFree(pairs); if (nn) (void)hipFree(nn); }
};

int have_device()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        fprintf(stderr, "[bsfm] FATAL: no usable HIP device; the matcher has no CPU fallback\n");
        return 0;
    }
    return 1;
}

}  // namespace

extern "C" int bsfm_match_keys_l2(int n1, const unsigned char* k1, int n2, const unsigned char* k2, double ratio,
                                  int* out_pairs, int max_out)
{
    if (!have_device()) return BSFM_ERROR;
    if (n1 < 0 || n2 < 2) {   // ANN aborts when asked for 2 neighbours among fewer points (ANN.h, annkPriSearch)
        fprintf(stderr, "[bsfm] bsfm_match_keys_l2: need at least 2 database keys (n2 = %d)\n", n2);
        return BSFM_ERROR;
    }
    if (n1 == 0) return 0;
    DevKeys d;
    const size_t tot = (size_t)n1 + n2;
    HIPM(hipMalloc((void**)&d.keys, (tot + KEY_SLACK) * 128)); HIPM(hipMalloc((void**)&d.qstat, (tot + KEY_SLACK) * sizeof(int)));
    HIPM(hipMalloc((void**)&d.pairs, sizeof(PairDesc))); HIPM(hipMalloc((void**)&d.nn, (size_t)n1 * sizeof(int)));
    HIPM(hipMemcpy(d.keys, k1, (size_t)n1 * 128, hipMemcpyHostToDevice));
    HIPM(hipMemcpy(d.keys + (size_t)n1 * 128, k2, (size_t)n2 * 128, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_key_stats, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, 0, d.keys, (int)tot, d.qstat);
    const PairDesc pd = { 0, n1, 0, 0 };
    HIPM(hipMemcpy(d.pairs, &pd, sizeof(pd), hipMemcpyHostToDevice));
    {   // one pair on its own is a pair the caller expects to match: the data-independent kernel unless pinned otherwise
        const bool rescan = match_mode() == MATCH_RESCAN;
        launch_match(rescan, (n1 + match_qb(rescan) - 1) / match_qb(rescan), 0, d.keys, d.qstat, d.pairs, 1, n1, n2, ratio * ratio, d.nn);
    }
    std::vector<int> nn(n1);
    HIPM(hipDeviceSynchronize());
    HIPM(hipMemcpy(nn.data(), d.nn, (size_t)n1 * sizeof(int), hipMemcpyDeviceToHost));
    int cnt = 0;
    for (int i = 0; i < n1; ++i)
        if (nn[i] >= 0) {
            if (cnt < max_out && out_pairs) { out_pairs[2 * cnt] = i; out_pairs[2 * cnt + 1] = nn[i]; }
            ++cnt;
        }
    return cnt;
}

extern "C" int bsfm_key_match_full(int num_images, const int* num_keys, const unsigned char* const* keys,
                                   double ratio, int window_radius, const char* out_path)
{
    return bsfm_key_match_full_sharded(num_images, num_keys, keys, ratio, window_radius, out_path, 0, 1);
}

// Concatenates the per-rank files of bsfm_key_match_full_sharded into the file one rank would have written: blocks are
// "j i\nN\n" + N lines, ascending in i inside every rank file; a k-way merge on i restores the reference's (i, j) order.
extern "C" int bsfm_merge_match_files(int count, const char* const* paths, const char* out_path)
{
    struct Src { FILE* f; int j, i, n; bool have; };
    std::vector<Src> src((size_t)count);
    auto advance = [](Src& s) { s.have = s.f && fscanf(s.f, "%d %d %d", &s.j, &s.i, &s.n) == 3; };
    for (int r = 0; r < count; ++r) {
        src[r].f = fopen(paths[r], "r");
        if (!src[r].f) { printf("Could not open %s for reading.\n", paths[r]); for (int q = 0; q < r; ++q) fclose(src[q].f); return BSFM_ERROR; }
        advance(src[r]);
    }
    FILE* out = fopen(out_path, "w");
    if (!out) { printf("Could not open %s for writing.\n", out_path); for (auto& s : src) fclose(s.f); return BSFM_ERROR; }
    int blocks = 0;
    for (;;) {
        int best = -1;
        for (int r = 0; r < count; ++r)
            if (src[r].have && (best < 0 || src[r].i < src[best].i || (src[r].i == src[best].i && src[r].j < src[best].j))) best = r;
        if (best < 0) break;
        Src& s = src[best];
        fprintf(out, "%d %d\n%d\n", s.j, s.i, s.n);
        for (int q = 0; q < s.n; ++q) { int a, b; if (fscanf(s.f, "%d %d", &a, &b) != 2) { s.n = -1; break; } fprintf(out, "%d %d\n", a, b); }
        if (s.n < 0) { fclose(out); for (auto& t : src) fclose(t.f); return BSFM_ERROR; }
        ++blocks;
        advance(s);
    }
    fclose(out);
    for (auto& s : src) fclose(s.f);
    return blocks;
}

// ---- resident key set: descriptors + per-key statistics stay in HBM across calls (the measurement boundary of bench.py:
// "inputs already resident in HBM when the timed region starts"; bsfm_key_match_full* = create + run + destroy)
// One slot of the search pipeline (two per key set): the pairs of one database image, their nearest neighbours, and the pinned buffers the
// accepted matches arrive in.  Kept with the key set (round 5): allocating 2 x 20 MB of pinned memory, the streams and the events inside
// every pass was ~20 ms of a 290 ms pass at config 5.
struct MatchSlot {
    PairDesc* h_pairs = nullptr; PairDesc* d_pairs = nullptr; int* d_nn = nullptr; int* d_cnt = nullptr;
    int* h_cnt = nullptr; int2* h_m = nullptr;           // pinned: accepted matches per pair / the pairs' (query, neighbour) lists back to back
    hipEvent_t done = nullptr, k0 = nullptr, k1 = nullptr; int image = -1; size_t npairs = 0; std::vector<int> js; bool busy = false;
    double dist = 0.0;
};
struct bsfm_match_set {
    MatchSlot slots[2];
    hipStream_t sts[2] = { nullptr, nullptr };    // one stream per slot: the tail of a launch (its last workgroups) overlaps the head of the next
    hipEvent_t base_ev = nullptr;                 // time zero of a pass on the device clock
    bool pipeline_ready = false;
    void release_pipeline()
    {
        for (MatchSlot& s : slots) {
            if (s.h_pairs) (void)hipHostFree(s.h_pairs);
            if (s.h_cnt) (void)hipHostFree(s.h_cnt);
            if (s.h_m) (void)hipHostFree(s.h_m);
            if (s.d_cnt) (void)hipFree(s.d_cnt);
            if (s.d_pairs) (void)hipFree(s.d_pairs);
            if (s.d_nn) (void)hipFree(s.d_nn);
            if (s.done) (void)hipEventDestroy(s.done);
            if (s.k0) (void)hipEventDestroy(s.k0);
            if (s.k1) (void)hipEventDestroy(s.k1);
            s = MatchSlot();
        }
        for (hipStream_t& q : sts) { if (q) (void)hipStreamDestroy(q); q = nullptr; }
        if (base_ev) (void)hipEventDestroy(base_ev);
        base_ev = nullptr; pipeline_ready = false;
    }
    ~bsfm_match_set() { release_pipeline(); }
    int num_images = 0;
    std::vector<int> num_keys;
    std::vector<size_t> off;
    size_t tot = 0;
    DevKeys d;
    // measurement of the last run: HIP-event time of the k_match_l2 launches, their distance count, pairs searched
    double kernel_ms = 0.0; double distances = 0.0; long long pairs = 0; int launches = 0, launches_rescan = 0;
};

extern "C" bsfm_match_set_t* bsfm_match_set_create(int num_images, const int* num_keys, const unsigned char* const* keys)
{
    if (!have_device() || num_images < 0) return nullptr;
    bsfm_match_set* ms = new bsfm_match_set();
    ms->num_images = num_images;
    ms->num_keys.assign(num_keys, num_keys + num_images);
    ms->off.assign((size_t)num_images + 1, 0);
    for (int i = 0; i < num_images; ++i) ms->off[i + 1] = ms->off[i] + (size_t)std::max(num_keys[i], 0);
    ms->tot = ms->off[num_images];
    if (ms->tot > 0x7fffffffULL) { fprintf(stderr, "[bsfm] too many keys\n"); delete ms; return nullptr; }
    if (ms->tot == 0) return ms;
    bool ok = hipMalloc((void**)&ms->d.keys, (ms->tot + KEY_SLACK) * 128) == hipSuccess && hipMalloc((void**)&ms->d.qstat, (ms->tot + KEY_SLACK) * sizeof(int)) == hipSuccess;
    for (int i = 0; i < num_images && ok; ++i)
        if (num_keys[i] > 0) ok = hipMemcpy(ms->d.keys + ms->off[i] * 128, keys[i], (size_t)num_keys[i] * 128, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(k_key_stats, dim3((unsigned)((ms->tot + 255) / 256)), dim3(256), 0, 0, ms->d.keys, (int)ms->tot, ms->d.qstat);
        ok = hipDeviceSynchronize() == hipSuccess;          // key upload + statistics (null stream) before any pipeline stream starts
    }
    if (!ok) { fprintf(stderr, "[bsfm] matcher: key upload failed\n"); delete ms; return nullptr; }
    return ms;
}

extern "C" void bsfm_match_set_destroy(bsfm_match_set_t* ms) { delete ms; }

extern "C" int bsfm_match_kernel(int mode)
{
    const int old = match_mode();
    if (mode >= MATCH_AUTO && mode <= MATCH_RESCAN) match_mode() = mode;
    return old;
}

extern "C" int bsfm_match_set_rescan_launches(const bsfm_match_set_t* ms) { return ms ? ms->launches_rescan : BSFM_ERROR; }

extern "C" int bsfm_match_set_stats(const bsfm_match_set_t* ms, double* kernel_ms, double* distances, long long* pairs, int* launches)
{
    if (!ms) return BSFM_ERROR;
    if (kernel_ms) *kernel_ms = ms->kernel_ms;
    if (distances) *distances = ms->distances;
    if (pairs) *pairs = ms->pairs;
    if (launches) *launches = ms->launches;
    return 0;
}

// rank / world_size: this call handles the database images i with i % world_size == rank (each with all its j < i), so
// the pair list is split without any exchange (matcher = embarrassingly parallel, descriptors replicated).
namespace {
struct MatchTable { std::vector<int> pi, pj, m, ptr; bool overflow = false; };
}
// out_path -> the reference's text; tab -> the same pairs in memory (either may be absent)
static int match_set_run_impl(bsfm_match_set_t* ms, double ratio, int window_radius, const char* out_path, MatchTable* tab, int rank, int world_size)
{
    if (!ms || world_size < 1 || rank < 0 || rank >= world_size) return BSFM_ERROR;
    FILE* f = out_path ? fopen(out_path, "w") : nullptr;
    if (out_path && !f) { printf("Could not open %s for writing.\n", out_path); return BSFM_ERROR; }   // KeyMatchFull.cpp:86-89
    if (tab) tab->ptr.assign(1, 0);
    const int num_images = ms->num_images;
    const int* num_keys = ms->num_keys.data();
    const std::vector<size_t>& off = ms->off;
    const size_t tot = ms->tot;
    DevKeys& d = ms->d;
    ms->kernel_ms = 0.0; ms->distances = 0.0; ms->pairs = 0; ms->launches = 0; ms->launches_rescan = 0;
    if (tot == 0) { if (f) fclose(f); return 0; }
    // Two-slot pipeline on one stream: while the GPU scans database image k, the host turns the accepted matches of image k-1
    // (compacted on the device, written by k_pair_write straight into the slot's pinned buffers) into text (own integer formatter).
    typedef MatchSlot Slot;
    Slot (&slots)[2] = ms->slots;
    hipStream_t (&sts)[2] = ms->sts;
    hipEvent_t& base_ev = ms->base_ev;
    double cover_end = 0.0;         // end of the union of scan intervals so far (ms after base_ev)
    double accept_rate = 0.0;       // accepted matches per query in the last drained launch (auto mode's signal)
    // BSFM_MATCH_STREAMS=1: everything on one stream (launches strictly one after the other: what a profiler's per-launch durations need)
    static const bool one_stream = [] { const char* e = getenv("BSFM_MATCH_STREAMS"); return e && atoi(e) == 1; }();
    auto release = [&] {};          // (the pipeline's buffers stay with the key set: bsfm_match_set::release_pipeline)
    bool ok = true;
    if (!ms->pipeline_ready) {
        ok = hipStreamCreateWithFlags(&sts[0], hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&sts[1], hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipEventCreate(&base_ev) == hipSuccess;
        for (Slot& s : slots) {
            ok = ok && hipHostMalloc((void**)&s.h_pairs, (size_t)num_images * sizeof(PairDesc)) == hipSuccess;
            ok = ok && hipHostMalloc((void**)&s.h_cnt, (size_t)num_images * sizeof(int)) == hipSuccess;
            ok = ok && hipHostMalloc((void**)&s.h_m, tot * sizeof(int2)) == hipSuccess;
            ok = ok && hipMalloc((void**)&s.d_cnt, (size_t)num_images * sizeof(int)) == hipSuccess;
            ok = ok && hipMalloc((void**)&s.d_pairs, (size_t)num_images * sizeof(PairDesc)) == hipSuccess;
            ok = ok && hipMalloc((void**)&s.d_nn, tot * sizeof(int)) == hipSuccess;
            ok = ok && hipEventCreateWithFlags(&s.done, hipEventDisableTiming) == hipSuccess;
            ok = ok && hipEventCreate(&s.k0) == hipSuccess && hipEventCreate(&s.k1) == hipSuccess;
        }
        if (!ok) { fprintf(stderr, "[bsfm] matcher: allocation failed\n"); ms->release_pipeline(); if (f) fclose(f); return BSFM_ERROR; }
        ms->pipeline_ready = true;
    }
    for (Slot& s : slots) { s.busy = false; s.image = -1; s.npairs = 0; s.dist = 0.0; }
    ok = hipEventRecord(base_ev, sts[0]) == hipSuccess;
    if (!ok) { fprintf(stderr, "[bsfm] matcher: HIP error\n"); if (f) fclose(f); return BSFM_ERROR; }
    int total_pairs_written = 0;
    std::vector<char> text;
    auto put_int = [&](int v, char sep) {
        char tmp[12]; int n = 0;
        unsigned u = (unsigned)v;
        do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
        while (n) text.push_back(tmp[--n]);
        text.push_back(sep);
    };
    auto drain = [&](Slot& s) -> bool {
        if (!s.busy) return true;
        if (hipEventSynchronize(s.done) != hipSuccess) return false;
        {   // scan-kernel time of the pass = the UNION of the launches' [start, end] intervals on the common clock (the two streams'
            // launches overlap by design; summing their durations would count the shared stretch twice)
            float t0 = 0.f, t1 = 0.f;
            if (hipEventElapsedTime(&t0, base_ev, s.k0) == hipSuccess && hipEventElapsedTime(&t1, base_ev, s.k1) == hipSuccess && t1 >= t0) {
                const double from = std::max((double)t0, cover_end);
                if ((double)t1 > from) ms->kernel_ms += (double)t1 - from;
                cover_end = std::max(cover_end, (double)t1);
                ms->distances += s.dist; ms->pairs += (long long)s.npairs; ms->launches++;
            }
        }
        text.clear();
        size_t seen_q = 0, seen_m = 0;
        const int2* mlist = s.h_m;                     // the pairs' lists follow one another in pair order (k_pair_write)
        for (size_t p = 0; p < s.npairs; ++p) {
            const int cnt = s.h_cnt[p];
            seen_q += (size_t)s.h_pairs[p].q_n; seen_m += (size_t)cnt;
            if (cnt >= 16) {   // KeyMatchFull.cpp:131-142: "j i\n", count, "idx_j idx_i" lines
                if (f) {
                    put_int(s.js[p], ' '); put_int(s.image, '\n'); put_int(cnt, '\n');
                    for (int t = 0; t < cnt; ++t) { put_int(mlist[t].x, ' '); put_int(mlist[t].y, '\n'); }
                }
                if (tab) {
                    tab->pi.push_back(s.js[p]); tab->pj.push_back(s.image);
                    for (int t = 0; t < cnt; ++t) { tab->m.push_back(mlist[t].x); tab->m.push_back(mlist[t].y); }
                    if (tab->m.size() / 2 > (size_t)INT_MAX) tab->overflow = true;
                    tab->ptr.push_back((int)(tab->m.size() / 2));
                }
                ++total_pairs_written;
            }
            mlist += cnt;
        }
        if (f && !text.empty()) fwrite(text.data(), 1, text.size(), f);
        if (seen_q) accept_rate = (double)seen_m / (double)seen_q;
        s.busy = false;
        return true;
    };
    int turn = 0;
    for (int i = 0; i < num_images && ok; ++i) {
        if (num_keys[i] == 0 || i % world_size != rank) continue;
        int start = 0;
        if (window_radius > 0) start = std::max(i - window_radius, 0);   // KeyMatchFull.cpp:117-119
        Slot& s = slots[turn & 1];
        hipStream_t st = sts[one_stream ? 0 : (turn & 1)];
        ok = drain(s);
        if (!ok) break;
        s.js.clear(); s.npairs = 0; s.image = i;
        const bool rescan = match_mode() == MATCH_RESCAN || (match_mode() == MATCH_AUTO && accept_rate <= MATCH_RESCAN_MAX_ACCEPT);
        const int qb = match_qb(rescan);
        int blk = 0; size_t out = 0;
        for (int j = start; j < i; ++j) {
            if (num_keys[j] == 0) continue;
            s.h_pairs[s.npairs++] = { (int)off[j], num_keys[j], (int)out, blk };
            s.js.push_back(j);
            blk += (num_keys[j] + qb - 1) / qb; out += (size_t)num_keys[j];
        }
        if (s.npairs == 0) continue;
        if (num_keys[i] < 2) {
            fprintf(stderr, "[bsfm] image %d has fewer than 2 keys: the reference's ANN search aborts here; skipped\n", i);
            continue;
        }
        ok = ok && hipMemcpyAsync(s.d_pairs, s.h_pairs, s.npairs * sizeof(PairDesc), hipMemcpyHostToDevice, st) == hipSuccess;
        ok = ok && hipEventRecord(s.k0, st) == hipSuccess;
        launch_match(rescan, blk, st, d.keys, d.qstat, s.d_pairs, (int)s.npairs, (int)off[i], num_keys[i], ratio * ratio, s.d_nn);
        ms->launches_rescan += rescan;
        ok = ok && hipEventRecord(s.k1, st) == hipSuccess;
        s.dist = (double)out * (double)num_keys[i];        // query keys of all pairs x database keys
        hipLaunchKernelGGL(k_pair_counts, dim3((unsigned)s.npairs), dim3(256), 0, st, s.d_pairs, s.d_nn, s.d_cnt);
        hipLaunchKernelGGL(k_pair_write, dim3((unsigned)s.npairs), dim3(256), 0, st, s.d_pairs, s.d_nn, s.d_cnt, s.h_cnt, s.h_m);
        ok = ok && hipEventRecord(s.done, st) == hipSuccess;
        s.busy = true;
        ++turn;
    }
    ok = ok && drain(slots[turn & 1]) && drain(slots[(turn + 1) & 1]);
    release();
    if (!ok) {
        // a failed pass may leave kernels of a busy slot in flight that still write its pinned buffers: nothing of the pipeline -- which now lives
        // with the match set and is reused by the next run -- may be touched again before both streams are idle
        for (int q = 0; q < 2; ++q) { (void)hipStreamSynchronize(sts[q]); slots[q].busy = false; }
        fprintf(stderr, "[bsfm] matcher: HIP error in the pair pipeline\n"); if (f) fclose(f); return BSFM_ERROR;
    }
    if (f) fclose(f);
    return total_pairs_written;
}

extern "C" int bsfm_match_set_run(bsfm_match_set_t* ms, double ratio, int window_radius, const char* out_path, int rank, int world_size)
{
    if (!out_path) return BSFM_ERROR;
    return match_set_run_impl(ms, ratio, window_radius, out_path, nullptr, rank, world_size);
}

// The same search with the result IN MEMORY (the match table feeds bsfm_fmatrix_ransac_batch / bsfm_compute_tracks
// without the text round trip of matches.init.txt, src/BundleIO.cpp:112-166).  Arrays are malloc'ed here: bsfm_free them.
extern "C" int bsfm_match_set_run_table(bsfm_match_set_t* ms, double ratio, int window_radius, int rank, int world_size,
                                        int** pair_i, int** pair_j, int** match_ptr, int** matches)
{
    if (!pair_i || !pair_j || !match_ptr || !matches) return BSFM_ERROR;
    MatchTable tab;
    const int rc = match_set_run_impl(ms, ratio, window_radius, nullptr, &tab, rank, world_size);
    if (rc < 0) return rc;
    if (tab.overflow) { fprintf(stderr, "[bsfm] bsfm_match_set_run_table: more than 2^31-1 matches; shard the run (rank / world_size)\n"); return BSFM_ERROR; }
    auto dup = [](const void* src, size_t bytes) { void* q = malloc(bytes ? bytes : 1); if (q && bytes) memcpy(q, src, bytes); return q; };
    if (tab.ptr.empty()) tab.ptr.assign(1, 0);
    *pair_i = static_cast<int*>(dup(tab.pi.data(), tab.pi.size() * sizeof(int)));
    *pair_j = static_cast<int*>(dup(tab.pj.data(), tab.pj.size() * sizeof(int)));
    *match_ptr = static_cast<int*>(dup(tab.ptr.data(), tab.ptr.size() * sizeof(int)));
    *matches = static_cast<int*>(dup(tab.m.data(), tab.m.size() * sizeof(int)));
    if (!*pair_i || !*pair_j || !*match_ptr || !*matches) { free(*pair_i); free(*pair_j); free(*match_ptr); free(*matches); return BSFM_ERROR; }
    return rc;
}

extern "C" void bsfm_free(void* p) { free(p); }

extern "C" int bsfm_key_match_full_sharded(int num_images, const int* num_keys, const unsigned char* const* keys,
                                           double ratio, int window_radius, const char* out_path, int rank, int world_size)
{
    if (world_size < 1 || rank < 0 || rank >= world_size) return BSFM_ERROR;
    if (!have_device()) return BSFM_ERROR;
    bsfm_match_set_t* ms = bsfm_match_set_create(num_images, num_keys, keys);
    if (!ms) return BSFM_ERROR;
    const int rc = bsfm_match_set_run(ms, ratio, window_radius, out_path, rank, world_size);
    bsfm_match_set_destroy(ms);
    return rc;
}
