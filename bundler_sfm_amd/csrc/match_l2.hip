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
#include <hip/hip_runtime.h>
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

#define HIPM(call)                                                                                   \
    do { hipError_t _e = (call); if (_e != hipSuccess) {                                             \
        fprintf(stderr, "[bsfm] HIP error %s at %s:%d\n", hipGetErrorName(_e), __FILE__, __LINE__); \
        return BSFM_ERROR; } } while (0)

// per key: q = |x|^2 - 256 * sum(x - 128)   (the query side subtracts the constant 2*128^3 later)
__global__ void k_key_stats(const unsigned char* __restrict__ keys, int n, int* __restrict__ q)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4* p = reinterpret_cast<const uint4*>(keys + (size_t)i * 128);
    int sq = 0, s = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const uint4 v = p[w];
        const unsigned u[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int b = 0; b < 4; ++b) { const int x = (u[c] >> (8 * b)) & 255; sq += x * x; s += x - 128; }
    }
    q[i] = sq - 256 * s;
}

struct PairDesc { int q_off, q_n, out_off, blk0; };   // query image: key offset / count; output offset; first block

__device__ __forceinline__ v4i load_frag(const unsigned char* __restrict__ keys, int key, int lane, int kstep)
{
    const uint4 v = *reinterpret_cast<const uint4*>(keys + (size_t)key * 128 + 64 * kstep + 16 * (lane >> 4));
    v4i r;
    r.x = (int)(v.x ^ 0x80808080u); r.y = (int)(v.y ^ 0x80808080u);
    r.z = (int)(v.z ^ 0x80808080u); r.w = (int)(v.w ^ 0x80808080u);
    return r;
}

// nn_out[out_off + query] = index of the accepted nearest neighbour in the database image, or -1.
__global__ __launch_bounds__(256) void k_match_l2(const unsigned char* __restrict__ keys, const int* __restrict__ qstat,
        const PairDesc* __restrict__ pairs, int npairs, int db_off, int db_n, double ratio_sq, int* __restrict__ nn_out, int one)
{
    constexpr int NG = QB / 16;                       // row groups of 16 queries held by every wave
    __shared__ int st_d0[QB][4], st_d1[QB][4], st_i0[QB][4];     // running (best, second, index) per query and wave
    __shared__ int s_pair;
    if (threadIdx.x == 0) {          // locate the pair this block belongs to
        int lo = 0, hi = npairs - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (pairs[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
        s_pair = lo;
    }
    for (int t = threadIdx.x; t < QB * 4; t += 256) { (&st_d0[0][0])[t] = BIG; (&st_d1[0][0])[t] = BIG; (&st_i0[0][0])[t] = -1; }
    __syncthreads();
    const PairDesc pd = pairs[s_pair];
    const int qbase = (blockIdx.x - pd.blk0) * QB;
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
        // software pipeline: the B fragments of tile t+1 are in flight while tile t is multiplied and ranked
        const int colf = seg + 16 * wave + (lane & 15);
        const int col0 = min(colf, db_n - 1);
        v4i nf0 = load_frag(dkeys, col0, lane, 0), nf1 = load_frag(dkeys, col0, lane, 1);
        int nqb = colf < db_n ? qstat[db_off + col0] + EBIAS : DEADQ;
        for (int tile = seg; tile < seg_end; tile += 64) {
            const v4i bf0 = nf0, bf1 = nf1;
            const int qbb = nqb;
            if (tile + 64 < seg_end) {
                const int coln = tile + 64 + 16 * wave + (lane & 15);
                const int colc = min(coln, db_n - 1);
                nf0 = load_frag(dkeys, colc, lane, 0); nf1 = load_frag(dkeys, colc, lane, 1);
                nqb = coln < db_n ? qstat[db_off + colc] + EBIAS : DEADQ;
            }
            // key = ((qb + BIAS - 2 dot) << TB) | tile = K - (dot << (TB + 1)) with K = ((qb + BIAS) << TB) | tile: the low TB
            // bits are untouched by the subtraction, so ONE 24-bit multiply-add per distance builds the packed key
            const int K = (int)(((unsigned)qbb << TB) | (unsigned)((tile - seg) >> 6));
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                v4i acc = { 0, 0, 0, 0 };
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(afrag[g][0], bf0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(afrag[g][1], bf1, acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // (no inline asm on the accumulator itself: the compiler must see the MFMA -> VALU dependency to place
                    //  the hazard wait states)
                    const unsigned key = (unsigned)(__mul24(acc[r], mscale) + K);        // v_mad_i32_i24
                    unsigned m;
                    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(m) : "v"(b0[g][r]), "v"(b1[g][r]), "v"(key));
                    b1[g][r] = m;
                    b0[g][r] = min(b0[g][r], key);
                }
            }
        }
        // The 16 column classes of a row sit in the 16 lanes (lane & 15) of a quarter wave: butterfly-merge their packed
        // pairs (the lane id of the best travels along), then lane 0 of each quarter folds the segment's result into the
        // wave's slot of the per-query state in LDS (d = e + qa; nobody else touches that slot: no barrier).
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                unsigned k0 = b0[g][r], k1 = b1[g][r];
                int l0 = lane & 15;
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) {
                    const unsigned o0 = (unsigned)__shfl_xor((int)k0, m, 64), o1 = (unsigned)__shfl_xor((int)k1, m, 64);
                    const int ol = __shfl_xor(l0, m, 64);
                    k1 = min(max(k0, o0), min(k1, o1));
                    const bool take = o0 < k0 || (o0 == k0 && ol < l0);
                    l0 = take ? ol : l0;
                    k0 = min(k0, o0);
                }
                if ((lane & 15) == 0) {
                    const int row = 16 * g + 4 * (lane >> 4) + r;
                    const int qrow = min(qbase + row, pd.q_n - 1);
                    const int qa = qstat[pd.q_off + qrow] - 2 * 128 * 128 * 128;
                    const unsigned e0 = k0 >> TB, e1 = k1 >> TB;
                    const int c0 = e0 >= DEADTHR ? BIG : (int)e0 - EBIAS + qa;
                    const int c1 = e1 >= DEADTHR ? BIG : (int)e1 - EBIAS + qa;
                    const int ci = seg + (int)((k0 & ((1u << TB) - 1)) << 6) + 16 * wave + l0;
                    const int d0 = st_d0[row][wave], d1 = st_d1[row][wave];
                    if (c0 < d0) { st_d1[row][wave] = min(d0, c1); st_d0[row][wave] = c0; st_i0[row][wave] = ci; }
                    else st_d1[row][wave] = min(d1, c0);
                }
            }
    }
    __syncthreads();
    if (threadIdx.x < QB) {
        const int row = threadIdx.x, q = qbase + row;
        if (q < pd.q_n) {
            int d0 = BIG, d1 = BIG, idx = -1;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int c0 = st_d0[row][w], c1 = st_d1[row][w], ci = st_i0[row][w];
                // insert the partial list (c0 <= c1) into the running (d0 <= d1)
                if (c0 < d0) { d1 = d0; d0 = c0; idx = ci; }
                else if (c0 < d1) d1 = c0;
                if (c1 < d1) d1 = c1;
            }
            bool ok;
            {
#pragma clang fp contract(off)
                ok = ((double)d0) < ratio_sq * ((double)d1);      // src/keys2a.cpp:362
            }
            nn_out[pd.out_off + q] = ok ? idx : -1;
        }
    }
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
    HIPM(hipMalloc((void**)&d.keys, tot * 128)); HIPM(hipMalloc((void**)&d.qstat, tot * sizeof(int)));
    HIPM(hipMalloc((void**)&d.pairs, sizeof(PairDesc))); HIPM(hipMalloc((void**)&d.nn, (size_t)n1 * sizeof(int)));
    HIPM(hipMemcpy(d.keys, k1, (size_t)n1 * 128, hipMemcpyHostToDevice));
    HIPM(hipMemcpy(d.keys + (size_t)n1 * 128, k2, (size_t)n2 * 128, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_key_stats, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, 0, d.keys, (int)tot, d.qstat);
    const PairDesc pd = { 0, n1, 0, 0 };
    HIPM(hipMemcpy(d.pairs, &pd, sizeof(pd), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_match_l2, dim3((n1 + QB - 1) / QB), dim3(256), 0, 0, d.keys, d.qstat, d.pairs, 1, n1, n2,
                       ratio * ratio, d.nn, 1);
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
struct bsfm_match_set {
    int num_images = 0;
    std::vector<int> num_keys;
    std::vector<size_t> off;
    size_t tot = 0;
    DevKeys d;
    // measurement of the last run: HIP-event time of the k_match_l2 launches, their distance count, pairs searched
    double kernel_ms = 0.0; double distances = 0.0; long long pairs = 0; int launches = 0;
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
    bool ok = hipMalloc((void**)&ms->d.keys, ms->tot * 128) == hipSuccess && hipMalloc((void**)&ms->d.qstat, ms->tot * sizeof(int)) == hipSuccess;
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
extern "C" int bsfm_match_set_run(bsfm_match_set_t* ms, double ratio, int window_radius, const char* out_path, int rank, int world_size)
{
    if (!ms || world_size < 1 || rank < 0 || rank >= world_size) return BSFM_ERROR;
    FILE* f = fopen(out_path, "w");
    if (!f) { printf("Could not open %s for writing.\n", out_path); return BSFM_ERROR; }   // KeyMatchFull.cpp:86-89
    const int num_images = ms->num_images;
    const int* num_keys = ms->num_keys.data();
    const std::vector<size_t>& off = ms->off;
    const size_t tot = ms->tot;
    DevKeys& d = ms->d;
    ms->kernel_ms = 0.0; ms->distances = 0.0; ms->pairs = 0; ms->launches = 0;
    if (tot == 0) { fclose(f); return 0; }
    // Two-slot pipeline on one stream: while the GPU scans database image k, the host turns the nearest-neighbour table of
    // image k-1 into text (own integer formatter: the text, not the search, was the larger part of the wall time).
    struct Slot {
        PairDesc* h_pairs = nullptr; PairDesc* d_pairs = nullptr; int* h_nn = nullptr; int* d_nn = nullptr;
        hipEvent_t done = nullptr, k0 = nullptr, k1 = nullptr; int image = -1; size_t npairs = 0; std::vector<int> js; bool busy = false;
        double dist = 0.0;
    } slots[2];
    hipStream_t st = nullptr;
    auto release = [&] {
        for (Slot& s : slots) {
            if (s.h_pairs) (void)hipHostFree(s.h_pairs);
            if (s.h_nn) (void)hipHostFree(s.h_nn);
            if (s.d_pairs) (void)hipFree(s.d_pairs);
            if (s.d_nn) (void)hipFree(s.d_nn);
            if (s.done) (void)hipEventDestroy(s.done);
            if (s.k0) (void)hipEventDestroy(s.k0);
            if (s.k1) (void)hipEventDestroy(s.k1);
        }
        if (st) (void)hipStreamDestroy(st);
    };
    bool ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
    for (Slot& s : slots) {
        ok = ok && hipHostMalloc((void**)&s.h_pairs, (size_t)num_images * sizeof(PairDesc)) == hipSuccess;
        ok = ok && hipHostMalloc((void**)&s.h_nn, tot * sizeof(int)) == hipSuccess;
        ok = ok && hipMalloc((void**)&s.d_pairs, (size_t)num_images * sizeof(PairDesc)) == hipSuccess;
        ok = ok && hipMalloc((void**)&s.d_nn, tot * sizeof(int)) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&s.done, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreate(&s.k0) == hipSuccess && hipEventCreate(&s.k1) == hipSuccess;
    }
    if (!ok) { fprintf(stderr, "[bsfm] matcher: allocation failed\n"); release(); fclose(f); return BSFM_ERROR; }
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
        { float kms = 0.f; if (hipEventElapsedTime(&kms, s.k0, s.k1) == hipSuccess && kms >= 0.f) { ms->kernel_ms += kms; ms->distances += s.dist; ms->pairs += (long long)s.npairs; ms->launches++; } }
        text.clear();
        for (size_t p = 0; p < s.npairs; ++p) {
            const int* row = s.h_nn + s.h_pairs[p].out_off;
            const int qn = s.h_pairs[p].q_n;
            int cnt = 0;
            for (int q = 0; q < qn; ++q) cnt += row[q] >= 0;
            if (cnt >= 16) {   // KeyMatchFull.cpp:131-142: "j i\n", count, "idx_j idx_i" lines
                put_int(s.js[p], ' '); put_int(s.image, '\n'); put_int(cnt, '\n');
                for (int q = 0; q < qn; ++q) if (row[q] >= 0) { put_int(q, ' '); put_int(row[q], '\n'); }
                ++total_pairs_written;
            }
        }
        if (!text.empty()) fwrite(text.data(), 1, text.size(), f);
        s.busy = false;
        return true;
    };
    int turn = 0;
    for (int i = 0; i < num_images && ok; ++i) {
        if (num_keys[i] == 0 || i % world_size != rank) continue;
        int start = 0;
        if (window_radius > 0) start = std::max(i - window_radius, 0);   // KeyMatchFull.cpp:117-119
        Slot& s = slots[turn & 1];
        ok = drain(s);
        if (!ok) break;
        s.js.clear(); s.npairs = 0; s.image = i;
        int blk = 0; size_t out = 0;
        for (int j = start; j < i; ++j) {
            if (num_keys[j] == 0) continue;
            s.h_pairs[s.npairs++] = { (int)off[j], num_keys[j], (int)out, blk };
            s.js.push_back(j);
            blk += (num_keys[j] + QB - 1) / QB; out += (size_t)num_keys[j];
        }
        if (s.npairs == 0) continue;
        if (num_keys[i] < 2) {
            fprintf(stderr, "[bsfm] image %d has fewer than 2 keys: the reference's ANN search aborts here; skipped\n", i);
            continue;
        }
        ok = ok && hipMemcpyAsync(s.d_pairs, s.h_pairs, s.npairs * sizeof(PairDesc), hipMemcpyHostToDevice, st) == hipSuccess;
        ok = ok && hipEventRecord(s.k0, st) == hipSuccess;
        hipLaunchKernelGGL(k_match_l2, dim3(blk), dim3(256), 0, st, d.keys, d.qstat, s.d_pairs, (int)s.npairs,
                           (int)off[i], num_keys[i], ratio * ratio, s.d_nn, 1);
        ok = ok && hipEventRecord(s.k1, st) == hipSuccess;
        s.dist = (double)out * (double)num_keys[i];        // query keys of all pairs x database keys
        ok = ok && hipMemcpyAsync(s.h_nn, s.d_nn, out * sizeof(int), hipMemcpyDeviceToHost, st) == hipSuccess;
        ok = ok && hipEventRecord(s.done, st) == hipSuccess;
        s.busy = true;
        ++turn;
    }
    ok = ok && drain(slots[turn & 1]) && drain(slots[(turn + 1) & 1]);
    release();
    if (!ok) { fprintf(stderr, "[bsfm] matcher: HIP error in the pair pipeline\n"); fclose(f); return BSFM_ERROR; }
    fclose(f);
    return total_pairs_written;
}

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
