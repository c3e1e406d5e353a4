// potrf_blocked.hip.h -- two-level blocking of the tiled Cholesky (round 2, opt-in: BSFM_CHOL=blocked).
//
// potrf_solve (potrf.hip.h) applies every 128-wide panel to the whole trailing matrix at once: one read + one write of every C tile
// per 128 accumulation steps.  At k = 128 that caps any kernel near 43 TFLOP/s (k_syrk_update 43, rocblas_dgemm 41, rocblas_dsyrk 31 on
// the 8 704^2 triangle; rocBLAS needs k >= 256 to pass 55, scripts/ubench_rocblas.cpp), and the bulk-bound steps are 6.4 of the 9.1 ms of a
// config-3 factorisation.  Here the tile columns are grouped into OUTER PANELS of D columns.  With [bs, be) the outer panel of step k:
//   chain / side  exactly potrf_solve's kernels (diagonal tile, first panel tile, rest of the panel, first trailing column);
//   N_cur(k)      panel k -> the remaining columns [k+2, be) of the current outer panel                          (stream s2, k = 128)
//   N_next(k)     panel k -> the columns [max(k+2, be), be + D) of the NEXT outer panel: the lookahead that lets the chain cross an
//                 outer boundary without waiting for a wide update                                              (stream s3, k = 128)
//   F(b)          the D finished panels of outer panel b -> every column >= be + D, ONCE, as rank-128 D updates  (stream s3):
//                 k_far_tri (own kernel, one read + one write of C per 128 D steps), or with BSFM_CHOL_FAR=rocblas one rocblas_dgemm per
//                 column strip of 8 tiles (plain library GEMM; dlopen'ed like the rocSOLVER cross-check backend).
// Every tile still receives every earlier panel exactly once; the order in which a tile receives them differs from potrf_solve's, so
// the two agree to rounding, and a run is bit-reproducible (stream order + events fix the order per tile).
// Tile (k+2, k+2) is skipped by the near launches of step k as in potrf_solve (the chain's own tile kernel applies panel k to it).
//
// MEASURED (config 3, profiles/r02_chol_blocked_timeline.txt): correct (tests/test_chol_gpu.py), and 2.3 x SLOWER than potrf_solve: 19.8 ms
// per solve instead of 8.5.  The wide GEMMs themselves run as expected (rocBLAS MT128x128 / MT128x64 kernels, 90-190 us per strip), but
// while one is resident the chain kernels take 4-10 x their usual time (first panel tile 64-100 us instead of 15, next-diagonal update
// 50-160 instead of 8, diagonal tile 63-220 instead of 52) and the chain period becomes 170-700 us: the library kernels' workgroups hold a
// CU's LDS for 100+ us, k_syrk_update's two 36 KB workgroups per CU leave room for a chain workgroup every ~25 us.  Keeping 16-64 CUs out
// of the wide stream's mask (BSFM_FAR_RESERVE_CUS) made it worse (25 ms), as did GPU_MAX_HW_QUEUES=8 (27 ms; with 4 queues the wide stream
// shares the side stream's queue).  The structure needs wide-update kernels that co-reside with the chain's -- an own 256-wide
// macro-tile kernel with a small LDS footprint -- before it can pay; kept opt-in as the correct starting point.
// With k_far_tri (same footprint as k_syrk_update, 58 TFLOP/s on the rank-512 update) it is 14.7-17 ms: the workgroups of a launch start
// and retire together, so no slot frees up for the 200-250 us a rank-512 workgroup lives, and the chain's kernels wait that long.
#pragma once
#include "potrf.hip.h"

namespace bsfm {

// panel k -> tiles (i, j), j0 <= j < j1, i >= j, operands from the compact panel copy `pk` (tile a = row k+1+a)
__global__ __launch_bounds__(512, BSFM_SYRK_WPS) void k_syrk_cols(double* __restrict__ S, int ld, int k, int nblk, int j0, int j1,
                                                                  const double* __restrict__ pk)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    int t = blockIdx.x, j = j0;
    while (j < j1 - 1 && t >= nblk - j) { t -= nblk - j; ++j; }
    const int i = j + t;
    if (i == k + 2 && j == k + 2) return;                // left to the chain (k_chain_tile32<1> of step k+1)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 64;
    double* Sij = S + ((size_t)i * POTRF_NB) * ld + (size_t)j * POTRF_NB;
    double acc[8][4];
    {
        const double* cp = Sij + (size_t)(wr + (lane >> 4)) * ld + wc + (lane & 15);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[q][u] = cp[16 * u];
            cp += 4 * (size_t)ld;
        }
    }
    constexpr size_t TL = (size_t)POTRF_NB * POTRF_NB;
    gemm_nt_128<true>(pk + (size_t)(i - k - 1) * TL, POTRF_NB, pk + (size_t)(j - k - 1) * TL, POTRF_NB, POTRF_NB, lds, acc);
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    double* Sl = Sij + (size_t)(((tid2 >> 7) << 5) + ((tid2 & 63) >> 4)) * ld + (((tid2 >> 6) & 1) << 6) + (tid2 & 15);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int u = 0; u < 4; ++u) Sl[16 * u] = acc[q][u];
        Sl += 4 * (size_t)ld;
    }
}

// panels [bs, be) -> the lower triangle (diagonal tiles included) of the tile block [c0, c1) x [c0, c1); operands straight from S
// (the be - bs factor tiles of a tile row are adjacent in memory: one K = 128 (be - bs) product per tile)
__global__ __launch_bounds__(512, BSFM_SYRK_WPS) void k_far_diag(double* __restrict__ S, int ld, int bs, int be, int c0)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int t = blockIdx.x;
    int a = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((a + 1) * (a + 2) / 2 <= t) ++a;
    while (a * (a + 1) / 2 > t) --a;
    const int b = t - a * (a + 1) / 2;
    const int i = c0 + a, j = c0 + b;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 64;
    double* Sij = S + ((size_t)i * POTRF_NB) * ld + (size_t)j * POTRF_NB;
    double acc[8][4];
    {
        const double* cp = Sij + (size_t)(wr + (lane >> 4)) * ld + wc + (lane & 15);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[q][u] = cp[16 * u];
            cp += 4 * (size_t)ld;
        }
    }
    gemm_nt_128<true>(S + ((size_t)i * POTRF_NB) * ld + (size_t)bs * POTRF_NB, ld, S + ((size_t)j * POTRF_NB) * ld + (size_t)bs * POTRF_NB, ld,
                      POTRF_NB * (be - bs), lds, acc, POTRF_NB);
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    double* Sl = Sij + (size_t)(((tid2 >> 7) << 5) + ((tid2 & 63) >> 4)) * ld + (((tid2 >> 6) & 1) << 6) + (tid2 & 15);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int u = 0; u < 4; ++u) Sl[16 * u] = acc[q][u];
        Sl += 4 * (size_t)ld;
    }
}

// panels [bs, be) -> every tile (i, j), i >= j >= f0: ONE read and ONE write of the C tile for (be - bs) x 128 accumulation steps.
// Operands: the compact panel copies of the ring (slot q & 7, tile a = row q + 1 + a), one 128-step product per panel with the
// accumulators kept -- with the 16x16x4 matrix instruction this loop holds 62 TFLOP/s (scripts/ubench_syrk.hip, "K x8"), and its
// workgroups are k_syrk_update's (36 KB of LDS, two per CU), so the chain's kernels still find room next to them.
__global__ __launch_bounds__(512, BSFM_SYRK_WPS) void k_far_tri(double* __restrict__ S, int ld, int bs, int be, int f0,
                                                                const double* __restrict__ ring, size_t pstride)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int t = blockIdx.x;
    int a = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((a + 1) * (a + 2) / 2 <= t) ++a;
    while (a * (a + 1) / 2 > t) --a;
    const int b = t - a * (a + 1) / 2;
    const int i = f0 + a, j = f0 + b;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 64;
    double* Sij = S + ((size_t)i * POTRF_NB) * ld + (size_t)j * POTRF_NB;
    double acc[8][4];
    {
        const double* cp = Sij + (size_t)(wr + (lane >> 4)) * ld + wc + (lane & 15);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[q][u] = cp[16 * u];
            cp += 4 * (size_t)ld;
        }
    }
    constexpr size_t TL = (size_t)POTRF_NB * POTRF_NB;
#pragma unroll 1
    for (int q = bs; q < be; ++q) {
        const double* pq = ring + (size_t)(q & 7) * pstride;
        gemm_nt_128<true>(pq + (size_t)(i - q - 1) * TL, POTRF_NB, pq + (size_t)(j - q - 1) * TL, POTRF_NB, POTRF_NB, lds, acc);
    }
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    double* Sl = Sij + (size_t)(((tid2 >> 7) << 5) + ((tid2 & 63) >> 4)) * ld + (((tid2 >> 6) & 1) << 6) + (tid2 & 15);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int u = 0; u < 4; ++u) Sl[16 * u] = acc[q][u];
        Sl += 4 * (size_t)ld;
    }
}

struct BlockedState {
    int enabled = 0;            // BSFM_CHOL=blocked
    int D = 4;                  // tile columns per outer panel (BSFM_CHOL_D)
    int strip = 8;              // tile columns per far strip (BSFM_CHOL_STRIP)
    int far_rocblas = 0;        // BSFM_CHOL_FAR=rocblas: wide updates as rocblas_dgemm strips instead of k_far_tri
    int min_tiles = 24;         // below this many tile columns the plain schedule runs
    hipStream_t s3 = nullptr; bool s3_masked = false;
    double* panel8 = nullptr;   // ring of 8 compact panel copies (the far stream may lag the chain by more than one outer panel)
    std::vector<hipEvent_t> ev3;   // one event per s3 launch group
    void* rb_lib = nullptr; void* rb_handle = nullptr;
    int (*rb_dgemm)(void*, int, int, int, int, int, const double*, const double*, int, const double*, int, const double*, double*, int) = nullptr;
    int (*rb_set_stream)(void*, hipStream_t) = nullptr;
    int (*rb_destroy)(void*) = nullptr;
    double far_ms = 0.0; double far_flops = 0.0; long long far_cnt = 0;
    hipEvent_t f0 = nullptr, f1 = nullptr;
    bool ready = false;
};

inline BlockedState& blocked_state(PotrfWorkspace& w)
{
    // one state per workspace, keyed by address (the workspace struct itself stays untouched for the other schedules)
    static std::mutex mu;
    static std::vector<std::pair<PotrfWorkspace*, BlockedState*>> all;
    std::lock_guard<std::mutex> lk(mu);
    for (auto& p : all) if (p.first == &w) return *p.second;
    all.emplace_back(&w, new BlockedState());
    return *all.back().second;
}

inline int blocked_init(PotrfWorkspace& w, BlockedState& b)
{
    if (b.ready) return 0;
    if (const char* e = getenv("BSFM_CHOL_D")) b.D = std::max(2, std::min(8, atoi(e)));
    if (const char* e = getenv("BSFM_CHOL_STRIP")) b.strip = std::max(2, atoi(e));
    if (const char* e = getenv("BSFM_CHOL_FAR")) b.far_rocblas = !strcmp(e, "rocblas");
    if (const char* e = getenv("BSFM_CHOL_MIN_TILES")) b.min_tiles = std::max(2 * b.D + 1, atoi(e));
    b.rb_lib = dlopen("librocblas.so", RTLD_NOW | RTLD_GLOBAL);
    if (!b.rb_lib) { fprintf(stderr, "[bsfm] BSFM_CHOL=blocked: librocblas.so unavailable (%s)\n", dlerror()); return -1; }
    auto create = (int (*)(void**))dlsym(b.rb_lib, "rocblas_create_handle");
    b.rb_destroy = (int (*)(void*))dlsym(b.rb_lib, "rocblas_destroy_handle");
    b.rb_set_stream = (int (*)(void*, hipStream_t))dlsym(b.rb_lib, "rocblas_set_stream");
    b.rb_dgemm = (decltype(b.rb_dgemm))dlsym(b.rb_lib, "rocblas_dgemm");
    if (!create || !b.rb_set_stream || !b.rb_dgemm || create(&b.rb_handle) != 0) { fprintf(stderr, "[bsfm] BSFM_CHOL=blocked: rocBLAS symbols missing\n"); return -1; }
    {   // The wide updates are long-running library kernels that fill every CU they may use: keep some CUs out of their reach so
        // that the chain / side kernels (16, 10, 1 workgroups; the diagonal tile needs a whole CU's LDS) never queue behind them
        int reserve = 0;
        if (const char* e = getenv("BSFM_FAR_RESERVE_CUS")) reserve = atoi(e);
        int dev = 0; (void)hipGetDevice(&dev);
        hipDeviceProp_t prop;
        if (reserve > 0 && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 2 * reserve) {
            const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
            std::vector<uint32_t> mask((size_t)words, 0u);
            for (int c = reserve; c < ncu; ++c) mask[(size_t)(c >> 5)] |= 1u << (c & 31);
            if (hipExtStreamCreateWithCUMask(&b.s3, (uint32_t)words, mask.data()) == hipSuccess) b.s3_masked = true;
            else b.s3 = nullptr;
        }
        if (!b.s3 && !(b.s3 = stream_pool().acquire())) return -1;
    }
    if (b.rb_set_stream(b.rb_handle, b.s3) != 0) return -1;
    const size_t tile = (size_t)POTRF_NB * POTRF_NB;
    if (hipMalloc((void**)&b.panel8, 8 * std::max<size_t>(1, (size_t)(w.nblk - 1)) * tile * sizeof(double)) != hipSuccess) return -1;
    b.ev3.resize((size_t)2 * w.nblk + 8);
    for (auto& e : b.ev3) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return -1;
    if (hipEventCreate(&b.f0) != hipSuccess || hipEventCreate(&b.f1) != hipSuccess) return -1;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_syrk_cols), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * 128 * GEMM_LDS_STRIDE * sizeof(double))) != hipSuccess) return -1;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_far_tri), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * 128 * GEMM_LDS_STRIDE * sizeof(double))) != hipSuccess) return -1;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_far_diag), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * 128 * GEMM_LDS_STRIDE * sizeof(double))) != hipSuccess) return -1;
    b.ready = true;
    return 0;
}

inline void blocked_release(PotrfWorkspace& w)
{
    BlockedState& b = blocked_state(w);
    if (b.s3) { (void)hipStreamSynchronize(b.s3); if (b.s3_masked) (void)hipStreamDestroy(b.s3); else stream_pool().release(b.s3); b.s3 = nullptr; }
    if (b.panel8) { (void)hipFree(b.panel8); b.panel8 = nullptr; }
    for (auto& e : b.ev3) if (e) (void)hipEventDestroy(e);
    b.ev3.clear();
    if (b.f0) (void)hipEventDestroy(b.f0);
    if (b.f1) (void)hipEventDestroy(b.f1);
    b.f0 = b.f1 = nullptr;
    if (b.rb_handle && b.rb_destroy) b.rb_destroy(b.rb_handle);
    b.rb_handle = nullptr;
    b.ready = false; b.enabled = 0;
}

// Same contract as potrf_solve.
inline int potrf_solve_blocked(PotrfWorkspace& w, BlockedState& B, double* S, int ld, int n, const double* E, double* x_out, int* d_info, hipStream_t st)
{
    const int nblk = (n + POTRF_NB - 1) / POTRF_NB;
    if (w.ev0) (void)hipEventRecord(w.ev0, st);
    w.sy_used = 0;
    const int D = B.D;
    const size_t lds_bytes = 2 * 128 * GEMM_LDS_STRIDE * sizeof(double);
    (void)hipMemsetAsync(w.etmp, 0, (size_t)ld * sizeof(double), st);
    (void)hipMemcpyAsync(w.etmp, E, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st);
    const size_t pstride = std::max<size_t>(1, (size_t)(w.nblk - 1)) * POTRF_NB * POTRF_NB;
    const size_t diag_lds = DG_LDS_DOUBLES * sizeof(double);
    const size_t lds64 = (64 + 128) * G64_STRIDE * sizeof(double);
    const size_t lds32 = T32_LDS_DOUBLES * sizeof(double);
    const size_t tl = (size_t)POTRF_NB * POTRF_NB;
    const double tile_flops = 2.0 * POTRF_NB * POTRF_NB * POTRF_NB;
    auto panel_of = [&](int k) { return B.panel8 + (size_t)(k & 7) * pstride; };
    (void)hipEventRecord(w.evU[w.nblk], st);                 // everything queued before the solve (S, E ready)
    (void)hipStreamWaitEvent(w.s2, w.evU[w.nblk], 0);
    (void)hipStreamWaitEvent(w.sd, w.evU[w.nblk], 0);
    (void)hipStreamWaitEvent(B.s3, w.evU[w.nblk], 0);
    // the last s3 event whose launches wrote tiles of column c (-1: none); s3 events are handed out in order
    std::vector<int> s3col((size_t)nblk, -1);
    int nev3 = 0;
    auto s3_mark = [&](int c0, int c1) {                     // records an event on s3 and tags columns [c0, c1)
        if (nev3 >= (int)B.ev3.size()) return;               // (cannot happen: <= 2 launch groups per step)
        (void)hipEventRecord(B.ev3[(size_t)nev3], B.s3);
        for (int c = c0; c < c1 && c < nblk; ++c) s3col[(size_t)c] = nev3;
        ++nev3;
    };
    auto wait_s3_cols = [&](hipStream_t s, int c0, int c1) { // s waits for everything s3 has done to columns [c0, c1)
        int last = -1;
        for (int c = std::max(c0, 0); c < c1 && c < nblk; ++c) last = std::max(last, s3col[(size_t)c]);
        if (last >= 0) (void)hipStreamWaitEvent(s, B.ev3[(size_t)last], 0);
    };
    const double alpha = -1.0, beta = 1.0;
    bool timed_far = false;
    hipLaunchKernelGGL(k_potrf_diag, dim3(1), dim3(512), diag_lds, st, S, ld, 0, n, w.linv, d_info, w.dbg);
    for (int k = 0; k + 1 < nblk; ++k) {
        const int T = nblk - k - 1;                          // tile rows below the diagonal tile k
        const int bs = (k / D) * D, be = std::min(bs + D, nblk);          // outer panel of this step
        double* pk = panel_of(k);
        const double* Lk = w.linv + (size_t)k * tl;
        // chain: first panel tile (its column k was completed by the side stream of step k-1)
        if (k > 0) (void)hipStreamWaitEvent(st, w.evC[k - 1], 0);
        hipLaunchKernelGGL(k_chain_tile32<0>, dim3(16), dim3(256), lds32, st, S, ld, k, Lk, pk, (const double*)nullptr, (const double*)nullptr);
        (void)hipEventRecord(w.evT[k], st);
        // side: rest of the panel and y_k, then the rest of the first trailing column
        (void)hipStreamWaitEvent(w.sd, w.evT[k], 0);
        hipLaunchKernelGGL(k_trsm_panel64, dim3(2 * (T - 1) + 2), dim3(256), lds64, w.sd, S, ld, k, Lk, pk, 1, 2 * (T - 1), w.etmp, w.y);
        (void)hipEventRecord(w.evP[k], w.sd);
        if (k > 0) (void)hipStreamWaitEvent(w.sd, w.evU[k - 1], 0);   // column k+1: the near launch of step k-1 (s2) ...
        wait_s3_cols(w.sd, k + 1, k + 2);                             // ... and whatever the far stream did to it
        hipLaunchKernelGGL(k_syrk_col64, dim3(2 * (T - 1) + T), dim3(256), lds64, w.sd, S, ld, k, pk, 1, 2 * (T - 1), w.etmp, w.y, 0);
        (void)hipEventRecord(w.evC[k], w.sd);
        // near, current outer panel: columns [k+2, be)
        (void)hipStreamWaitEvent(w.s2, w.evP[k], 0);
        if (k + 2 < be) {
            wait_s3_cols(w.s2, k + 2, be);
            int cnt = 0;
            for (int j = k + 2; j < be; ++j) cnt += nblk - j;
            const bool timed = w.syrk_events > 0;
            if (timed) (void)hipEventRecord(w.sy0[w.sy_used], w.s2);
            hipLaunchKernelGGL(k_syrk_cols, dim3(cnt), dim3(512), lds_bytes, w.s2, S, ld, k, nblk, k + 2, be, (const double*)pk);
            if (timed) { (void)hipEventRecord(w.sy1[w.sy_used], w.s2); w.sy_flops[w.sy_used++] = tile_flops * (cnt - 1); }
        }
        (void)hipEventRecord(w.evU[k], w.s2);
        // near, next outer panel (lookahead): columns [max(k+2, be), be + D) on the far stream, behind the wide update they follow
        {
            const int j0 = std::max(k + 2, be), j1 = std::min(be + D, nblk);
            if (j0 < j1) {
                (void)hipStreamWaitEvent(B.s3, w.evP[k], 0);
                int cnt = 0;
                for (int j = j0; j < j1; ++j) cnt += nblk - j;
                hipLaunchKernelGGL(k_syrk_cols, dim3(cnt), dim3(512), lds_bytes, B.s3, S, ld, k, nblk, j0, j1, (const double*)pk);
                s3_mark(j0, j1);
            }
        }
        // wide update: once the outer panel is complete, its D panels -> every column past the next outer panel
        if (k == be - 1 && be + D < nblk) {
            const int f0 = be + D;
            (void)hipStreamWaitEvent(B.s3, w.evP[k], 0);             // the last panel of the outer panel (the earlier ones precede it on sd)
            if (!timed_far) (void)hipEventRecord(B.f0, B.s3);
            double flops = 0.0;
            if (!B.far_rocblas) {
                const int Tf = nblk - f0;
                hipLaunchKernelGGL(k_far_tri, dim3(Tf * (Tf + 1) / 2), dim3(512), lds_bytes, B.s3, S, ld, bs, be, f0, (const double*)B.panel8, pstride);
                flops += tile_flops * (be - bs) * (Tf * (Tf + 1) / 2);
            } else
            for (int c0 = f0; c0 < nblk; c0 += B.strip) {
                // One GEMM per column strip, from the strip's own first row down: the part above the diagonal inside the strip's
                // diagonal block is computed too (5 % extra flops at 8-tile strips) and lands in tiles / half tiles nobody reads --
                // separate 36-workgroup launches for the diagonal blocks ran at 1/7 of the device, one after the other.
                // row-major C(rows >= c0, cols [c0, c1)) -= A_R A_C^T  ==  column-major C^T (W x R) -= op_T(A_C) * A_R^T
                const int c1 = std::min(c0 + B.strip, nblk), wt = c1 - c0;
                const int Wc = wt * POTRF_NB, R = (nblk - c0) * POTRF_NB, K = (be - bs) * POTRF_NB;
                const double* Ac = S + ((size_t)c0 * POTRF_NB) * ld + (size_t)bs * POTRF_NB;
                double* Cr = S + ((size_t)c0 * POTRF_NB) * ld + (size_t)c0 * POTRF_NB;
                if (B.rb_dgemm(B.rb_handle, 112 /* transpose */, 111 /* none */, Wc, R, K, &alpha, Ac, ld, Ac, ld, &beta, Cr, ld) != 0) {
                    fprintf(stderr, "[bsfm] rocblas_dgemm failed in the wide update\n");
                    return -1;
                }
                flops += 2.0 * Wc * (double)R * K;
            }
            if (!timed_far) { (void)hipEventRecord(B.f1, B.s3); B.far_flops = flops; timed_far = true; }
            s3_mark(f0, nblk);
        }
        // chain: next diagonal tile (S_{k+1,k+1} takes panels k-1 and k here, see potrf_solve)
        wait_s3_cols(st, k + 1, k + 2);
        hipLaunchKernelGGL(k_chain_tile32<1>, dim3(10), dim3(256), lds32, st, S, ld, k, Lk, pk,
                           k > 0 ? (const double*)panel_of(k - 1) : (const double*)nullptr, (const double*)nullptr);
        hipLaunchKernelGGL(k_potrf_diag, dim3(1), dim3(512), diag_lds, st, S, ld, k + 1, n, w.linv, d_info, w.dbg);
    }
    // everything has drained when the streams meet again
    if (nblk > 1) { (void)hipStreamWaitEvent(st, w.evC[nblk - 2], 0); (void)hipStreamWaitEvent(st, w.evU[nblk - 2], 0); }
    if (nev3 > 0) (void)hipStreamWaitEvent(st, B.ev3[(size_t)nev3 - 1], 0);
    hipLaunchKernelGGL(k_fwd_last, dim3(1), dim3(256), 0, st, w.linv + (size_t)(nblk - 1) * tl,
                       w.etmp + (size_t)(nblk - 1) * POTRF_NB, w.y + (size_t)(nblk - 1) * POTRF_NB);
    (void)hipMemsetAsync(w.bflags, 0, (size_t)(w.nblk + 1) * sizeof(int), st);
    hipLaunchKernelGGL(k_bwd_persistent, dim3(nblk), dim3(256), 0, st, S, ld, nblk, w.linv, w.y, w.xs, w.bflags, w.bflags + w.nblk);
    (void)hipMemcpyAsync(x_out, w.xs, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st);
    if (w.ev1) (void)hipEventRecord(w.ev1, st);
    return 0;
}

}  // namespace bsfm
