// ubench_syrk.hip -- where does the bulk trailing-update kernel lose time?  Runs k_syrk_update (part 2, T = 69 tile
// rows = the first step of config 3) alone on the whole device and compares with variants that drop the epilogue,
// drop the prologue latency (K repeated 8x) or drop the global loads, and an XCD-aware tile order (no gain in the
// real factorisation: 35.0 TFLOP/s with and without, so it is not in the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I bundler_sfm_amd/csrc -I include scripts/ubench_syrk.hip -o /tmp/ubench_syrk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <dlfcn.h>
#include "potrf.hip.h"
using namespace bsfm;

template <int MODE>   // 0: as shipped  1: no C epilogue (one value per lane written)  2: K loop 8x (K=1024)  3: 8x + no epilogue  6 / 7: K loop 2x / 4x
__global__ __launch_bounds__(512, BSFM_SYRK_WPS) void k_var(double* __restrict__ S, int ld, int k, const double* __restrict__ panel, double* sink)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int t = blockIdx.x;
    int a = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((a + 1) * (a + 2) / 2 <= t) ++a;
    while (a * (a + 1) / 2 > t) --a;
    int b = t - a * (a + 1) / 2;
    ++a; ++b;
    const int i = k + 1 + a, j = k + 1 + b;
    double acc[8][4];
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[q][u] = 0.0;
    if (MODE == 4) { a = 1; b = 1; }          // every workgroup reads the same two tiles: pure L2 hits, no epilogue
    const int reps = (MODE == 2 || MODE == 3) ? 8 : (MODE == 6 ? 2 : (MODE == 7 ? 4 : 1));
    if (MODE == 5) {                           // lda = 0: all rows alias one 128-byte line (L1 hits): no memory cost at all
        gemm_nt_128(panel, 0, panel, 0, POTRF_NB, lds, acc);
    } else
#pragma unroll 1
    for (int r = 0; r < reps; ++r)
        gemm_nt_128(panel + (size_t)a * POTRF_NB * POTRF_NB, POTRF_NB, panel + (size_t)b * POTRF_NB * POTRF_NB, POTRF_NB, POTRF_NB, lds, acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 64;
    if (MODE == 1 || MODE >= 3) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int u = 0; u < 4; ++u) s += acc[q][u];
        if (s == 12345.678) sink[threadIdx.x] = s;
        return;
    }
    double* Sij = S + ((size_t)i * POTRF_NB) * ld + (size_t)j * POTRF_NB;
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = wr + 4 * q + (lane >> 4), col = wc + 16 * u + (lane & 15);
            Sij[(size_t)row * ld + col] -= acc[q][u];
        }
}

// MODE 6: workgroups alternate between "accumulators start as C" (store-only epilogue) and "accumulators start at 0"
// (load / subtract / store epilogue): the two workgroups of a CU then have their C traffic at opposite ends of their life.
template <int SEL>   // 0: by blockIdx parity  1: all prologue (= shipped)  2: all epilogue
__global__ __launch_bounds__(512, BSFM_SYRK_WPS) void k_mix(double* __restrict__ S, int ld, int k, const double* __restrict__ panel)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int t = blockIdx.x + 1;
    int a = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((a + 1) * (a + 2) / 2 <= t) ++a;
    while (a * (a + 1) / 2 > t) --a;
    int b = t - a * (a + 1) / 2;
    ++a; ++b;
    const int i = k + 1 + a, j = k + 1 + b;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 64;
    double* Sij = S + ((size_t)i * POTRF_NB) * ld + (size_t)j * POTRF_NB;
    const bool pro = SEL == 1 || (SEL == 0 && (blockIdx.x & 1) == 0);
    double acc[8][4];
    {
        const double* cp = Sij + (size_t)(wr + (lane >> 4)) * ld + wc + (lane & 15);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[q][u] = pro ? cp[16 * u] : 0.0;
            cp += 4 * (size_t)ld;
        }
    }
    gemm_nt_128<true>(panel + (size_t)a * POTRF_NB * POTRF_NB, POTRF_NB, panel + (size_t)b * POTRF_NB * POTRF_NB, POTRF_NB, POTRF_NB, lds, acc);
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    double* Sl = Sij + (size_t)(((tid2 >> 7) << 5) + ((tid2 & 63) >> 4)) * ld + (((tid2 >> 6) & 1) << 6) + (tid2 & 15);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int u = 0; u < 4; ++u) Sl[16 * u] = pro ? acc[q][u] : Sl[16 * u] + acc[q][u];
        Sl += 4 * (size_t)ld;
    }
}

// XCD-aware order: supertiles of G x G tiles, the list cut into nx contiguous chunks, chunk x served by the workgroups
// with blockIdx % nx == x (round-robin dispatch of workgroups over the XCDs).
static void build_map(int n, int G, int nx, std::vector<int2>& out)
{
    std::vector<int2> L;
    for (int A = 0; A < n; A += G)
        for (int B = 0; B <= A; B += G)
            for (int a = A; a < std::min(n, A + G); ++a)
                for (int b = B; b < std::min(n, B + G); ++b)
                    if (b <= a) L.push_back(make_int2(a, b));
    const int N = (int)L.size();
    out.assign(N, make_int2(0, 0));
    int pos = 0;
    for (int x = 0; x < nx; ++x) {
        const int cnt = (N - x + nx - 1) / nx;
        for (int l = 0; l < cnt; ++l) out[(size_t)l * nx + x] = L[pos++];
    }
}

__global__ __launch_bounds__(512, BSFM_SYRK_WPS) void k_mapped(double* __restrict__ S, int ld, int k, const double* __restrict__ panel, const int2* __restrict__ map)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int2 ab = map[blockIdx.x];
    const int a = ab.x + 1, b = ab.y + 1;
    const int i = k + 1 + a, j = k + 1 + b;
    double acc[8][4];
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[q][u] = 0.0;
    gemm_nt_128(panel + (size_t)a * POTRF_NB * POTRF_NB, POTRF_NB, panel + (size_t)b * POTRF_NB * POTRF_NB, POTRF_NB, POTRF_NB, lds, acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 64;
    double* Sij = S + ((size_t)i * POTRF_NB) * ld + (size_t)j * POTRF_NB;
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = wr + 4 * q + (lane >> 4), col = wc + 16 * u + (lane & 15);
            Sij[(size_t)row * ld + col] -= acc[q][u];
        }
}

__global__ void k_probe(int* out)
{
    if (threadIdx.x == 0) {
        unsigned xcc = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | ((4 - 1) << 11));
        unsigned hw = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | ((32 - 1) << 11));
        out[2 * blockIdx.x] = (int)xcc; out[2 * blockIdx.x + 1] = (int)hw;
    }
}

static void probe(hipStream_t st, const char* name)
{
    const int nb = 64; int* d; hipMalloc((void**)&d, 2 * nb * sizeof(int));
    hipLaunchKernelGGL(k_probe, dim3(nb), dim3(64), 0, st, d);
    hipStreamSynchronize(st);
    int h[2 * nb]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%s: blockIdx -> XCC_ID:", name);
    for (int i = 0; i < 32; ++i) printf(" %d", h[2 * i]);
    printf("\n   cu_id(hw_id[11:8]) se(hw_id[15:13]):");
    for (int i = 0; i < 32; ++i) printf(" %d/%d", (h[2 * i + 1] >> 8) & 15, (h[2 * i + 1] >> 13) & 7);
    printf("\n");
    hipFree(d);
}

int main()
{
    const int nblk = 71, ld = nblk * POTRF_NB, T = nblk - 1;
    double *S, *panel, *sink;
    hipMalloc((void**)&S, (size_t)ld * ld * 8); hipMemset(S, 0, (size_t)ld * ld * 8);
    hipMalloc((void**)&panel, (size_t)2 * T * POTRF_NB * POTRF_NB * 8);
    hipMalloc((void**)&sink, 4096);
    std::vector<double> h((size_t)2 * T * POTRF_NB * POTRF_NB);
    for (size_t q = 0; q < h.size(); ++q) h[q] = 1e-3 * (double)((q * 2654435761u) % 1000);
    hipMemcpy(panel, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    const size_t lds_bytes = 2 * 128 * GEMM_LDS_STRIDE * sizeof(double);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = T * (T - 1) / 2 - 1;       // part 2 leaves tile 0 of the triangle to the chain
    const double flop1 = 2.0 * 128 * 128 * 128 * grid;
    auto run = [&](const char* name, auto launch, double flop) {
        for (int w = 0; w < 3; ++w) launch();
        hipDeviceSynchronize();
        float best = 1e9f;
        for (int r = 0; r < 10; ++r) {
            hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
        }
        printf("%-34s grid %5d  %8.3f ms  %7.2f TFLOP/s\n", name, grid, best, flop / (best * 1e-3) / 1e12);
    };
    {   // warm-up ramp: the same launch 40 times, time of each (the first runs in a process are ~10 % slower than the steady state)
        printf("ramp (ms):");
        for (int r = 0; r < 40; ++r) {
            hipEventRecord(e0, 0); hipLaunchKernelGGL(k_syrk_update<false>, dim3(grid), dim3(512), lds_bytes, 0, S, ld, 0, panel, 2, (const double*)nullptr); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); printf(" %.3f", ms);
        }
        printf("\n");
    }
    run("shipped k_syrk_update part 2", [&] { hipLaunchKernelGGL(k_syrk_update<false>, dim3(grid), dim3(512), lds_bytes, 0, S, ld, 0, panel, 2, (const double*)nullptr); }, flop1);
    // (a rank-256 two-panel variant measured 46.6 vs 42.9 TFLOP/s here; dropped, see potrf.hip.h)
    run("same, mode 0 copy", [&] { hipLaunchKernelGGL(k_var<0>, dim3(grid), dim3(512), lds_bytes, 0, S, ld, 0, panel, sink); }, flop1);
    run("no C epilogue", [&] { hipLaunchKernelGGL(k_var<1>, dim3(grid), dim3(512), lds_bytes, 0, S, ld, 0, panel, sink); }, flop1);
    run("K x8 (1024) with epilogue", [&] { hipLaunchKernelGGL(k_var<2>, dim3(grid), dim3(512), lds_bytes, 0, S, ld, 0, panel, sink); }, 8 * flop1);
    run("K x2 (256) with epilogue", [&] { hipLaunchKernelGGL(k_var<6>, dim3(grid), dim3(512), lds_bytes, 0, S, ld, 0, panel, sink); }, 2 * flop1);
    run("K x4 (512) with epilogue", [&] { hipLaunchKernelGGL(k_var<7>, dim3(grid), dim3(512), lds_bytes, 0, S, ld, 0, panel, sink); }, 4 * flop1);
    {   // one workgroup per CU (90 KB of dynamic LDS requested): what a wide update would reach if it left half of every CU to the chain
        const size_t big = 90 * 1024;
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_var<7>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)big);
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_var<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)big);
        run("K x4 (512), ONE workgroup per CU", [&] { hipLaunchKernelGGL(k_var<7>, dim3(grid), dim3(512), big, 0, S, ld, 0, panel, sink); }, 4 * flop1);
        run("K = 128, ONE workgroup per CU", [&] { hipLaunchKernelGGL(k_var<0>, dim3(grid), dim3(512), big, 0, S, ld, 0, panel, sink); }, flop1);
    }
    run("K x8, no epilogue", [&] { hipLaunchKernelGGL(k_var<3>, dim3(grid), dim3(512), lds_bytes, 0, S, ld, 0, panel, sink); }, 8 * flop1);
    run("all WGs same tiles (L2 hits), no epi", [&] { hipLaunchKernelGGL(k_var<4>, dim3(grid), dim3(512), lds_bytes, 0, S, ld, 0, panel, sink); }, flop1);
    run("lda=0 (L1 hits), no epilogue", [&] { hipLaunchKernelGGL(k_var<5>, dim3(grid), dim3(512), lds_bytes, 0, S, ld, 0, panel, sink); }, flop1);
    run("mix: prologue / epilogue by parity", [&] { hipLaunchKernelGGL(k_mix<0>, dim3(grid), dim3(512), lds_bytes, 0, S, ld, 0, panel); }, flop1);
    run("mix kernel, all prologue", [&] { hipLaunchKernelGGL(k_mix<1>, dim3(grid), dim3(512), lds_bytes, 0, S, ld, 0, panel); }, flop1);
    run("mix kernel, all epilogue", [&] { hipLaunchKernelGGL(k_mix<2>, dim3(grid), dim3(512), lds_bytes, 0, S, ld, 0, panel); }, flop1);
    run("shipped k_syrk_update again", [&] { hipLaunchKernelGGL(k_syrk_update<false>, dim3(grid), dim3(512), lds_bytes, 0, S, ld, 0, panel, 2, (const double*)nullptr); }, flop1);
    run("mix kernel, all prologue again", [&] { hipLaunchKernelGGL(k_mix<1>, dim3(grid), dim3(512), lds_bytes, 0, S, ld, 0, panel); }, flop1);
    for (int G : {8}) {
        std::vector<int2> hm; build_map(T - 1, G, 8, hm);
        int2* dm; hipMalloc((void**)&dm, hm.size() * sizeof(int2)); hipMemcpy(dm, hm.data(), hm.size() * sizeof(int2), hipMemcpyHostToDevice);
        char nm[64]; snprintf(nm, sizeof nm, "XCD-aware supertile G=%d", G);
        run(nm, [&] { hipLaunchKernelGGL(k_mapped, dim3(grid), dim3(512), lds_bytes, 0, S, ld, 0, panel, dm); }, flop1);
        hipFree(dm);
    }
    probe(0, "null stream");
    {
        hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
        const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
        std::vector<uint32_t> mask(words, 0u);
        for (int c = 32; c < ncu; ++c) mask[c >> 5] |= 1u << (c & 31);
        hipStream_t sm; if (hipExtStreamCreateWithCUMask(&sm, (uint32_t)words, mask.data()) == hipSuccess) {
            probe(sm, "CU-masked stream (bits 32..255)");
            std::vector<int2> hm; build_map(T - 1, 8, 8, hm);
            int2* dm; hipMalloc((void**)&dm, hm.size() * sizeof(int2)); hipMemcpy(dm, hm.data(), hm.size() * sizeof(int2), hipMemcpyHostToDevice);
            for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_mapped, dim3(grid), dim3(512), lds_bytes, sm, S, ld, 0, panel, dm);
            hipStreamSynchronize(sm);
            hipEventRecord(e0, sm); hipLaunchKernelGGL(k_mapped, dim3(grid), dim3(512), lds_bytes, sm, S, ld, 0, panel, dm); hipEventRecord(e1, sm); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); printf("masked stream, G=8 map: %.3f ms %.2f TFLOP/s\n", ms, flop1 / (ms * 1e-3) / 1e12);
            hipEventRecord(e0, sm); hipLaunchKernelGGL(k_syrk_update<false>, dim3(grid), dim3(512), lds_bytes, sm, S, ld, 0, panel, 2, (const double*)nullptr); hipEventRecord(e1, sm); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1); printf("masked stream, shipped order: %.3f ms %.2f TFLOP/s\n", ms, flop1 / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
