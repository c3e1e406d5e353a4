// ubench_syrk_stagger.hip -- round 3: are the two workgroups of a CU in LOCKSTEP in the bulk trailing update (all load C, all multiply,
// all store), and what does de-phasing them buy?  k_syrk_update's launch of the first step of config 3 (T = 69) alone on the device:
//   * census: which (XCC, SE, CU) does workgroup b of a 512-thread / 36 KB-LDS launch land on -- do b and b + 256 share a CU?
//   * persistent form (grid 512, workgroup b takes tiles b, b + 512, ...) with the second-slot workgroups (b >= 256) delayed by d us
//     at the start: total time INCLUDING the delay, d = 0 .. 30;
//   * the same with the tile order handed out by an atomic counter.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I bundler_sfm_amd/csrc -I include scripts/ubench_syrk_stagger.hip -o /tmp/ubench_syrk_stagger
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <map>
#include "potrf.hip.h"
using namespace bsfm;

__device__ __forceinline__ void tile_of(int t, int& a, int& b)
{
    a = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((a + 1) * (a + 2) / 2 <= t) ++a;
    while (a * (a + 1) / 2 > t) --a;
    b = t - a * (a + 1) / 2;
    ++a; ++b;
}

__device__ __forceinline__ void one_tile(double* __restrict__ S, int ld, int k, const double* __restrict__ panel, int t, double* lds)
{
    int a, b;
    tile_of(t, a, b);
    const int i = k + 1 + a, j = k + 1 + b;
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
    gemm_nt_128<true>(panel + (size_t)a * POTRF_NB * POTRF_NB, POTRF_NB, panel + (size_t)b * POTRF_NB * POTRF_NB, POTRF_NB, POTRF_NB, lds, acc);
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

// DYN = 0: static stride; 1: atomic tile counter
template <int DYN>
__global__ __launch_bounds__(512, BSFM_SYRK_WPS) void k_pers(double* __restrict__ S, int ld, int k, const double* __restrict__ panel, int ntiles,
                                                               int delay_ticks, int slot_split, int* counter)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ int next;
    if (delay_ticks > 0 && (int)blockIdx.x >= slot_split) {
        if (threadIdx.x == 0) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < delay_ticks) __builtin_amdgcn_s_sleep(8); }
        __syncthreads();
    }
    if (DYN == 0) {
        for (int t = blockIdx.x + 1; t <= ntiles; t += gridDim.x) { one_tile(S, ld, k, panel, t, lds); __syncthreads(); }
    } else {
        for (;;) {
            if (threadIdx.x == 0) next = atomicAdd(counter, 1);
            __syncthreads();
            const int t = next + 1;
            if (t > ntiles) break;
            one_tile(S, ld, k, panel, t, lds);
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(512, BSFM_SYRK_WPS) void k_census(int* out, long long* stamps)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | ((4 - 1) << 11));
        const unsigned hw = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | ((32 - 1) << 11));
        out[2 * blockIdx.x] = (int)xcc; out[2 * blockIdx.x + 1] = (int)hw;
        stamps[blockIdx.x] = wall_clock64();
        const long long t0 = wall_clock64(); while (wall_clock64() - t0 < 2000) __builtin_amdgcn_s_sleep(8);      // 20 us: keeps every slot taken
        lds[0] = 1.0;
    }
}

int main()
{
    const int nblk = 71, ld = nblk * POTRF_NB, T = nblk - 1;
    double *S, *panel;
    hipMalloc((void**)&S, (size_t)ld * ld * 8); hipMemset(S, 0, (size_t)ld * ld * 8);
    hipMalloc((void**)&panel, (size_t)2 * T * POTRF_NB * POTRF_NB * 8);
    std::vector<double> h((size_t)2 * T * POTRF_NB * POTRF_NB);
    for (size_t q = 0; q < h.size(); ++q) h[q] = 1e-3 * (double)((q * 2654435761u) % 1000);
    hipMemcpy(panel, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    const size_t lds_bytes = 2 * 128 * GEMM_LDS_STRIDE * sizeof(double);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int* counter; hipMalloc((void**)&counter, 4);
    {   // census
        const int nb = 1024; int* d; long long* st; hipMalloc((void**)&d, 2 * nb * sizeof(int)); hipMalloc((void**)&st, nb * 8);
        hipLaunchKernelGGL(k_census, dim3(nb), dim3(512), lds_bytes, 0, d, st);
        hipDeviceSynchronize();
        std::vector<int> hh(2 * nb); std::vector<long long> hs(nb);
        hipMemcpy(hh.data(), d, hh.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hs.data(), st, nb * 8, hipMemcpyDeviceToHost);
        auto cuid = [&](int b) { const int hw = hh[2 * b + 1]; return (hh[2 * b] << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15); };
        std::map<int, std::vector<int>> by;
        for (int b = 0; b < 512; ++b) by[cuid(b)].push_back(b);
        int pairs256 = 0, singles = 0, others = 0;
        for (auto& kv : by) { if (kv.second.size() == 2 && kv.second[1] - kv.second[0] == 256) ++pairs256; else if (kv.second.size() == 1) ++singles; else ++others; }
        printf("census (first 512 workgroups): %zu distinct CU ids; pairs (b, b+256) on one CU: %d, CUs with one WG: %d, other: %d\n", by.size(), pairs256, singles, others);
        printf("  examples:"); int shown = 0; for (auto& kv : by) { if (shown++ >= 12) break; printf(" [%x:", kv.first); for (int b : kv.second) printf(" %d", b); printf("]"); } printf("\n");
        long long tmin = hs[0]; for (int b = 0; b < nb; ++b) tmin = std::min(tmin, hs[b]);
        int late = 0; for (int b = 0; b < nb; ++b) if (hs[b] - tmin > 1000) ++late;
        printf("  workgroups of the 1024 that started > 10 us after the first: %d (second wave of dispatch)\n", late);
    }
    const int ntiles = T * (T - 1) / 2 - 1;
    const double flop1 = 2.0 * 128 * 128 * 128 * ntiles;
    auto run = [&](const char* name, auto launch) {
        for (int w = 0; w < 3; ++w) launch();
        hipDeviceSynchronize();
        float best = 1e9f, sum = 0.f;
        for (int r = 0; r < 12; ++r) {
            hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms); sum += ms;
        }
        printf("%-52s %8.3f ms best %8.3f avg  %7.2f TFLOP/s (best)\n", name, best, sum / 12, flop1 / (best * 1e-3) / 1e12);
    };
    run("k_syrk_update part 2, 8-byte C loads and stores", [&] { hipLaunchKernelGGL((k_syrk_update<false, 0>), dim3(ntiles), dim3(512), lds_bytes, 0, S, ld, 0, panel, 2, (const double*)nullptr); });
    run("k_syrk_update part 2, 16-byte C loads", [&] { hipLaunchKernelGGL((k_syrk_update<false, 1>), dim3(ntiles), dim3(512), lds_bytes, 0, S, ld, 0, panel, 2, (const double*)nullptr); });
    run("k_syrk_update part 2, 16-byte C stores", [&] { hipLaunchKernelGGL((k_syrk_update<false, 2>), dim3(ntiles), dim3(512), lds_bytes, 0, S, ld, 0, panel, 2, (const double*)nullptr); });
    run("k_syrk_update part 2, 16-byte C loads + stores", [&] { hipLaunchKernelGGL((k_syrk_update<false, 3>), dim3(ntiles), dim3(512), lds_bytes, 0, S, ld, 0, panel, 2, (const double*)nullptr); });
    run("k_syrk_update part 2, 16-byte, non-temporal", [&] { hipLaunchKernelGGL((k_syrk_update<true, 3>), dim3(ntiles), dim3(512), lds_bytes, 0, S, ld, 0, panel, 2, (const double*)nullptr); });
    run("k_syrk_update part 2, 8-byte again", [&] { hipLaunchKernelGGL((k_syrk_update<false, 0>), dim3(ntiles), dim3(512), lds_bytes, 0, S, ld, 0, panel, 2, (const double*)nullptr); });
    run("k_syrk_update part 2, 16-byte loads + stores again", [&] { hipLaunchKernelGGL((k_syrk_update<false, 3>), dim3(ntiles), dim3(512), lds_bytes, 0, S, ld, 0, panel, 2, (const double*)nullptr); });
    if (getenv("UB_ONLY_C16")) return 0;
    for (int split : {256, 1}) {
        for (int d : {0, 5, 10, 15, 20, 25, 30}) {
            char nm[96]; snprintf(nm, sizeof nm, "persistent static, %s delayed %2d us", split == 256 ? "b >= 256" : "odd b (ctrl)", d);
            if (split == 256) run(nm, [&] { hipLaunchKernelGGL(k_pers<0>, dim3(512), dim3(512), lds_bytes, 0, S, ld, 0, panel, ntiles, d * 100, 256, counter); });
            else if (d == 15) run("persistent static, CONTROL: b >= 128 delayed 15 us", [&] { hipLaunchKernelGGL(k_pers<0>, dim3(512), dim3(512), lds_bytes, 0, S, ld, 0, panel, ntiles, d * 100, 128, counter); });
        }
    }
    for (int d : {0, 10, 15, 20, 25}) {
        char nm[96]; snprintf(nm, sizeof nm, "persistent atomic counter, b >= 256 delayed %2d us", d);
        run(nm, [&] { hipMemsetAsync(counter, 0, 4, 0); hipLaunchKernelGGL(k_pers<1>, dim3(512), dim3(512), lds_bytes, 0, S, ld, 0, panel, ntiles, d * 100, 256, counter); });
    }
    // smaller launches (T = 40, 25): where the device is no longer full for many rounds
    for (int Ts : {40, 25}) {
        const int nt = Ts * (Ts - 1) / 2 - 1;
        const double fl = 2.0 * 128 * 128 * 128 * nt;
        for (int v = 0; v < 3; ++v) {
            float best = 1e9f;
            for (int r = 0; r < 12; ++r) {
                hipMemsetAsync(counter, 0, 4, 0);
                hipEventRecord(e0, 0);
                if (v == 0) hipLaunchKernelGGL((k_syrk_update<false, 0>), dim3(nt), dim3(512), lds_bytes, 0, S, ld, 0, panel, 2, (const double*)nullptr);
                if (v == 1) hipLaunchKernelGGL(k_pers<1>, dim3(std::min(512, nt)), dim3(512), lds_bytes, 0, S, ld, 0, panel, nt, 0, 256, counter);
                if (v == 2) hipLaunchKernelGGL(k_pers<1>, dim3(std::min(512, nt)), dim3(512), lds_bytes, 0, S, ld, 0, panel, nt, 1500, 256, counter);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
            }
            printf("T = %d (%d tiles) %-28s %8.3f ms %7.2f TFLOP/s\n", Ts, nt, v == 0 ? "shipped" : v == 1 ? "persistent atomic" : "persistent atomic + 15 us", best, fl / (best * 1e-3) / 1e12);
        }
    }
    return 0;
}
