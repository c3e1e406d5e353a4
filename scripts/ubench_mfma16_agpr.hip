// Is v_mfma_f64_16x16x4_f64 full rate on gfx950 when its accumulators live in AGPRs?  (With VGPR accumulators the plain builtin loop
// measured 36 TFLOP/s, ubench_fp64.hip, while rocBLAS' MI16x16x4 kernels reach 67.)
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_mfma16_agpr.hip -o scripts/_bin/ubench_mfma16_agpr
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC, bool AGPR, bool VARY>
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters, double seed)
{
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4d){ 0.0, 0.0, 0.0, 0.0 };
    double a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = seed * 1e-3 * (threadIdx.x + i); b[i] = 1e-3 * (seed + threadIdx.x - i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            const double av = VARY ? a[i & 3] : a[0], bv = VARY ? b[(i >> 2) & 3] : b[0];
            if (AGPR) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(av), "v"(bv));
            else asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(av), "v"(bv));
        }
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, bool AGPR>
__global__ __launch_bounds__(256) void k_mfma4(double* out, int iters, double seed)
{
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    double a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = seed * 1e-3 * (threadIdx.x + i); b[i] = 1e-3 * (seed + threadIdx.x - i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (AGPR) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[i & 3]), "v"(b[(i >> 2) & 3]));
            else asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 3]), "v"(b[(i >> 2) & 3]));
        }
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F> float timeit(F f)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) f();
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = best < ms ? best : ms; }
    return best;
}
int main()
{
    double* out; hipMalloc(&out, sizeof(double) * 256 * 4096);
    const int iters = 2000, blocks = 1024;
    auto rep = [&](const char* name, float ms, int nacc) { printf("%-44s %.3f ms  %6.2f TFLOP/s\n", name, ms, (double)blocks * 4 * iters * nacc * 2048.0 / ms / 1e9); };
    rep("16x16x4 VGPR acc x16, same operands", timeit([&] { hipLaunchKernelGGL((k_mfma<16, false, false>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); }), 16);
    rep("16x16x4 AGPR acc x16, same operands", timeit([&] { hipLaunchKernelGGL((k_mfma<16, true, false>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); }), 16);
    rep("16x16x4 VGPR acc x16, 4 x 4 operands", timeit([&] { hipLaunchKernelGGL((k_mfma<16, false, true>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); }), 16);
    rep("16x16x4 AGPR acc x16, 4 x 4 operands", timeit([&] { hipLaunchKernelGGL((k_mfma<16, true, true>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); }), 16);
    rep("16x16x4 AGPR acc x8", timeit([&] { hipLaunchKernelGGL((k_mfma<8, true, true>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); }), 8);
    rep("16x16x4 AGPR acc x4", timeit([&] { hipLaunchKernelGGL((k_mfma<4, true, true>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); }), 4);
    rep("16x16x4 VGPR acc x4", timeit([&] { hipLaunchKernelGGL((k_mfma<4, false, true>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); }), 4);
    auto rep4 = [&](const char* name, float ms, int nacc) { printf("%-44s %.3f ms  %6.2f TFLOP/s\n", name, ms, (double)blocks * 4 * iters * nacc * 512.0 / ms / 1e9); };
    rep4("4x4x4_4b VGPR acc x32", timeit([&] { hipLaunchKernelGGL((k_mfma4<32, false>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); }), 32);
    rep4("4x4x4_4b AGPR acc x32", timeit([&] { hipLaunchKernelGGL((k_mfma4<32, true>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); }), 32);
    return 0;
}
