// FP64 throughput microbenchmarks for gfx950: v_mfma_f64_16x16x4_f64 vs v_fma_f64 (VALU), per-CU and chip.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters, double seed)
{
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4d){ seed, seed, seed, seed };
    double a = seed * threadIdx.x, b = seed + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
typedef double v1d;
__global__ __launch_bounds__(256) void k_mfma4(double* out, int iters, double seed)
{
    double acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = seed;
    double a = seed * threadIdx.x, b = seed + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k_fma(double* out, int iters, double seed)
{
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = seed + i;
    double a = 1.0 + 1e-9 * threadIdx.x, b = 1e-9 * seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F> float timeit(F f)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main()
{
    double* out; hipMalloc(&out, sizeof(double) * 256 * 4096);
    const int iters = 4000;
    for (int blocks : { 256, 512, 1024 }) {
        float ms = timeit([&] { hipLaunchKernelGGL(k_mfma<16>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); });
        double flops = (double)blocks * 4 * iters * 16 * 2048.0;
        double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 16 * (blocks / 256.0));   // per MFMA per SIMD if 1 wave/SIMD/block
        printf("mfma_f64_16x16x4 x16acc blocks=%d: %.3f ms  %.2f TFLOP/s  (~%.1f cyc/MFMA/SIMD @2.4GHz)\n", blocks, ms, flops / ms / 1e9, cyc);
    }
    { float ms = timeit([&] { hipLaunchKernelGGL(k_mfma<4>, dim3(1024), dim3(256), 0, 0, out, iters, 1.0); });
      printf("mfma_f64_16x16x4 x4acc blocks=1024: %.3f ms  %.2f TFLOP/s\n", ms, 1024.0 * 4 * iters * 4 * 2048.0 / ms / 1e9); }
    { float ms = timeit([&] { hipLaunchKernelGGL(k_mfma4, dim3(1024), dim3(256), 0, 0, out, iters, 1.0); });
      printf("mfma_f64_4x4x4 x16acc blocks=1024: %.3f ms  %.2f TFLOP/s\n", ms, 1024.0 * 4 * iters * 16 * 512.0 / ms / 1e9); }
    for (int blocks : { 256, 1024, 2048 }) {
        float ms = timeit([&] { hipLaunchKernelGGL(k_fma<32>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); });
        double flops = (double)blocks * 256 * iters * 32 * 2.0;
        printf("v_fma_f64 x32acc blocks=%d: %.3f ms  %.2f TFLOP/s\n", blocks, ms, flops / ms / 1e9);
    }
    return 0;
}
