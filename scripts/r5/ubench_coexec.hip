// Does the FP64 matrix pipe of gfx950 run concurrently with VALU work of OTHER waves on the same SIMD?
// Workgroups of 512 threads = 8 waves = 2 per SIMD.  mode bits: waves with (wave & 1) == 0 run `a`, the others run `b`:
//   0 idle, 1 = 16 independent v_mfma_f64_16x16x4 per iteration, 2 = v_fma_f64 chain x32, 3 = DPP quad broadcasts (v_mov_b32_dpp) x32,
//   4 = v_mfma_f64_4x4x4_4b x16, 5 = ds_read_b128 x8 per iteration
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double work(int mode, int iters, double seed, d2* lds)
{
    double s = 0.0;
    if (mode == 1) {
        v4d acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = (v4d){ seed, seed, seed, seed };
        double a = seed * threadIdx.x, b = seed + threadIdx.x;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else if (mode == 2) {
        double acc[32];
        for (int i = 0; i < 32; ++i) acc[i] = seed + i;
        double a = 1.0 + 1e-9 * threadIdx.x, b = 1e-9 * seed;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) acc[i] = __builtin_fma(acc[i], a, b);
        }
        for (int i = 0; i < 32; ++i) s += acc[i];
    } else if (mode == 3) {
        int v[32];
        for (int i = 0; i < 32; ++i) v[i] = threadIdx.x + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) v[i] = __builtin_amdgcn_mov_dpp(v[i], 0x55, 0xf, 0xf, true) + 1;
        }
        for (int i = 0; i < 32; ++i) s += v[i];
    } else if (mode == 4) {
        double acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = seed;
        double a = seed * threadIdx.x, b = seed + threadIdx.x;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) s += acc[i];
    } else if (mode == 5) {
        d2 t = { 0.0, 0.0 };
        const int lane = threadIdx.x & 63;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { const d2 r = lds[(lane + 64 * i + it) & 1023]; t.x += r.x; t.y += r.y; }
        }
        s = t.x + t.y;
    }
    return s;
}

__global__ __launch_bounds__(512) void k_mix(double* out, int iters, double seed, int a, int b)
{
    __shared__ d2 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 512) lds[i] = (d2){ seed, seed };
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mode = ((wave >> 2) & 1) ? b : a;        // waves 0-3 (one per SIMD): a; waves 4-7: b
    out[blockIdx.x * 512 + threadIdx.x] = work(mode, iters, seed, lds);
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
    double* out; hipMalloc(&out, sizeof(double) * 512 * 1024);
    const int iters = 2000, blocks = 256;        // one workgroup per CU
    const char* names[] = { "idle", "mfma16x16x4 x16", "v_fma_f64 x32", "dpp mov x32", "mfma4x4x4_4b x16", "ds_read_b128 x8" };
    const int pairs[][2] = { {1,0},{2,0},{3,0},{4,0},{5,0},{1,1},{1,2},{1,3},{1,5},{4,2},{4,3},{4,5},{2,3},{2,2},{2,5} };
    for (auto& p : pairs) {
        float ms = timeit([&] { hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(512), 0, 0, out, iters, 1.0, p[0], p[1]); });
        printf("%-18s + %-18s : %.3f ms\n", names[p[0]], names[p[1]], ms);
    }
    return 0;
}
