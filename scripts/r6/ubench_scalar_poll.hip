// EXPERIMENT (round 6): can a wave poll a word another XCD writes through the SCALAR memory path (own queue: lgkmcnt, not behind the wave's vector
// prefetches)?  Producer workgroup (block 0) waits ~20 us, then stores a value with an agent-scope vector store.  Consumers (blocks 1 .. N, on
// other CUs / XCDs) poll with (mode 0) agent-scope vector loads, (mode 1) s_load_dwordx2 glc, (mode 2) buffer_inv sc1 + s_load_dwordx2 glc, and
// report how many 100 MHz ticks after the producer's store they saw it (or that they never did within the bound).
// The word lives in (mem 0) ordinary hipMalloc memory, (mem 1) hipExtMallocWithFlags(hipDeviceMallocUncached), (mem 2) ...Finegrained.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__device__ __forceinline__ long long wall() { return (long long)__builtin_readcyclecounter(); }
__device__ __forceinline__ long long wall100() { long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }

__global__ __launch_bounds__(64) void k_poll(unsigned long long* word, long long* out, int mode, long long delay_ticks)
{
    const int b = blockIdx.x;
    if (b == 0) {
        const long long t0 = wall100();
        while (wall100() - t0 < delay_ticks) __builtin_amdgcn_s_sleep(10);
        const long long ts = wall100();
        if (threadIdx.x == 0) { __hip_atomic_store(word, 0x1234567812345678ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); out[0] = ts; }
        return;
    }
    unsigned long long v = 0;
    long long seen = -1;
    unsigned xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    for (int it = 0; it < 200000; ++it) {
        if (mode == 0) v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else {
            if (mode == 2) asm volatile("buffer_inv sc1" ::: "memory");
            unsigned long long sv;
            asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(sv) : "s"(word) : "memory");
            v = sv;
        }
        if (v == 0x1234567812345678ull) { seen = wall100(); break; }
    }
    if (threadIdx.x == 0) { out[2 * b] = seen; out[2 * b + 1] = (long long)(xcc & 15u); }
}

int main()
{
    const int NB = 33;
    long long* d_out; hipMalloc(&d_out, sizeof(long long) * 2 * NB);
    const char* mem_name[3] = { "hipMalloc", "uncached", "fine-grained" };
    const char* mode_name[3] = { "vector load, agent scope", "s_load glc", "buffer_inv sc1 + s_load glc" };
    for (int mem = 0; mem < 3; ++mem) {
        unsigned long long* w = nullptr;
        hipError_t e = hipSuccess;
        if (mem == 0) e = hipMalloc(&w, 4096);
        else e = hipExtMallocWithFlags((void**)&w, 4096, mem == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
        if (e != hipSuccess) { printf("%s: allocation failed (%s)\n", mem_name[mem], hipGetErrorString(e)); continue; }
        for (int mode = 0; mode < 3; ++mode) {
            long long h[2 * NB];
            double worst = 0, sum = 0; int cnt = 0, never = 0;
            for (int rep = 0; rep < 5; ++rep) {
                hipMemset(w, 0, 4096); hipMemset(d_out, 0xff, sizeof(long long) * 2 * NB); hipDeviceSynchronize();
                hipLaunchKernelGGL(k_poll, dim3(NB), dim3(64), 0, 0, w, d_out, mode, 2000LL);
                hipDeviceSynchronize();
                hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
                for (int b = 1; b < NB; ++b) {
                    if (h[2 * b] < 0) { ++never; continue; }
                    const double us = (h[2 * b] - h[0]) / 100.0;
                    sum += us; ++cnt; if (us > worst) worst = us;
                }
            }
            printf("%-12s | %-28s | seen by %3d of %3d pollers, mean %.2f us after the store, worst %.2f us; never within the bound: %d\n",
                   mem_name[mem], mode_name[mode], cnt, cnt + never, cnt ? sum / cnt : -1.0, worst, never);
        }
        hipFree(w);
    }
    return 0;
}
