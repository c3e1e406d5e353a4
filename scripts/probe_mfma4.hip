// Empirical lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950: one-hot A lane x one-hot B lane -> which D lane.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int* out)
{
    const int l = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
            double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            unsigned long long m = __ballot(d != 0.0);
            if (l == 0) out[la * 64 + lb] = m ? (__builtin_ctzll(m) | (__builtin_popcountll(m) << 8)) : -1;
        }
}
int main()
{
    int* d; hipMalloc(&d, 4096 * sizeof(int));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    static int h[4096]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int la = 0; la < 64; ++la) {
        printf("A%02d:", la);
        for (int lb = 0; lb < 64; ++lb) if (h[la * 64 + lb] >= 0) printf(" B%02d->D%02d(x%d)", lb, h[la * 64 + lb] & 255, h[la * 64 + lb] >> 8);
        printf("\n");
    }
    return 0;
}
