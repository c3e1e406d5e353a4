// Lane layout of v_mfma_f64_16x16x4_f64 on gfx950, probed: D = A (16 x 4) * B (4 x 16).
//   hipcc --offload-arch=gfx950 -O3 scripts/probe_mfma16.hip -o scripts/_bin/probe_mfma16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(const double* a_in, const double* b_in, double* d_out)
{
    v4d acc = { 0.0, 0.0, 0.0, 0.0 };
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a_in[threadIdx.x], b_in[threadIdx.x], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d_out[4 * threadIdx.x + r] = acc[r];
}
int main()
{
    double ha[64], hb[64], hd[256], *a, *b, *d;
    hipMalloc(&a, 512); hipMalloc(&b, 512); hipMalloc(&d, 2048);
    // hypothesis: A[i][k] at lane i + 16 k; B[k][j] at lane j + 16 k; D[i][j]: lane l, register r: i = 4 (l / 16) + r, j = l % 16
    double A[16][4], Bm[4][16], D[16][16];
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i][k] = 1.0 + i + 0.01 * k;
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) Bm[k][j] = 2.0 + 0.1 * j + 3.0 * k;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { D[i][j] = 0; for (int k = 0; k < 4; ++k) D[i][j] += A[i][k] * Bm[k][j]; }
    for (int l = 0; l < 64; ++l) { ha[l] = A[l % 16][l / 16]; hb[l] = Bm[l / 16][l % 16]; }
    hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd, d, 2048, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) { const double want = D[4 * (l / 16) + r][l % 16]; if (fabs(hd[4 * l + r] - want) > 1e-9) ++bad; }
    printf("hypothesis (A lane = i + 16 k, B lane = j + 16 k, D[4 (l/16) + r][l %% 16]): %s (%d mismatches)\n", bad ? "WRONG" : "confirmed", bad);
    if (bad) {
        // brute force: where does D[i][j] land?
        for (int l = 0; l < 8; ++l) { printf("lane %d:", l); for (int r = 0; r < 4; ++r) { int fi = -1, fj = -1; for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if (fabs(hd[4 * l + r] - D[i][j]) < 1e-9) { fi = i; fj = j; } printf(" r%d=(%d,%d)", r, fi, fj); } printf("\n"); }
        for (int l = 16; l < 20; ++l) { printf("lane %d:", l); for (int r = 0; r < 4; ++r) { int fi = -1, fj = -1; for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if (fabs(hd[4 * l + r] - D[i][j]) < 1e-9) { fi = i; fj = j; } printf(" r%d=(%d,%d)", r, fi, fj); } printf("\n"); }
    }
    return 0;
}
