// ubench_rocblas.cpp -- what does the vendor library reach on the shapes of the bulk trailing update?
//   C (n x n, lower) -= A (n x k) A^T  as rocblas_dsyrk and as rocblas_dgemm (full square), n = 128 T, k = 128 / 256.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_rocblas.cpp -lrocblas -o scripts/_bin/ubench_rocblas
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <vector>
#include <algorithm>

int main()
{
    rocblas_handle h; rocblas_create_handle(&h);
    const int nmax = 128 * 70, kmax = 512;
    double *C, *A;
    hipMalloc((void**)&C, (size_t)nmax * nmax * 8); hipMemset(C, 0, (size_t)nmax * nmax * 8);
    hipMalloc((void**)&A, (size_t)nmax * kmax * 8);
    std::vector<double> ha((size_t)nmax * kmax);
    for (size_t q = 0; q < ha.size(); ++q) ha[q] = 1e-3 * (double)((q * 2654435761u) % 1000);
    hipMemcpy(A, ha.data(), ha.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double alpha = -1.0, beta = 1.0;
    for (int T : { 68, 50, 30 })
        for (int k : { 128, 256, 512 }) {
            const int n = 128 * T;
            auto time = [&](auto f) {
                for (int w = 0; w < 3; ++w) f();
                hipDeviceSynchronize();
                float best = 1e9f;
                for (int r = 0; r < 8; ++r) { hipEventRecord(e0, 0); f(); hipEventRecord(e1, 0); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms); }
                return best;
            };
            // column-major: C = C - A A^T with A n x k (lda = n)
            const float t_syrk = time([&] { rocblas_dsyrk(h, rocblas_fill_lower, rocblas_operation_none, n, k, &alpha, A, n, &beta, C, nmax); });
            const float t_gemm = time([&] { rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_transpose, n, n, k, &alpha, A, n, A, n, &beta, C, nmax); });
            // row-major panel (k contiguous per row, as the factorisation stores it): A^T is k x n column-major with lda = k
            const float t_syrk_t = time([&] { rocblas_dsyrk(h, rocblas_fill_lower, rocblas_operation_transpose, n, k, &alpha, A, k, &beta, C, nmax); });
            const double f_syrk = (double)n * n * k, f_gemm = 2.0 * n * n * k;
            printf("T %2d (n %5d) k %3d: dsyrk(N) %7.3f ms %6.1f TFLOP/s | dsyrk(T) %7.3f ms %6.1f | dgemm full square %7.3f ms %6.1f TFLOP/s\n", T, n, k,
                   t_syrk, f_syrk / (t_syrk * 1e-3) / 1e12, t_syrk_t, f_syrk / (t_syrk_t * 1e-3) / 1e12, t_gemm, f_gemm / (t_gemm * 1e-3) / 1e12);
        }
    // the shapes a two-level blocked factorisation would use: row-major S (ld = 9088), strip of W columns x R rows below it,
    // C_rect -= A_R A_C^T  ==  column-major  C^T (W x R) -= op_T(A_C: k x W, ld) * A_R^T (k x R, ld)   (TN gemm, everything with ld = 9088)
    {
        const int ld = 9088;
        double* S; hipMalloc((void**)&S, (size_t)ld * ld * 8); hipMemset(S, 0, (size_t)ld * ld * 8);
        for (int W : { 512, 1024, 2048 })
            for (int Rq : { 7000, 4000, 2000 })
                for (int k : { 256, 512 }) {
                    const int R = std::min(Rq, ld - 1024 - W);
                    const double* Ac = S + (size_t)1024 * ld;              // rows 1024 .. 1024+W, columns 0 .. k
                    const double* Ar = S + (size_t)(1024 + W) * ld;        // rows below
                    double* Cr = S + (size_t)(1024 + W) * ld + 1024;       // C block: rows below, columns 1024 .. 1024+W
                    auto f = [&] { rocblas_dgemm(h, rocblas_operation_transpose, rocblas_operation_none, W, R, k, &alpha, Ac, ld, Ar, ld, &beta, Cr, ld); };
                    for (int w = 0; w < 3; ++w) f();
                    hipDeviceSynchronize();
                    float best = 1e9f;
                    for (int r = 0; r < 8; ++r) { hipEventRecord(e0, 0); f(); hipEventRecord(e1, 0); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms); }
                    printf("TN strip W %4d x R %4d, k %3d (ld %d): %7.3f ms %6.1f TFLOP/s\n", W, R, k, ld, best, 2.0 * W * R * k / (best * 1e-3) / 1e12);
                }
    }
    return 0;
}
