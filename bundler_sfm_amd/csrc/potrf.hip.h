// potrf.hip.h -- dense FP64 Cholesky solve of the reduced camera system S x = E on gfx950.
//
// Replaces sba_Axb_Chol = LAPACK dpotrf("U") + dpotrs (lib/sba-1.5/sba_lapack.c:374-485, called from
// lib/sba-1.5/sba_levmar.c:1368).  S is symmetric, so the reference's column-major "U" factorisation
// is this file's row-major LOWER factorisation S = L L^T; only the lower triangle of S is read.
//
// MI355X design (the one MFMA-bound contraction of the LM iteration, ~N^3/3 FP64 flop):
//   * right-looking tiled factorisation, tile NB = 128; S is padded to a multiple of NB (identity in
//     the padding) so every kernel works on full tiles;
//   * diagonal tile: ONE 1024-thread workgroup, register-tiled (4x4 cyclic per thread) fused
//     potrf + triangular inverse with a single barrier per column (pivot column / inverse row are
//     double-buffered through LDS);
//   * panel: L_ik = S_ik * inv(L_kk)^T is a GEMM on v_mfma_f64_16x16x4_f64 (no triangular solve);
//     it also writes a compact copy of the panel (contiguous 128 KB tiles) that the trailing update reads;
//   * trailing update S_ij -= L_ik L_jk^T on the lower triangle: 128x128 tile per 256-thread workgroup,
//     4 waves x (4x4 MFMA tiles of 16x16) = 64 FP64 accumulators per lane, K staged through LDS in
//     16-wide chunks (row stride padded to 18 doubles => conflict-free ds_read_b64 of the fragments),
//     next chunk prefetched into registers while the MFMAs of the current one issue;
//   * forward / backward substitution use the stored inverse diagonal tiles: one launch per tile step.
// f64 MFMA fragment layout (cdna_hip_programming.md section 3): A[l&15][l>>4], B[l>>4][l&15],
// C/D row = (l>>4) + 4*reg, col = l&15.
#pragma once
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cmath>

namespace bsfm {

constexpr int POTRF_NB = 128;
constexpr int GEMM_KC = 16;
constexpr int GEMM_LDS_STRIDE = GEMM_KC + 2;

typedef double v4d __attribute__((ext_vector_type(4)));

struct PotrfWorkspace {
    int ld = 0, nblk = 0, backend = 0;
    double* panel = nullptr;   // (nblk-1) tiles of NB x NB, compact copy of the current panel
    double* linv = nullptr;    // nblk tiles: inverse of each diagonal factor tile
    double* y = nullptr;       // ld
    double* xs = nullptr;      // ld
    double* etmp = nullptr;    // ld (rhs working copy)
    // rocSOLVER cross-check backend
    void* rs_lib = nullptr; void* rb_lib = nullptr; void* rb_handle = nullptr;
    int (*rs_potrf)(void*, int, int, double*, int, int*) = nullptr;
    int (*rs_potrs)(void*, int, int, int, double*, int, double*, int) = nullptr;
    int (*rb_set_stream)(void*, hipStream_t) = nullptr;
    int (*rb_destroy)(void*) = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double ms = 0.0; int cnt = 0;
};

// ------------------------------------------------------------------------------------------------
// C(128x128) = A(128xK, row-major lda) * B(128xK, row-major ldb)^T, per-wave 64x64 quadrant in acc[4][4].
__device__ __forceinline__ void gemm_nt_128(const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb,
                                            int K, double* __restrict__ lds, v4d (&acc)[4][4])
{
    double* As = lds;
    double* Bs = lds + 128 * GEMM_LDS_STRIDE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;
    // staging: 128 rows x 16 doubles = 1024 double2 per operand -> 4 per thread
    double2 pa0, pa1, pa2, pa3, pb0, pb1, pb2, pb3;
    const int srow = tid >> 3, sc2 = (tid & 7) * 2;     // rows srow + 32 q
    const double* Ag = A + (size_t)srow * lda + sc2;
    const double* Bg = B + (size_t)srow * ldb + sc2;
    double* Asw = As + srow * GEMM_LDS_STRIDE + sc2;
    double* Bsw = Bs + srow * GEMM_LDS_STRIDE + sc2;
#define BSFM_GLOAD(kc)                                                                      \
    pa0 = *reinterpret_cast<const double2*>(Ag + (kc));                                     \
    pa1 = *reinterpret_cast<const double2*>(Ag + (size_t)32 * lda + (kc));                  \
    pa2 = *reinterpret_cast<const double2*>(Ag + (size_t)64 * lda + (kc));                  \
    pa3 = *reinterpret_cast<const double2*>(Ag + (size_t)96 * lda + (kc));                  \
    pb0 = *reinterpret_cast<const double2*>(Bg + (kc));                                     \
    pb1 = *reinterpret_cast<const double2*>(Bg + (size_t)32 * ldb + (kc));                  \
    pb2 = *reinterpret_cast<const double2*>(Bg + (size_t)64 * ldb + (kc));                  \
    pb3 = *reinterpret_cast<const double2*>(Bg + (size_t)96 * ldb + (kc));
    BSFM_GLOAD(0)
    for (int kc = 0; kc < K; kc += GEMM_KC) {
        __syncthreads();          // previous chunk fully consumed
        *reinterpret_cast<double2*>(Asw) = pa0;
        *reinterpret_cast<double2*>(Asw + 32 * GEMM_LDS_STRIDE) = pa1;
        *reinterpret_cast<double2*>(Asw + 64 * GEMM_LDS_STRIDE) = pa2;
        *reinterpret_cast<double2*>(Asw + 96 * GEMM_LDS_STRIDE) = pa3;
        *reinterpret_cast<double2*>(Bsw) = pb0;
        *reinterpret_cast<double2*>(Bsw + 32 * GEMM_LDS_STRIDE) = pb1;
        *reinterpret_cast<double2*>(Bsw + 64 * GEMM_LDS_STRIDE) = pb2;
        *reinterpret_cast<double2*>(Bsw + 96 * GEMM_LDS_STRIDE) = pb3;
        __syncthreads();
        if (kc + GEMM_KC < K) { BSFM_GLOAD(kc + GEMM_KC) }
#pragma unroll
        for (int kk = 0; kk < GEMM_KC; kk += 4) {
            double a[4], b[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                a[t] = As[(wr + 16 * t + (lane & 15)) * GEMM_LDS_STRIDE + kk + (lane >> 4)];
                b[t] = Bs[(wc + 16 * t + (lane & 15)) * GEMM_LDS_STRIDE + kk + (lane >> 4)];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    acc[t][u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], b[u], acc[t][u], 0, 0, 0);
        }
    }
#undef BSFM_GLOAD
}

// Panel: X_i = S_ik * Linv_k^T for i = k+1 .. nblk-1; writes X back into S (it is L) and into the compact panel.
__global__ __launch_bounds__(256, 2) void k_trsm_panel(double* __restrict__ S, int ld, int k,
        const double* __restrict__ Linv, double* __restrict__ panel)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int i = k + 1 + blockIdx.x;
    double* Sik = S + ((size_t)i * POTRF_NB) * ld + (size_t)k * POTRF_NB;
    v4d acc[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[t][u] = (v4d){ 0.0, 0.0, 0.0, 0.0 };
    gemm_nt_128(Sik, ld, Linv, POTRF_NB, POTRF_NB, lds, acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;
    double* Pt = panel + (size_t)blockIdx.x * POTRF_NB * POTRF_NB;
    __syncthreads();   // every wave has finished reading S_ik through LDS before it is overwritten
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int row = wr + 16 * t + (lane >> 4) + 4 * g, col = wc + 16 * u + (lane & 15);
                const double v = acc[t][u][g];
                Sik[(size_t)row * ld + col] = v;
                Pt[row * POTRF_NB + col] = v;
            }
}

// Trailing update: S_ij -= P_a P_b^T for k < j <= i (a = i-k-1, b = j-k-1), one tile per workgroup.
__global__ __launch_bounds__(256, 2) void k_syrk_update(double* __restrict__ S, int ld, int k, const double* __restrict__ panel)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int t = blockIdx.x;
    int a = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((a + 1) * (a + 2) / 2 <= t) ++a;
    while (a * (a + 1) / 2 > t) --a;
    const int b = t - a * (a + 1) / 2;
    const int i = k + 1 + a, j = k + 1 + b;
    v4d acc[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[q][u] = (v4d){ 0.0, 0.0, 0.0, 0.0 };
    gemm_nt_128(panel + (size_t)a * POTRF_NB * POTRF_NB, POTRF_NB, panel + (size_t)b * POTRF_NB * POTRF_NB, POTRF_NB,
                POTRF_NB, lds, acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;
    double* Sij = S + ((size_t)i * POTRF_NB) * ld + (size_t)j * POTRF_NB;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int row = wr + 16 * q + (lane >> 4) + 4 * g, col = wc + 16 * u + (lane & 15);
                Sij[(size_t)row * ld + col] -= acc[q][u][g];
            }
}

// Diagonal tile: fused Cholesky + inverse of the factor, 1024 threads as a 32x32 grid, thread (ti,tj) owns
// rows ti+32a, cols tj+32b (a,b < 4) of both the tile A and X (X starts as I and ends as inv(L)).
// Per column j: s = 1/sqrt(pivot); L[:,j] = raw*s; X[j,:] = raw*s; A -= l l^T ; X -= l xrow.
// The raw column j+1 of A and raw row j+1 of X are published to LDS right after the update of step j,
// so each column costs ONE barrier.  n_valid rows/cols of the tile belong to S, the rest is identity padding.
__global__ __launch_bounds__(1024) void k_potrf_diag(double* __restrict__ S, int ld, int k, int n_total,
        double* __restrict__ Linv, int* __restrict__ info)
{
    __shared__ double colbuf[2][POTRF_NB];
    __shared__ double rowbuf[2][POTRF_NB];
    const int ti = threadIdx.x >> 5, tj = threadIdx.x & 31;
    const int base = k * POTRF_NB;
    double* T = S + (size_t)base * ld + base;
    double a[4][4], x[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = ti + 32 * p, c = tj + 32 * q;
            double v = 0.0;
            if (base + r < n_total && base + c < n_total) { if (c <= r) v = T[(size_t)r * ld + c]; }
            else if (r == c) v = 1.0;
            a[p][q] = v;
            x[p][q] = (r == c) ? 1.0 : 0.0;
        }
    // publish raw column 0 / row 0
    if (tj == 0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) colbuf[0][ti + 32 * p] = a[p][0];
    }
    if (ti == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) rowbuf[0][tj + 32 * q] = x[0][q];
    }
    __syncthreads();
    for (int j = 0; j < POTRF_NB; ++j) {
        const int cur = j & 1, nxt = cur ^ 1;
        const double piv = colbuf[cur][j];
        if (!(piv > 0.0) && threadIdx.x == 0 && base + j < n_total) atomicCAS(info, 0, base + j + 1);
        const double s = 1.0 / sqrt(piv);
        double lr[4], lc[4], xr[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) lr[p] = colbuf[cur][ti + 32 * p] * s;
#pragma unroll
        for (int q = 0; q < 4; ++q) { lc[q] = colbuf[cur][tj + 32 * q] * s; xr[q] = rowbuf[cur][tj + 32 * q] * s; }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = ti + 32 * p, c = tj + 32 * q;
                if (c == j) { if (r >= j) a[p][q] = lr[p]; }          // final L column j
                else if (r > j && c > j) a[p][q] -= lr[p] * lc[q];
                if (r == j) x[p][q] = xr[q];                           // final inverse row j
                else if (r > j) x[p][q] -= lr[p] * xr[q];
            }
        if (j + 1 < POTRF_NB) {
            const int jn = j + 1;
            const int jq = jn >> 5;   // static selects: a runtime register-array index would go to scratch
            if (tj == (jn & 31)) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const double v = jq == 0 ? a[p][0] : (jq == 1 ? a[p][1] : (jq == 2 ? a[p][2] : a[p][3]));
                    colbuf[nxt][ti + 32 * p] = v;
                }
            }
            if (ti == (jn & 31)) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double v = jq == 0 ? x[0][q] : (jq == 1 ? x[1][q] : (jq == 2 ? x[2][q] : x[3][q]));
                    rowbuf[nxt][tj + 32 * q] = v;
                }
            }
        }
        __syncthreads();
    }
    double* Li = Linv + (size_t)k * POTRF_NB * POTRF_NB;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = ti + 32 * p, c = tj + 32 * q;
            if (c <= r) T[(size_t)r * ld + c] = a[p][q];
            Li[r * POTRF_NB + c] = (c <= r) ? x[p][q] : 0.0;
        }
}

// forward substitution step: tiles i > k do E_i -= L_ik y_k ; tile k+1 then computes y_{k+1} = Linv_{k+1} E_{k+1}.
// k == -1 bootstraps y_0 = Linv_0 E_0.
__global__ __launch_bounds__(256) void k_fwd_step(const double* __restrict__ S, int ld, int k, const double* __restrict__ Linv,
                                                  double* __restrict__ E, double* __restrict__ y)
{
    __shared__ double vec[POTRF_NB];
    __shared__ double red[2][POTRF_NB];
    const int i = k + 1 + blockIdx.x;
    const int r = threadIdx.x & 127, h = threadIdx.x >> 7;
    if (k >= 0) {
        if (threadIdx.x < POTRF_NB) vec[threadIdx.x] = y[(size_t)k * POTRF_NB + threadIdx.x];
        __syncthreads();
        const double* Lr = S + ((size_t)i * POTRF_NB + r) * ld + (size_t)k * POTRF_NB + 64 * h;
        double s = 0.0;
#pragma unroll 8
        for (int c = 0; c < 64; ++c) s += Lr[c] * vec[64 * h + c];
        red[h][r] = s;
        __syncthreads();
        if (threadIdx.x < POTRF_NB) E[(size_t)i * POTRF_NB + r] -= red[0][r] + red[1][r];
        if (blockIdx.x != 0) return;
        __syncthreads();
    }
    if (threadIdx.x < POTRF_NB) vec[threadIdx.x] = E[(size_t)i * POTRF_NB + threadIdx.x];
    __syncthreads();
    const double* Li = Linv + (size_t)i * POTRF_NB * POTRF_NB + (size_t)r * POTRF_NB + 64 * h;
    double s = 0.0;
#pragma unroll 8
    for (int c = 0; c < 64; ++c) s += Li[c] * vec[64 * h + c];
    red[h][r] = s;
    __syncthreads();
    if (threadIdx.x < POTRF_NB) y[(size_t)i * POTRF_NB + r] = red[0][r] + red[1][r];
}

// backward substitution step: tiles kk < i do y_kk -= L_{i,kk}^T x_i ; tile i-1 then computes
// x_{i-1} = Linv_{i-1}^T y_{i-1}.  i == nblk bootstraps x_{nblk-1}.
__global__ __launch_bounds__(256) void k_bwd_step(const double* __restrict__ S, int ld, int i, int nblk,
        const double* __restrict__ Linv, double* __restrict__ y, double* __restrict__ x)
{
    __shared__ double vec[POTRF_NB];
    __shared__ double red[2][POTRF_NB];
    const int kk = i - 1 - blockIdx.x;
    const int c = threadIdx.x & 127, h = threadIdx.x >> 7;
    if (i < nblk) {
        if (threadIdx.x < POTRF_NB) vec[threadIdx.x] = x[(size_t)i * POTRF_NB + threadIdx.x];
        __syncthreads();
        const double* Lc = S + ((size_t)i * POTRF_NB + 64 * h) * ld + (size_t)kk * POTRF_NB + c;
        double s = 0.0;
#pragma unroll 8
        for (int r = 0; r < 64; ++r) s += Lc[(size_t)r * ld] * vec[64 * h + r];
        red[h][c] = s;
        __syncthreads();
        if (threadIdx.x < POTRF_NB) y[(size_t)kk * POTRF_NB + c] -= red[0][c] + red[1][c];
        if (blockIdx.x != 0) return;
        __syncthreads();
    }
    if (threadIdx.x < POTRF_NB) vec[threadIdx.x] = y[(size_t)kk * POTRF_NB + threadIdx.x];
    __syncthreads();
    const double* Li = Linv + (size_t)kk * POTRF_NB * POTRF_NB + (size_t)(64 * h) * POTRF_NB + c;
    double s = 0.0;
#pragma unroll 8
    for (int r = 0; r < 64; ++r) s += Li[(size_t)r * POTRF_NB] * vec[64 * h + r];
    red[h][c] = s;
    __syncthreads();
    if (threadIdx.x < POTRF_NB) x[(size_t)kk * POTRF_NB + c] = red[0][c] + red[1][c];
}

// ------------------------------------------------------------------------------------------------
inline void potrf_free(PotrfWorkspace& w)
{
    if (w.panel) (void)hipFree(w.panel);
    if (w.linv) (void)hipFree(w.linv);
    if (w.y) (void)hipFree(w.y);
    if (w.xs) (void)hipFree(w.xs);
    if (w.etmp) (void)hipFree(w.etmp);
    if (w.rb_handle && w.rb_destroy) w.rb_destroy(w.rb_handle);
    if (w.ev0) (void)hipEventDestroy(w.ev0);
    if (w.ev1) (void)hipEventDestroy(w.ev1);
    w = PotrfWorkspace();
}

inline int potrf_init(PotrfWorkspace& w, int ld, int backend)
{
    w.ld = ld; w.nblk = ld / POTRF_NB; w.backend = backend;
    const size_t tile = (size_t)POTRF_NB * POTRF_NB;
    if (hipMalloc((void**)&w.panel, std::max<size_t>(1, (size_t)(w.nblk - 1)) * tile * sizeof(double)) != hipSuccess) return -1;
    if (hipMalloc((void**)&w.linv, (size_t)w.nblk * tile * sizeof(double)) != hipSuccess) return -1;
    if (hipMalloc((void**)&w.y, (size_t)ld * sizeof(double)) != hipSuccess) return -1;
    if (hipMalloc((void**)&w.xs, (size_t)ld * sizeof(double)) != hipSuccess) return -1;
    if (hipMalloc((void**)&w.etmp, (size_t)ld * sizeof(double)) != hipSuccess) return -1;
    (void)hipEventCreate(&w.ev0); (void)hipEventCreate(&w.ev1);
    if (backend == 1) {
        // cross-check backend only: rocSOLVER through dlopen, never linked
        w.rb_lib = dlopen("librocblas.so", RTLD_NOW | RTLD_GLOBAL);
        w.rs_lib = dlopen("librocsolver.so", RTLD_NOW | RTLD_GLOBAL);
        if (!w.rb_lib || !w.rs_lib) { fprintf(stderr, "[bsfm] rocSOLVER cross-check backend unavailable: %s\n", dlerror()); return -1; }
        auto create = (int (*)(void**))dlsym(w.rb_lib, "rocblas_create_handle");
        w.rb_destroy = (int (*)(void*))dlsym(w.rb_lib, "rocblas_destroy_handle");
        w.rb_set_stream = (int (*)(void*, hipStream_t))dlsym(w.rb_lib, "rocblas_set_stream");
        w.rs_potrf = (int (*)(void*, int, int, double*, int, int*))dlsym(w.rs_lib, "rocsolver_dpotrf");
        w.rs_potrs = (int (*)(void*, int, int, int, double*, int, double*, int))dlsym(w.rs_lib, "rocsolver_dpotrs");
        if (!create || !w.rs_potrf || !w.rs_potrs || !w.rb_set_stream || create(&w.rb_handle) != 0) {
            fprintf(stderr, "[bsfm] rocSOLVER symbols missing\n"); return -1;
        }
    }
    return 0;
}

// Solves S x = E (n valid rows, S padded to ld); S is destroyed, E is preserved. info: 0 or dpotrf's k.
inline int potrf_solve(PotrfWorkspace& w, double* S, int ld, int n, const double* E, double* x_out, int* d_info, hipStream_t st)
{
    if (n <= 0) return 0;
    if (w.ev0) (void)hipEventRecord(w.ev0, st);
    if (w.backend == 1) {
        w.rb_set_stream(w.rb_handle, st);
        (void)hipMemcpyAsync(x_out, E, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st);
        // column-major "upper" of a symmetric matrix == our row-major lower (rocblas_fill_upper = 121)
        if (w.rs_potrf(w.rb_handle, 121, n, S, ld, d_info) != 0) return -1;
        if (w.rs_potrs(w.rb_handle, 121, n, 1, S, ld, x_out, n) != 0) return -1;
        if (w.ev1) { (void)hipEventRecord(w.ev1, st); }
        return 0;
    }
    const int nblk = (n + POTRF_NB - 1) / POTRF_NB;
    const size_t lds_bytes = 2 * 128 * GEMM_LDS_STRIDE * sizeof(double);
    (void)hipMemsetAsync(w.etmp, 0, (size_t)ld * sizeof(double), st);
    (void)hipMemcpyAsync(w.etmp, E, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st);
    for (int k = 0; k < nblk; ++k) {
        hipLaunchKernelGGL(k_potrf_diag, dim3(1), dim3(1024), 0, st, S, ld, k, n, w.linv, d_info);
        const int T = nblk - k - 1;
        if (T > 0) {
            hipLaunchKernelGGL(k_trsm_panel, dim3(T), dim3(256), lds_bytes, st, S, ld, k,
                               w.linv + (size_t)k * POTRF_NB * POTRF_NB, w.panel);
            hipLaunchKernelGGL(k_syrk_update, dim3(T * (T + 1) / 2), dim3(256), lds_bytes, st, S, ld, k, w.panel);
        }
    }
    for (int k = -1; k < nblk - 1; ++k)
        hipLaunchKernelGGL(k_fwd_step, dim3(k < 0 ? 1 : nblk - k - 1), dim3(256), 0, st, S, ld, k, w.linv, w.etmp, w.y);
    for (int i = nblk; i >= 1; --i)
        hipLaunchKernelGGL(k_bwd_step, dim3(i == nblk ? 1 : i), dim3(256), 0, st, S, ld, i, nblk, w.linv, w.y, w.xs);
    (void)hipMemcpyAsync(x_out, w.xs, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st);
    if (w.ev1) (void)hipEventRecord(w.ev1, st);
    return 0;
}

inline void potrf_collect_time(PotrfWorkspace& w)
{
    float ms = 0.f;
    if (w.ev0 && w.ev1 && hipEventElapsedTime(&ms, w.ev0, w.ev1) == hipSuccess && ms >= 0.f) { w.ms += ms; w.cnt++; }
}

}  // namespace bsfm
