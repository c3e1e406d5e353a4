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
//     8 waves x (32 x 64) = 32 FP64 accumulators per lane on v_mfma_f64_4x4x4_4b (the full-rate FP64
//     matrix instruction of gfx950: 72.7 TFLOP/s measured vs 36 for v_mfma_f64_16x16x4), K staged through LDS in
//     16-wide chunks (row stride padded to 18 doubles => conflict-free ds_read_b64 of the fragments),
//     next chunk prefetched into registers while the MFMAs of the current one issue;
//   * forward / backward substitution use the stored inverse diagonal tiles: one launch per tile step.
// v_mfma_f64_4x4x4_4b lane layout (probed on MI355X, scripts/probe_mfma4.hip), lane l = 16k + 4g + r:
// A[g][i=r][k], B[g][k][j=r], D[g][i][j] at lane 16i + 4g + j, for the 4 independent blocks g.
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
    // per-launch HIP-event timing of the trailing-update kernel (the roofline kernel bench.py reports)
    hipEvent_t* sy0 = nullptr; hipEvent_t* sy1 = nullptr; int sy_used = 0;
    double syrk_ms = 0.0; long long syrk_cnt = 0;
};

// ------------------------------------------------------------------------------------------------
// C(128x128) = A(128xK, row-major lda) * B(128xK, row-major ldb)^T, per-wave 64x64 quadrant in acc[4][4].
__device__ __forceinline__ void gemm_nt_128(const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb,
                                            int K, double* __restrict__ lds, double (&acc)[8][4])
{
    double* As = lds;
    double* Bs = lds + 128 * GEMM_LDS_STRIDE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 64;   // 8 waves: 4 (rows) x 2 (cols), 32 x 64 each
    // staging: 128 rows x 16 doubles = 1024 double2 per operand -> 2 per thread (512 threads)
    double2 pa0, pa1, pb0, pb1;
    const int srow = tid >> 3, sc2 = (tid & 7) * 2;     // rows srow + 64 q
    const double* Ag = A + (size_t)srow * lda + sc2;
    const double* Bg = B + (size_t)srow * ldb + sc2;
    double* Asw = As + srow * GEMM_LDS_STRIDE + sc2;
    double* Bsw = Bs + srow * GEMM_LDS_STRIDE + sc2;
#define BSFM_GLOAD(kc)                                                                      \
    pa0 = *reinterpret_cast<const double2*>(Ag + (kc));                                     \
    pa1 = *reinterpret_cast<const double2*>(Ag + (size_t)64 * lda + (kc));                  \
    pb0 = *reinterpret_cast<const double2*>(Bg + (kc));                                     \
    pb1 = *reinterpret_cast<const double2*>(Bg + (size_t)64 * ldb + (kc));
    BSFM_GLOAD(0)
    for (int kc = 0; kc < K; kc += GEMM_KC) {
        __syncthreads();          // previous chunk fully consumed
        *reinterpret_cast<double2*>(Asw) = pa0;
        *reinterpret_cast<double2*>(Asw + 64 * GEMM_LDS_STRIDE) = pa1;
        *reinterpret_cast<double2*>(Bsw) = pb0;
        *reinterpret_cast<double2*>(Bsw + 64 * GEMM_LDS_STRIDE) = pb1;
        __syncthreads();
        if (kc + GEMM_KC < K) { BSFM_GLOAD(kc + GEMM_KC) }
#pragma unroll
        for (int kk = 0; kk < GEMM_KC; kk += 4) {
            // v_mfma_f64_4x4x4_4b: 4 independent 4x4x4 blocks g = (lane>>2)&3 per instruction, the full-rate FP64
            // matrix op on gfx950 (measured 72.7 TFLOP/s vs 36 for v_mfma_f64_16x16x4).  Lane l = 16k + 4g + r:
            //   A[g][i=r][k], B[g][k][j=r]  ->  D[g][i][j] at lane 16i + 4g + j   (probed: scripts/probe_mfma4.hip).
            // B fragment: 16 output columns (4 per block); A fragment: 4 output rows replicated over the blocks.
            double b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                b[u] = Bs[(wc + 16 * u + (lane & 15)) * GEMM_LDS_STRIDE + kk + (lane >> 4)];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const double a = As[(wr + 4 * t + (lane & 3)) * GEMM_LDS_STRIDE + kk + (lane >> 4)];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    acc[t][u] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b[u], acc[t][u], 0, 0, 0);
            }
        }
    }
#undef BSFM_GLOAD
}

// Panel: X_i = S_ik * Linv_k^T for i = k+1 .. nblk-1; writes X back into S (it is L) and into the compact panel.
__global__ __launch_bounds__(512, 4) void k_trsm_panel(double* __restrict__ S, int ld, int k,
        const double* __restrict__ Linv, double* __restrict__ panel)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int i = k + 1 + blockIdx.x;
    double* Sik = S + ((size_t)i * POTRF_NB) * ld + (size_t)k * POTRF_NB;
    double acc[8][4];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[t][u] = 0.0;
    gemm_nt_128(Sik, ld, Linv, POTRF_NB, POTRF_NB, lds, acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 64;
    double* Pt = panel + (size_t)blockIdx.x * POTRF_NB * POTRF_NB;
    __syncthreads();   // every wave has finished reading S_ik through LDS before it is overwritten
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = wr + 4 * t + (lane >> 4), col = wc + 16 * u + (lane & 15);
            const double v = acc[t][u];
            Sik[(size_t)row * ld + col] = v;
            Pt[row * POTRF_NB + col] = v;
        }
}

// Trailing update: S_ij -= P_a P_b^T for k < j <= i (a = i-k-1, b = j-k-1), one tile per workgroup.
__global__ __launch_bounds__(512, 4) void k_syrk_update(double* __restrict__ S, int ld, int k, const double* __restrict__ panel)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int t = blockIdx.x;
    int a = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((a + 1) * (a + 2) / 2 <= t) ++a;
    while (a * (a + 1) / 2 > t) --a;
    const int b = t - a * (a + 1) / 2;
    const int i = k + 1 + a, j = k + 1 + b;
    double acc[8][4];
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[q][u] = 0.0;
    gemm_nt_128(panel + (size_t)a * POTRF_NB * POTRF_NB, POTRF_NB, panel + (size_t)b * POTRF_NB * POTRF_NB, POTRF_NB,
                POTRF_NB, lds, acc);
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

// Diagonal tile: fused Cholesky + inverse of the factor in ONE 512-thread workgroup (the serial critical path of
// the factorisation).  Threads form a 16 x 32 grid; thread (ti,tj) owns rows ti+16p (p<8), cols tj+32q (q<4) of the
// tile A and of X (X starts as I and ends as inv(L)): 32+32 FP64 accumulators in registers, block-cyclic so that
// the shrinking trailing matrix stays balanced.  Per column j:
//   s = 1/sqrt(pivot) (computed ONCE by the owner of (j,j) when it publishes the column), l = raw column * s,
//   xr = raw X row * s;  A[p>=jb][q>=jq] -= l l^T ; X[p>=jb][q<=jq] -= l xr   (finished row/column blocks are
//   skipped with wave-uniform predicates, so the issued FMA count follows the ~1/3 triangular work).
// Raw column j+1 / X row j+1 / next pivot scale are published right after the update => ONE barrier per column.
__device__ __forceinline__ double rsqrt_f64(double v)
{
    double s = __builtin_amdgcn_rsq(v);          // v_rsq_f64 seed, two Newton steps to full precision
    s = s * (1.5 - 0.5 * v * s * s);
    s = s * (1.5 - 0.5 * v * s * s);
    return s;
}

__global__ __launch_bounds__(512) void k_potrf_diag(double* __restrict__ S, int ld, int k, int n_total,
        double* __restrict__ Linv, int* __restrict__ info)
{
    __shared__ double colbuf[2][POTRF_NB];
    __shared__ double rowbuf[2][POTRF_NB];
    __shared__ double svec[POTRF_NB];      // 1/sqrt(pivot_j): finished columns of A / rows of X stay RAW in
                                           // registers and are scaled once, at the store
    const int ti = threadIdx.x >> 5, tj = threadIdx.x & 31;
    const int base = k * POTRF_NB;
    double* T = S + (size_t)base * ld + base;
    double a[8][4], x[8][4];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = ti + 16 * p, c = tj + 32 * q;
            double v = 0.0;
            if (base + r < n_total && base + c < n_total) { if (c <= r) v = T[(size_t)r * ld + c]; }
            else if (r == c) v = 1.0;
            a[p][q] = v;
            x[p][q] = (r == c) ? 1.0 : 0.0;
        }
    if (tj == 0) {
#pragma unroll
        for (int p = 0; p < 8; ++p) colbuf[0][ti + 16 * p] = a[p][0];
    }
    if (ti == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) rowbuf[0][tj + 32 * q] = x[0][q];
    }
    if (threadIdx.x == 0) {
        const double piv = a[0][0];
        if (!(piv > 0.0) && base < n_total) atomicCAS(info, 0, base + 1);
        svec[0] = rsqrt_f64(piv);
    }
    __syncthreads();

    // one column step for a fixed (compile-time) column-block index JQ = j >> 5
#define BSFM_DIAG_STEP(JQ)                                                                                   \
    {                                                                                                        \
        const double s = svec[j];                                                                            \
        const int jb = j >> 4;                                                                               \
        double lc[4], xr[4];                                                                                 \
        _Pragma("unroll") for (int q = JQ; q < 4; ++q) lc[q] = colbuf[cur][tj + 32 * q] * s;                 \
        if (tj <= (j & 31)) lc[JQ] = 0.0;                                                                    \
        _Pragma("unroll") for (int q = 0; q <= JQ; ++q) xr[q] = rowbuf[cur][tj + 32 * q] * s;                \
        _Pragma("unroll") for (int p = 0; p < 8; ++p) {                                                      \
            if (p >= jb) {                                                                                   \
                double l = colbuf[cur][ti + 16 * p] * s;                                                     \
                if (p == jb && ti <= (j & 15)) l = 0.0;                                                      \
                _Pragma("unroll") for (int q = JQ; q < 4; ++q) a[p][q] -= l * lc[q];                         \
                _Pragma("unroll") for (int q = 0; q <= JQ; ++q) x[p][q] -= l * xr[q];                        \
            }                                                                                                \
        }                                                                                                    \
        if (j + 1 < POTRF_NB) {                                                                              \
            const int jn = j + 1, nb = jn >> 4;                                                              \
            constexpr int NQ0 = JQ, NQ1 = (JQ < 3) ? JQ + 1 : 3;                                             \
            const bool nextq = (jn >> 5) != JQ;                                                              \
            if (tj == (jn & 31)) {                                                                           \
                _Pragma("unroll") for (int p = 0; p < 8; ++p) {                                              \
                    const double v = nextq ? a[p][NQ1] : a[p][NQ0];                                          \
                    colbuf[nxt][ti + 16 * p] = v;                                                            \
                    if (p == nb && ti == (jn & 15)) {                                                        \
                        if (!(v > 0.0) && base + jn < n_total) atomicCAS(info, 0, base + jn + 1);            \
                        svec[jn] = rsqrt_f64(v);                                                             \
                    }                                                                                        \
                }                                                                                            \
            }                                                                                                \
            if (ti == (jn & 15)) {                                                                           \
                _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                              \
                    double v = x[0][q];                                                                      \
                    _Pragma("unroll") for (int p = 1; p < 8; ++p) if (p == nb) v = x[p][q];                  \
                    rowbuf[nxt][tj + 32 * q] = v;                                                            \
                }                                                                                            \
            }                                                                                                \
        }                                                                                                    \
        __syncthreads();                                                                                     \
    }

#pragma unroll 1
    for (int j = 0; j < 32; ++j) { const int cur = j & 1, nxt = cur ^ 1; BSFM_DIAG_STEP(0) }
#pragma unroll 1
    for (int j = 32; j < 64; ++j) { const int cur = j & 1, nxt = cur ^ 1; BSFM_DIAG_STEP(1) }
#pragma unroll 1
    for (int j = 64; j < 96; ++j) { const int cur = j & 1, nxt = cur ^ 1; BSFM_DIAG_STEP(2) }
#pragma unroll 1
    for (int j = 96; j < 128; ++j) { const int cur = j & 1, nxt = cur ^ 1; BSFM_DIAG_STEP(3) }
#undef BSFM_DIAG_STEP

    double* Li = Linv + (size_t)k * POTRF_NB * POTRF_NB;
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = ti + 16 * p, c = tj + 32 * q;
            if (c <= r) T[(size_t)r * ld + c] = a[p][q] * svec[c];       // L = raw column * 1/sqrt(pivot)
            Li[r * POTRF_NB + c] = (c <= r) ? x[p][q] * svec[r] : 0.0;   // inv(L) row r = raw row * 1/sqrt(pivot_r)
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
    for (int i = 0; w.sy0 && i < w.nblk; ++i) { (void)hipEventDestroy(w.sy0[i]); (void)hipEventDestroy(w.sy1[i]); }
    delete[] w.sy0; delete[] w.sy1;
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
    w.sy0 = new hipEvent_t[w.nblk]; w.sy1 = new hipEvent_t[w.nblk];
    for (int i = 0; i < w.nblk; ++i) { (void)hipEventCreate(&w.sy0[i]); (void)hipEventCreate(&w.sy1[i]); }
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
    w.sy_used = 0;
    const size_t lds_bytes = 2 * 128 * GEMM_LDS_STRIDE * sizeof(double);
    (void)hipMemsetAsync(w.etmp, 0, (size_t)ld * sizeof(double), st);
    (void)hipMemcpyAsync(w.etmp, E, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st);
    for (int k = 0; k < nblk; ++k) {
        hipLaunchKernelGGL(k_potrf_diag, dim3(1), dim3(512), 0, st, S, ld, k, n, w.linv, d_info);
        const int T = nblk - k - 1;
        if (T > 0) {
            hipLaunchKernelGGL(k_trsm_panel, dim3(T), dim3(512), lds_bytes, st, S, ld, k,
                               w.linv + (size_t)k * POTRF_NB * POTRF_NB, w.panel);
            (void)hipEventRecord(w.sy0[k], st);
            hipLaunchKernelGGL(k_syrk_update, dim3(T * (T + 1) / 2), dim3(512), lds_bytes, st, S, ld, k, w.panel);
            (void)hipEventRecord(w.sy1[k], st);
            w.sy_used = k + 1;
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
    for (int i = 0; i < w.sy_used; ++i)
        if (hipEventElapsedTime(&ms, w.sy0[i], w.sy1[i]) == hipSuccess && ms >= 0.f) { w.syrk_ms += ms; w.syrk_cnt++; }
    w.sy_used = 0;
}

}  // namespace bsfm
