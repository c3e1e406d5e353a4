// potrf.hip.h -- dense FP64 Cholesky solve of the reduced camera system S x = E on gfx950.
//
// Replaces sba_Axb_Chol = LAPACK dpotrf("U") + dpotrs (lib/sba-1.5/sba_lapack.c:374-485, called from
// lib/sba-1.5/sba_levmar.c:1368).  S is symmetric, so the reference's column-major "U" factorisation
// is this file's row-major LOWER factorisation S = L L^T; only the lower triangle of S is read.
//
// MI355X design (the one MFMA-bound contraction of the LM iteration, ~N^3/3 FP64 flop):
//   * right-looking tiled factorisation, tile NB = 128; S is padded to a multiple of NB (identity in
//     the padding) so every kernel works on full tiles;
//   * what bounds it at N = 9000 is the serial chain diag(k) -> panel(k) -> update of tile (k+1,k+1) -> diag(k+1), so the
//     schedule (potrf_solve) runs on three streams and keeps everything else off that chain:
//       chain  : k_potrf_diag (ONE 512-thread workgroup, tile in LDS as 8x8 blocks of 16x16: 16x16 diagonal blocks
//                factored by one wave with v_readlane broadcasts, rows below solved by in-register substitution, trailing
//                blocks and the block-wise inverse of the factor as 16x16x16 products on the 4x4x4 MFMA), then
//                k_chain_tile32<0> (ONLY the first panel tile) and k_chain_tile32<1> (next diagonal tile -= two panels);
//       side   : the rest of the panel (L_ik = S_ik * inv(L_kk)^T is a GEMM, no triangular solve; 64-row halves with the
//                whole K range prefetched into registers; also writes the compact panel copy, ring of 4 buffers) and the
//                rest of the first trailing column -- while the next diagonal tile is being factored;
//       bulk   : the trailing update of all other tiles (k_syrk_update), the MFMA bulk;
//   * trailing update S_ij -= L_ik L_jk^T on the lower triangle: 128x128 tile per 512-thread workgroup,
//     8 waves x (32 x 64) = 32 FP64 accumulators per lane on v_mfma_f64_4x4x4_4b (the full-rate FP64
//     matrix instruction of gfx950: 72.7 TFLOP/s measured vs 36 for v_mfma_f64_16x16x4), K staged through LDS in
//     16-wide chunks (row stride padded to 18 doubles => conflict-free ds_read_b64 of the fragments),
//     next chunk prefetched into registers while the MFMAs of the current one issue; the accumulators start as the C
//     tile (store-only epilogue);
//   * forward substitution rides on the side stream (y_k in the panel launch, E updates in the
//     first-trailing-column launch); backward substitution is ONE persistent launch with flag hand-offs.
// v_mfma_f64_4x4x4_4b lane layout (probed on MI355X, scripts/probe_mfma4.hip), lane l = 16k + 4g + r:
// A[g][i=r][k], B[g][k][j=r], D[g][i][j] at lane 16i + 4g + j, for the 4 independent blocks g.
#pragma once
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <vector>
#include <cstring>
#include <algorithm>
#include <mutex>
#include "devcache.h"

struct bsfm_comm;      // include/bsfm.h: one rank's communicator (comm.hip)

namespace bsfm {

// Process-wide pool of plain non-blocking streams.  An incremental reconstruction calls run_sfm hundreds of times on small
// problems; creating and destroying the three streams of a problem cost ~7 ms + ~7 ms per call (hipStreamCreateWithFlags /
// hipStreamDestroy, measured with rocprofv3 --hip-trace) -- more than the LM iterations themselves at 10 cameras.
// Streams go back to the pool drained (the owner synchronises before releasing); per device.
struct StreamPool {
    std::mutex mu;
    std::vector<std::pair<int, hipStream_t>> idle;
    hipStream_t acquire()
    {
        int dev = 0; (void)hipGetDevice(&dev);
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t i = 0; i < idle.size(); ++i)
                if (idle[i].first == dev) { hipStream_t s = idle[i].second; idle.erase(idle.begin() + (long)i); return s; }
        }
        hipStream_t s = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
        return s;
    }
    void release(hipStream_t s)
    {
        if (!s) return;
        int dev = 0; (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lk(mu);
        if (idle.size() < 16) idle.emplace_back(dev, s); else (void)hipStreamDestroy(s);
    }
};
inline StreamPool& stream_pool() { static StreamPool* p = new StreamPool(); return *p; }   // leaked on purpose: no teardown order issues at exit

constexpr int POTRF_NB = 128;
constexpr int POTRF_NARROW = 8;      // upper bound of BSFM_NARROW (steps with at most that many tile rows run single-stream, k_narrow_tiles)
constexpr int POTRF_MAX_TILES = 240;   // k_bwd_persistent needs one resident workgroup per tile column (256 CUs)
#ifndef BSFM_SYRK_WPS
#define BSFM_SYRK_WPS 4      // waves per SIMD the bulk tile kernel is compiled for (4 = two workgroups per CU)
#endif
#ifndef BSFM_GEMM_MFMA16
#define BSFM_GEMM_MFMA16 1
#endif
#ifndef BSFM_GEMM_KC
#define BSFM_GEMM_KC 16
#endif
constexpr int GEMM_KC = BSFM_GEMM_KC;             // K chunk of the bulk 128x128 tile kernel
#ifndef BSFM_GEMM_STRIDE
#define BSFM_GEMM_STRIDE (GEMM_KC + (GEMM_KC == 16 ? 2 : 4))
#endif
constexpr int GEMM_LDS_STRIDE = BSFM_GEMM_STRIDE;   // 18: best for both fragment forms (16x16x4, kernel alone at k = 1 024: 18 -> 62.2, 20 -> 59.4 TFLOP/s); 36: 16-byte aligned rows
constexpr int G64_KC = 16;                        // the latency-critical 64-row chain kernels keep 16-wide chunks
constexpr int G64_STRIDE = G64_KC + 2;

typedef double v4d __attribute__((ext_vector_type(4)));

struct FlowWorkspace;          // chol_flow.hip.h: the tile-dataflow factorisation (round 4, the default for systems of two or more tiles)
struct PotrfWorkspace;
inline int flow_solve_dispatch(PotrfWorkspace& w, double* S, int ld, int n, const double* E, double* x_out, int* d_info, hipStream_t st);
inline void flow_release(PotrfWorkspace& w);
inline int flow_solve_one(PotrfWorkspace& w, double* S, int ld, int n, const double* E, double* x_out, int* d_info, hipStream_t st);

struct PotrfWorkspace {
    FlowWorkspace* flow = nullptr;
    ::bsfm_comm* dist_comm = nullptr;      // != nullptr: systems of two or more tiles are factored by this communicator's ranks together (chol_flow.hip.h: FlowDist)
    bool solve_one_attr = false;   // k_flow_solve_one's dynamic-LDS attribute has been set on this workspace's device
    int use_flow = 1;           // BSFM_CHOL=streams selects the three-stream schedule of rounds 1-3 below (kept as the A/B reference)
    int ld = 0, nblk = 0, backend = 0;
    double* panel = nullptr;   // 4 x (nblk-1) tiles of NB x NB: compact copies of the last panels (ring, k & 3)
    hipStream_t s2 = nullptr;  // bulk-update stream of the lookahead schedule
    bool s2_masked = false;    // created with a CU mask (BSFM_PANEL_CUS): not a pool stream
    hipStream_t sd = nullptr;  // side stream: the part of panel k / first trailing column the NEXT diagonal tile does not need
    hipEvent_t* evP = nullptr; hipEvent_t* evU = nullptr;   // panel k complete (side stream) / trailing update k done
    hipEvent_t* evT = nullptr; hipEvent_t* evC = nullptr;   // first panel tile of step k ready (chain) / first trailing column done (side)
    // Tile ENVELOPE of the matrix (opt-in, solver.hip: BSFM_SOLVER_ENVELOPE): env_rows[k] = number of tile rows below diagonal tile k
    // that can hold a non-zero of the factor (rows k+1 .. k+env_rows[k]; Cholesky without pivoting creates no fill outside the envelope
    // of the matrix).  Empty = dense (every step works on all rows below k).  d_last[k] = k + env_rows[k] for the backward substitution.
    std::vector<int> env_rows;
    int* d_last = nullptr;
    double* linv = nullptr;    // nblk tiles: inverse of each diagonal factor tile
    double* y = nullptr;       // ld
    double* xs = nullptr;      // ld
    double* xu = nullptr;      // ld: the solution vector k_bwd_scalar polls through the scalar path (a buffer of its own: nothing else ever reads it cached)
    double* etmp = nullptr;    // ld (rhs working copy)
    int* bflags = nullptr;     // nblk + 1: hand-off flags of the persistent backward substitution (+ timeout word)
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
    double* sy_flops = nullptr; double syrk_flops = 0.0;     // flops of each timed launch / running sum
    int syrk_events = 1;        // HIP-event timing of every n-th bulk launch (BSFM_SYRK_EVENTS=n; 0 = none)
    int narrow = 3;             // steps with at most this many tile rows run single-stream (k_narrow_tiles); BSFM_NARROW = 0 .. 8.  Measured
                                // (profiles/r03_narrow_steps.txt): 8 vs 0: 50 / 100 / 200 / 400 cameras 0.603 / 0.891 / 1.641 / 3.139 vs 0.620 / 0.939 / 1.694 /
                                // 3.205 ms per iteration, but the envelope solve of the banded scene (4-5 rows per step) 5.36 vs 5.22 ms: the three-stream
                                // step hides its panel / column work behind the next diagonal tile, the single-stream step does not -- so only the very
                                // narrow steps (where there is next to nothing to hide) take this path by default
    int timing = 1;             // 0: no timing events at all (small problems: solver.hip turns it off below 100 000 observations)
    int syrk_nt = 0;            // non-temporal C traffic in the bulk kernel (BSFM_SYRK_NT=1; measured neutral: 8.44 vs 8.52 ms per solve)
    long long* dbg = nullptr;   // optional device buffer: cycle stamps of k_potrf_diag phases (BSFM_DEBUG_DIAG=1)
};

// ------------------------------------------------------------------------------------------------
// C(128x128) = A(128xK, row-major lda) * B(128xK, row-major ldb)^T, per-wave 64x64 quadrant in acc[4][4].
template <bool NEG_A = false>
__device__ __forceinline__ void gemm_nt_128(const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb,
                                            int K, double* __restrict__ lds, double (&acc)[8][4], long seg2 = 0)
{
    // K > 128: columns 128.. of BOTH operands continue at element offset seg2 from their first column (second panel
    // buffer of a two-panel update; the offset is wave-uniform and identical for A and B, so it costs no VGPRs)
    double* As = lds;
    double* Bs = lds + 128 * GEMM_LDS_STRIDE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 64;   // 8 waves: 4 (rows) x 2 (cols), 32 x 64 each
    // staging: 128 rows x GEMM_KC doubles per operand and chunk, 512 threads
    constexpr int DPR = GEMM_KC / 2;                    // double2 per row
    constexpr int RPP = 512 / DPR;                      // rows per pass
    constexpr int NPASS = 128 / RPP;
    double pa[NPASS][2], pb[NPASS][2];                  // plain doubles (double2 arrays would land in scratch)
    const int srow = tid / DPR, sc2 = (tid % DPR) * 2;
    const double* Ag = A + (size_t)srow * lda + sc2;
    const double* Bg = B + (size_t)srow * ldb + sc2;
    double* Asw = As + srow * GEMM_LDS_STRIDE + sc2;
    double* Bsw = Bs + srow * GEMM_LDS_STRIDE + sc2;
#define BSFM_GLOAD(kc)                                                                                    \
    _Pragma("unroll") for (int q = 0; q < NPASS; ++q) {                                                   \
        const long ko_ = (kc) >= POTRF_NB ? (long)(kc) - POTRF_NB + seg2 : (long)(kc);                    \
        const double2 ta = *reinterpret_cast<const double2*>(Ag + (size_t)(RPP * q) * lda + ko_);          \
        const double2 tb = *reinterpret_cast<const double2*>(Bg + (size_t)(RPP * q) * ldb + ko_);          \
        pa[q][0] = ta.x; pa[q][1] = ta.y; pb[q][0] = tb.x; pb[q][1] = tb.y;                               \
    }
    BSFM_GLOAD(0)
    for (int kc = 0; kc < K; kc += GEMM_KC) {
        __syncthreads();          // previous chunk fully consumed
#pragma unroll
        for (int q = 0; q < NPASS; ++q) {
            *reinterpret_cast<double2*>(Asw + RPP * q * GEMM_LDS_STRIDE) = NEG_A ? make_double2(-pa[q][0], -pa[q][1]) : make_double2(pa[q][0], pa[q][1]);
            *reinterpret_cast<double2*>(Bsw + RPP * q * GEMM_LDS_STRIDE) = make_double2(pb[q][0], pb[q][1]);
        }
        __syncthreads();
        if (kc + GEMM_KC < K) { BSFM_GLOAD(kc + GEMM_KC) }
#if BSFM_GEMM_MFMA16
        // v_mfma_f64_16x16x4_f64 with the accumulators in VGPRs: 77.9 TFLOP/s on gfx950 (scripts/ubench_mfma16_agpr.hip; round 1 had
        // measured 36 and dismissed the instruction -- that loop's accumulators had been allocated to AGPRs, which halves its rate).
        // A[i][k] at lane i + 16 k, B[k][j] at lane j + 16 k, D[i][j]: register r of lane l holds row 4 r + (l >> 4), column l & 15
        // (scripts/probe_mfma16.hip) -- for a 16-row block that is exactly the acc[t = 4 bi + r][u] layout of the 4x4x4 form below, so the
        // C loads / stores of every caller stay as they are.  Per k-step of 4: 2 + 4 fragment reads for 8 matrix instructions (the 4x4x4 form:
        // 8 + 4 reads for 32).
#pragma unroll
        for (int kk = 0; kk < GEMM_KC; kk += 4) {
            typedef double v4d_ __attribute__((ext_vector_type(4)));
            double b[4], a2[2];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                b[u] = Bs[(wc + 16 * u + (lane & 15)) * GEMM_LDS_STRIDE + kk + (lane >> 4)];
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
                a2[bi] = As[(wr + 16 * bi + (lane & 15)) * GEMM_LDS_STRIDE + kk + (lane >> 4)];
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    v4d_ c = { acc[4 * bi][u], acc[4 * bi + 1][u], acc[4 * bi + 2][u], acc[4 * bi + 3][u] };
                    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[bi], b[u], c, 0, 0, 0);
                    acc[4 * bi][u] = c[0]; acc[4 * bi + 1][u] = c[1]; acc[4 * bi + 2][u] = c[2]; acc[4 * bi + 3][u] = c[3];
                }
        }
#else
#pragma unroll
        for (int kk = 0; kk < GEMM_KC; kk += 4) {
            // v_mfma_f64_4x4x4_4b: 4 independent 4x4x4 blocks g = (lane>>2)&3 per instruction.  Lane l = 16k + 4g + r:
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
#endif
    }
#undef BSFM_GLOAD
}

// Latency-critical variant for the panel chain: C(64 x 128) = A(64 x K) * B(128 x K)^T by a 256-thread workgroup
// (4 waves, 32 x 64 each).  The tiles of the serial chain (panel solve, first trailing column) are split into two
// such halves so that twice as many CUs share the chain's MFMA work.
__device__ __forceinline__ void gemm_nt_64(const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb,
                                           double* __restrict__ lds, double (&acc)[8][4])
{
    // K = 128 fixed.  The whole K range of both operands is fetched into registers up front (48 double2 per lane:
    // these workgroups run one wave per SIMD, so the 512-VGPR budget is available) -- the chain kernels are latency-,
    // not throughput-bound, and this removes the per-chunk global-load wait.
    double* As = lds;
    double* Bs = lds + 64 * G64_STRIDE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 64;
    const int srow = tid >> 3, sc2 = (tid & 7) * 2;     // 32 rows per pass
    const double* Ag = A + (size_t)srow * lda + sc2;
    const double* Bg = B + (size_t)srow * ldb + sc2;
    double* Asw = As + srow * G64_STRIDE + sc2;
    double* Bsw = Bs + srow * G64_STRIDE + sc2;
    double pa[8][2][2], pb[8][4][2];      // plain doubles: arrays of the double2 vector struct are not promoted to registers
#pragma clang loop unroll(full)
    for (int ch = 0; ch < 8; ++ch) {
#pragma unroll
        for (int q = 0; q < 2; ++q) { const double2 t2 = *reinterpret_cast<const double2*>(Ag + (size_t)(32 * q) * lda + G64_KC * ch); pa[ch][q][0] = t2.x; pa[ch][q][1] = t2.y; }
#pragma unroll
        for (int q = 0; q < 4; ++q) { const double2 t2 = *reinterpret_cast<const double2*>(Bg + (size_t)(32 * q) * ldb + G64_KC * ch); pb[ch][q][0] = t2.x; pb[ch][q][1] = t2.y; }
    }
#pragma clang loop unroll(full)
    for (int ch = 0; ch < 8; ++ch) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) *reinterpret_cast<double2*>(Asw + 32 * q * G64_STRIDE) = make_double2(pa[ch][q][0], pa[ch][q][1]);
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<double2*>(Bsw + 32 * q * G64_STRIDE) = make_double2(pb[ch][q][0], pb[ch][q][1]);
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < G64_KC; kk += 4) {
            double b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                b[u] = Bs[(wc + 16 * u + (lane & 15)) * G64_STRIDE + kk + (lane >> 4)];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const double a = As[(wr + 4 * t + (lane & 3)) * G64_STRIDE + kk + (lane >> 4)];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    acc[t][u] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b[u], acc[t][u], 0, 0, 0);
            }
        }
    }
}

// Panel, two halves per tile: workgroup (tile, half) computes rows 64*half .. +63 of X_i = S_ik * Linv_k^T.
// y_k = inv(L_kk) * E_k for one tile (forward substitution rides on the factorisation chain: E_k is final once the
// first-trailing-column launch of step k-1 has run).
__device__ __forceinline__ void fwd_tile_solve(const double* __restrict__ Linv, const double* __restrict__ Ek,
                                               double* __restrict__ yk, double* __restrict__ lds)
{
    double* vec = lds; double* red = lds + POTRF_NB;
    const int r = threadIdx.x & 127, h = threadIdx.x >> 7;
    if (threadIdx.x < POTRF_NB) vec[threadIdx.x] = Ek[threadIdx.x];
    __syncthreads();
    const double* Li = Linv + (size_t)r * POTRF_NB + 64 * h;
    double s = 0.0;
#pragma unroll 8
    for (int c = 0; c < 64; ++c) s += Li[c] * vec[64 * h + c];
    red[h * POTRF_NB + r] = s;
    __syncthreads();
    if (threadIdx.x < POTRF_NB) yk[r] = red[r] + red[POTRF_NB + r];
}

__global__ __launch_bounds__(256) void k_fwd_last(const double* __restrict__ Linv, const double* __restrict__ Ek, double* __restrict__ yk)
{
    __shared__ double sm[3 * POTRF_NB];
    fwd_tile_solve(Linv, Ek, yk, sm);
}

__global__ __launch_bounds__(256, 1) void k_trsm_panel64(double* __restrict__ S, int ld, int k,
        const double* __restrict__ Linv, double* __restrict__ panel, int tile0, int nwork, const double* __restrict__ E, double* __restrict__ y)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if ((int)blockIdx.x == nwork) {          // the extra workgroup: y_k = inv(L_kk) E_k
        fwd_tile_solve(Linv, E + (size_t)k * POTRF_NB, y + (size_t)k * POTRF_NB, lds);
        return;
    }
    if ((int)blockIdx.x == nwork + 1) {      // second extra workgroup: L_{k+1,k} = slot 0 of the compact panel (k_chain_tile32<0>) -> S
        double* dst = S + ((size_t)(k + 1) * POTRF_NB) * ld + (size_t)k * POTRF_NB;
        for (int idx = threadIdx.x; idx < POTRF_NB * POTRF_NB / 2; idx += 256) {
            const int r = idx >> 6, c2 = (idx & 63) * 2;
            *reinterpret_cast<double2*>(dst + (size_t)r * ld + c2) = *reinterpret_cast<const double2*>(panel + (size_t)r * POTRF_NB + c2);
        }
        return;
    }
    const int tile = tile0 + (blockIdx.x >> 1), half = blockIdx.x & 1;
    double* Sik = S + ((size_t)(k + 1 + tile) * POTRF_NB + 64 * half) * ld + (size_t)k * POTRF_NB;
    double acc[8][4];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[t][u] = 0.0;
    gemm_nt_64(Sik, ld, Linv, POTRF_NB, lds, acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 64;
    double* Pt = panel + (size_t)tile * POTRF_NB * POTRF_NB + (size_t)(64 * half) * POTRF_NB;
    __syncthreads();
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

// First trailing column, two halves per tile: S_{k+1+a, k+1} -= P_a P_0^T.
__global__ __launch_bounds__(256, 1) void k_syrk_col64(double* __restrict__ S, int ld, int k, const double* __restrict__ panel,
        int a0, int ngemm, double* __restrict__ E, const double* __restrict__ y, int cofs)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if ((int)blockIdx.x >= ngemm) {
        // extra workgroups (one per tile row, off the GEMM workgroups' critical path): forward substitution
        // E_{k+1+a} -= L_{k+1+a,k} * y_k, 2 lanes per row
        const int a = blockIdx.x - ngemm;
        const int row = threadIdx.x >> 1, part = threadIdx.x & 1;
        const double* Pr = panel + (size_t)a * POTRF_NB * POTRF_NB + (size_t)row * POTRF_NB + 64 * part;
        const double* yk = y + (size_t)k * POTRF_NB + 64 * part;
        double sacc = 0.0;
#pragma unroll 16
        for (int c = 0; c < 64; ++c) sacc += Pr[c] * yk[c];
        sacc += __shfl_xor(sacc, 1, 64);
        if (part == 0) E[(size_t)(k + 1 + a) * POTRF_NB + row] -= sacc;
        return;
    }
    const int a = a0 + (blockIdx.x >> 1), half = blockIdx.x & 1;
    double acc[8][4];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[t][u] = 0.0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 64;
    double* Sij = S + ((size_t)(k + 1 + a) * POTRF_NB + 64 * half) * ld + (size_t)(k + 1 + cofs) * POTRF_NB;
    double cin[8][4];                     // the C tile is fetched up front as well
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u)
            cin[t][u] = Sij[(size_t)(wr + 4 * t + (lane >> 4)) * ld + wc + 16 * u + (lane & 15)];
    gemm_nt_64(panel + (size_t)a * POTRF_NB * POTRF_NB + (size_t)(64 * half) * POTRF_NB, POTRF_NB,
               panel + (size_t)cofs * POTRF_NB * POTRF_NB, POTRF_NB, lds, acc);
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = wr + 4 * t + (lane >> 4), col = wc + 16 * u + (lane & 15);
            Sij[(size_t)row * ld + col] = cin[t][u] - acc[t][u];
        }
}

// (Tried and dropped: no-return global_atomic_add_f64 of -acc instead of a load/subtract/store epilogue -- bit-identical,
// but slower, 31.1 vs 34.8 TFLOP/s on config 3: L2 atomic throughput becomes the limit.)
// The two single-tile products ON the serial chain, cut into 32x32 blocks so that 16 (10) compute units share one
// 128x128x128 product and no wave issues more than 128 matrix instructions (a 64-row half per workgroup keeps one wave
// busy for ~7 us of pure MFMA issue; here it is < 1 us and the kernel is bounded by one global-load round trip):
//   MODE 0 (grid 16): P_0 = S_{k+1,k} inv(L_kk)^T, written to S and to slot 0 of the compact panel.  inv(L_kk) is lower
//                     triangular: output column block bc only needs k < 32 (bc + 1);
//   MODE 1 (grid 10): S_{k+1,k+1} -= P_0 P_0^T, lower-triangle blocks only (all the diagonal-tile kernel reads); with
//                     prev != nullptr ALSO -= Q_1 Q_1^T, Q = panel k-1 (and with prev2 the tile of panel k-2): the bulk
//                     launches leave this one tile to the chain, so the next diagonal tile never waits for a bulk launch
//                     that has just started.
// The whole K range of both operands goes to LDS in one step (row stride 132 doubles: conflict-free A fragments).
constexpr int T32_STRIDE = 132;
constexpr int T32_LDS_DOUBLES = 2 * 32 * T32_STRIDE;
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_chain_tile32(   // (2: a 256-register budget keeps the 16x16x4 accumulators in VGPRs; their AGPR form is half rate)
        double* __restrict__ S, int ld, int k, const double* __restrict__ Linv,
                                                         double* __restrict__ panel, const double* __restrict__ prev,
                                                         const double* __restrict__ prev2)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    int br, bc;
    if (MODE == 0) { br = blockIdx.x >> 2; bc = blockIdx.x & 3; }
    else { const int t = blockIdx.x; br = t < 1 ? 0 : t < 3 ? 1 : t < 6 ? 2 : 3; bc = t - br * (br + 1) / 2; }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = (wave >> 1) * 16, wc = (wave & 1) * 16;
    constexpr size_t TL = (size_t)POTRF_NB * POTRF_NB;
    // product segments: MODE 0 one (tile x inverse factor, K = 32 (bc + 1)); MODE 1 up to three panels, K = 128 each:
    // P0 of panel k, tile 1 of panel k-1, tile 2 of panel k-2 (= the rows of S_{k+1,*} in those panels)
    const int nseg = MODE == 0 ? 1 : (prev ? (prev2 ? 3 : 2) : 1);
    const int K0 = MODE == 0 ? 32 * (bc + 1) : POTRF_NB;
    double* As = lds; double* Bs = lds + 32 * T32_STRIDE;
    double* Ct = S + ((size_t)(k + 1) * POTRF_NB + 32 * br) * ld + (size_t)(k + (MODE == 0 ? 0 : 1)) * POTRF_NB + 32 * bc;
    double cin[4];
    if (MODE == 1) {
#pragma unroll
        for (int t = 0; t < 4; ++t) cin[t] = Ct[(size_t)(wr + 4 * t + (lane >> 4)) * ld + wc + (lane & 15)];
    }
    // 32 rows x K doubles per operand: thread -> (row = tid >> 3, 16-byte columns (tid & 7) + 8 q)
    const int row = tid >> 3, c2 = (tid & 7) * 2;
    double pa[8][2], pb[8][2];
#define BSFM_T32_FETCH(sg)                                                                                           \
    {                                                                                                                \
        const double* A_; const double* B_; int lda_;                                                                \
        if (MODE == 0) {                                                                                             \
            A_ = S + ((size_t)(k + 1) * POTRF_NB + 32 * br) * ld + (size_t)k * POTRF_NB; lda_ = ld;                  \
            B_ = Linv + (size_t)(32 * bc) * POTRF_NB;                                                                \
        } else {                                                                                                     \
            const double* base_ = (sg) == 0 ? (const double*)panel : ((sg) == 1 ? prev + TL : prev2 + 2 * TL);       \
            A_ = base_ + (size_t)(32 * br) * POTRF_NB; B_ = base_ + (size_t)(32 * bc) * POTRF_NB; lda_ = POTRF_NB;   \
        }                                                                                                            \
        _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                                              \
            if (16 * q < K0) {                                                                                       \
                const double2 ta = *reinterpret_cast<const double2*>(A_ + (size_t)row * lda_ + 16 * q + c2);         \
                const double2 tb = *reinterpret_cast<const double2*>(B_ + (size_t)row * POTRF_NB + 16 * q + c2);     \
                pa[q][0] = ta.x; pa[q][1] = ta.y; pb[q][0] = tb.x; pb[q][1] = tb.y;                                  \
            }                                                                                                        \
        }                                                                                                            \
    }
    BSFM_T32_FETCH(0)
    double acc[4] = { 0.0, 0.0, 0.0, 0.0 };
#if !BSFM_GEMM_MFMA16
    const double* ap = As + (wr + (lane & 3)) * T32_STRIDE + (lane >> 4);
#endif
    const double* bp = Bs + (wc + (lane & 15)) * T32_STRIDE + (lane >> 4);
#pragma unroll 1
    for (int sg = 0; sg < nseg; ++sg) {
        if (sg > 0) __syncthreads();             // the previous segment has been consumed
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (16 * q < K0) {
                *reinterpret_cast<double2*>(As + row * T32_STRIDE + 16 * q + c2) = make_double2(pa[q][0], pa[q][1]);
                *reinterpret_cast<double2*>(Bs + row * T32_STRIDE + 16 * q + c2) = make_double2(pb[q][0], pb[q][1]);
            }
        }
        __syncthreads();
        if (sg + 1 < nseg) BSFM_T32_FETCH(sg + 1)      // in flight during this segment's products
#if BSFM_GEMM_MFMA16
        {   // one v_mfma_f64_16x16x4 per k-step (the wave's 16 x 16 block is one accumulator tuple): 2 LDS reads instead of 5
            typedef double v4d_ __attribute__((ext_vector_type(4)));
            const double* ap16 = As + (wr + (lane & 15)) * T32_STRIDE + (lane >> 4);
            // two accumulator tuples (even / odd k-steps): a dependent 16x16x4 waits for its predecessor's write-back, two chains hide it
            v4d_ c = { acc[0], acc[1], acc[2], acc[3] }, c2 = { 0.0, 0.0, 0.0, 0.0 };
            for (int kk = 0; kk < K0; kk += 8) {
                c = __builtin_amdgcn_mfma_f64_16x16x4f64(ap16[kk], bp[kk], c, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ap16[kk + 4], bp[kk + 4], c2, 0, 0, 0);
            }
            acc[0] = c[0] + c2[0]; acc[1] = c[1] + c2[1]; acc[2] = c[2] + c2[2]; acc[3] = c[3] + c2[3];
        }
#else
        for (int kk = 0; kk < K0; kk += 4) {
            const double b = bp[kk];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(ap[4 * t * T32_STRIDE + kk], b, acc[t], 0, 0, 0);
        }
#endif
    }
#undef BSFM_T32_FETCH
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int rw = wr + 4 * t + (lane >> 4), col = wc + (lane & 15);
        if (MODE == 0) {
            // NOT written back to S here: the other column blocks of this row block still read the tile (in-place hazard
            // across workgroups); the side stream's panel kernel copies slot 0 of the compact panel to S afterwards
            panel[(size_t)(32 * br + rw) * POTRF_NB + 32 * bc + col] = acc[t];
        } else {
            Ct[(size_t)rw * ld + col] = cin[t] - acc[t];
        }
    }
}

// ---- NARROW STEPS (round 3) ------------------------------------------------------------------------------------------------
// A step with few tile rows below its diagonal tile (<= POTRF_NARROW: every step of a small problem or of the envelope solver on a
// banded scene, and the last steps of the dense factorisation) has no bulk worth a stream of its own; what it pays in the three-stream
// schedule is the plumbing -- two launch boundaries and two cross-stream event hops (~16 us) next to 58 us of kernels.  Such a step runs
// on the chain stream ALONE as three launches, no events, no owed-tile bookkeeping:
//   k_narrow_tiles<0>  every panel tile P_a = S_{k+1+a,k} inv(L_kk)^T in 32 x 32 blocks (T x 16 workgroups) -> compact panel; + y_k
//   k_narrow_tiles<1>  every trailing tile S_{k+1+a,k+1+b} -= P_a P_b^T, b <= a, in 32 x 32 blocks; + the panel copies -> S, + E -= P y_k
//   k_potrf_diag       the next diagonal tile
// Same 32 x 32-block products as k_chain_tile32 (whole K range of both operands to LDS in one step, 16x16x4 on two accumulator chains).
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_narrow_tiles(double* __restrict__ S, int ld, int k, int T, const double* __restrict__ Linv,
        double* __restrict__ panel, double* __restrict__ E, double* __restrict__ y)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr size_t TL = (size_t)POTRF_NB * POTRF_NB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int a = 0, b = 0, blk;
    if (MODE == 0) {
        if ((int)blockIdx.x == T * 16) { fwd_tile_solve(Linv, E + (size_t)k * POTRF_NB, y + (size_t)k * POTRF_NB, lds); return; }   // y_k = inv(L_kk) E_k
        a = blockIdx.x >> 4; blk = blockIdx.x & 15;
    } else {
        const int ntri = T * (T + 1) / 2;
        int w = (int)blockIdx.x - ntri * 16;
        if (w >= 0) {
            if (w < T) {                 // panel tile w -> S (the factor itself lives in S: backward substitution, exports)
                double* dst = S + ((size_t)(k + 1 + w) * POTRF_NB) * ld + (size_t)k * POTRF_NB;
                const double* src = panel + (size_t)w * TL;
                for (int idx = tid; idx < POTRF_NB * POTRF_NB / 2; idx += 256) {
                    const int r = idx >> 6, c2 = (idx & 63) * 2;
                    *reinterpret_cast<double2*>(dst + (size_t)r * ld + c2) = *reinterpret_cast<const double2*>(src + (size_t)r * POTRF_NB + c2);
                }
                return;
            }
            w -= T;                      // forward substitution: E_{k+1+w} -= P_w y_k, 2 lanes per row
            const int row = tid >> 1, part = tid & 1;
            const double* Pr = panel + (size_t)w * TL + (size_t)row * POTRF_NB + 64 * part;
            const double* yk = y + (size_t)k * POTRF_NB + 64 * part;
            double sacc = 0.0;
#pragma unroll 16
            for (int c = 0; c < 64; ++c) sacc += Pr[c] * yk[c];
            sacc += __shfl_xor(sacc, 1, 64);
            if (part == 0) E[(size_t)(k + 1 + w) * POTRF_NB + row] -= sacc;
            return;
        }
        const int t = blockIdx.x >> 4; blk = blockIdx.x & 15;
        a = (int)((sqrtf(8.0f * t + 1.0f) - 1.0f) * 0.5f);
        while ((a + 1) * (a + 2) / 2 <= t) ++a;
        while (a * (a + 1) / 2 > t) --a;
        b = t - a * (a + 1) / 2;
    }
    const int br = blk >> 2, bc = blk & 3;
    if (MODE == 1 && a == b && bc > br) return;              // diagonal tiles: the lower-triangle blocks are all the next kernels read
    const int wr = (wave >> 1) * 16, wc = (wave & 1) * 16;
    const int K0 = MODE == 0 ? 32 * (bc + 1) : POTRF_NB;     // inv(L_kk) is lower triangular: column block bc only needs k < 32 (bc + 1)
    double* As = lds; double* Bs = lds + 32 * T32_STRIDE;
    const double* A_; const double* B_; int lda_;
    if (MODE == 0) { A_ = S + ((size_t)(k + 1 + a) * POTRF_NB + 32 * br) * ld + (size_t)k * POTRF_NB; lda_ = ld; B_ = Linv + (size_t)(32 * bc) * POTRF_NB; }
    else { A_ = panel + (size_t)a * TL + (size_t)(32 * br) * POTRF_NB; lda_ = POTRF_NB; B_ = panel + (size_t)b * TL + (size_t)(32 * bc) * POTRF_NB; }
    double* Ct = S + ((size_t)(k + 1 + a) * POTRF_NB + 32 * br) * ld + (size_t)(k + 1 + b) * POTRF_NB + 32 * bc;     // MODE 1 only
    double cin[4] = { 0.0, 0.0, 0.0, 0.0 };
    if (MODE == 1) {
#pragma unroll
        for (int t = 0; t < 4; ++t) cin[t] = Ct[(size_t)(wr + 4 * t + (lane >> 4)) * ld + wc + (lane & 15)];
    }
    const int row = tid >> 3, c2 = (tid & 7) * 2;
    double pa[8][2], pb[8][2];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        if (16 * q < K0) {
            const double2 ta = *reinterpret_cast<const double2*>(A_ + (size_t)row * lda_ + 16 * q + c2);
            const double2 tb = *reinterpret_cast<const double2*>(B_ + (size_t)row * POTRF_NB + 16 * q + c2);
            pa[q][0] = ta.x; pa[q][1] = ta.y; pb[q][0] = tb.x; pb[q][1] = tb.y;
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        if (16 * q < K0) {
            *reinterpret_cast<double2*>(As + row * T32_STRIDE + 16 * q + c2) = make_double2(pa[q][0], pa[q][1]);
            *reinterpret_cast<double2*>(Bs + row * T32_STRIDE + 16 * q + c2) = make_double2(pb[q][0], pb[q][1]);
        }
    }
    __syncthreads();
    typedef double v4d_ __attribute__((ext_vector_type(4)));
    const double* ap16 = As + (wr + (lane & 15)) * T32_STRIDE + (lane >> 4);
    const double* bp = Bs + (wc + (lane & 15)) * T32_STRIDE + (lane >> 4);
    v4d_ c = { 0.0, 0.0, 0.0, 0.0 }, cc = { 0.0, 0.0, 0.0, 0.0 };
    for (int kk = 0; kk < K0; kk += 8) {
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(ap16[kk], bp[kk], c, 0, 0, 0);
        cc = __builtin_amdgcn_mfma_f64_16x16x4f64(ap16[kk + 4], bp[kk + 4], cc, 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int rw = wr + 4 * t + (lane >> 4), col = wc + (lane & 15);
        const double v = c[t] + cc[t];
        if (MODE == 0) panel[(size_t)a * TL + (size_t)(32 * br + rw) * POTRF_NB + 32 * bc + col] = v;
        else Ct[(size_t)rw * ld + col] = cin[t] - v;
    }
}

// (Also tried and dropped: an XCD-aware blockIdx -> tile order (super-tiles of 8x8 tiles per XCD chunk).  +8-11 % for the
// kernel alone on the whole device (scripts/ubench_syrk.hip), exactly 0 in the factorisation -- with the CU-masked bulk
// stream (35.0 TFLOP/s either way) and again without the mask (38.4 vs 38.5).  Nor do 64-row half-tile workgroups for the
// late, device-underfilling launches: 37.7-38.5 TFLOP/s with a threshold of 24-32 tile rows vs 38.0 without.)
// Trailing update: S_ij -= P_a P_b^T for k < j <= i (a = i-k-1, b = j-k-1), one tile per workgroup.
// part 1 = only the first trailing column (b == 0, grid T): the tiles the NEXT panel needs (lookahead stream);
// part 2 = every other tile (b >= 1) except S_{k+2,k+2}, grid T(T-1)/2 - 1: the bulk, on the update stream.
// NT: the C tile is touched exactly once per launch (128 KB in, 128 KB out per workgroup, 64 workgroups in flight per XCD = 16 MB
// against 4 MB of L2) while the panel tiles are re-read by every workgroup of a tile row / column: non-temporal loads and stores
// for C keep the use-once stream from evicting the panel out of L2 (r01 counters: the surplus of HBM traffic over the algorithmic
// C bytes was panel tiles missing L2).  Measured at config 3: 38.7 vs 39.2 TFLOP/s for the kernel, 8.44 vs 8.52 ms for the solve --
// neutral, so the plain variant stays the default; BSFM_SYRK_NT=1 selects this one (same binary).
// pprev != nullptr: a PAIRED launch -- panel k-1 (compact copy pprev, tile a+1 = the same tile row) is applied in the same pass over
// C: one read and one write of the tile for 256 accumulation steps (with the 16x16x4 loop: 61 instead of 44 TFLOP/s for the kernel alone).
// (Tried and dropped, round 3: 16-byte C traffic.  The 16x16x4 accumulator layout gives a lane ONE double of four different rows, so the
// prologue / epilogue issue 32 eight-byte loads / stores per lane; adjacent lanes swapped one register of each row pair through a DPP quad
// permute so that every lane held two adjacent columns of one row and moved 16 bytes per instruction (16 instead of 32, same 128-byte
// segments).  Kernel alone 43.0-45.7 vs 43.5-46.9 TFLOP/s, in the factorisation 38.9 vs 39.6 TFLOP/s, solve 8.44 vs 8.34 ms
// (profiles/r03_syrk_c16_*): the epilogue is not issue-bound here.  Also round 3: de-phasing the two workgroups of a CU (b and b + 256
// DO share a CU) by delaying one of them, persistent workgroups with static or atomic tile hand-out -- 39-43 vs 43.9 TFLOP/s
// (profiles/r03_syrk_stagger_ubench.txt).)
template <bool NT>
__global__ __launch_bounds__(512, BSFM_SYRK_WPS) void k_syrk_update(double* __restrict__ S, int ld, int k, const double* __restrict__ panel, int part,
                                                                    const double* __restrict__ pprev)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    int a, b;
    if (part == 1) { a = blockIdx.x; b = 0; }
    else {
        // part 2: tile 0 of the triangle = S_{k+2,k+2} is left to the chain (k_chain_tile32<1>); part 3 (panel engine): every tile
        const int t = blockIdx.x + (part == 2 ? 1 : 0);
        a = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
        while ((a + 1) * (a + 2) / 2 <= t) ++a;
        while (a * (a + 1) / 2 > t) --a;
        b = t - a * (a + 1) / 2;
        ++a; ++b;      // triangle of size T-1 shifted past the first column
    }
    const int i = k + 1 + a, j = k + 1 + b;
    // The accumulators START as the C tile and the A operand is negated while it is staged into LDS, so the matrix
    // cores produce S_ij - P_a P_b^T directly: the C loads overlap the first panel-chunk loads (no extra registers)
    // and the epilogue is stores only -- a workgroup no longer waits for its 128 KB C tile at the end
    // (measured on the tile kernel alone: 42 -> 51 TFLOP/s is the cost of a load/subtract/store epilogue).
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 64;
    double* Sij = S + ((size_t)i * POTRF_NB) * ld + (size_t)j * POTRF_NB;
    double acc[8][4];
    {
        const double* cp = Sij + (size_t)(wr + (lane >> 4)) * ld + wc + (lane & 15);   // rows 4 ld apart
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[q][u] = NT ? __builtin_nontemporal_load(cp + 16 * u) : cp[16 * u];
            cp += 4 * (size_t)ld;
        }
    }
    gemm_nt_128<true>(panel + (size_t)a * POTRF_NB * POTRF_NB, POTRF_NB, panel + (size_t)b * POTRF_NB * POTRF_NB, POTRF_NB,
                      POTRF_NB, lds, acc);
    if (pprev)
        gemm_nt_128<true>(pprev + (size_t)(a + 1) * POTRF_NB * POTRF_NB, POTRF_NB, pprev + (size_t)(b + 1) * POTRF_NB * POTRF_NB, POTRF_NB,
                          POTRF_NB, lds, acc);
    // the store address is recomputed from an opaque copy of the thread id: keeping the load addresses alive across
    // the K loop would push the kernel over its 128-VGPR budget (and the staging registers into scratch)
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    double* Sl = Sij + (size_t)(((tid2 >> 7) << 5) + ((tid2 & 63) >> 4)) * ld + (((tid2 >> 6) & 1) << 6) + (tid2 & 15);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { if (NT) __builtin_nontemporal_store(acc[q][u], Sl + 16 * u); else Sl[16 * u] = acc[q][u]; }
        Sl += 4 * (size_t)ld;
    }
}

// (Tried three times and dropped: rank-256 bulk launches that apply panels k and k+1 in one pass over C (pprev).  The kernel alone runs at
// 46.6 (4x4x4 loop) / 61 (16x16x4 loop) instead of 42.9 / 44 TFLOP/s, but the factorisation got slower every time -- two-stream schedule
// 10.9 vs 10.1 ms, three-stream schedule 9.5 vs 9.0 ms, round 2 with 30 / 40 / 50 paired tile rows 8.99 / 8.83 / 8.67 vs 8.46 ms: the bulk
// stream idles while the chain and side streams prepare the second panel of a pair, the paired launches reach 43, not 61 TFLOP/s (two
// different panels per tile), and their two-round workgroups hold the slots the chain's kernels wait for twice as long.  The host
// side of that schedule was removed in round 3; the kernel keeps its pprev argument.)

// (Tried and dropped, round 2: TWO bulk streams, tile rows shared out by the parity of the global tile row, so that the ragged last
// round of one launch -- a launch runs ceil(tiles / 512) rounds of workgroups, 4.45 -> 5 at T = 68 -- is filled by the other stream's
// launch.  Solve 10.5 ms instead of 8.55 with 4 hardware queues, 13.2 ms with GPU_MAX_HW_QUEUES=8: the two launches interleave on
// every CU and each runs at 23-27 TFLOP/s.)
// (Tried and dropped: taking the tile inverse off the chain -- the chain kernel factors only, a side-stream launch inverts,
// and the chain's own panel tile comes from a 16-row block substitution against L (8 workgroups, inverse diagonal blocks
// from the factorisation).  Correct, but ~10 % slower at every size from 2 to 70 tile columns: the extra chain -> side
// event hop (~12 us) and the separate 150 KB-LDS launch cost more than the 12.5 us of inverse they remove from the chain.)
// Diagonal tile: Cholesky factor AND its inverse in ONE 512-thread workgroup -- the serial critical path of the
// factorisation, so it is blocked to keep the truly serial work tiny:
//   * the 128x128 tile lives in LDS (row stride 132 doubles) as 8x8 blocks of 16x16;
//   * phase A, per block column s: wave 0 factors the 16x16 diagonal block with one lane per row (v_readlane
//     broadcasts, no LDS traffic); the rows of the panel blocks below are solved against it by one thread each (forward
//     substitution in registers) while an idle wave derives inv(L_ss) the same way; the trailing updates
//     T[I][J] -= T[I][s] T[J][s]^T are 16x16x16 products on v_mfma_f64_4x4x4_4b (16 instructions each), done by waves
//     1..7 WHILE wave 0 already updates and factors the next diagonal block (51 -> 47 us per tile);
//   * phase B, inverse: X = inv(L) by block forward substitution, ONE WAVE PER BLOCK COLUMN J and no barriers:
//     X[I][J] = -inv(L_II) * sum_{K=J}^{I-1} L[I][K] X[K][J]; X[K][J]^T is parked in the unused upper block T[J][K]
//     so that every product is of the A * Bt^T ("NT") form the MFMA fragments read directly from LDS.
constexpr int DG_TS = 134;                                   // LDS row stride of the tile (doubles): 132 -> 134 takes the inverse phase from 12.6 to 10.6 us (bank conflicts of the 16x16 operand reads; 136 is 10 us WORSE, 129 / 130 / 138 / 140 within 1 us of 134)
constexpr int DG_LDS_DOUBLES = POTRF_NB * DG_TS + 8 * 256;

__device__ __forceinline__ double rsqrt_f64(double v)
{
    double s = __builtin_amdgcn_rsq(v);          // v_rsq_f64 seed, two Newton steps to full precision
    s = s * (1.5 - 0.5 * v * s * s);
    s = s * (1.5 - 0.5 * v * s * s);
    return s;
}

// acc[a] += A(16x16, row stride sa) * Bt(16x16, row stride sb)^T ; acc[a] holds rows 4a + (lane>>4), col lane&15.
// (v_mfma_f64_16x16x4: its accumulator registers are exactly this layout, and a 16 x 16 x 16 product takes 8 LDS reads and 4 matrix
//  instructions instead of 20 and 16 with the 4x4x4 form -- the diagonal-tile kernel is sensitive to LDS latency, not to the matrix pipe.)
__device__ __forceinline__ void mma16_nt(double (&acc)[4], const double* A, int sa, const double* Bt, int sb, int lane)
{
#if BSFM_GEMM_MFMA16
    typedef double v4d_ __attribute__((ext_vector_type(4)));
    v4d_ c = { acc[0], acc[1], acc[2], acc[3] };
    double av[4], bv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        av[q] = A[(lane & 15) * sa + 4 * q + (lane >> 4)];
        bv[q] = Bt[(lane & 15) * sb + 4 * q + (lane >> 4)];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) c = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], c, 0, 0, 0);
    acc[0] = c[0]; acc[1] = c[1]; acc[2] = c[2]; acc[3] = c[3];
#else
#pragma unroll
    for (int kk = 0; kk < 16; kk += 4) {
        const double b = Bt[(lane & 15) * sb + kk + (lane >> 4)];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const double av = A[(4 * a + (lane & 3)) * sa + kk + (lane >> 4)];
            acc[a] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b, acc[a], 0, 0, 0);
        }
    }
#endif
}

// ---- A1: wave 0 factors the diagonal block (s,s) in registers, ONE LANE PER ROW (lanes 16.. mirror lane & 15):
// lane r holds D[r][0..15]; per column the pivot and the column entries D[c][j] come by v_readlane (static lane
// ids -> SGPR operands, no LDS round trip), the multiplier uses 1/pivot (v_rcp_f64 + 2 Newton steps), and all
// square roots are deferred to the end of the block (L = raw column * 1/sqrt(pivot)).  Entries above the
// diagonal take part in the updates unmasked: they are never read.
__device__ __forceinline__ void diag_factor_block(double* __restrict__ T, double* __restrict__ svec, int* __restrict__ info,
                                                  int base, int n_total, int s, int lane)
{
    double* B = T + (16 * s) * DG_TS + 16 * s;
    const int r = lane & 15;
    double d[16], pv[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) d[c] = B[r * DG_TS + c];
#define BSFM_RDLANE(v, l) __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), (l)), __builtin_amdgcn_readlane(__double2loint(v), (l)))
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double piv = BSFM_RDLANE(d[j], j);
        pv[j] = piv;
        double inv = __builtin_amdgcn_rcp(piv);
        inv = fma(fma(-piv, inv, 1.0), inv, inv);
        inv = fma(fma(-piv, inv, 1.0), inv, inv);
        const double lr = d[j] * inv;
#pragma unroll
        for (int c = j + 1; c < 16; ++c) { const double sc = BSFM_RDLANE(d[j], c); d[c] -= lr * sc; }
    }
    // first non-positive pivot = dpotrf's info
    if (lane == 0) {
        int bad = -1;
#pragma unroll
        for (int j = 15; j >= 0; --j) if (!(pv[j] > 0.0)) bad = j;
        if (bad >= 0 && base + 16 * s + bad < n_total) atomicCAS(info, 0, base + 16 * s + bad + 1);
    }
    double myp = pv[0];
#pragma unroll
    for (int c = 1; c < 16; ++c) myp = (r == c) ? pv[c] : myp;
    const double myrs = rsqrt_f64(myp);                       // 1 / L[r][r]
    if (lane < 16) {
        svec[16 * s + r] = myrs;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const double rs_c = BSFM_RDLANE(myrs, c);
            B[r * DG_TS + c] = (c <= r) ? d[c] * rs_c : 0.0;
        }
    }
#undef BSFM_RDLANE
}

__device__ __forceinline__ void diag_tile_body(double* __restrict__ dlds, double* __restrict__ S, int ld, int k, int n_total,
        double* __restrict__ Linv, int* __restrict__ info, long long* __restrict__ dbg)
{
    long long t0 = 0; if (dbg && threadIdx.x == 0) t0 = wall_clock64();
    double* T = dlds;                                  // [128][DG_TS]
    double* Di = dlds + POTRF_NB * DG_TS;              // 8 blocks of 16x16: inverse diagonal blocks
    // statically declared => guaranteed LDS address space (a volatile generic pointer compiled to FLAT ops + waits)
    __shared__ double svec[POTRF_NB];      // raw pivots (statically declared => LDS address space)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int base = k * POTRF_NB;
    double* G = S + (size_t)base * ld + base;
    __syncthreads();

    // load the tile (lower triangle; identity in the padding beyond n_total)
    {
        double v[32];
#pragma unroll
        for (int it = 0; it < 32; ++it) {          // all 32 loads in flight before the first LDS store
            const int idx = tid + 512 * it, r = idx >> 7, c = idx & 127;
            v[it] = 0.0;
            if (base + r < n_total && base + c < n_total) { if (c <= r) v[it] = G[(size_t)r * ld + c]; }
            else if (r == c) v[it] = 1.0;
        }
#pragma unroll
        for (int it = 0; it < 32; ++it) {
            const int idx = tid + 512 * it, r = idx >> 7, c = idx & 127;
            T[r * DG_TS + c] = v[it];
        }
    }
    __syncthreads();

    if (dbg && threadIdx.x == 0) dbg[0] = wall_clock64() - t0;
    long long tA1 = 0, tA2 = 0, tA3 = 0, tq = 0;
    // Block-column schedule (two barriers per block column):
    //   phase P: all waves solve the rows below block (s,s) against L_ss (A2) while wave 7 derives inv(L_ss);
    //   phase Q: wave 0 alone applies column s to block (s+1,s+1) and immediately factors it (A1 of the next column) --
    //            meanwhile waves 1..7 apply column s to all the other trailing blocks (A3).  The serial 16x16
    //            factorisation thus overlaps the MFMA updates instead of following them.
    if (wave == 0) diag_factor_block(T, svec, info, base, n_total, 0, lane);
    __syncthreads();
    for (int s = 0; s < 8; ++s) {
        if (dbg && threadIdx.x == 0) tq = wall_clock64();
        // ---- A2: rows of the panel blocks below, x <- x * inv(L_ss)^T by forward substitution, ONE THREAD PER ROW with
        // the 16 entries in registers (L_ss and 1/diag are wave-uniform LDS broadcasts): no cross-lane traffic at all.
        // Wave 7 meanwhile runs the SAME substitution on the rows of the identity: row c of inv(L_ss)^T = column c of
        // inv(L_ss), the diagonal block of the inverse that phase B needs -- off the critical path.
        {
            const double* Lb = T + (16 * s) * DG_TS + 16 * s;
            const int nrows = (7 - s) * 16;
            const bool inv_job = (wave == 7 && lane < 16);
            if (inv_job || tid < nrows) {
                double* X = T + (size_t)(16 * (s + 1) + tid) * DG_TS + 16 * s;
                double t[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) t[c] = inv_job ? ((c == lane) ? 1.0 : 0.0) : X[c];
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const double xc = t[c] * svec[16 * s + c];
                    t[c] = xc;
#pragma unroll
                    for (int kq = c + 1; kq < 16; ++kq) t[kq] -= xc * Lb[kq * DG_TS + c];
                }
                if (inv_job) {
#pragma unroll
                    for (int rr = 0; rr < 16; ++rr) Di[s * 256 + rr * 16 + lane] = t[rr];     // inv(L_ss)[rr][lane]
                } else {
#pragma unroll
                    for (int c = 0; c < 16; ++c) X[c] = t[c];
                }
            }
        }
        __syncthreads();
        if (dbg && threadIdx.x == 0) { const long long t = wall_clock64(); tA2 += t - tq; tq = t; }
        // ---- phase Q
        if (wave == 0) {
            if (s < 7) {
                double acc[4] = { 0.0, 0.0, 0.0, 0.0 };
                const double* Ps = T + (16 * (s + 1)) * DG_TS + 16 * s;
                mma16_nt(acc, Ps, DG_TS, Ps, DG_TS, lane);
                double* C = T + (16 * (s + 1)) * DG_TS + 16 * (s + 1);
#pragma unroll
                for (int q = 0; q < 4; ++q) C[(4 * q + (lane >> 4)) * DG_TS + (lane & 15)] -= acc[q];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                diag_factor_block(T, svec, info, base, n_total, s + 1, lane);
            }
        } else {
            // A3: trailing blocks (I,J), s < J <= I < 8, except (s+1,s+1) which wave 0 has taken
            const int rem = 7 - s;
            const int cnt = rem * (rem + 1) / 2;
            for (int idx = wave; idx < cnt; idx += 7) {          // idx 0 = block (s+1,s+1); waves 1..7 start at 1..7
                int a = (int)((sqrtf(8.0f * idx + 1.0f) - 1.0f) * 0.5f);
                while ((a + 1) * (a + 2) / 2 <= idx) ++a;
                while (a * (a + 1) / 2 > idx) --a;
                const int b = idx - a * (a + 1) / 2;
                const int I = s + 1 + a, J = s + 1 + b;
                double acc[4] = { 0.0, 0.0, 0.0, 0.0 };
                mma16_nt(acc, T + (16 * I) * DG_TS + 16 * s, DG_TS, T + (16 * J) * DG_TS + 16 * s, DG_TS, lane);
                double* C = T + (16 * I) * DG_TS + 16 * J;
#pragma unroll
                for (int q = 0; q < 4; ++q) C[(4 * q + (lane >> 4)) * DG_TS + (lane & 15)] -= acc[q];
            }
        }
        __syncthreads();
        if (dbg && threadIdx.x == 0) { const long long t = wall_clock64(); tA1 += t - tq; tq = t; }
    }
    if (dbg && threadIdx.x == 0) { dbg[4] = tA1; dbg[5] = tA2; dbg[6] = tA3; }
    if (dbg && threadIdx.x == 0) dbg[1] = wall_clock64() - t0;
    // ---- write L back to S (lower triangle incl. diagonal)
#pragma unroll 8
    for (int it = 0; it < 32; ++it) {
        const int idx = tid + 512 * it, r = idx >> 7, c = idx & 127;
        if (c <= r) G[(size_t)r * ld + c] = T[r * DG_TS + c];
    }
    __syncthreads();
    if (dbg && threadIdx.x == 0) dbg[2] = wall_clock64() - t0;
    // ---- phase B: inverse, wave J owns block column J
    double* Li = Linv + (size_t)k * POTRF_NB * POTRF_NB;
    {
        const int J = wave;
        // X[J][J]^T = inv(L_JJ)^T parked in T[J][J] (L_JJ itself is no longer needed), and written out
        double* DJ = T + (16 * J) * DG_TS + 16 * J;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 4 * q + (lane >> 4), c = lane & 15;
            const double v = Di[J * 256 + r * 16 + c];
            Li[(16 * J + r) * POTRF_NB + 16 * J + c] = v;
            DJ[c * DG_TS + r] = v;
        }
        // zero blocks above the diagonal in the output
        for (int Iu = 0; Iu < J; ++Iu)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                Li[(16 * Iu + 4 * q + (lane >> 4)) * POTRF_NB + 16 * J + (lane & 15)] = 0.0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int I = J + 1; I < 8; ++I) {
            double acc[4] = { 0.0, 0.0, 0.0, 0.0 };
            for (int K = J; K < I; ++K)
                mma16_nt(acc, T + (16 * I) * DG_TS + 16 * K, DG_TS, T + (16 * J) * DG_TS + 16 * K, DG_TS, lane);
            double* U = T + (16 * J) * DG_TS + 16 * I;       // upper block (J,I): receives acc^T, then X[I][J]^T
#pragma unroll
            for (int q = 0; q < 4; ++q) U[(lane & 15) * DG_TS + 4 * q + (lane >> 4)] = acc[q];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            double res[4] = { 0.0, 0.0, 0.0, 0.0 };
            mma16_nt(res, Di + I * 256, 16, U, DG_TS, lane);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = 4 * q + (lane >> 4), c = lane & 15;
                U[c * DG_TS + r] = -res[q];
                Li[(16 * I + r) * POTRF_NB + 16 * J + c] = -res[q];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    if (dbg && threadIdx.x == 0) dbg[3] = wall_clock64() - t0;
}

__global__ __launch_bounds__(512) void k_potrf_diag(double* __restrict__ S, int ld, int k, int n_total,
        double* __restrict__ Linv, int* __restrict__ info, long long* __restrict__ dbg)
{
    extern __shared__ __attribute__((aligned(16))) double dlds[];
    diag_tile_body(dlds, S, ld, k, n_total, Linv, info, dbg);
}

// Systems of ONE tile (<= 128 unknowns: up to 14 cameras at cnp = 9, the first rounds of every incremental reconstruction): factor,
// inverse and both substitutions in the one workgroup that holds the tile -- one launch instead of nine stream operations
// (rhs copy, diagonal tile, forward tile, flag memset, persistent backward kernel, time-out fold, solution copy ...).
// x = inv(L)^T (inv(L) E): the inverse factor has just been written by this workgroup (and was never read before in this launch).
__global__ __launch_bounds__(512) void k_potrf_solve_one(double* __restrict__ S, int ld, int n_total, double* __restrict__ Linv,
        int* __restrict__ info, const double* __restrict__ E, double* __restrict__ x)
{
    extern __shared__ __attribute__((aligned(16))) double dlds[];
    diag_tile_body(dlds, S, ld, 0, n_total, Linv, info, nullptr);
    __threadfence_block();
    __syncthreads();
    double* vec = dlds; double* red = dlds + POTRF_NB; double* yv = dlds + 5 * POTRF_NB;     // the tile in LDS is no longer needed
    const int r = threadIdx.x & 127, h = threadIdx.x >> 7;                                  // 4 quarter-sums per row
    if (threadIdx.x < POTRF_NB) vec[threadIdx.x] = threadIdx.x < n_total ? E[threadIdx.x] : 0.0;
    __syncthreads();
    {
        const double* Li = Linv + (size_t)r * POTRF_NB + 32 * h;
        double s = 0.0;
#pragma unroll 8
        for (int c = 0; c < 32; ++c) s += Li[c] * vec[32 * h + c];
        red[h * POTRF_NB + r] = s;
    }
    __syncthreads();
    if (threadIdx.x < POTRF_NB) yv[r] = (red[r] + red[POTRF_NB + r]) + (red[2 * POTRF_NB + r] + red[3 * POTRF_NB + r]);
    __syncthreads();
    {
        double s = 0.0;                                                                      // x[r] = sum_q inv(L)[q][r] y[q]
#pragma unroll 8
        for (int q = 0; q < 32; ++q) s += Linv[(size_t)(32 * h + q) * POTRF_NB + r] * yv[32 * h + q];
        red[h * POTRF_NB + r] = s;
    }
    __syncthreads();
    if (threadIdx.x < POTRF_NB && threadIdx.x < n_total) x[r] = (red[r] + red[POTRF_NB + r]) + (red[2 * POTRF_NB + r] + red[3 * POTRF_NB + r]);
}

// Backward substitution x = L^-T y as ONE persistent launch (replaces nblk dependent launches).
// Workgroup kk owns tile column kk, all nblk workgroups are resident at once (nblk <= #CUs, 256 threads, no big LDS):
//   for i = nblk-1 .. kk+1 :  wait for x_i  ->  y_kk -= L_{i,kk}^T x_i        (tile (i,kk) prefetched into registers
//   x_kk = inv(L_kk)^T y_kk  ->  publish                                       while waiting; inv(L_kk) lives in registers)
// Hand-off follows the guide's write-through recipe (cdna_hip_programming.md, Guideline 16, R1): the producer stores
// x_kk with agent-scope relaxed atomic stores (sc1, write-through), every storing wave drains vmcnt, one lane stores
// the flag; consumers poll the flag relaxed from one lane and then read x_i with agent-scope (sc1) loads, which bypass
// the possibly stale L1.  Flags are zeroed by a memset node before every launch; epoch = 1.
__global__ __launch_bounds__(256) void k_bwd_persistent(const double* __restrict__ S, int ld, int nblk,
        const double* __restrict__ Linv, const double* __restrict__ y, double* x, int* flags, int* timeout, const int* __restrict__ last_row)
{
    __shared__ double yk[POTRF_NB];
    __shared__ double xi[POTRF_NB];
    __shared__ double red[POTRF_NB];
    const int kk = nblk - 1 - (int)blockIdx.x;
    const int c = threadIdx.x & 127, h = threadIdx.x >> 7;
    double lreg[64], tcur[64];
    {
        const double* Li = Linv + (size_t)kk * POTRF_NB * POTRF_NB + (size_t)(64 * h) * POTRF_NB + c;
#pragma unroll
        for (int r = 0; r < 64; ++r) lreg[r] = Li[(size_t)r * POTRF_NB];
    }
    if (threadIdx.x < POTRF_NB) yk[threadIdx.x] = y[(size_t)kk * POTRF_NB + threadIdx.x];
    __syncthreads();
    const int itop = last_row ? last_row[kk] : nblk - 1;      // tile envelope: the tiles (i, kk) beyond it are structurally zero
    for (int i = itop; i > kk; --i) {
        {   // tile (i, kk), rows of this half, column c
            const double* Lc = S + ((size_t)i * POTRF_NB + 64 * h) * ld + (size_t)kk * POTRF_NB + c;
#pragma unroll
            for (int r = 0; r < 64; ++r) tcur[r] = Lc[(size_t)r * ld];
        }
        if (threadIdx.x == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(&flags[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 26)) { atomicExch(timeout, 1); break; }     // bounded spin
            }
        }
        __syncthreads();
        if (threadIdx.x < POTRF_NB)
            xi[threadIdx.x] = __hip_atomic_load(&x[(size_t)i * POTRF_NB + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        double sacc = 0.0;
#pragma unroll
        for (int r = 0; r < 64; ++r) sacc += tcur[r] * xi[64 * h + r];
        if (h == 1) red[c] = sacc;
        __syncthreads();
        if (h == 0) yk[c] -= sacc + red[c];
        __syncthreads();
    }
    double sacc = 0.0;
#pragma unroll
    for (int r = 0; r < 64; ++r) sacc += lreg[r] * yk[64 * h + r];
    if (h == 1) red[c] = sacc;
    __syncthreads();
    if (h == 0) __hip_atomic_store(&x[(size_t)kk * POTRF_NB + c], sacc + red[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every storing wave drains its write-through stores
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&flags[kk], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// A bounded spin of k_bwd_persistent that expired leaves its word set: the solve then reports POTRF_INFO_TIMEOUT through dpotrf's
// info word (solver.hip treats a negative info as fatal) instead of handing back a silently wrong x.
constexpr int POTRF_INFO_TIMEOUT = -7;
__global__ void k_fold_timeout(const int* __restrict__ timeout, int* __restrict__ info)
{
    if (*timeout != 0) *info = POTRF_INFO_TIMEOUT;
}

// ------------------------------------------------------------------------------------------------
inline void potrf_free(PotrfWorkspace& w)
{
    flow_release(w);
    bsfm::dev_free(w.panel, true);
    bsfm::dev_free(w.linv, true);
    bsfm::dev_free(w.y, true);
    bsfm::dev_free(w.xs, true);
    if (w.xu) { (void)hipFree(w.xu); w.xu = nullptr; }
    bsfm::dev_free(w.etmp, true);
    bsfm::dev_free(w.bflags, true);
    bsfm::dev_free(w.d_last, true);
    if (w.rb_handle && w.rb_destroy) w.rb_destroy(w.rb_handle);
    if (w.ev0) (void)hipEventDestroy(w.ev0);
    if (w.ev1) (void)hipEventDestroy(w.ev1);
    for (int i = 0; w.sy0 && i < w.nblk; ++i) { (void)hipEventDestroy(w.sy0[i]); (void)hipEventDestroy(w.sy1[i]); }
    delete[] w.sy0; delete[] w.sy1; delete[] w.sy_flops;
    for (int i = 0; w.evP && i <= w.nblk; ++i) {
        (void)hipEventDestroy(w.evP[i]); (void)hipEventDestroy(w.evU[i]); (void)hipEventDestroy(w.evT[i]); (void)hipEventDestroy(w.evC[i]);
    }
    delete[] w.evP; delete[] w.evU; delete[] w.evT; delete[] w.evC;
    if (w.s2) { if (w.s2_masked) (void)hipStreamDestroy(w.s2); else stream_pool().release(w.s2); }
    stream_pool().release(w.sd);
    w = PotrfWorkspace();
}

inline int potrf_init(PotrfWorkspace& w, int ld, int backend)
{
    w.ld = ld; w.nblk = ld / POTRF_NB; w.backend = backend;
    const size_t tile = (size_t)POTRF_NB * POTRF_NB;
    if (bsfm::dev_alloc((void**)&w.panel, 4 * std::max<size_t>(1, (size_t)(w.nblk - 1)) * tile * sizeof(double)) != hipSuccess) return -1;
    {   // Optional CU reservation (BSFM_PANEL_CUS=n masks n CUs out of the bulk stream, hipExtStreamCreateWithCUMask).
        // It was essential for the two-stream schedule (the 150 KB-LDS diagonal-tile workgroup could never be placed
        // while 2 400 bulk workgroups were queued: 12.4 vs 13.9 ms).  With the three-stream schedule the chain only
        // needs small kernels while a bulk launch is in flight, and no reservation is faster at every size measured
        // (config 3: 8.9 vs 9.2 ms; 2 000 cameras: 44.5 vs 48.3 ms; 3 000: 136 vs 148 ms) -- default 0.
        int reserve = 0;
        if (const char* e = getenv("BSFM_PANEL_CUS")) reserve = atoi(e);
        hipDeviceProp_t prop; int dev = 0; (void)hipGetDevice(&dev);
        hipError_t rc = hipErrorUnknown;
        // only systems with enough tiles to be bulk-bound profit from the reservation; small ones use a plain stream
        if (reserve > 0 && w.nblk >= 16 && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 2 * reserve) {
            const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
            std::vector<uint32_t> mask(words, 0u);
            for (int c = reserve; c < ncu; ++c) mask[c >> 5] |= 1u << (c & 31);
            rc = hipExtStreamCreateWithCUMask(&w.s2, (uint32_t)words, mask.data());
        }
        w.s2_masked = (rc == hipSuccess);
        if (rc != hipSuccess && !(w.s2 = stream_pool().acquire())) return -1;
    }
    if (!(w.sd = stream_pool().acquire())) return -1;
    w.evP = new hipEvent_t[w.nblk + 1]; w.evU = new hipEvent_t[w.nblk + 1];
    w.evT = new hipEvent_t[w.nblk + 1]; w.evC = new hipEvent_t[w.nblk + 1];
    for (int i = 0; i <= w.nblk; ++i) {
        if (hipEventCreateWithFlags(&w.evP[i], hipEventDisableTiming) != hipSuccess) return -1;
        if (hipEventCreateWithFlags(&w.evU[i], hipEventDisableTiming) != hipSuccess) return -1;
        if (hipEventCreateWithFlags(&w.evT[i], hipEventDisableTiming) != hipSuccess) return -1;
        if (hipEventCreateWithFlags(&w.evC[i], hipEventDisableTiming) != hipSuccess) return -1;
    }
    if (bsfm::dev_alloc((void**)&w.linv, (size_t)w.nblk * tile * sizeof(double)) != hipSuccess) return -1;
    if (hipMemset(w.linv, 0, (size_t)w.nblk * tile * sizeof(double)) != hipSuccess) return -1;      // the dataflow POTRF never writes the (zero) upper triangle of an inverse factor
    if (const char* e = getenv("BSFM_CHOL")) w.use_flow = strcmp(e, "streams") != 0;
    if (bsfm::dev_alloc((void**)&w.y, (size_t)ld * sizeof(double)) != hipSuccess) return -1;
    if (bsfm::dev_alloc((void**)&w.xs, (size_t)ld * sizeof(double)) != hipSuccess) return -1;
    if (w.xu) { (void)hipFree(w.xu); w.xu = nullptr; }
    // ORDINARY device memory by default.  Uncached memory (BSFM_XU_UNCACHED=1) makes the scalar polls see a store 0.66 us after it instead of 1.3 us
    // (k_bwd_scalar 181 instead of 242 us at 71 tile columns) -- but buffers of that type, allocated and freed between solves, gave WRONG RESULTS in
    // ~1 % of a randomised sweep when the chain's hand-offs used them (profiles/r06_chain_data_flags.txt): not a memory type to ship a solution vector in.
    if (getenv("BSFM_XU_UNCACHED") && atoi(getenv("BSFM_XU_UNCACHED")) != 0) {
        if (hipExtMallocWithFlags((void**)&w.xu, (size_t)ld * sizeof(double), hipDeviceMallocUncached) != hipSuccess) { w.xu = nullptr; (void)hipGetLastError(); }
    } else if (hipMalloc((void**)&w.xu, (size_t)ld * sizeof(double)) != hipSuccess) { w.xu = nullptr; (void)hipGetLastError(); }
    if (bsfm::dev_alloc((void**)&w.etmp, (size_t)ld * sizeof(double)) != hipSuccess) return -1;
    if (bsfm::dev_alloc((void**)&w.bflags, (size_t)(w.nblk + 1) * sizeof(int)) != hipSuccess) return -1;
    if (const char* e = getenv("BSFM_SYRK_EVENTS")) w.syrk_events = std::max(0, atoi(e));
    (void)hipEventCreate(&w.ev0); (void)hipEventCreate(&w.ev1);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf_diag), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(DG_LDS_DOUBLES * sizeof(double))) != hipSuccess) return -1;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf_solve_one), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(DG_LDS_DOUBLES * sizeof(double))) != hipSuccess) return -1;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_tile32<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(T32_LDS_DOUBLES * sizeof(double))) != hipSuccess) return -1;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_tile32<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(T32_LDS_DOUBLES * sizeof(double))) != hipSuccess) return -1;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_narrow_tiles<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(T32_LDS_DOUBLES * sizeof(double))) != hipSuccess) return -1;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_narrow_tiles<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(T32_LDS_DOUBLES * sizeof(double))) != hipSuccess) return -1;
    if (const char* e = getenv("BSFM_NARROW")) w.narrow = std::max(0, std::min(POTRF_NARROW, atoi(e)));
    w.sy0 = new hipEvent_t[w.nblk]; w.sy1 = new hipEvent_t[w.nblk]; w.sy_flops = new double[w.nblk];
    for (int i = 0; i < w.nblk; ++i) { (void)hipEventCreate(&w.sy0[i]); (void)hipEventCreate(&w.sy1[i]); }
    if (const char* e = getenv("BSFM_SYRK_NT")) w.syrk_nt = atoi(e) != 0;
    if (getenv("BSFM_DEBUG_DIAG")) { (void)hipMalloc((void**)&w.dbg, 8 * sizeof(long long)); }
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
    if (w.ev0 && w.timing) (void)hipEventRecord(w.ev0, st);
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
    if (nblk > POTRF_MAX_TILES && !w.use_flow) {      // (the dataflow path runs its backward substitution in waves of POTRF_MAX_TILES columns)
        fprintf(stderr, "[bsfm] reduced camera system of order %d exceeds the %d-tile residency limit of the persistent "
                        "backward substitution\n", n, POTRF_MAX_TILES);
        return -1;
    }
    w.sy_used = 0;
    if (nblk == 1 && w.use_flow) {      // one tile: everything in one launch, the tile factorisation of the dataflow kernel (27 us; the round-2 one below: 42)
        const int rc = flow_solve_one(w, S, ld, n, E, x_out, d_info, st);
        if (w.ev1 && w.timing) (void)hipEventRecord(w.ev1, st);
        return rc;
    }
    if (nblk == 1) {         // (BSFM_CHOL=streams: the A/B reference)
        hipLaunchKernelGGL(k_potrf_solve_one, dim3(1), dim3(512), DG_LDS_DOUBLES * sizeof(double), st, S, ld, n, w.linv, d_info, E, x_out);
        if (w.ev1 && w.timing) (void)hipEventRecord(w.ev1, st);
        return 0;
    }
    if (w.use_flow) {
        const int rc = flow_solve_dispatch(w, S, ld, n, E, x_out, d_info, st);
        if (w.ev1 && w.timing) (void)hipEventRecord(w.ev1, st);
        return rc;
    }
    const size_t lds_bytes = 2 * 128 * GEMM_LDS_STRIDE * sizeof(double);
    (void)hipMemsetAsync(w.etmp, 0, (size_t)ld * sizeof(double), st);
    (void)hipMemcpyAsync(w.etmp, E, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st);
    // Lookahead schedule on three streams.  With L = the factor, P_a = tile (k+1+a, k) of panel k:
    //   st (chain) : diag(k) -> P_0 = S_{k+1,k} inv(L_kk)^T -> S_{k+1,k+1} -= P_0 P_0^T -> diag(k+1) ...
    //                the serial critical path touches only ONE tile of the panel and ONE tile of the next column;
    //   sd (side)  : the other panel tiles P_1.. (+ y_k of the fused forward substitution), then the rest of the first
    //                trailing column S_{k+1+a,k+1} -= P_a P_0^T (+ E -= P y_k): runs while diag(k+1) is busy;
    //   s2 (bulk)  : S_ij -= P_a P_b^T for the columns past k+1, needs the whole panel k      (the MFMA bulk).
    // The compact panel copies live in a ring of 4 buffers (k & 3): panel k+1 is written while the bulk still reads k.
    const size_t pstride = std::max<size_t>(1, (size_t)(w.nblk - 1)) * POTRF_NB * POTRF_NB;
    const size_t diag_lds = DG_LDS_DOUBLES * sizeof(double);
    const size_t lds64 = (64 + 128) * G64_STRIDE * sizeof(double);
    const size_t lds32 = T32_LDS_DOUBLES * sizeof(double);
    const size_t tl = (size_t)POTRF_NB * POTRF_NB;
    const double tile_flops = 2.0 * POTRF_NB * POTRF_NB * POTRF_NB;
    auto panel_of = [&](int k) { return w.panel + (size_t)(k & 3) * pstride; };
    (void)hipEventRecord(w.evU[w.nblk], st);                 // everything queued before the solve (S, E ready)
    (void)hipStreamWaitEvent(w.s2, w.evU[w.nblk], 0);
    (void)hipStreamWaitEvent(w.sd, w.evU[w.nblk], 0);
    // Tile envelope (opt-in): step k only touches the env_rows[k] tile rows below k that can hold a non-zero of the factor.
    const bool env = (int)w.env_rows.size() >= nblk && w.d_last != nullptr;
    auto rows_below = [&](int k) { return env ? std::min(w.env_rows[k], nblk - k - 1) : nblk - k - 1; };
    hipLaunchKernelGGL(k_potrf_diag, dim3(1), dim3(512), diag_lds, st, S, ld, 0, n, w.linv, d_info, w.dbg);
    for (int k = 0; k + 1 < nblk; ++k) {
        const int T = rows_below(k);                         // tile rows below the diagonal tile k (all of them unless an envelope is set)
        const int Tprev = k > 0 ? rows_below(k - 1) : 0;
        double* pk = panel_of(k);
        const double* Lk = w.linv + (size_t)k * tl;
        // what panel k-1 still owes tile (k+1, k+1): its tile 1 = row k+1 (the bulk launches leave that one tile to the chain)
        const double* owed = (k > 0 && Tprev >= 2) ? (const double*)panel_of(k - 1) + tl : (const double*)nullptr;
        const bool prev_narrow = k > 0 && Tprev <= w.narrow;
        if (prev_narrow) owed = nullptr;                     // a narrow step applies its panel to EVERY trailing tile: nothing is owed
        if (T <= w.narrow) {
            // ---- narrow step: chain stream only.  Whatever the previous (wide) step left on the other streams must have landed first.
            if (k > 0 && !prev_narrow) { (void)hipStreamWaitEvent(st, w.evC[k - 1], 0); (void)hipStreamWaitEvent(st, w.evU[k - 1], 0); }
            if (owed)      // (the wide step before launched its bulk over ALL its tiles when it knew this step to be narrow: see below)
                owed = nullptr;
            hipLaunchKernelGGL(k_narrow_tiles<0>, dim3(T * 16 + 1), dim3(256), lds32, st, S, ld, k, T, Lk, pk, w.etmp, w.y);
            if (T > 0)
                hipLaunchKernelGGL(k_narrow_tiles<1>, dim3(T * (T + 1) / 2 * 16 + 2 * T), dim3(256), lds32, st, S, ld, k, T, Lk, pk, w.etmp, w.y);
            const bool next_wide = k + 2 < nblk && rows_below(k + 1) > w.narrow;
            if (next_wide || k + 2 >= nblk) {                // only a following wide step (or the tail of the solve) looks at this step's events
                (void)hipEventRecord(w.evT[k], st); (void)hipEventRecord(w.evP[k], st);
                (void)hipEventRecord(w.evC[k], st); (void)hipEventRecord(w.evU[k], st);
            }
            hipLaunchKernelGGL(k_potrf_diag, dim3(1), dim3(512), diag_lds, st, S, ld, k + 1, n, w.linv, d_info, w.dbg);
            continue;
        }
        const bool next_narrow = k + 2 < nblk && rows_below(k + 1) <= w.narrow;
        // chain: first panel tile (its column k was completed by the side stream of step k-1)
        if (k > 0) (void)hipStreamWaitEvent(st, w.evC[k - 1], 0);
        hipLaunchKernelGGL(k_chain_tile32<0>, dim3(16), dim3(256), lds32, st, S, ld, k, Lk, pk, (const double*)nullptr, (const double*)nullptr);
        (void)hipEventRecord(w.evT[k], st);
        // side: rest of the panel and y_k (the extra workgroup), then the rest of the first trailing column
        (void)hipStreamWaitEvent(w.sd, w.evT[k], 0);
        hipLaunchKernelGGL(k_trsm_panel64, dim3(2 * (T - 1) + 2), dim3(256), lds64, w.sd, S, ld, k, Lk, pk, 1, 2 * (T - 1), w.etmp, w.y);
        (void)hipEventRecord(w.evP[k], w.sd);
        if (k > 0) (void)hipStreamWaitEvent(w.sd, w.evU[k - 1], 0);   // column k+1 was last written by the bulk of step k-1
        hipLaunchKernelGGL(k_syrk_col64, dim3(2 * (T - 1) + T), dim3(256), lds64, w.sd, S, ld, k, pk, 1, 2 * (T - 1), w.etmp, w.y, 0);
        (void)hipEventRecord(w.evC[k], w.sd);
        // bulk
        (void)hipStreamWaitEvent(w.s2, w.evP[k], 0);
        if (T > 2 || (next_narrow && T == 2)) {
            const double* pprev = nullptr;
            const bool timed = w.timing && w.syrk_events > 0 && (k % w.syrk_events) == 0;
            if (timed) (void)hipEventRecord(w.sy0[w.sy_used], w.s2);
            // part 2 leaves tile (k+2, k+2) to the chain's k_chain_tile32<1> of the next step; a NARROW next step has no such kernel: part 3
            const int ntile = T * (T - 1) / 2 - (next_narrow ? 0 : 1);
            const dim3 bg(ntile);
            if (w.syrk_nt) hipLaunchKernelGGL(k_syrk_update<true>, bg, dim3(512), lds_bytes, w.s2, S, ld, k, pk, next_narrow ? 3 : 2, pprev);
            else hipLaunchKernelGGL(k_syrk_update<false>, bg, dim3(512), lds_bytes, w.s2, S, ld, k, pk, next_narrow ? 3 : 2, pprev);
            if (timed) { (void)hipEventRecord(w.sy1[w.sy_used], w.s2); w.sy_flops[w.sy_used++] = tile_flops * ntile; }
        }
        (void)hipEventRecord(w.evU[k], w.s2);
        // chain: next diagonal tile
        // S_{k+1,k+1} takes panels k-1 and k here; the last bulk launch that touched it is the one of step k-2, and that one
        // is already ordered before this point: the chain waited for evC[k-1] above, and the side stream recorded it after
        // having waited for evU[k-2] itself.  (Tried: hipStreamWriteValue32 / hipStreamWaitValue32 on signal memory instead
        // of the chain <-> side events: no faster.)
        hipLaunchKernelGGL(k_chain_tile32<1>, dim3(10), dim3(256), lds32, st, S, ld, k, Lk, pk,
                           owed ? owed - tl : (const double*)nullptr, (const double*)nullptr);
        hipLaunchKernelGGL(k_potrf_diag, dim3(1), dim3(512), diag_lds, st, S, ld, k + 1, n, w.linv, d_info, w.dbg);
    }
    // y of the last tile: E_last is final once the side stream has drained
    if (nblk > 1) { (void)hipStreamWaitEvent(st, w.evC[nblk - 2], 0); (void)hipStreamWaitEvent(st, w.evU[nblk - 2], 0); }
    hipLaunchKernelGGL(k_fwd_last, dim3(1), dim3(256), 0, st, w.linv + (size_t)(nblk - 1) * tl,
                       w.etmp + (size_t)(nblk - 1) * POTRF_NB, w.y + (size_t)(nblk - 1) * POTRF_NB);
    // persistent backward substitution: all nblk workgroups must be resident (one per tile column)
    (void)hipMemsetAsync(w.bflags, 0, (size_t)(w.nblk + 1) * sizeof(int), st);
    hipLaunchKernelGGL(k_bwd_persistent, dim3(nblk), dim3(256), 0, st, S, ld, nblk, w.linv, w.y, w.xs, w.bflags, w.bflags + w.nblk,
                       (const int*)(env ? w.d_last : nullptr));
    hipLaunchKernelGGL(k_fold_timeout, dim3(1), dim3(1), 0, st, (const int*)(w.bflags + w.nblk), d_info);     // a hand-off that never arrived must not pass as a solution
    (void)hipMemcpyAsync(x_out, w.xs, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st);
    if (w.ev1 && w.timing) (void)hipEventRecord(w.ev1, st);
    if (w.dbg) {
        long long h[8]; (void)hipStreamSynchronize(st); (void)hipMemcpy(h, w.dbg, sizeof(h), hipMemcpyDeviceToHost);
        fprintf(stderr, "[bsfm] diag tile stamps (100 MHz ticks): load %lld, phaseA %lld (A1 %lld A2 %lld A3 %lld), store %lld, phaseB %lld\n", h[0], h[1], h[4], h[5], h[6], h[2], h[3]);
    }
    return 0;
}

inline void potrf_collect_time(PotrfWorkspace& w)
{
    float ms = 0.f;
    if (!w.timing) { w.sy_used = 0; return; }
    if (w.ev0 && w.ev1 && hipEventElapsedTime(&ms, w.ev0, w.ev1) == hipSuccess && ms >= 0.f) { w.ms += ms; w.cnt++; }
    for (int i = 0; i < w.sy_used; ++i)
        if (hipEventElapsedTime(&ms, w.sy0[i], w.sy1[i]) == hipSuccess && ms >= 0.f) { w.syrk_ms += ms; w.syrk_cnt++; w.syrk_flops += w.sy_flops[i]; }
    w.sy_used = 0;
}

}  // namespace bsfm
