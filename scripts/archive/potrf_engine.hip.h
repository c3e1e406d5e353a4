// potrf_engine.hip.h -- the serial chain of the tiled Cholesky as TWO PERSISTENT KERNELS with in-kernel flags (round 2).
//
// Round 1's schedule (potrf.hip.h:potrf_solve) drives diag(k) -> panel(k) -> first trailing column -> diag(k+1) with five kernel
// launches per tile column on two streams; its chain period (~86 us per column at config 3: 50 us diagonal tile incl. the explicit
// inverse, two 8-14 us single-tile kernels, ~7 us of cross-stream event latency at each of three hand-offs) bounds the last ~30
// tile columns and the small reduced systems of incremental Bundler outright.  Here the chain never leaves the GPU:
//
//   k_engine_diag   ONE workgroup (512 threads, tile in LDS): for k = 0 .. nblk-1: wait for tile (k,k) -> factor it; every block
//                   column s of the factor is PUBLISHED as soon as it is final (write-through stores + flag, diag_tile_body<true>),
//                   so the panel runs one block column behind the factorisation; then off the critical path: the explicit inverse
//                   (only the backward substitution needs it now) and y_k = inv(L_kk) E_k of the fused forward substitution.
//   k_engine_panel  W workgroups (512 threads).  Worker w owns the tile ROWS i = w+1, w+1+W, ... and walks each row through
//                   TRSM(i,k): L_ik = S_ik L_kk^-T by block substitution against the published block columns (accumulators hold the
//                              tile; L_ik goes to S and to the compact panel copy the bulk kernel reads),
//                   UPD(i,k+1): S_i,k+1 -= L_ik L_k+1,k^T  (the first trailing column; for i = k+1 this is the next diagonal tile),
//                   EUP(i,k):  E_i -= L_ik y_k,
//                   picking, among its rows, the lowest one whose next step is ready (a row near the diagonal is on the chain).
//   bulk stream     per column k:  k_wait_flag(panel k complete) -> k_syrk_update (columns >= k+2) -> k_set_flag(bulk_done = k+1):
//                   the MFMA bulk stays an ordinary launch, gated by two one-thread kernels instead of stream events.
// Hand-offs follow the guide's recipe (cdna_hip_programming.md, Guideline 16): producer stores -> every storing wave drains ->
// barrier -> one lane: agent-scope release (not needed after write-through stores) -> relaxed agent flag store; consumer: ONE lane
// polls relaxed with s_sleep -> ONE agent-scope acquire -> barrier -> plain loads.  Every spin is bounded: on a time-out the
// kernels leave, info becomes POTRF_INFO_TIMEOUT and the LM driver fails loudly.  All flags are zeroed by a memset node ahead of
// the launches.  Summation order per tile is fixed (panels in ascending k), so results are bit-identical from run to run.
#pragma once
#include "potrf.hip.h"

namespace bsfm {

constexpr int POTRF_INFO_TIMEOUT = -2;
constexpr unsigned ENGINE_SPIN_LIMIT = 1u << 23;        // x (one relaxed load + s_sleep): seconds; a healthy hand-off takes microseconds

struct EngineFlags {
    int* lflag;        // 8 * nblk: block column s of factor tile k published
    int* tile_ready;   // nblk: diagonal tile (k,k) has received every update
    int* p_ready;      // nblk: number of panel tiles of column k completed
    int* p0_ready;     // nblk: the first panel tile (row k+1) of column k is in the compact panel
    int* yflag;        // nblk: y_k published
    int* e_ready;      // nblk: E_k has received every update
    int* bulk_done;    // 1: number of bulk launches completed
    int* timeout;      // 1
};

__device__ __forceinline__ int ld_flag(const int* f) { return __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_flag(int* f, int v) { __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One lane waits until *f >= target; returns false after the spin limit (and raises the time-out word).
__device__ __forceinline__ bool spin_ge(const int* f, int target, int* timeout)
{
    unsigned spins = 0;
    while (ld_flag(f) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > ENGINE_SPIN_LIMIT || ((spins & 1023u) == 0 && ld_flag(timeout) != 0)) { atomicExch(timeout, 1); return false; }
    }
    return true;
}

// Workgroup-wide wait: lane 0 polls, one agent-scope acquire, barrier.  `ok` lives in LDS.
__device__ __forceinline__ bool wg_wait_ge(const int* f, int target, int* timeout, int* ok)
{
    if (threadIdx.x == 0) {
        const bool good = spin_ge(f, target, timeout);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *ok = good ? 1 : 0;
    }
    __syncthreads();
    const bool r = *ok != 0;
    __syncthreads();
    return r;
}

// Publish after PLAIN stores by the whole workgroup: drain, barrier, one release, then the flag operation by lane 0.
__device__ __forceinline__ void wg_release()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // restated: the compiler may drop the fence's own wait (guide, pitfall 12)
    }
}

__global__ void k_wait_flag(const int* flag, int target, int* timeout)
{
    (void)spin_ge(flag, target, timeout);
}
__global__ void k_set_flag(int* flag, int value)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    st_flag(flag, value);
}

// a time-out anywhere in the engine becomes the solve's info code (the LM driver treats it as a fatal error)
__global__ void k_engine_finish(const int* timeout, int* info)
{
    if (threadIdx.x == 0 && ld_flag(timeout) != 0) *info = POTRF_INFO_TIMEOUT;
}

// ------------------------------------------------------------------------------------------------ diagonal-tile engine
__global__ __launch_bounds__(512) void k_engine_diag(double* __restrict__ S, int ld, int nblk, int n_total, double* __restrict__ Linv,
        double* __restrict__ dinv, const double* E, double* y, int* __restrict__ info, EngineFlags F, long long* __restrict__ dbg)
{
    extern __shared__ __attribute__((aligned(16))) double dlds[];
    __shared__ int ok;
    __shared__ double yred[4 * POTRF_NB];
    for (int k = 0; k < nblk; ++k) {
        if (k > 0 && !wg_wait_ge(&F.tile_ready[k], 1, F.timeout, &ok)) break;
        if (dbg && threadIdx.x == 0) dbg[8 * k + 0] = wall_clock64();
        diag_tile_body<true>(dlds, S, ld, k, n_total, Linv, info, nullptr, dinv, F.lflag);
        if (dbg && threadIdx.x == 0) dbg[8 * k + 2] = wall_clock64();
        // ---- y_k = inv(L_kk) E_k (fused forward substitution): E_k is final once the worker of row k has applied y_0 .. y_k-1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this workgroup's own stores of inv(L_kk) ...
        __syncthreads();
        if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // ... are re-read below by other lanes: drop stale L1 lines
        __syncthreads();
        if (k > 0 && !wg_wait_ge(&F.e_ready[k], 1, F.timeout, &ok)) break;
        {
            double* vec = dlds;                                 // the LDS tile is free again (its head is re-used here)
            const int r = threadIdx.x & 127, h = threadIdx.x >> 7;          // 4 quarter rows of 32 columns each
            __syncthreads();
            if (threadIdx.x < POTRF_NB) vec[threadIdx.x] = E[(size_t)k * POTRF_NB + threadIdx.x];
            __syncthreads();
            const double* Li = Linv + (size_t)k * POTRF_NB * POTRF_NB + (size_t)r * POTRF_NB + 32 * h;
            double sacc = 0.0;
#pragma unroll 8
            for (int c = 0; c < 32; ++c) sacc += Li[c] * vec[32 * h + c];
            yred[h * POTRF_NB + r] = sacc;
            __syncthreads();
            if (threadIdx.x < POTRF_NB)
                __hip_atomic_store(&y[(size_t)k * POTRF_NB + r], (yred[r] + yred[POTRF_NB + r]) + (yred[2 * POTRF_NB + r] + yred[3 * POTRF_NB + r]),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) st_flag(&F.yflag[k], 1);
            if (dbg && threadIdx.x == 0) dbg[8 * k + 3] = wall_clock64();
        }
    }
}

// ------------------------------------------------------------------------------------------------ panel workers
constexpr int EP_STRIDE = 18;                                   // LDS row stride of the 128 x 16 operand slabs (conflict-free fragments)
constexpr int EP_LDS_DOUBLES = 2 * 128 * GEMM_LDS_STRIDE + 128 * EP_STRIDE + 16 * EP_STRIDE;

// L_ik = S_ik L_kk^-T for one tile by right-looking block substitution, one block column s per published flag:
//   X_s <- X_s inv(L_ss)^T ;  X_J -= X_s L_Js^T  (J > s).
// The tile lives in the accumulators (wave (wr, wc) holds rows wr .. wr+31, columns wc .. wc+63 as acc[t][u]: row wr + 4 t + (lane >> 4),
// column wc + 16 u + (lane & 15)).  Finished block columns are written to S (the factor) and to the compact panel copy `P`.
__device__ __forceinline__ bool engine_trsm(double* __restrict__ Sik, int ld, double* __restrict__ P, const double* Lkk, const double* dinv_k,
                                            const int* lflag_k, int* timeout, double* __restrict__ lds, int* ok)
{
    double* As = lds;                                            // -X_s, 128 x 16 (stride GEMM_LDS_STRIDE)
    double* Bs = lds + 128 * GEMM_LDS_STRIDE;                    // block column s of L_kk, rows = output columns
    double* Xr = lds + 2 * 128 * GEMM_LDS_STRIDE;                // raw X_s, 128 x 16 (stride EP_STRIDE)
    double* Ds = Xr + 128 * EP_STRIDE;                           // inv(L_ss), 16 x 16 (stride EP_STRIDE)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 64;
    double acc[8][4];
    {
        const double* cp = Sik + (size_t)(wr + (lane >> 4)) * ld + wc + (lane & 15);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[q][u] = cp[16 * u];
            cp += 4 * (size_t)ld;
        }
    }
    for (int s = 0; s < 8; ++s) {
        if (!wg_wait_ge(&lflag_k[s], 1, timeout, ok)) return false;
        // stage inv(L_ss) and block column s of L_kk (rows 16 (s+1) .. 127; rows above stay zero: those output columns are final)
        for (int idx = tid; idx < 128 * 16; idx += 512) {
            const int r = idx >> 4, c = idx & 15;
            Bs[r * GEMM_LDS_STRIDE + c] = r >= 16 * (s + 1) ? Lkk[(size_t)r * ld + 16 * s + c] : 0.0;
        }
        if (tid < 256) Ds[(tid >> 4) * EP_STRIDE + (tid & 15)] = dinv_k[s * 256 + tid];
        // the waves that hold column block s park it
        if ((wave & 1) == (s >> 2)) {
            const int u = s & 3;
#pragma unroll
            for (int q = 0; q < 8; ++q) Xr[(wr + 4 * q + (lane >> 4)) * EP_STRIDE + (lane & 15)] = acc[q][u];
        }
        __syncthreads();
        // X_s <- X_s inv(L_ss)^T: wave w takes the 16 rows 16 w .. 16 w + 15
        {
            double res[4] = { 0.0, 0.0, 0.0, 0.0 };
            mma16_nt(res, Xr + (16 * wave) * EP_STRIDE, EP_STRIDE, Ds, EP_STRIDE, lane);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int r = 16 * wave + 4 * a + (lane >> 4), c = lane & 15;
                As[r * GEMM_LDS_STRIDE + c] = -res[a];
                Sik[(size_t)r * ld + 16 * s + c] = res[a];
                P[(size_t)r * POTRF_NB + 16 * s + c] = res[a];
            }
        }
        __syncthreads();
        // X_J -= X_s L_Js^T for the column blocks J > s this wave holds
        if (s < 7) {
#pragma unroll
            for (int kk = 0; kk < 16; kk += 4) {
                double b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) b[u] = Bs[(wc + 16 * u + (lane & 15)) * GEMM_LDS_STRIDE + kk + (lane >> 4)];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const double a = As[(wr + 4 * t + (lane & 3)) * GEMM_LDS_STRIDE + kk + (lane >> 4)];
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[t][u] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b[u], acc[t][u], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    return true;
}

// C -= A B^T for one 128 x 128 tile (A, B: compact panel tiles, row stride 128), C in S.
__device__ __forceinline__ void engine_update(double* __restrict__ Cij, int ld, const double* __restrict__ A, const double* __restrict__ B,
                                              double* __restrict__ lds)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 64;
    double acc[8][4];
    {
        const double* cp = Cij + (size_t)(wr + (lane >> 4)) * ld + wc + (lane & 15);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[q][u] = cp[16 * u];
            cp += 4 * (size_t)ld;
        }
    }
    gemm_nt_128<true>(A, POTRF_NB, B, POTRF_NB, POTRF_NB, lds, acc);
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    double* Sl = Cij + (size_t)(((tid2 >> 7) << 5) + ((tid2 & 63) >> 4)) * ld + (((tid2 >> 6) & 1) << 6) + (tid2 & 15);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int u = 0; u < 4; ++u) Sl[16 * u] = acc[q][u];
        Sl += 4 * (size_t)ld;
    }
}

__global__ __launch_bounds__(512, 4) void k_engine_panel(double* __restrict__ S, int ld, int nblk, int W, double* __restrict__ panel,
        size_t pstride, const double* __restrict__ dinv, double* E, const double* y, EngineFlags F, long long* __restrict__ dbg)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int MAXROWS = 64;
    // progress of this worker's rows i = w + 1 + W q, kept by lane 0: next tile step (column tstep, phase 0 = TRSM / 1 = UPD) and
    // next forward-substitution update
    __shared__ int ok, pick_row, pick_op, pick_k, nrows_s, tstep[MAXROWS], tphase[MAXROWS], estep[MAXROWS];
    const int w = blockIdx.x;
    constexpr size_t TL = (size_t)POTRF_NB * POTRF_NB;
    if (threadIdx.x == 0) {
        int nrows = 0;
        for (int i = w + 1; i < nblk && nrows < MAXROWS; i += W) { tstep[nrows] = 0; tphase[nrows] = 0; estep[nrows] = 0; ++nrows; }
        nrows_s = nrows;
    }
    __syncthreads();
    unsigned idle = 0;
    for (;;) {
        // ---- pick: lane 0 scans the rows from the diagonal outwards (lowest row first = closest to the critical path)
        if (threadIdx.x == 0) {
            int row = -1, op = -1, kk = 0, left = 0;
            const int nrows = nrows_s;
            const int bd = ld_flag(F.bulk_done);
            for (int q = 0; q < nrows; ++q) {
                const int i = w + 1 + W * q;
                if (tstep[q] < i || estep[q] < i) ++left;
                if (row < 0 && tstep[q] < i) {
                    const int k = tstep[q];
                    if (tphase[q] == 0) { if (ld_flag(&F.lflag[8 * k]) >= 1) { row = q; op = 0; kk = k; } }
                    else if (bd >= k && (i == k + 1 || ld_flag(&F.p0_ready[k]) >= 1)) { row = q; op = 1; kk = k; }
                }
            }
            for (int q = 0; q < nrows && row < 0; ++q) {          // nothing on the tile side: a forward-substitution update
                const int i = w + 1 + W * q;
                const int k = estep[q];
                if (k < i && k < tstep[q] && ld_flag(&F.yflag[k]) >= 1) { row = q; op = 2; kk = k; }
            }
            if (row >= 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            pick_row = left == 0 ? -2 : row; pick_op = op; pick_k = kk;
        }
        __syncthreads();
        const int q = pick_row, op = pick_op, k = pick_k;
        __syncthreads();
        if (q == -2) return;                                     // every row of this worker has reached the diagonal
        if (q < 0) {
            __builtin_amdgcn_s_sleep(4);
            if (++idle > ENGINE_SPIN_LIMIT || ((idle & 255u) == 0 && ld_flag(F.timeout) != 0)) { if (threadIdx.x == 0) atomicExch(F.timeout, 1); return; }
            continue;
        }
        idle = 0;
        const int i = w + 1 + W * q;
        if (dbg && threadIdx.x == 0 && i == k + 1) dbg[8 * k + (op == 0 ? 4 : 6)] = op == 2 ? dbg[8 * k + 6] : wall_clock64();
        if (op == 0) {                                           // TRSM(i, k)
            double* Pk = panel + (size_t)(k & 3) * pstride + (size_t)(i - k - 1) * TL;
            if (!engine_trsm(S + ((size_t)i * POTRF_NB) * ld + (size_t)k * POTRF_NB, ld, Pk, S + ((size_t)k * POTRF_NB) * ld + (size_t)k * POTRF_NB,
                             dinv + (size_t)k * 2048, F.lflag + 8 * k, F.timeout, lds, &ok)) return;
            wg_release();
            if (threadIdx.x == 0) {
                if (i == k + 1) st_flag(&F.p0_ready[k], 1);
                atomicAdd(&F.p_ready[k], 1);
                tphase[q] = 1;
                if (dbg && i == k + 1) dbg[8 * k + 5] = wall_clock64();
            }
        } else if (op == 1) {                                    // UPD(i, k+1): first trailing column (the next diagonal tile for i = k+1)
            const double* Pk = panel + (size_t)(k & 3) * pstride;
            engine_update(S + ((size_t)i * POTRF_NB) * ld + (size_t)(k + 1) * POTRF_NB, ld, Pk + (size_t)(i - k - 1) * TL, Pk, lds);
            wg_release();
            if (threadIdx.x == 0) {
                if (i == k + 1) st_flag(&F.tile_ready[i], 1);
                tstep[q] = k + 1; tphase[q] = 0;
                if (dbg && i == k + 1) dbg[8 * k + 7] = wall_clock64();
            }
        } else {                                                 // EUP(i, k): E_i -= L_ik y_k  (L_ik from S: the compact copy is a ring)
            const int row = threadIdx.x >> 2, part = threadIdx.x & 3;
            const double* pr = S + ((size_t)i * POTRF_NB + row) * ld + (size_t)k * POTRF_NB + 32 * part;
            const double* yk = y + (size_t)k * POTRF_NB + 32 * part;
            double sacc = 0.0;
#pragma unroll 8
            for (int c = 0; c < 32; ++c) sacc += pr[c] * yk[c];
            sacc += __shfl_xor(sacc, 1, 64);
            sacc += __shfl_xor(sacc, 2, 64);
            if (part == 0) E[(size_t)i * POTRF_NB + row] -= sacc;
            if (k + 1 == i) wg_release();                        // E_i is final: the diagonal engine may form y_i
            if (threadIdx.x == 0) {
                if (k + 1 == i) st_flag(&F.e_ready[i], 1);
                estep[q] = k + 1;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ host
// Solves S x = E like potrf_solve (S destroyed, E preserved, info = 0 / dpotrf's k / POTRF_INFO_TIMEOUT).
inline int potrf_solve_engine(PotrfWorkspace& w, double* S, int ld, int n, const double* E, double* x_out, int* d_info, hipStream_t st)
{
    const int nblk = (n + POTRF_NB - 1) / POTRF_NB;
    if (!w.engine_attr_set) {        // 150 KB of dynamic LDS for the diagonal-tile engine (once per workspace)
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_engine_diag), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(DG_LDS_DOUBLES * sizeof(double))) != hipSuccess) return -1;
        w.engine_attr_set = 1;
    }
    if (w.ev0) (void)hipEventRecord(w.ev0, st);
    w.sy_used = 0;
    const size_t pstride = std::max<size_t>(1, (size_t)(w.nblk - 1)) * POTRF_NB * POTRF_NB;
    const double tile_flops = 2.0 * POTRF_NB * POTRF_NB * POTRF_NB;
    EngineFlags F;
    F.lflag = w.eflags; F.tile_ready = F.lflag + 8 * w.nblk; F.p_ready = F.tile_ready + w.nblk; F.p0_ready = F.p_ready + w.nblk;
    F.yflag = F.p0_ready + w.nblk; F.e_ready = F.yflag + w.nblk; F.bulk_done = F.e_ready + w.nblk; F.timeout = F.bulk_done + 1;
    (void)hipMemsetAsync(w.eflags, 0, (size_t)(13 * w.nblk + 8) * sizeof(int), st);
    (void)hipMemsetAsync(w.etmp, 0, (size_t)ld * sizeof(double), st);
    (void)hipMemcpyAsync(w.etmp, E, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st);
    (void)hipEventRecord(w.evU[w.nblk], st);                 // S, E and the cleared flags are ready
    (void)hipStreamWaitEvent(w.s2, w.evU[w.nblk], 0);
    (void)hipStreamWaitEvent(w.sd, w.evU[w.nblk], 0);
    const int W = std::max(1, std::min(w.engine_workers, nblk - 1));
    const size_t diag_lds = DG_LDS_DOUBLES * sizeof(double);
    const size_t panel_lds = EP_LDS_DOUBLES * sizeof(double);
    // the two persistent kernels first (they must be resident before the gated bulk launches fill the device)
    hipLaunchKernelGGL(k_engine_diag, dim3(1), dim3(512), diag_lds, st, S, ld, nblk, n, w.linv, w.dinv, (const double*)w.etmp, w.y, d_info, F, w.edbg);
    if (nblk > 1)
        hipLaunchKernelGGL(k_engine_panel, dim3(W), dim3(512), panel_lds, w.sd, S, ld, nblk, W, w.panel, pstride, (const double*)w.dinv,
                           w.etmp, (const double*)w.y, F, w.edbg);
    const size_t lds_bytes = 2 * 128 * GEMM_LDS_STRIDE * sizeof(double);
    for (int k = 0; k + 1 < nblk; ++k) {
        const int T = nblk - k - 1;                          // panel tiles of column k (rows k+1 .. nblk-1)
        hipLaunchKernelGGL(k_wait_flag, dim3(1), dim3(1), 0, w.s2, (const int*)(F.p_ready + k), T, F.timeout);
        if (T >= 2) {
            const int tiles = T * (T - 1) / 2;               // columns k+2 .. : every tile (i, j), i >= j, diagonal tiles included
            (void)hipEventRecord(w.sy0[w.sy_used], w.s2);
            if (w.syrk_nt) hipLaunchKernelGGL(k_syrk_update<true>, dim3(tiles), dim3(512), lds_bytes, w.s2, S, ld, k, w.panel + (size_t)(k & 3) * pstride, 3, (const double*)nullptr);
            else hipLaunchKernelGGL(k_syrk_update<false>, dim3(tiles), dim3(512), lds_bytes, w.s2, S, ld, k, w.panel + (size_t)(k & 3) * pstride, 3, (const double*)nullptr);
            (void)hipEventRecord(w.sy1[w.sy_used], w.s2);
            w.sy_flops[w.sy_used++] = tile_flops * tiles;
        }
        hipLaunchKernelGGL(k_set_flag, dim3(1), dim3(1), 0, w.s2, F.bulk_done, k + 1);
    }
    // everything has drained when the three streams meet again
    (void)hipEventRecord(w.evC[0], w.sd); (void)hipEventRecord(w.evU[0], w.s2);
    (void)hipStreamWaitEvent(st, w.evC[0], 0); (void)hipStreamWaitEvent(st, w.evU[0], 0);
    hipLaunchKernelGGL(k_engine_finish, dim3(1), dim3(64), 0, st, (const int*)F.timeout, d_info);
    (void)hipMemsetAsync(w.bflags, 0, (size_t)(w.nblk + 1) * sizeof(int), st);
    hipLaunchKernelGGL(k_bwd_persistent, dim3(nblk), dim3(256), 0, st, S, ld, nblk, w.linv, w.y, w.xs, w.bflags, w.bflags + w.nblk);
    (void)hipMemcpyAsync(x_out, w.xs, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st);
    if (w.ev1) (void)hipEventRecord(w.ev1, st);
    if (w.edbg) {        // BSFM_DEBUG_ENGINE=1: per-column stamps (100 MHz ticks) of the chain
        std::vector<long long> h((size_t)8 * w.nblk);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h.data(), w.edbg, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        const long long t0 = h[0];
        fprintf(stderr, "[bsfm] engine chain (us from the first diagonal tile): k: diag start | body end | y | P0 trsm start / done | next-diag update start / done\n");
        for (int k = 0; k < nblk; k += std::max(1, nblk / 24)) {
            auto us = [&](int q) { return h[8 * k + q] ? (h[8 * k + q] - t0) * 0.01 : -1.0; };
            fprintf(stderr, "  %3d: %9.1f | %9.1f | %9.1f | %9.1f / %9.1f | %9.1f / %9.1f\n", k, us(0), us(2), us(3), us(4), us(5), us(6), us(7));
        }
        (void)hipMemset(w.edbg, 0, h.size() * sizeof(long long));
    }
    return 0;
}

// Round 2 default: the panel engine; BSFM_CHOL=streams (or the rocSOLVER cross-check backend) keeps round 1's schedule.
}  // namespace bsfm
#include "potrf_blocked.hip.h"
namespace bsfm {

inline int potrf_solve_auto(PotrfWorkspace& w, double* S, int ld, int n, const double* E, double* x_out, int* d_info, hipStream_t st)
{
    if (n > 0 && w.backend == 0) {
        const char* e = getenv("BSFM_CHOL");
        if (e && !strcmp(e, "blocked")) {          // two-level blocking (potrf_blocked.hip.h), opt-in
            BlockedState& b = blocked_state(w);
            const int nblk = (n + POTRF_NB - 1) / POTRF_NB;
            if (!b.ready && blocked_init(w, b) != 0) return -1;
            if (nblk >= b.min_tiles && nblk <= POTRF_MAX_TILES) return potrf_solve_blocked(w, b, S, ld, n, E, x_out, d_info, st);
        }
    }
    if (n > 0 && w.backend == 0 && w.use_engine && (n + POTRF_NB - 1) / POTRF_NB <= POTRF_MAX_TILES)
        return potrf_solve_engine(w, S, ld, n, E, x_out, d_info, st);
    return potrf_solve(w, S, ld, n, E, x_out, d_info, st);
}

}  // namespace bsfm
