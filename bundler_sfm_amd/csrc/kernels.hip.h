// kernels.hip.h -- HIP kernels of the LM normal-equations path (gfx950 / CDNA4, wave64, FP64).
//
// Reference behaviour restated (paths relative to the reference tree):
//   residuals / cost          sba_motstr_Qs + nrmL2xmy      lib/sba-1.5/sba_levmar_wrap.c:73-104, sba_levmar.c:159-207
//   Jacobian                  sba_motstr_Qs_fdjac / projac  lib/sba-1.5/sba_levmar_wrap.c:163-259
//   U_j, ea_j                 lib/sba-1.5/sba_levmar.c:919-964
//   V_i, eb_i                 lib/sba-1.5/sba_levmar.c:987-1030
//   (V_i + mu I)^-1           lib/sba-1.5/sba_levmar.c:1138-1162 (sba_symat_invert_BK)
//   S_jk, e_j (Schur)         lib/sba-1.5/sba_levmar.c:1182-1339          (task kernel: schur.hip.h)
//   camera-only step          lib/sba-1.5/sba_levmar.c:2499-2513          (k_cam_solve, fix_points mode)
//   db_i (back-substitution)  lib/sba-1.5/sba_levmar.c:1393-1433
//   step norms, dL            lib/sba-1.5/sba_levmar.c:1444-1527
//   Snavely stop rule         lib/sba-1.5/sba_levmar.c:1552-1561
//
// Data layout in HBM (all FP64 / int32, observation order == the reference's CRS order, i.e. point-major,
// camera ascending within a point -- sba_levmar.c:653-663):
//   x[2*nvis]       measurements            obs_cam[nvis], obs_pt[nvis], rowptr[n+1]   point-major index
//   camptr[m+1], camobs[nvis]               camera-major secondary index (replaces sba_crsm_col_elmidxs'
//                                           per-call binary searches, sba_crsm.c:183-212)
//   camtab[m*72]    per-camera derived row  (model.hip.h)
//   campos[nvis], cam_pt[nvis], cam_cam[nvis]   observation -> camera-major position; position -> point / camera
//   EVERYTHING per observation lives in CAMERA-major order (position t = campos[k] of observation k; round 3):
//   xc[2*nvis]      the measurements, permuted once at problem_create;  e / hx[2*nvis]  residuals at p / at the trial point
//   Ac[nvis*2cnp]   A_ij as cnp 16-byte chunks (A[0][c], A[1][c]): the two image rows of a column side by side, which is
//                   what every consumer multiplies together (U_j = sum of chunk outer products, Yh = M * chunk, A da)
//   Bc[nvis*8]      64-byte record: B_ij (2 x 3 row-major) || e_ij -- the per-point kernels (V_i / eb_i) fetch exactly one
//                   64-byte sector per observation instead of 48 + 16 bytes out of two 192-byte-strided streams
//   Cc[nvis*8]      per solve attempt: C_ij = B_ij V*_i^-1 (2 x 3) || C_ij eb_i (k_schur_prep): the Schur tasks need no
//                   V*^-1 / eb gathers and no per-triple 3 x 3 work
//   The camera-side kernels (Jacobian, residual, U_j / ea_j, Schur tasks) stream or gather these by t; the point-side ones
//   (V_i / eb_i, back-substitution) gather records through campos[].
//   U[m*cnp*cnp], ea[m*cnp], V[n*6] (packed upper), Vinv[n*6], eb[n*3], S[ld*ld], E[ld]
// W_ij = A_ij^T B_ij is never materialised (1.08 GB at 5 M observations): every consumer uses the factored
// form, e.g. Y_ij W_ik^T = A_ij^T (C_ij B_ik^T) A_ik with a 2x2 core, and W_ij^T da = B_ij^T (A_ij da).
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include "model.hip.h"
#include "index_build.h"      // SchurTask, SCHUR_CHUNK, the device-side index construction

namespace bsfm {

struct DevProblem {
    ModelCfg cfg;
    int n, m, mcon, nvis;
    const double* x;              // measurements in observation (CRS) order: only the growing / shrinking of a problem reads it
    const double* xc;             // measurements in camera-major order
    const int* obs_cam; const int* obs_pt; const int* rowptr;
    const int* camptr; const int* camobs;
    const int* campos;            // obs k -> its position in camera-major order (inverse of camobs)
    const int* cam_pt;            // camera-major position -> point index
    const int* cam_cam;           // camera-major position -> camera index
    const double* Rinit; const double* finit;
    // constraints
    const unsigned char* ccon; const double* cval; const double* cw;   // m*cnp (may be null)
    const unsigned char* pcon; const double* pval; double pweight;     // n, 3n (may be null)
    double nvis_global;
    // work arrays
    double* Ac; double* Bc; double* Cc; double* U; double* ea; double* V; double* Vinv; double* eb;
};

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
// value of lane Q of every quad (DPP quad_perm [Q, Q, Q, Q]): two 32-bit moves, no LDS
template <int Q>
__device__ __forceinline__ double quad_bcast_d(double v)
{
    constexpr int ctrl = Q | (Q << 2) | (Q << 4) | (Q << 6);
    // mov_dpp with bound_ctrl: every lane of a quad permutation has a valid source, so there is no "old" value to keep -- the
    // update_dpp(0, ...) form cost one extra v_mov_b32 per half (24 per pass of the Schur kernels)
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), ctrl, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), ctrl, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_down(v, o, 64); v = (t > v) ? t : v; }
    return v;
}

__device__ __forceinline__ double block_max4(double s, double* sm)      // result valid in thread 0
{
    { double t;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { t = __shfl_down(s, o, 64); s = t > s ? t : s; } }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    double q = sm[0];
    for (int i = 1; i < 4; ++i) q = sm[i] > q ? sm[i] : q;
    return q;
}
__device__ __forceinline__ double block_sum4(double s, double* sm)
{
    s = wave_sum(s);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// ---------------------------------------------------------------------------------------------------
// "The last workgroup finishes the job" (round 5): the second-stage reductions -- one-block kernels of a few microseconds that only
// existed to wait for every workgroup of the kernel before them -- run inside that kernel, in the workgroup that arrives last.  An
// iteration of a Bundler-sized problem (14 - 50 cameras, src/BundleFast.cpp:263-438 calls run_sfm hundreds of times at that size) is
// nothing but launch latencies: ~25 launches of 3.5 - 16 us.  Same partitions, same order of the sums, executed by one workgroup: the
// values are bit-identical to the two-launch form (which remains for ticket == nullptr).
// Visibility without fences (as in chol_flow.hip.h): the partial results are written with agent-scope (sc1) stores, every storing
// thread drains them (s_waitcnt vmcnt(0)), the workgroup synchronises, ONE thread takes a ticket with an agent-scope atomic; the
// workgroup that draws the last ticket reads the partials with agent-scope loads and puts the ticket back to zero for the next launch.
__device__ __forceinline__ double ld_agent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Two levels for big grids: atomics on ONE address are served one at a time (measured on the headline's k_residual, 9 000 workgroups:
// one ticket word made the kernel 261 us instead of 111 -- about 17 ns per ticket, more than the workgroups take to arrive), so
// workgroups draw from the word of their group of 64 (TICKET_STRIDE words apart: a 128-byte line each, different channels) and only
// the last of a group goes on to the top-level word.
constexpr int TICKET_GROUP = 64, TICKET_STRIDE = 32;
__host__ __device__ constexpr size_t ticket_group_words(size_t grid) { return ((grid + TICKET_GROUP - 1) / TICKET_GROUP) * TICKET_STRIDE; }
__device__ __forceinline__ bool last_block_arrives(unsigned* ticket, unsigned total, unsigned* groups = nullptr)
{
    __shared__ int s_last_block;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        bool last = false;
        if (!groups || total <= (unsigned)TICKET_GROUP) {
            const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = t == total - 1u;
            if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const unsigned g = blockIdx.x / (unsigned)TICKET_GROUP, ng = (total + TICKET_GROUP - 1u) / (unsigned)TICKET_GROUP;
            const unsigned in_group = min((unsigned)TICKET_GROUP, total - g * (unsigned)TICKET_GROUP);
            unsigned* gw = groups + (size_t)g * TICKET_STRIDE;
            const unsigned t = __hip_atomic_fetch_add(gw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == in_group - 1u) {
                __hip_atomic_store(gw, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned t2 = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = t2 == ng - 1u;
                if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        s_last_block = last ? 1 : 0;
    }
    __syncthreads();
    return s_last_block != 0;
}

// ---------------------------------------------------------------------------------------------------
// per-camera derived table (trig happens only here).  A row is FOUR independent parts -- the rotation with its derivative factor and
// the scalars; the three perturbed rotations of the forward differences -- and a thread computes one part (round 6: one thread per camera
// spent ~20 us on its four Rodrigues formulas one after the other, on the critical path of every solve attempt of a small problem:
// k_backsub's finishing workgroup builds the trial point's table).  Same functions on the same inputs: the values do not change.
constexpr int CT_PARTS = 4;
__device__ __forceinline__ void cam_table_part(const ModelCfg& cfg, int j, int part, const double* __restrict__ pa, const double* __restrict__ Rinit,
                            const double* __restrict__ finit, const double* __restrict__ known, int with_fd, double* __restrict__ camtab)
{
    const double* a = pa + (size_t)j * cfg.cnp;
    const double* R0 = Rinit + (size_t)j * 9;
    double* ct = camtab + (size_t)j * CT_STRIDE;
    double Ri[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) Ri[k] = R0[k];
    if (part == 0) {
        double R[9], Q[9];
        rot_update(Ri, a[3], a[4], a[5], R);
        rot_deriv_factor(Ri, R, a[3], a[4], a[5], Q);
#pragma unroll
        for (int k = 0; k < 9; ++k) { ct[CT_R + k] = R[k]; ct[CT_Q + k] = Q[k]; }
        for (int k = 0; k < 9; ++k) ct[CT_A + k] = (k < cfg.cnp) ? a[k] : 0.0;
        int col = 6;
        double f = finit[j];
        if (cfg.est_focal) { f = a[6] / cfg.f_scale; col = 7; }
        ct[CT_F] = f;
        ct[CT_K1] = cfg.undistort ? a[col] / cfg.k_scale : 0.0;
        ct[CT_K2] = cfg.undistort ? a[col + 1] / cfg.k_scale : 0.0;
        ct[30] = finit[j]; ct[31] = 0.0;
        for (int k = 0; k < CT_EXT; ++k) ct[CT_KN + k] = known ? known[(size_t)j * CT_EXT + k] : 0.0;   // extended-model block (model.hip.h)
        if (with_fd)
            for (int k = 0; k < 9; ++k) {
                double d = 0.0;
                if (k < cfg.cnp) { d = 1E-04 * a[k]; d = fabs(d); if (d < 1E-06) d = 1E-06; }
                ct[CT_D + k] = d;
            }
    } else if (with_fd) {
        const int k = part - 1;
        double d = 1E-04 * a[3 + k]; d = fabs(d); if (d < 1E-06) d = 1E-06;       // (= ct[CT_D + 3 + k]: cnp >= 6 always)
        double Rp[9];
        rot_update(Ri, k == 0 ? a[3] + d : a[3], k == 1 ? a[4] + d : a[4], k == 2 ? a[5] + d : a[5], Rp);
        for (int q = 0; q < 9; ++q) ct[CT_RP + 9 * k + q] = Rp[q];
    }
}
__global__ void k_cam_table(ModelCfg cfg, int m, const double* __restrict__ pa, const double* __restrict__ Rinit,
                            const double* __restrict__ finit, const double* __restrict__ known, int with_fd, double* __restrict__ camtab)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m * CT_PARTS) return;
    cam_table_part(cfg, t % m, t / m, pa, Rinit, finit, known, with_fd, camtab);      // (neighbouring lanes = neighbouring cameras, the SAME part: no divergence)
}

// ---------------------------------------------------------------------------------------------------
// residuals e = x - proj(p): one thread per CAMERA-major position t (round 3; it was one per observation in CRS order, where the 64
// lanes of a wave read ~10 different 576-byte camera-table rows -- 0.19 of the HBM roof): neighbouring threads share the camera, so
// the table row is one broadcast; xc / e are streamed, the point (24 bytes, L2-resident array) is gathered.  Block partial of
// sum e^2; with e_prev != null also the block partial of the Snavely pct-change maximum (element-wise, so the order is free).
constexpr int RES_BLOCK = 256;
template <bool KNOWN>
__global__ __launch_bounds__(RES_BLOCK) void k_residual(ModelCfg cfg, int nvis,
        const double* __restrict__ xc, const int* __restrict__ cam_cam, const double* __restrict__ ptc,
        const double* __restrict__ camtab,
        double* __restrict__ e_out, const double* __restrict__ e_prev, double eps5,
        double* __restrict__ part_cost, double* __restrict__ part_pct,
        unsigned* __restrict__ ticket /* null: k_reduce_sum_max follows */, unsigned* __restrict__ ticket_groups,
        double* __restrict__ out_sum, double* __restrict__ out_max)
{
    __shared__ double sm[2 * (RES_BLOCK / 64)];
    const int k = blockIdx.x * RES_BLOCK + threadIdx.x;
    double c = 0.0, pct = 0.0;
    if (k < nvis) {
        const double* ct = camtab + (size_t)cam_cam[k] * CT_STRIDE;
        // the point of observation k from the CAMERA-major mirror of the point array (round 5: streamed, 32 bytes; the gather of 24 bytes
        // out of the 12 MB point array fetched a 128-byte line per observation: 2.6 x the algorithmic traffic of this kernel)
        const double2 b01 = reinterpret_cast<const double2*>(ptc)[2 * (size_t)k], b2_ = reinterpret_cast<const double2*>(ptc)[2 * (size_t)k + 1];
        const double b[3] = { b01.x, b01.y, b2_.x };
        double h0, h1;
        project_row<KNOWN>(cfg, ct, b[0], b[1], b[2], h0, h1);
        const double2 xx = reinterpret_cast<const double2*>(xc)[k];
        const double e0 = xx.x - h0, e1 = xx.y - h1;
        reinterpret_cast<double2*>(e_out)[k] = make_double2(e0, e1);
        c = e0 * e0 + e1 * e1;
        if (e_prev) {
            const double2 eo = reinterpret_cast<const double2*>(e_prev)[k];
            // sba_levmar.c:1552-1561: skipped when both are (signed) below eps5
            if (!(eo.x < eps5 && e0 < eps5)) { const double q = fabs((eo.x - e0) / eo.x); if (q > pct) pct = q; }
            if (!(eo.y < eps5 && e1 < eps5)) { const double q = fabs((eo.y - e1) / eo.y); if (q > pct) pct = q; }
        }
    }
    c = wave_sum(c); pct = wave_max(pct);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sm[w] = c; sm[RES_BLOCK / 64 + w] = pct; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0, q = 0.0;
        for (int i = 0; i < RES_BLOCK / 64; ++i) { s += sm[i]; q = sm[RES_BLOCK / 64 + i] > q ? sm[RES_BLOCK / 64 + i] : q; }
        st_agent(part_cost + blockIdx.x, s);
        if (part_pct) st_agent(part_pct + blockIdx.x, q);
    }
    if (!ticket || !last_block_arrives(ticket, gridDim.x, ticket_groups)) return;
    {   // k_reduce_sum_max, by the workgroup that arrived last (same partition, same order)
        __shared__ double sm2[4];
        const int count = (int)gridDim.x;
        double s = 0.0, m = 0.0;
#pragma unroll 8
        for (int t = threadIdx.x; t < count; t += 256) { s += ld_agent(part_cost + t); if (part_pct) { const double v = ld_agent(part_pct + t); m = v > m ? v : m; } }
        const double rs = block_sum4(s, sm2), rm = block_max4(m, sm2);
        if (threadIdx.x == 0) { *out_sum = rs; if (part_pct) *out_max = rm; }
    }
}

// camera-major copy of a per-observation array of 16-byte items (measurements at problem_create; exports go the other way)
__global__ __launch_bounds__(256) void k_permute16(int nvis, const int* __restrict__ src_of, const double* __restrict__ src, double* __restrict__ dst)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < nvis) reinterpret_cast<double2*>(dst)[t] = reinterpret_cast<const double2*>(src)[src_of[t]];
}

// Camera-major mirror of the point part of a parameter vector: 32 bytes (x, y, z, 0) per observation.  Built by this gather when a
// parameter vector comes from outside (lm_begin, reset), kept up to date by k_backsub, which scatters every new trial point to the
// positions of its observations -- the projection kernels (residual, Jacobian) then STREAM their points.
__global__ __launch_bounds__(256) void k_point_mirror(int nvis, const int* __restrict__ cam_pt, const double* __restrict__ pb, double* __restrict__ ptc)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nvis) return;
    const double* b = pb + (size_t)cam_pt[t] * 3;
    reinterpret_cast<double2*>(ptc)[2 * (size_t)t] = make_double2(b[0], b[1]);
    reinterpret_cast<double2*>(ptc)[2 * (size_t)t + 1] = make_double2(b[2], 0.0);
}

// ---------------------------------------------------------------------------------------------------
// Jacobian: one thread per CAMERA-major position t.  Neighbouring threads share the camera (its 72-double table row is one
// broadcast), gather their point (24 bytes, the point array is L2-resident) and write consecutive records of the two streams
// Ac (cnp chunks (A[0][c], A[1][c])) and Bc (B || e, 64 bytes; e is copied from the residual array so that the per-point kernels
// find it in the same sector as B).
// (The forward-difference variant takes 164 VGPRs = three waves per SIMD.  Capped at 128 -- __launch_bounds__(256, 4) -- it spills 32
// registers and the phase goes from 0.38 to 0.51 ms at config 3: measured in round 4, not kept.)
template <int CNP, bool FD, bool KNOWN>
__global__ __launch_bounds__(256) void k_jacobian(ModelCfg cfg, int nvis,
        const int* __restrict__ cam_cam, const double* __restrict__ ptc,
        const double* __restrict__ camtab, const double* __restrict__ e,
        double* __restrict__ Ac, double* __restrict__ Bc)
{
    const int tt = blockIdx.x * 256 + threadIdx.x;
    const int t = min(tt, nvis - 1);                         // the last workgroup's surplus threads repeat the last record (never stored)
    const double* ct = camtab + (size_t)cam_cam[t] * CT_STRIDE;
    const double2 b01 = reinterpret_cast<const double2*>(ptc)[2 * (size_t)t], b2_ = reinterpret_cast<const double2*>(ptc)[2 * (size_t)t + 1];
    const double b[3] = { b01.x, b01.y, b2_.x };             // the point from the camera-major mirror (streamed; was a 24-byte gather)
    const double2 et = reinterpret_cast<const double2*>(e)[t];
    double A[2 * CNP], B[6], x0, x1;
    if (FD) jac_fd<CNP, KNOWN>(cfg, ct, b[0], b[1], b[2], A, B, x0, x1);
    else    jac_analytic<CNP>(cfg, ct, b[0], b[1], b[2], A, B, x0, x1);
    // The 256 records of a workgroup are contiguous in both streams: they are transposed through LDS so that every store
    // instruction writes 64 x 16 consecutive bytes (a lane-per-record store touches 64 different cache lines per instruction,
    // 16 bytes of each: the kernel was bound by those write transactions, 0.43 ms for 960 MB).
    // The two streams go through the SAME staging area one after the other (37 KB: three workgroups per CU; both at once were 57 KB:
    // two, and the kernel went from 0.31 to 0.41 ms).
    constexpr int LA = CNP + 1 - (CNP & 1);                  // LDS row stride in chunks (odd: conflict-free column writes)
    constexpr int LB = 5;
    __shared__ double2 stage[256 * LA];
    double2* mineA = stage + threadIdx.x * LA;
#pragma unroll
    for (int q = 0; q < CNP; ++q) mineA[q] = make_double2(A[q], A[CNP + q]);
    __syncthreads();
    const int t0 = blockIdx.x * 256;
    const int nrec = min(256, nvis - t0);
    double2* outa = reinterpret_cast<double2*>(Ac + (size_t)t0 * 2 * CNP);
    for (int c = threadIdx.x; c < nrec * CNP; c += 256) {
        const int rec = c / CNP, part = c - rec * CNP;
        outa[c] = stage[rec * LA + part];
    }
    __syncthreads();
    static_assert(LB <= LA, "the B || e records reuse the staging area of the A chunks");
    double2* mineB = stage + threadIdx.x * LB;
#pragma unroll
    for (int q = 0; q < 3; ++q) mineB[q] = make_double2(B[2 * q], B[2 * q + 1]);
    mineB[3] = et;
    __syncthreads();
    double2* outb = reinterpret_cast<double2*>(Bc + (size_t)t0 * 8);
    for (int c = threadIdx.x; c < nrec * 4; c += 256) outb[c] = stage[(c >> 2) * LB + (c & 3)];
}

// N 16-byte loads of 2N consecutive doubles.  Jacobian records are 16-byte aligned (even length, 2*cnp even), which
// the compiler cannot know from a double*: spelled out, a 192-byte record costs 12 load instructions instead of 24 --
// the per-observation kernels are bound by the number of line requests on the vector-memory path, not by bytes.
template <int N>
__device__ __forceinline__ void load_pairs(const double* __restrict__ p, double* __restrict__ out)
{
#pragma unroll
    for (int q = 0; q < N; ++q) {
        const double2 t = reinterpret_cast<const double2*>(p)[q];
        out[2 * q] = t.x; out[2 * q + 1] = t.y;
    }
}

// ---------------------------------------------------------------------------------------------------
// V_i (packed upper: 00 01 02 11 12 22), eb_i: one thread per point walks its CRS row.
// Round 6: with part != nullptr the kernel also does k_iter_partials' and k_iter_final's job (sba_levmar.c:1085-1128: max |eb|, the largest
// diagonal entry of V, sum p_b^2 while the values are in registers; the workgroup that arrives last reduces them, adds the camera side and
// the constraint cost) and clears the flags of the iteration's first solve attempt: two stream operations fewer per iteration.
struct IterFinalArgs {       // what the finishing workgroup needs
    const double* pa; int have_points, point_part_slot;
    int s_eabinf_a, s_eabinf_b, s_maxdiag_u, s_maxdiag_v, s_pl2_a, s_pl2_b, s_ccost;
    double* scal;
    int point_blocks;      // workgroups of k_point_blocks that do points; ONE MORE behind them does the camera side of the scalars (0: the grid is points only)
};
__device__ __forceinline__ void iter_final_body(const DevProblem& P, const double* __restrict__ pa, const double* __restrict__ pb,
        const double* __restrict__ part, int nparts, int have_points, int point_part_slot,
        int s_eabinf_a, int s_eabinf_b, int s_maxdiag_u, int s_maxdiag_v, int s_pl2_a, int s_pl2_b, int s_ccost,
        double* __restrict__ scal, bool agent_loads, int which = 3 /* bit 0: the point side's partials, bit 1: the camera side and the constraint cost */);

// LPP lanes per point (round 6): 1 = a thread walks its point's row, four observations per trip (positions first, then the four records; past the
// row's end the last position is read again and its terms dropped, so the loads stay unconditional); 4 = the small problems, where the kernel is
// nothing but the latency of that chain of dependent gathers (see k_backsub): lane q takes observations q, q + 4, q + 8 -- one trip for up to twelve --
// and the quad adds its partial sums in a fixed order.  With part != nullptr the camera side of the iteration's scalars is done by one more workgroup
// behind the points' (beside them, not after them: it needs U and ea of the kernel before, nothing of this one).
constexpr int PB_TRIP = 3;
template <int CNP, int LPP>
__global__ __launch_bounds__(256) void k_point_blocks(DevProblem P, const double* __restrict__ pb,
        double* __restrict__ part = nullptr /* [3][point blocks] */, unsigned* __restrict__ ticket = nullptr, unsigned* __restrict__ ticket_groups = nullptr,
        IterFinalArgs fa = IterFinalArgs(), int* __restrict__ flags_to_clear = nullptr)
{
    const int nbp = part ? fa.point_blocks : (int)gridDim.x;
    if ((int)blockIdx.x >= nbp) {
        iter_final_body(P, fa.pa, pb, part, nbp, fa.have_points, fa.point_part_slot, fa.s_eabinf_a, fa.s_eabinf_b, fa.s_maxdiag_u, fa.s_maxdiag_v,
                        fa.s_pl2_a, fa.s_pl2_b, fa.s_ccost, fa.scal, true, 2);
        return;
    }
    const int gi = blockIdx.x * 256 + threadIdx.x;
    const int ireal = gi / LPP, q = gi % LPP;
    const bool active = ireal < P.n && q == 0;
    const int i = ireal < P.n ? ireal : P.n - 1;             // (quads past the end compute along: the quad sums need every lane)
    double it_a = 0.0, it_v = -DBL_MAX, it_s = 0.0;
    {
        double v00 = 0, v01 = 0, v02 = 0, v11 = 0, v12 = 0, v22 = 0, g0 = 0, g1 = 0, g2 = 0;
        const int k0 = P.rowptr[i], k1 = P.rowptr[i + 1];
        constexpr int TRIP = LPP == 1 ? 4 : PB_TRIP;
        for (int kb = k0; kb < k1; kb += LPP * TRIP) {
            int pos[TRIP];
#pragma unroll
            for (int u = 0; u < TRIP; ++u) pos[u] = P.campos[min(kb + LPP * u + q, k1 - 1)];
            double R[TRIP][8];                                             // B (2 x 3) || e: one 64-byte sector each
#pragma unroll
            for (int u = 0; u < TRIP; ++u) load_pairs<4>(P.Bc + (size_t)pos[u] * 8, R[u]);
#pragma unroll
            for (int u = 0; u < TRIP; ++u) {
                if (kb + LPP * u + q < k1) {
                    const double b0 = R[u][0], b1 = R[u][1], b2 = R[u][2], b3 = R[u][3], b4 = R[u][4], b5 = R[u][5];
                    const double e0 = R[u][6], e1 = R[u][7];
                    v00 += b0 * b0 + b3 * b3; v01 += b0 * b1 + b3 * b4; v02 += b0 * b2 + b3 * b5;
                    v11 += b1 * b1 + b4 * b4; v12 += b1 * b2 + b4 * b5; v22 += b2 * b2 + b5 * b5;
                    g0 += b0 * e0 + b3 * e1; g1 += b1 * e0 + b4 * e1; g2 += b2 * e0 + b5 * e1;
                }
            }
        }
        if (LPP == 4) {
#define BSFM_QSUM(x) x = (quad_bcast_d<0>(x) + quad_bcast_d<1>(x)) + (quad_bcast_d<2>(x) + quad_bcast_d<3>(x))
            BSFM_QSUM(v00); BSFM_QSUM(v01); BSFM_QSUM(v02); BSFM_QSUM(v11); BSFM_QSUM(v12); BSFM_QSUM(v22); BSFM_QSUM(g0); BSFM_QSUM(g1); BSFM_QSUM(g2);
#undef BSFM_QSUM
        }
        if (active) {
            if (P.pcon && P.pcon[i]) {   // sba_levmar.c:1017-1028 (weights scale with the job-wide nvis)
                const double w = P.nvis_global * P.pweight;
                v00 += w; v11 += w; v22 += w;
                g0 += w * (P.pval[3 * i] - pb[3 * i]);
                g1 += w * (P.pval[3 * i + 1] - pb[3 * i + 1]);
                g2 += w * (P.pval[3 * i + 2] - pb[3 * i + 2]);
            }
            double* V = P.V + (size_t)i * 6;
            V[0] = v00; V[1] = v01; V[2] = v02; V[3] = v11; V[4] = v12; V[5] = v22;
            double* g = P.eb + (size_t)i * 3;
            g[0] = g0; g[1] = g1; g[2] = g2;
            if (part) {
                const double a0 = fabs(g0), a1 = fabs(g1), a2 = fabs(g2);
                it_a = a0 > a1 ? a0 : a1; it_a = it_a > a2 ? it_a : a2;
                it_v = v00 > v11 ? v00 : v11; it_v = it_v > v22 ? it_v : v22;
                const double p0 = pb[3 * (size_t)i], p1 = pb[3 * (size_t)i + 1], p2 = pb[3 * (size_t)i + 2];
                it_s = p0 * p0 + p1 * p1 + p2 * p2;
            }
        }
    }
    if (!part) return;
    __shared__ double sm[4];
    const double ra = block_max4(it_a, sm), rv = block_max4(it_v, sm), rs = block_sum4(it_s, sm);
    if (threadIdx.x == 0) {
        st_agent(part + blockIdx.x, ra); st_agent(part + (size_t)nbp + blockIdx.x, rv); st_agent(part + 2 * (size_t)nbp + blockIdx.x, rs);
    }
    if (!last_block_arrives(ticket, (unsigned)nbp, ticket_groups)) return;
    iter_final_body(P, fa.pa, pb, part, nbp, fa.have_points, fa.point_part_slot, fa.s_eabinf_a, fa.s_eabinf_b, fa.s_maxdiag_u, fa.s_maxdiag_v,
                    fa.s_pl2_a, fa.s_pl2_b, fa.s_ccost, fa.scal, true, 1);
    if (flags_to_clear && threadIdx.x < 4) flags_to_clear[threadIdx.x] = 0;
}

// ---------------------------------------------------------------------------------------------------
// U_j (upper triangle accumulated, mirrored on store), ea_j.  CAM_SPLIT 256-thread blocks per camera each walk one slice
// of the camera's (contiguous, camera-major) records: per-thread register partials, wave shuffle reduction, LDS across
// the 4 waves, one partial row per block; k_cam_blocks_fin adds the CAM_SPLIT rows in slice order (deterministic).
// One block per camera left a third of the device idle: 1000 blocks of ~166 VGPRs fill 768 slots, then 232.
constexpr int CAM_SPLIT = 4;
__device__ __forceinline__ void cam_slice(const int* __restrict__ camptr, int j, int s, int& t0, int& t1)
{
    const int a = camptr[j], b = camptr[j + 1];
    const int chunk = (b - a + CAM_SPLIT - 1) / CAM_SPLIT;
    t0 = min(b, a + s * chunk); t1 = min(b, t0 + chunk);
}

template <int CNP>
__global__ __launch_bounds__(256) void k_cam_blocks(DevProblem P, const double* __restrict__ e, double* __restrict__ part,
                                                    unsigned* __restrict__ cam_ticket /* per camera; null: k_cam_blocks_fin follows */,
                                                    const double* __restrict__ pa_con = nullptr /* != null: k_cam_constraints' terms are added here (single GPU) */)
{
    constexpr int NU = CNP * (CNP + 1) / 2;
    constexpr int NV = NU + CNP;
    __shared__ double sm[4][NV];
    const int j = blockIdx.x / CAM_SPLIT, s = blockIdx.x % CAM_SPLIT;
    double acc[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) acc[q] = 0.0;
    if (j >= P.mcon) {
        int t0, t1;
        cam_slice(P.camptr, j, s, t0, t1);
        for (int t = t0 + threadIdx.x; t < t1; t += 256) {
            double a[2 * CNP], ee[2];                                  // a[2 c] = A[0][c], a[2 c + 1] = A[1][c]
            load_pairs<CNP>(P.Ac + (size_t)t * 2 * CNP, a);
            load_pairs<1>(e + 2 * (size_t)t, ee);                      // camera-major residuals: streamed, no gather
            const double e0 = ee[0], e1 = ee[1];
            int u = 0;
#pragma unroll
            for (int r = 0; r < CNP; ++r) {
#pragma unroll
                for (int c = r; c < CNP; ++c) { acc[u] += a[2 * r] * a[2 * c] + a[2 * r + 1] * a[2 * c + 1]; ++u; }
            }
#pragma unroll
            for (int r = 0; r < CNP; ++r) acc[NU + r] += a[2 * r] * e0 + a[2 * r + 1] * e1;
        }
    }
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < NV; ++q) { const double v = wave_sum(acc[q]); if (lane == 0) sm[w][q] = v; }
    __syncthreads();
    if (threadIdx.x < NV)
        st_agent(part + (size_t)blockIdx.x * NV + threadIdx.x, (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]));
    if (!cam_ticket || !last_block_arrives(cam_ticket + j, CAM_SPLIT)) return;
    {   // k_cam_blocks_fin for camera j: the CAM_SPLIT slice rows in slice order (the same sums, the same order)
        const double* pj = part + (size_t)j * CAM_SPLIT * NV;
        if (threadIdx.x < CNP * CNP) {
            const int r = threadIdx.x / CNP, c = threadIdx.x % CNP;
            const int rr = r < c ? r : c, cc = r < c ? c : r;
            const int u = rr * CNP - rr * (rr - 1) / 2 + (cc - rr);
            double v = 0.0;
#pragma unroll
            for (int sl = 0; sl < CAM_SPLIT; ++sl) v += ld_agent(pj + sl * NV + u);
            if (pa_con && r == c && j >= P.mcon && P.ccon[(size_t)j * CNP + r]) v += P.cw[(size_t)j * CNP + r];      // sba_levmar.c:953-962 (k_cam_constraints)
            P.U[(size_t)j * CNP * CNP + threadIdx.x] = v;
        } else if (threadIdx.x < CNP * CNP + CNP) {
            const int u = NU + (threadIdx.x - CNP * CNP);
            double v = 0.0;
#pragma unroll
            for (int sl = 0; sl < CAM_SPLIT; ++sl) v += ld_agent(pj + sl * NV + u);
            const size_t t = (size_t)j * CNP + (threadIdx.x - CNP * CNP);
            if (pa_con && j >= P.mcon && P.ccon[t]) { const double diff = P.cval[t] - pa_con[t]; v += P.cw[t] * diff; }
            P.ea[t] = v;
        }
    }
}

template <int CNP>
__global__ __launch_bounds__(128) void k_cam_blocks_fin(DevProblem P, const double* __restrict__ part)
{
    constexpr int NU = CNP * (CNP + 1) / 2;
    constexpr int NV = NU + CNP;
    const int j = blockIdx.x;
    const double* pj = part + (size_t)j * CAM_SPLIT * NV;
    if (threadIdx.x < CNP * CNP) {
        const int r = threadIdx.x / CNP, c = threadIdx.x % CNP;
        const int rr = r < c ? r : c, cc = r < c ? c : r;
        const int u = rr * CNP - rr * (rr - 1) / 2 + (cc - rr);
        double v = 0.0;
#pragma unroll
        for (int s = 0; s < CAM_SPLIT; ++s) v += pj[s * NV + u];
        P.U[(size_t)j * CNP * CNP + threadIdx.x] = v;
    } else if (threadIdx.x < CNP * CNP + CNP) {
        const int u = NU + (threadIdx.x - CNP * CNP);
        double v = 0.0;
#pragma unroll
        for (int s = 0; s < CAM_SPLIT; ++s) v += pj[s * NV + u];
        P.ea[(size_t)j * CNP + (threadIdx.x - CNP * CNP)] = v;
    }
}

// camera constraints into U_j / ea_j (sba_levmar.c:953-962); replicated cameras => every rank applies it
// AFTER the cross-rank reduction of U and ea.
__global__ void k_cam_constraints(DevProblem P, const double* __restrict__ pa)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int cnp = P.cfg.cnp;
    if (t >= P.m * cnp) return;
    const int j = t / cnp, jj = t % cnp;
    if (j < P.mcon || !P.ccon[t]) return;
    const double diff = P.cval[t] - pa[t];
    P.U[(size_t)j * cnp * cnp + jj * cnp + jj] += P.cw[t];
    P.ea[t] += P.cw[t] * diff;
}

// ---------------------------------------------------------------------------------------------------
// (V_i + mu I)^-1, closed form for the symmetric 3x3 (the reference goes through dsytrf/dsytri);
// a non-finite or zero determinant raises the "singular" flag => more damping (sba_levmar.c:1156-1161).
// (one function for both users -- the stand-alone kernel and k_schur_prep's fused form -- so that the same source expressions, hence the
//  same contractions, produce V*^-1: the two forms are bit-identical)
__device__ __forceinline__ bool sym3_invert(const double* __restrict__ v, double mu, double (&o)[6])
{
    const double a = v[0] + mu, b = v[1], c = v[2], d = v[3] + mu, e = v[4], f = v[5] + mu;
    const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
    const double det = a * c00 + b * c01 + c * c02;
    const double id = 1.0 / det;
    o[0] = c00 * id; o[1] = c01 * id; o[2] = c02 * id;
    o[3] = (a * f - c * c) * id; o[4] = (b * c - a * e) * id; o[5] = (a * d - b * b) * id;
    return !(fabs(det) > 0.0) || !isfinite(id);
}
__global__ __launch_bounds__(256) void k_point_invert(int n, double mu, const double* __restrict__ V,
                                                      double* __restrict__ Vinv, int* __restrict__ flag)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double inv[6];
    const bool singular = sym3_invert(V + (size_t)i * 6, mu, inv);
    double* o = Vinv + (size_t)i * 6;
#pragma unroll
    for (int q = 0; q < 6; ++q) o[q] = inv[q];
    if (singular) atomicOr(flag, 1);
}

// S is rebuilt for every solve attempt (the factorisation overwrites it and fills in its zero blocks).  Only the lower
// triangle of 128x128 tiles is read by the factorisation, so only that half is cleared: one workgroup per tile,
// 16-byte stores.  (The strictly upper tiles keep whatever the mirror writes of the assembly put there.)  Virtual blocks ntl .. do
// k_rhs_init's job.  Round 6: the body is a function so that k_schur_prep can run it in workgroups appended to its own grid (one launch
// fewer per solve attempt; k_zero_lower_tiles below is the stand-alone form).
struct ZeroTilesArgs {
    double* S; int ld, ntl, count, off, add_ea; const double* ea; double* E; const int* spos; int cnp;
    int first_block;      // k_schur_prep: its workgroups from this index on run zero_tiles_body (0 = none appended)
};
__device__ __forceinline__ void zero_tiles_body(const ZeroTilesArgs& z, int vb)
{
    if (vb >= z.ntl) {      // blocks ntl .. : k_rhs_init's job (round 5: one launch fewer per attempt)
        const int t = (vb - z.ntl) * 256 + threadIdx.x;
        if (t >= z.count) return;
        const int dst = z.spos ? z.spos[t / z.cnp] * z.cnp + t % z.cnp : t;
        z.E[dst] = z.add_ea ? z.ea[z.off + t] : 0.0;
        return;
    }
    int a = (int)((sqrt(8.0 * vb + 1.0) - 1.0) * 0.5);
    while ((a + 1) * (a + 2) / 2 <= vb) ++a;
    while (a * (a + 1) / 2 > vb) --a;
    const int b = vb - a * (a + 1) / 2;          // tile (a, b), b <= a
    double* T = z.S + ((size_t)a * 128) * z.ld + (size_t)b * 128;
    for (int idx = threadIdx.x; idx < 128 * 64; idx += 256) {
        const int r = idx >> 6, c2 = (idx & 63) * 2;
        *reinterpret_cast<double2*>(T + (size_t)r * z.ld + c2) = make_double2(0.0, 0.0);
    }
}

// Per solve attempt (V*^-1 depends on mu): C_ij = B_ij V*_i^-1 (2 x 3) and r_ij = C_ij eb_i for every observation, camera-major,
// one 64-byte record.  With it a Schur task forms its 2 x 2 core as C_ij B_ik^T (12 FMAs) from two 48-byte operands instead of
// gathering V*_i^-1 and doing the 3 x 3 product per co-visibility triple (a point with d cameras is in d (d + 1) / 2 triples), and
// the reduced right-hand side needs no eb gather: e_j -= sum_i A_ij^T r_ij (sba_levmar.c:1195-1216, 1320-1339: Y_ij = W_ij V*_i^-1).
// Walked in POINT-major (CRS) order with FOUR lanes per observation: V*_i^-1 and eb_i of consecutive observations are the same or the
// next point (streamed; in camera-major order every camera sweeps the whole 36 MB of point data past L2: 0.28 ms), the B || e record
// is ONE 64-byte sector fetched by the quad (lane q loads chunk q, the six B values go round by DPP), and lane q stores chunk q of the
// C || r record: one whole 64-byte sector per quad, gathered and scattered through campos[].
// Round 6: with V != nullptr the kernel ALSO does k_point_invert's job (one launch fewer per solve attempt, and at 14 cameras a launch is
// 3 % of the iteration): every quad inverts V*_i itself from the six numbers it would otherwise have loaded as V*^-1, and the quad of a
// point's FIRST observation stores the inverse (k_backsub reads it) and raises the "singular" flag.  Needs every point to have an
// observation (the index build reports empty rows; then the stand-alone kernel runs).
__global__ __launch_bounds__(256) void k_schur_prep(int nvis, const int* __restrict__ obs_pt, const int* __restrict__ campos,
        const double* __restrict__ Bc, const double* __restrict__ Vinv, const double* __restrict__ eb, double* __restrict__ Cc,
        const double* __restrict__ V = nullptr, double mu = 0.0, const int* __restrict__ rowptr = nullptr, double* __restrict__ Vinv_out = nullptr,
        int* __restrict__ flag = nullptr, ZeroTilesArgs z = ZeroTilesArgs())
{
    if (z.first_block > 0 && (int)blockIdx.x >= z.first_block) { zero_tiles_body(z, (int)blockIdx.x - z.first_block); return; }      // (appended workgroups: k_zero_lower_tiles' job)
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t k = min(g >> 2, (size_t)nvis - 1);               // surplus quads of the last workgroup repeat the last observation
    const int q = (int)(g & 3);
    const size_t t = (size_t)campos[k];
    const int i = obs_pt[k];
    const double2 mine = reinterpret_cast<const double2*>(Bc)[t * 4 + q];
    double vi[6];
    if (V) {
        double vv[6];
        load_pairs<3>(V + (size_t)i * 6, vv);
        const bool singular = sym3_invert(vv, mu, vi);
        if (q == 0 && (g >> 2) < (size_t)nvis && (int)k == rowptr[i]) {
            double* o = Vinv_out + (size_t)i * 6;
#pragma unroll
            for (int c = 0; c < 6; ++c) o[c] = vi[c];
            if (singular) atomicOr(flag, 1);
        }
    } else load_pairs<3>(Vinv + (size_t)i * 6, vi);
    const double b0 = quad_bcast_d<0>(mine.x), b1 = quad_bcast_d<0>(mine.y), b2 = quad_bcast_d<1>(mine.x);
    const double b3 = quad_bcast_d<1>(mine.y), b4 = quad_bcast_d<2>(mine.x), b5 = quad_bcast_d<2>(mine.y);
    const double i00 = vi[0], i01 = vi[1], i02 = vi[2], i11 = vi[3], i12 = vi[4], i22 = vi[5];
    const double c00 = b0 * i00 + b1 * i01 + b2 * i02;
    const double c01 = b0 * i01 + b1 * i11 + b2 * i12;
    const double c02 = b0 * i02 + b1 * i12 + b2 * i22;
    const double c10 = b3 * i00 + b4 * i01 + b5 * i02;
    const double c11 = b3 * i01 + b4 * i11 + b5 * i12;
    const double c12 = b3 * i02 + b4 * i12 + b5 * i22;
    double2 o;
    if (q == 0) o = make_double2(c00, c01);
    else if (q == 1) o = make_double2(c02, c10);
    else if (q == 2) o = make_double2(c11, c12);
    else {
        const double e0 = eb[(size_t)i * 3], e1 = eb[(size_t)i * 3 + 1], e2 = eb[(size_t)i * 3 + 2];
        o = make_double2(c00 * e0 + c01 * e1 + c02 * e2, c10 * e0 + c11 * e1 + c12 * e2);
    }
    if ((g >> 2) < (size_t)nvis) reinterpret_cast<double2*>(Cc)[t * 4 + q] = o;
}

// ---------------------------------------------------------------------------------------------------
// Schur complement: the task kernel lives in schur.hip.h; here are the assembly kernels.
// S_jk = [j==k](U_j + mu I) - sum_tasks partial ; mirrored into S_kj (sba_levmar.c:1274-1316).
template <int CNP>
__global__ __launch_bounds__(128) void k_schur_assemble(int nblk, const int* __restrict__ blk_j, const int* __restrict__ blk_k,
        const int2* __restrict__ blk_range, const double* __restrict__ partials, const double* __restrict__ epart,
        const double* __restrict__ U, double mu, int mcon, double* __restrict__ S, int ld, double* __restrict__ E,
        const int* __restrict__ spos, int m_fill = 0, const int* __restrict__ camptr = nullptr)
{
    // spos (envelope solver): position of free camera c = j - mcon in the reordered system; nullptr = natural order
    const int b = blockIdx.x;
    if (b >= nblk) {
        // blocks nblk .. : k_schur_diag_fill's job (round 5: one launch fewer per attempt) -- the diagonal block of a camera that has no
        // (j, j) block in the triple list (no observations)
        const int j = mcon + (b - nblk);
        if (j >= m_fill || threadIdx.x >= CNP * CNP) return;
        if (camptr[j + 1] > camptr[j]) return;
        const int row = threadIdx.x / CNP, col = threadIdx.x % CNP;
        double v = U[(size_t)j * CNP * CNP + threadIdx.x];
        if (row == col) v += mu;
        const size_t pj = spos ? spos[j - mcon] : j - mcon;
        S[(pj * CNP + row) * ld + pj * CNP + col] = v;
        return;
    }
    const int j = blk_j[b], k = blk_k[b];
    const int pj = spos ? spos[j - mcon] : j - mcon, pk = spos ? spos[k - mcon] : k - mcon;
    const int2 sl = blk_range[b];               // the block's slots: its tasks of k_schur_tasks, in task order (index_build.hip)
    if (j == k && threadIdx.x >= CNP * CNP && threadIdx.x < CNP * CNP + CNP) {   // e_j -= this block's tasks (E was set to ea by k_rhs_init)
        const int q = threadIdx.x - CNP * CNP;
        double se = 0.0;
        for (int t = sl.x; t < sl.y; ++t) se += epart[(size_t)t * CNP + q];
        E[(size_t)pj * CNP + q] -= se;
    }
    if (threadIdx.x >= CNP * CNP) return;
    const int row = threadIdx.x / CNP, col = threadIdx.x % CNP;
    double s = 0.0;
    for (int t = sl.x; t < sl.y; ++t) s += partials[(size_t)t * CNP * CNP + threadIdx.x];
    double v = -s;
    if (j == k) { v += U[(size_t)j * CNP * CNP + threadIdx.x]; if (row == col) v += mu; }
    const size_t rj = (size_t)pj * CNP + row, ck = (size_t)pk * CNP + col;
    S[rj * ld + ck] = v;
    if (j != k) S[ck * ld + rj] = v;
}

// Multi-GPU job: this rank's block sums go to their slot of the union structure (the buffer that is all-reduced) ...
template <int CNP>
__global__ __launch_bounds__(128) void k_schur_pack(int nblk, const int* __restrict__ blk_j, const int* __restrict__ blk_k,
        const int2* __restrict__ blk_range, const double* __restrict__ partials, const double* __restrict__ epart,
        const int* __restrict__ gidx, double* __restrict__ G, int mcon, double* __restrict__ E)
{
    const int b = blockIdx.x;
    if (b >= nblk) return;
    const int2 sl = blk_range[b];
    if (blk_j[b] == blk_k[b] && threadIdx.x >= CNP * CNP && threadIdx.x < CNP * CNP + CNP) {
        const int q = threadIdx.x - CNP * CNP;
        double se = 0.0;
        for (int t = sl.x; t < sl.y; ++t) se += epart[(size_t)t * CNP + q];
        E[(size_t)(blk_j[b] - mcon) * CNP + q] -= se;
    }
    if (threadIdx.x >= CNP * CNP) return;
    double s = 0.0;
    for (int t = sl.x; t < sl.y; ++t) s += partials[(size_t)t * CNP * CNP + threadIdx.x];
    G[(size_t)gidx[b] * CNP * CNP + threadIdx.x] = s;
}

// ... and after the exchange every rank assembles the same S from the summed blocks.
template <int CNP>
__global__ __launch_bounds__(128) void k_schur_unpack(int ngblk, const int* __restrict__ gj, const int* __restrict__ gk,
        const double* __restrict__ G, const double* __restrict__ U, double mu, int mcon, double* __restrict__ S, int ld,
        const int* __restrict__ spos)
{
    const int g = blockIdx.x;
    if (g >= ngblk || threadIdx.x >= CNP * CNP) return;
    const int j = gj[g], k = gk[g];
    const int row = threadIdx.x / CNP, col = threadIdx.x % CNP;
    double v = -G[(size_t)g * CNP * CNP + threadIdx.x];
    if (j == k) { v += U[(size_t)j * CNP * CNP + threadIdx.x]; if (row == col) v += mu; }
    const size_t rj = (size_t)(spos ? spos[j - mcon] : j - mcon) * CNP + row, ck = (size_t)(spos ? spos[k - mcon] : k - mcon) * CNP + col;
    S[rj * ld + ck] = v;
    if (j != k) S[ck * ld + rj] = v;
}

// Cameras that own no (j,j) block in the triple list (no observation) still need U_j + mu I.
// camptr == nullptr: fill EVERY diagonal block (multi-GPU path: k_schur_unpack then overwrites those with a block).
template <int CNP>
__global__ void k_schur_diag_fill(int m, int mcon, const int* __restrict__ camptr, const double* __restrict__ U,
                                  double mu, double* __restrict__ S, int ld, const int* __restrict__ spos)
{
    const int j = mcon + blockIdx.x;
    if (j >= m || threadIdx.x >= CNP * CNP) return;
    if (camptr && camptr[j + 1] > camptr[j]) return;    // has a (j,j) block in the triple list
    const int row = threadIdx.x / CNP, col = threadIdx.x % CNP;
    double v = U[(size_t)j * CNP * CNP + threadIdx.x];
    if (row == col) v += mu;
    const size_t pj = spos ? spos[j - mcon] : j - mcon;
    S[(pj * CNP + row) * ld + pj * CNP + col] = v;
}

__global__ __launch_bounds__(256) void k_zero_lower_tiles(double* __restrict__ S, int ld, int ntl = 0x7fffffff, int count = 0, int off = 0, int add_ea = 0,
        const double* __restrict__ ea = nullptr, double* __restrict__ E = nullptr, const int* __restrict__ spos = nullptr, int cnp = 0)
{
    ZeroTilesArgs z; z.S = S; z.ld = ld; z.ntl = ntl; z.count = count; z.off = off; z.add_ea = add_ea; z.ea = ea; z.E = E; z.spos = spos; z.cnp = cnp; z.first_block = 0;
    zero_tiles_body(z, (int)blockIdx.x);
}

// Reduced right-hand side: E_j starts as ea_j (on the rank that contributes U/ea to a multi-GPU sum, else 0); the tasks of
// the diagonal blocks subtract sum_i A_ij^T (B_ij V*_i^-1 eb_i) in k_schur_assemble / k_schur_pack (sba_levmar.c:1320-1339).
__global__ __launch_bounds__(256) void k_rhs_init(int count, int off, int add_ea, const double* __restrict__ ea, double* __restrict__ E,
                                                  const int* __restrict__ spos, int cnp)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= count) return;
    const int dst = spos ? spos[t / cnp] * cnp + t % cnp : t;
    E[dst] = add_ea ? ea[off + t] : 0.0;
}

// envelope solver: the solution comes back in the reordered system; dpa[c] = x[spos[c]] for the free cameras
__global__ __launch_bounds__(256) void k_unpermute_step(int count, int cnp, const int* __restrict__ spos, const double* __restrict__ x, double* __restrict__ dpa)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < count) dpa[t] = x[(size_t)spos[t / cnp] * cnp + t % cnp];
}

// sum of the three block-partial rows of k_backsub + the camera part of the step (k_step_sums' job; see there)
__device__ __forceinline__ void step_sums_body(int count, int fixed, double mu, const double* __restrict__ pa,
        const double* __restrict__ dpa, const double* __restrict__ ea, double* __restrict__ pdpa, double* __restrict__ out3,
        const double* __restrict__ part, int nbp, double* __restrict__ pt3, bool agent_loads)
{
    __shared__ double sm[6][4];
    double s_dp = 0, s_p = 0, s_dl = 0;
    for (int t = threadIdx.x; t < count; t += 256) {
        const double d = (t < fixed) ? 0.0 : dpa[t], p = pa[t];
        pdpa[t] = p + d;
        s_dp += d * d; s_p += p * p; s_dl += d * (mu * d + ea[t]);
    }
    double v[6] = { s_dp, s_p, s_dl, 0.0, 0.0, 0.0 };
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double a = 0.0;
#pragma unroll 8
        for (int t = threadIdx.x; t < nbp; t += 256) a += agent_loads ? ld_agent(part + (size_t)c * nbp + t) : part[(size_t)c * nbp + t];
        v[3 + c] = a;
    }
    // six block sums through ONE exchange (round 6: six block_sum4 were twelve barriers on the way to every trial point of a small problem); the
    // order inside each sum is block_sum4's: lanes of a wave, then (w0 + w1) + (w2 + w3)
#pragma unroll
    for (int c = 0; c < 6; ++c) v[c] = wave_sum(v[c]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int c = 0; c < 6; ++c) sm[c][threadIdx.x >> 6] = v[c];
    }
    __syncthreads();
    const double r0 = (sm[0][0] + sm[0][1]) + (sm[0][2] + sm[0][3]), r1 = (sm[1][0] + sm[1][1]) + (sm[1][2] + sm[1][3]), r2 = (sm[2][0] + sm[2][1]) + (sm[2][2] + sm[2][3]);
    double q[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) q[c] = (sm[3 + c][0] + sm[3 + c][1]) + (sm[3 + c][2] + sm[3 + c][3]);
    if (threadIdx.x == 0) { out3[0] = r0; out3[1] = r1; out3[2] = r2; pt3[0] = q[0]; pt3[1] = q[1]; pt3[2] = q[2]; }
}
struct StepFinalArgs {       // k_step_sums' and k_cam_table's arguments, for the workgroup of k_backsub that arrives last
    int count, fixed; const double* pa; double* pdpa; double* out3; double* pt3;
    const double* known; int with_fd; double* camtab_trial;
    unsigned* ticket_groups;      // last_block_arrives' group words for this grid (null: one level)
    int point_blocks;             // workgroups of k_backsub that do points; one more behind them builds the trial point's camera table (camtab_trial != null)
};

// ---------------------------------------------------------------------------------------------------
// First pass of the back-substitution on big problems (round 5): w_k = W_k^T da_j = B_k^T (A_k da_j) for every observation, one thread
// per CAMERA-major position.  The thread-per-point kernel below gathers the 144 + 48 bytes of (A, B) of each of its observations from
// the camera-major streams -- whole 128-byte lines for pieces of them: 1.9 x the bytes (1 974 MB against 1 046 at config 3, 0.50 ms).
// Here the two streams are read as they lie (the 256 records of a workgroup are contiguous: A through LDS with 16 consecutive bytes per
// lane, the way k_jacobian wrote them), da_j is one broadcast per camera, and 32 bytes per observation go out for the gather of the
// second pass.  Same operations in the same order as the one-pass form: bit-identical.  (Four lanes per point in the second pass -- more
// gathers in flight -- measured slower, 270 against 232 us: that pass is bound by the number of scattered requests, not by their latency.)
template <int CNP>
__global__ __launch_bounds__(256) void k_backsub_obs(int nvis, int mcon, const int* __restrict__ cam_cam, const double* __restrict__ Ac,
        const double* __restrict__ Bc, const double* __restrict__ dpa, double* __restrict__ wobs)
{
    constexpr int LA = CNP + 1 - (CNP & 1);                  // LDS row stride in 16-byte chunks (odd: conflict-free)
    __shared__ double2 stage[256 * LA];
    const int t0 = blockIdx.x * 256;
    const int nrec = min(256, nvis - t0);
    const double2* src = reinterpret_cast<const double2*>(Ac + (size_t)t0 * 2 * CNP);
    for (int c = threadIdx.x; c < nrec * CNP; c += 256) {
        const int rec = c / CNP, part = c - rec * CNP;
        stage[rec * LA + part] = src[c];
    }
    __syncthreads();
    const int t = t0 + threadIdx.x;
    if (t >= nvis) return;
    const int j = cam_cam[t];
    double w0 = 0.0, w1 = 0.0, w2 = 0.0;
    if (j >= mcon) {
        double B[6];
        load_pairs<3>(Bc + (size_t)t * 8, B);
        const double* da = dpa + (size_t)j * CNP;
        const double2* mine = stage + threadIdx.x * LA;
        double q0 = 0, q1 = 0;
#pragma unroll
        for (int c = 0; c < CNP; ++c) { const double2 a = mine[c]; q0 += a.x * da[c]; q1 += a.y * da[c]; }
        w0 = B[0] * q0 + B[3] * q1; w1 = B[1] * q0 + B[4] * q1; w2 = B[2] * q0 + B[5] * q1;
    }
    // (Round 6, measured and dropped: writing the product to the observation's POINT-major slot so that the second pass streams its row instead of
    //  gathering 32 bytes out of a 128-byte line.  The scattered 32-byte stores of this pass cost more than the gathers of the next: back-substitution
    //  0.50 -> 0.63 ms at config 3.)
    reinterpret_cast<double2*>(wobs)[2 * (size_t)t] = make_double2(w0, w1);
    reinterpret_cast<double2*>(wobs)[2 * (size_t)t + 1] = make_double2(w2, 0.0);
}

// ---------------------------------------------------------------------------------------------------
// db_i = V*_i^-1 (eb_i - sum_j W_ij^T da_j), W_ij^T da_j = B_ij^T (A_ij da_j); thread per point.
// Also writes pdp_b = p_b + db and block partials of sum db^2, sum p_b^2 and sum db (mu db + eb).
// Round 6, the small problems (one-pass form, below 400 000 observations) are latency: a 14-camera problem is 6 workgroups whose threads each walked
// a chain of 2 d_i dependent gathers, every one a miss of the XCD's L2 (the records were written on other XCDs): 12.7 us of a 26 us kernel, and the
// last workgroup then spent 6.3 us on the trial point's camera table (profiles/r06_small_problem_latency.txt).  Now FOUR lanes share a point -- lane q
// takes observations q, q + 4, q + 8 of the row, three per trip: positions, then records, one trip for up to 12 observations; the quad adds its four
// partial products in a fixed order -- and the camera table is built by ONE MORE workgroup at the end of the grid, beside the gathers instead of behind
// them (it needs p_a + dp_a only, which it forms itself).
constexpr int BS_LANES = 4, BS_TRIP = 3;
constexpr int BS_TABLE_MAX = 256 * 9;      // camera parameters the table workgroup stages in LDS (the host builds the table here below 256 cameras)
template <int CNP, bool TWO_PASS /* the per-observation products come from k_backsub_obs (wobs); false: computed here */>
__global__ __launch_bounds__(256) void k_backsub(DevProblem P, double mu, const double* __restrict__ dpa,
        const double* __restrict__ pb, double* __restrict__ dpb, double* __restrict__ pdpb,
        double* __restrict__ part /* [3][fa.point_blocks] */, double* __restrict__ ptc_out /* camera-major mirror of pdpb, or null */,
        const double* __restrict__ wobs /* k_backsub_obs' products (camera-major), or null: computed here */,
        unsigned* __restrict__ ticket /* null: k_step_sums and k_cam_table follow */, StepFinalArgs fa)
{
    __shared__ double sm[3][4];
    const int nbp = fa.point_blocks;
    if ((int)blockIdx.x >= nbp) {
        // ---- the table workgroup: trial camera parameters into LDS, then the four parts of every camera's row
        __shared__ double s_pd[BS_TABLE_MAX];
        for (int t = threadIdx.x; t < fa.count; t += 256) s_pd[t] = fa.pa[t] + ((t < fa.fixed) ? 0.0 : dpa[t]);      // = pdp_a as step_sums_body forms it
        __syncthreads();
        for (int t = threadIdx.x; t < P.m * CT_PARTS; t += 256)
            cam_table_part(P.cfg, t % P.m, t / P.m, s_pd, P.Rinit, P.finit, fa.known, fa.with_fd, fa.camtab_trial);
        return;
    }
    constexpr int LPP = TWO_PASS ? 1 : BS_LANES;
    const int gi = blockIdx.x * 256 + threadIdx.x;
    const int ireal = gi / LPP, q = gi % LPP;
    const bool active = ireal < P.n && q == 0;               // the lane that owns the point's results
    const int i = ireal < P.n ? ireal : P.n - 1;             // (the other lanes of a quad, and quads past the end, compute along: the quad sums need every lane)
    double s_dp = 0.0, s_p = 0.0, s_dl = 0.0;
    {
        const double* g = P.eb + (size_t)i * 3;
        double w0 = 0, w1 = 0, w2 = 0;
        const int k0 = P.rowptr[i], k1 = P.rowptr[i + 1];
        // (everything that does not depend on the sum below is requested in front of it: the loop is a chain of dependent gathers)
        const double g0 = g[0], g1 = g[1], g2 = g[2];
        const double* vi = P.Vinv + (size_t)i * 6;
        const double vi0 = vi[0], vi1 = vi[1], vi2 = vi[2], vi3 = vi[3], vi4 = vi[4], vi5 = vi[5];
        const double p0 = pb[3 * (size_t)i], p1 = pb[3 * (size_t)i + 1], p2 = pb[3 * (size_t)i + 2];
        if (TWO_PASS) {
            for (int k = k0; k < k1; ++k) {
                const double2* wk = reinterpret_cast<const double2*>(wobs) + 2 * (size_t)P.campos[k];
                const double2 a = wk[0], b = wk[1];
                w0 += a.x; w1 += a.y; w2 += b.x;             // (a fixed camera's product is +0.0: the sum is what the skip below leaves)
            }
        } else {
            for (int kb = k0; kb < k1; kb += BS_LANES * BS_TRIP) {
                int jj[BS_TRIP], tt[BS_TRIP];
#pragma unroll
                for (int u = 0; u < BS_TRIP; ++u) { const int kc = min(kb + BS_LANES * u + q, k1 - 1); jj[u] = P.obs_cam[kc]; tt[u] = P.campos[kc]; }
                double A[BS_TRIP][2 * CNP], B[BS_TRIP][6], da[BS_TRIP][CNP];
#pragma unroll
                for (int u = 0; u < BS_TRIP; ++u) {
                    load_pairs<CNP>(P.Ac + (size_t)tt[u] * 2 * CNP, A[u]);
                    load_pairs<3>(P.Bc + (size_t)tt[u] * 8, B[u]);
#pragma unroll
                    for (int c = 0; c < CNP; ++c) da[u][c] = dpa[(size_t)jj[u] * CNP + c];
                }
#pragma unroll
                for (int u = 0; u < BS_TRIP; ++u) {
                    if (kb + BS_LANES * u + q < k1 && jj[u] >= P.mcon) {
                        double q0 = 0, q1 = 0;
#pragma unroll
                        for (int c = 0; c < CNP; ++c) { q0 += A[u][2 * c] * da[u][c]; q1 += A[u][2 * c + 1] * da[u][c]; }
                        w0 += B[u][0] * q0 + B[u][3] * q1; w1 += B[u][1] * q0 + B[u][4] * q1; w2 += B[u][2] * q0 + B[u][5] * q1;
                    }
                }
            }
            // the quad's four partial sums in a fixed order; every lane ends up with the point's sum
            w0 = (quad_bcast_d<0>(w0) + quad_bcast_d<1>(w0)) + (quad_bcast_d<2>(w0) + quad_bcast_d<3>(w0));
            w1 = (quad_bcast_d<0>(w1) + quad_bcast_d<1>(w1)) + (quad_bcast_d<2>(w1) + quad_bcast_d<3>(w1));
            w2 = (quad_bcast_d<0>(w2) + quad_bcast_d<1>(w2)) + (quad_bcast_d<2>(w2) + quad_bcast_d<3>(w2));
        }
        const double r0 = g0 - w0, r1 = g1 - w1, r2 = g2 - w2;
        const double d0 = vi0 * r0 + vi1 * r1 + vi2 * r2;
        const double d1 = vi1 * r0 + vi3 * r1 + vi4 * r2;
        const double d2 = vi2 * r0 + vi4 * r1 + vi5 * r2;
        if (active) {
            dpb[3 * (size_t)i] = d0; dpb[3 * (size_t)i + 1] = d1; dpb[3 * (size_t)i + 2] = d2;
            pdpb[3 * (size_t)i] = p0 + d0; pdpb[3 * (size_t)i + 1] = p1 + d1; pdpb[3 * (size_t)i + 2] = p2 + d2;
        }
        if (ptc_out && ireal < P.n) {      // the trial point to the camera-major positions of its observations (32-byte records; campos[] is L1 / L2 warm)
            const double2 v01 = make_double2(p0 + d0, p1 + d1), v2 = make_double2(p2 + d2, 0.0);
            for (int k = k0 + 4 * q; k < k1; k += 4 * LPP) {      // four positions per trip and lane (past the end: the last one again -- the same record twice)
                int pos[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) pos[u] = P.campos[min(k + u, k1 - 1)];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    double2* dst = reinterpret_cast<double2*>(ptc_out) + 2 * (size_t)pos[u];
                    dst[0] = v01; dst[1] = v2;
                }
            }
        }
        if (active) {
            s_dp = d0 * d0 + d1 * d1 + d2 * d2;
            s_p = p0 * p0 + p1 * p1 + p2 * p2;
            s_dl = d0 * (mu * d0 + g0) + d1 * (mu * d1 + g1) + d2 * (mu * d2 + g2);
        }
    }
    s_dp = wave_sum(s_dp); s_p = wave_sum(s_p); s_dl = wave_sum(s_dl);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sm[0][w] = s_dp; sm[1][w] = s_p; sm[2][w] = s_dl; }
    __syncthreads();
    if (threadIdx.x < 3)
        st_agent(part + (size_t)threadIdx.x * nbp + blockIdx.x,
                 (sm[threadIdx.x][0] + sm[threadIdx.x][1]) + (sm[threadIdx.x][2] + sm[threadIdx.x][3]));
    if (!ticket || !last_block_arrives(ticket, (unsigned)nbp, fa.ticket_groups)) return;
    // k_step_sums (camera part of the step, sums of the point partials), by the point workgroup that arrived last
    step_sums_body(fa.count, fa.fixed, mu, fa.pa, dpa, P.ea, fa.pdpa, fa.out3, part, nbp, fa.pt3, true);
}

// camera part of the step: pdp_a = p_a + dp_a and sum dpa^2, sum pa^2, sum dpa (mu dpa + ea) (single block).
__global__ __launch_bounds__(256) void k_cam_step(int count, int fixed, double mu, const double* __restrict__ pa,
        const double* __restrict__ dpa, const double* __restrict__ ea, double* __restrict__ pdpa, double* __restrict__ out3)
{
    __shared__ double sm[3][4];
    double s_dp = 0, s_p = 0, s_dl = 0;
#pragma unroll 4
    for (int t = threadIdx.x; t < count; t += 256) {
        const double d = (t < fixed) ? 0.0 : dpa[t], p = pa[t];
        pdpa[t] = p + d;
        s_dp += d * d; s_p += p * p; s_dl += d * (mu * d + ea[t]);
    }
    s_dp = wave_sum(s_dp); s_p = wave_sum(s_p); s_dl = wave_sum(s_dl);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sm[0][w] = s_dp; sm[1][w] = s_p; sm[2][w] = s_dl; }
    __syncthreads();
    if (threadIdx.x < 3) out3[threadIdx.x] = (sm[threadIdx.x][0] + sm[threadIdx.x][1]) + (sm[threadIdx.x][2] + sm[threadIdx.x][3]);
}

// ---------------------------------------------------------------------------------------------------
// deterministic second-stage reductions (single block, fixed order)
// constraint cost: sum_j w (c - p)^2 over constrained camera params (sba_levmar.c:809-826) and
// nvis * w * (c - p)^2 over constrained points (sba_levmar.c:829-842); single block.
__global__ __launch_bounds__(256) void k_constraint_cost(DevProblem P, const double* __restrict__ pa,
        const double* __restrict__ pb, int with_cams, double* __restrict__ out)
{
    __shared__ double sm[4];
    double s = 0.0;
    if (P.ccon && with_cams)
        for (int t = threadIdx.x; t < P.m * P.cfg.cnp; t += 256)
            if (P.ccon[t]) { const double d = P.cval[t] - pa[t]; s += P.cw[t] * d * d; }
    if (P.pcon)
        for (int i = threadIdx.x; i < P.n; i += 256)
            if (P.pcon[i]) {
                for (int q = 0; q < 3; ++q) { const double d = P.pval[3 * i + q] - pb[3 * i + q]; s += P.nvis_global * P.pweight * d * d; }
            }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *out = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// ---------------------------------------------------------------------------------------------------
// Post-solve statistics of RunSFM_SBA (src/Bundle.cpp:659-913) on the resident problem.
// dist = |x - proj(p)| per observation, stored in observation order and at its camera-major position
__global__ __launch_bounds__(256) void k_obs_dist(int nvis, const double* __restrict__ e, const int* __restrict__ campos,
                                                  double* __restrict__ dist, double* __restrict__ dist_cm)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nvis) return;
    const int t = campos[k];                                    // the residuals live in camera-major order
    const double2 ee = reinterpret_cast<const double2*>(e)[t];
    const double d = sqrt(ee.x * ee.x + ee.y * ee.y);
    dist[k] = d; dist_cm[t] = d;
}

// Per camera: the k-th smallest distance for k = iround(0.8 n) and iround(0.5 n) (kth_element_copy, lib/imagelib/
// qsort.c:152-203 -- returns 0.0 when k >= n, which happens for n <= 2), the mean, and the outlier threshold
// clamp(1.2 * 2.0 * kth80, min_thr, max_thr) (Bundle.cpp:761-771).  Exact selection without sorting: distances are >= 0,
// so their bit patterns order like unsigned integers; 8 passes of an 8-bit histogram narrow the prefix of the k-th key.
__global__ __launch_bounds__(256) void k_cam_dist_stats(int m, const int* __restrict__ camptr, const double* __restrict__ dist_cm,
        double min_thr, double max_thr, int* __restrict__ nobs, double* __restrict__ mean, double* __restrict__ kth80,
        double* __restrict__ kth50, double* __restrict__ thresh)
{
    __shared__ unsigned hist[256];
    __shared__ unsigned long long s_prefix;
    __shared__ int s_k;
    __shared__ double sm[4];
    const int j = blockIdx.x;
    const int t0 = camptr[j], n = camptr[j + 1] - t0;
    const unsigned long long* keys = reinterpret_cast<const unsigned long long*>(dist_cm) + t0;
    double res[2] = { 0.0, 0.0 };
    for (int which = 0; which < 2; ++which) {
        const double frac = which == 0 ? 0.8 : 0.5;
        const int kk = (int)(frac * n + 0.5);                  // iround, lib/imagelib/util.c:75-81 (n >= 0)
        if (kk >= n) continue;                                  // "[kth_element] Error: k should be < n" -> 0.0
        if (threadIdx.x == 0) { s_prefix = 0ull; s_k = kk; }
        __syncthreads();
        for (int shift = 56; shift >= 0; shift -= 8) {
            hist[threadIdx.x] = 0u;
            __syncthreads();
            const unsigned long long prefix = s_prefix;
            const unsigned long long himask = shift == 56 ? 0ull : (~0ull << (shift + 8));
            for (int t = threadIdx.x; t < n; t += 256) {
                const unsigned long long key = keys[t];
                if ((key & himask) == prefix) atomicAdd(&hist[(unsigned)(key >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                int k = s_k; unsigned b = 0;
                while (k >= (int)hist[b]) { k -= (int)hist[b]; ++b; }
                s_k = k; s_prefix = prefix | ((unsigned long long)b << shift);
            }
            __syncthreads();
        }
        res[which] = __longlong_as_double((long long)s_prefix);
        __syncthreads();
    }
    double sacc = 0.0;
    for (int t = threadIdx.x; t < n; t += 256) sacc += dist_cm[t0 + t];
    sacc = wave_sum(sacc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = sacc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double tot = (sm[0] + sm[1]) + (sm[2] + sm[3]);
        nobs[j] = n; mean[j] = n > 0 ? tot / n : 0.0;
        kth80[j] = res[0]; kth50[j] = res[1];
        double th = 1.2 * 2.0 * res[0];
        th = th < min_thr ? min_thr : (th > max_thr ? max_thr : th);
        thresh[j] = th;
    }
}

// A point is an outlier when one of its observations lies above its camera's threshold; the reference records the error
// of the first such observation in camera order (Bundle.cpp:806-821) = ascending camera index = CRS row order.
// Constrained points are exempt -- the reference only tests the x component of the constraint (Bundle.cpp:800-804).
__global__ __launch_bounds__(256) void k_point_outliers(int n, const int* __restrict__ rowptr, const int* __restrict__ obs_cam,
        const double* __restrict__ dist, const double* __restrict__ thresh, const double* __restrict__ pval,
        unsigned char* __restrict__ flag, double* __restrict__ err)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned char f = 0; double ev = 0.0;
    if (!(pval && pval[3 * (size_t)i] != 0.0)) {
        for (int k = rowptr[i]; k < rowptr[i + 1]; ++k)
            if (dist[k] > thresh[obs_cam[k]]) { f = 1; ev = dist[k]; break; }
    }
    flag[i] = f; err[i] = ev;
}

// Ray-angle pruning of RemoveBadPointsAndCameras (src/Bundle.cpp:4190-4261): per point the largest angle between the rays
// to two of its cameras, r = (X - c) * (1 / |X - c|) (matrix_diff, matrix_norm, matrix_scale), angle = acos(clamp(r1.r2,
// -1 + 1e-8, 1 - 1e-8)).  The unit rays are formed once per observation (the reference re-forms the second ray inside the
// pair loop -- same values); acos is monotone, so the largest angle is the acos of the smallest clamped dot product.
__global__ __launch_bounds__(256) void k_obs_rays(int nvis, int cnp, const int* __restrict__ obs_cam, const int* __restrict__ obs_pt,
        const double* __restrict__ pa, const double* __restrict__ pb, double* __restrict__ rays)
{
#pragma clang fp contract(off)
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nvis) return;
    const double* c = pa + (size_t)obs_cam[k] * cnp;
    const double* X = pb + (size_t)obs_pt[k] * 3;
    const double r0 = X[0] - c[0], r1 = X[1] - c[1], r2 = X[2] - c[2];
    double sum = 0.0;
    sum += r0 * r0; sum += r1 * r1; sum += r2 * r2;
    const double s = 1.0 / sqrt(sum);
    rays[3 * (size_t)k] = r0 * s; rays[3 * (size_t)k + 1] = r1 * s; rays[3 * (size_t)k + 2] = r2 * s;
}

__global__ __launch_bounds__(256) void k_point_ray_angle(int n, const int* __restrict__ rowptr, const double* __restrict__ rays,
        double half_thr_deg, double* __restrict__ angle_deg, unsigned char* __restrict__ prune)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int k0 = rowptr[i], k1 = rowptr[i + 1];
    const double hi = 1.0 - 1.0e-8, lo = -1.0 + 1.0e-8;
    bool any = false;
    double dmin = hi;
    for (int a = k0; a < k1; ++a) {
        const double a0 = rays[3 * (size_t)a], a1 = rays[3 * (size_t)a + 1], a2 = rays[3 * (size_t)a + 2];
        for (int b = a + 1; b < k1; ++b) {
            double dot = 0.0;
            dot += a0 * rays[3 * (size_t)b]; dot += a1 * rays[3 * (size_t)b + 1]; dot += a2 * rays[3 * (size_t)b + 2];
            dot = dot < lo ? lo : (dot > hi ? hi : dot);
            if (!any || dot < dmin) dmin = dot;
            any = true;
        }
    }
    // max_angle starts at 0.0 and only grows (Bundle.cpp:4205,4225-4227); acos(hi) > 0, so any pair sets it
    const double deg = any ? acos(dmin) * (180.0 / 3.14159265358979323846) : 0.0;
    angle_deg[i] = deg;
    prune[i] = (k1 > k0 && deg < half_thr_deg) ? 1 : 0;
}

// Camera-only refinement (points fixed): (U_j + mu I) da_j = ea_j, one thread per camera, Cholesky in registers
// (the reference calls sba_Axb_Chol on every U_j, sba_levmar.c:2499-2513); a non-positive pivot raises flag[1].
template <int CNP>
__global__ __launch_bounds__(64) void k_cam_solve(DevProblem P, double mu, double* __restrict__ dpa, int* __restrict__ flag)
{
    const int j = blockIdx.x * 64 + threadIdx.x;
    if (j >= P.m) return;
    double* out = dpa + (size_t)j * CNP;
    if (j < P.mcon) {
#pragma unroll
        for (int q = 0; q < CNP; ++q) out[q] = 0.0;
        return;
    }
    const double* U = P.U + (size_t)j * CNP * CNP;
    double l[CNP][CNP], x[CNP];
    bool bad = false;
#pragma unroll
    for (int c = 0; c < CNP; ++c) {
        double d = U[c * CNP + c] + mu;
#pragma unroll
        for (int q = 0; q < c; ++q) d -= l[c][q] * l[c][q];
        if (!(d > 0.0)) bad = true;
        const double s = sqrt(d);
        l[c][c] = s;
        const double inv = 1.0 / s;
#pragma unroll
        for (int r = c + 1; r < CNP; ++r) {
            double v = U[r * CNP + c];
#pragma unroll
            for (int q = 0; q < c; ++q) v -= l[r][q] * l[c][q];
            l[r][c] = v * inv;
        }
    }
    if (bad) { atomicOr(flag + 1, 1); return; }
#pragma unroll
    for (int r = 0; r < CNP; ++r) {          // L y = ea
        double v = P.ea[(size_t)j * CNP + r];
#pragma unroll
        for (int q = 0; q < r; ++q) v -= l[r][q] * x[q];
        x[r] = v / l[r][r];
    }
#pragma unroll
    for (int r = CNP - 1; r >= 0; --r) {     // L^T da = y
        double v = x[r];
#pragma unroll
        for (int q = r + 1; q < CNP; ++q) v -= l[q][r] * x[q];
        x[r] = v / l[r][r];
    }
#pragma unroll
    for (int q = 0; q < CNP; ++q) out[q] = x[q];
}

// ---------------------------------------------------------------------------------------------------
// Per-iteration scalars (sba_levmar.c:1085-1128).  The point side's block partials come from k_point_blocks (round 6: it was a kernel of
// its own, k_iter_partials); k_iter_final is the stand-alone finisher of the camera-only problem (have_points = 0).
__global__ __launch_bounds__(256) void k_iter_final(DevProblem P, const double* __restrict__ pa, const double* __restrict__ pb,
        const double* __restrict__ part, int have_points, int point_part_slot,
        int s_eabinf_a, int s_eabinf_b, int s_maxdiag_u, int s_maxdiag_v, int s_pl2_a, int s_pl2_b, int s_ccost,
        double* __restrict__ scal)
{
    iter_final_body(P, pa, pb, part, 0, have_points, point_part_slot, s_eabinf_a, s_eabinf_b, s_maxdiag_u, s_maxdiag_v, s_pl2_a, s_pl2_b, s_ccost, scal, false);
}
__device__ __forceinline__ void iter_final_body(const DevProblem& P, const double* __restrict__ pa, const double* __restrict__ pb,
        const double* __restrict__ part, int nparts, int have_points, int point_part_slot,
        int s_eabinf_a, int s_eabinf_b, int s_maxdiag_u, int s_maxdiag_v, int s_pl2_a, int s_pl2_b, int s_ccost,
        double* __restrict__ scal, bool agent_loads, int which)
{
    __shared__ double sm[4];
    const int cnp = P.cfg.cnp, t = threadIdx.x;
    if (have_points && (which & 1)) {      // part = [3][nparts]: the block partials of k_point_blocks, reduced in a fixed order
        double a = 0.0, v = 0.0, s = 0.0;                                         // (maxima of non-negative quantities: a 0-based maximum is exact)
        for (int q = t; q < nparts; q += 256) {
            const double pa_ = agent_loads ? ld_agent(part + q) : part[q];
            const double pv_ = agent_loads ? ld_agent(part + (size_t)nparts + q) : part[(size_t)nparts + q];
            a = pa_ > a ? pa_ : a; v = pv_ > v ? pv_ : v;
            s += agent_loads ? ld_agent(part + 2 * (size_t)nparts + q) : part[2 * (size_t)nparts + q];
        }
        const double ra = block_max4(a, sm), rv = block_max4(v, sm), rs = block_sum4(s, sm);
        if (t == 0) { scal[s_eabinf_b] = ra; scal[s_maxdiag_v] = rv; scal[s_pl2_b] = rs; }
    }
    if (!(which & 2)) return;
    double ea = 0.0, ud = -DBL_MAX, ps = 0.0;
#pragma unroll 4
    for (int q = t; q < P.m * cnp; q += 256) {
        const double x = fabs(P.ea[q]); ea = x > ea ? x : ea;
        ps += pa[q] * pa[q];
    }
#pragma unroll 4
    for (int q = P.mcon * cnp + t; q < P.m * cnp; q += 256) {
        const int j = q / cnp, jj = q % cnp;
        const double x = P.U[(size_t)j * cnp * cnp + jj * cnp + jj];
        ud = x > ud ? x : ud;
    }
    const double rea = block_max4(ea, sm), rud = block_max4(ud, sm), rps = block_sum4(ps, sm);
    // constraint cost: cameras (sba_levmar.c:809-826) + points (sba_levmar.c:829-842); the point part separately for the
    // multi-GPU sum (cameras are replicated, points sharded)
    double cc = 0.0, cp = 0.0;
    if (P.ccon)
        for (int q = t; q < P.m * cnp; q += 256)
            if (P.ccon[q]) { const double d = P.cval[q] - pa[q]; cc += P.cw[q] * d * d; }
    if (P.pcon)
        for (int i = t; i < P.n; i += 256)
            if (P.pcon[i])
                for (int q = 0; q < 3; ++q) { const double d = P.pval[3 * i + q] - pb[3 * i + q]; const double w = P.nvis_global * P.pweight * d * d; cc += w; cp += w; }
    const double rcc = block_sum4(cc, sm), rcp = block_sum4(cp, sm);
    if (t == 0) {
        scal[s_eabinf_a] = rea; scal[s_maxdiag_u] = rud; scal[s_pl2_a] = rps; scal[s_ccost] = rcc;
        if (point_part_slot >= 0) scal[point_part_slot] = rcp;
    }
}

// sum of the three block-partial rows of k_backsub + the camera part of the step in ONE launch (was four):
// out3 = sum dpa^2, sum pa^2, sum dpa (mu dpa + ea); pt3 = sums of the point partials.
__global__ __launch_bounds__(256) void k_step_sums(int count, int fixed, double mu, const double* __restrict__ pa,
        const double* __restrict__ dpa, const double* __restrict__ ea, double* __restrict__ pdpa, double* __restrict__ out3,
        const double* __restrict__ part, int nbp, double* __restrict__ pt3)
{
    step_sums_body(count, fixed, mu, pa, dpa, ea, pdpa, out3, part, nbp, pt3, false);
}

// cost sum and Snavely pct-change max of the residual pass in one launch
__global__ __launch_bounds__(256) void k_reduce_sum_max(const double* __restrict__ in_sum, const double* __restrict__ in_max,
        int count, double* __restrict__ out_sum, double* __restrict__ out_max)
{
    __shared__ double sm[4];
    double s = 0.0, m = 0.0;
#pragma unroll 8
    for (int t = threadIdx.x; t < count; t += 256) { s += in_sum[t]; if (in_max) { const double v = in_max[t]; m = v > m ? v : m; } }
    const double rs = block_sum4(s, sm), rm = block_max4(m, sm);
    if (threadIdx.x == 0) { *out_sum = rs; if (in_max) *out_max = rm; }
}

// expand packed V (+mu) to the reference's full symmetric 3x3 per point (test/export helper)
__global__ void k_expand_v(int n, double mu, const double* __restrict__ V, double* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* v = V + (size_t)i * 6;
    double* o = out + (size_t)i * 9;
    o[0] = v[0] + mu; o[1] = v[1]; o[2] = v[2];
    o[3] = v[1]; o[4] = v[3] + mu; o[5] = v[4];
    o[6] = v[2]; o[7] = v[4]; o[8] = v[5] + mu;
}

}  // namespace bsfm
