// compsolve.hip.h -- reduced camera system that falls apart into independent camera groups.
//
// The reference always factors the dense (m*cnp)^2 matrix S (sba_Axb_Chol, lib/sba-1.5/sba_lapack.c:374-485).  S_jk is
// non-zero only when cameras j and k share a point (lib/sba-1.5/sba_levmar.c:1231-1302); when the co-visibility graph has
// several connected components, S is block diagonal up to a permutation of the cameras and Cholesky creates no fill between
// the components: every group can be solved on its own, with the same result up to summation order.  This is what the
// synthetic generator of SURVEY 8(d) produces (cameras (j0 + d m/deg) mod m: m/deg... groups of deg cameras); real
// reconstructions are connected, so this path is OPT-IN (bsfm_options_t.reduced_solver = BSFM_SOLVER_AUTO) and the dense
// MFMA Cholesky of potrf.hip.h stays the default and the benchmarked path.
//
// One workgroup per group: the group's rows/columns of S (lower triangle) are gathered into LDS, factored there
// (right-looking, column by column, 16 x 16 thread tiling of the trailing update), forward/backward substitution on the
// gathered right-hand side, scatter of the solution.  Groups up to COMP_MAX_DIM unknowns (LDS-resident); a problem with a
// larger group uses the dense path.  A non-positive pivot reports its 1-based row in S through `info`, like dpotrf.
#pragma once
#include "potrf.hip.h"

namespace bsfm {

constexpr int COMP_MAX_DIM = 128;

__global__ __launch_bounds__(256) void k_comp_solve(int cnp, int maxdim, int stride, const int* __restrict__ comp_ptr,
        const int* __restrict__ comp_cams, const double* __restrict__ S, int ld, const double* __restrict__ E,
        double* __restrict__ x, int* __restrict__ info)
{
    extern __shared__ double cs_sm[];
    double* A = cs_sm;                               // n x n, row stride `stride` (odd: column walks are conflict-free)
    double* b = A + (size_t)maxdim * stride;
    int* gi = reinterpret_cast<int*>(b + maxdim);    // row of S behind each local row
    const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
    const int c0 = comp_ptr[blockIdx.x], nc = comp_ptr[blockIdx.x + 1] - c0, n = nc * cnp;
    for (int r = tid; r < n; r += 256) {
        const int cam = r / cnp;
        gi[r] = comp_cams[c0 + cam] * cnp + (r - cam * cnp);
    }
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += 256) {
        const int r = idx / n, cc = idx - r * n;
        if (cc <= r) A[r * stride + cc] = S[(size_t)gi[r] * ld + gi[cc]];
    }
    for (int r = tid; r < n; r += 256) b[r] = E[gi[r]];
    bool failed = false;
    for (int k = 0; k < n; ++k) {
        __syncthreads();
        const double d = A[k * stride + k];
        if (!(d > 0.0)) { failed = true; if (tid == 0) atomicCAS(info, 0, gi[k] + 1); break; }   // uniform: everybody reads the same LDS word
        const double s = sqrt(d);
        __syncthreads();
        for (int i = k + tid; i < n; i += 256) A[i * stride + k] = (i == k) ? s : A[i * stride + k] / s;
        __syncthreads();
        for (int i = k + 1 + ti; i < n; i += 16) {
            const double lik = A[i * stride + k];
            for (int j = k + 1 + tj; j <= i; j += 16) A[i * stride + j] -= lik * A[j * stride + k];
        }
    }
    if (failed) return;
    for (int k = 0; k < n; ++k) {                    // L y = b
        __syncthreads();
        const double yk = b[k] / A[k * stride + k];
        __syncthreads();
        for (int i = k + tid; i < n; i += 256) b[i] = (i == k) ? yk : b[i] - A[i * stride + k] * yk;
    }
    for (int k = n - 1; k >= 0; --k) {               // L^T x = y
        __syncthreads();
        const double xk = b[k] / A[k * stride + k];
        __syncthreads();
        for (int i = tid; i <= k; i += 256) b[i] = (i == k) ? xk : b[i] - A[k * stride + i] * xk;
    }
    __syncthreads();
    for (int r = tid; r < n; r += 256) x[gi[r]] = b[r];
}

struct CompSolver {
    int ncomp = 0, maxdim = 0, stride = 0;
    size_t lds = 0;
    int *d_ptr = nullptr, *d_cams = nullptr;
    bool active = false;
};

inline void comp_free(CompSolver& cs)
{
    bsfm::dev_free(cs.d_ptr, true);
    bsfm::dev_free(cs.d_cams, true);
    cs = CompSolver();
}

// Connected components of the block structure (bj[b], bk[b]) over `mm` cameras (indices relative to the first free camera).
// Returns 0 and leaves cs.active false when the dense path should be used (one component, or a group too large for LDS).
inline int comp_setup(CompSolver& cs, int mm, int cnp, const std::vector<int>& bj, const std::vector<int>& bk, int mcon)
{
    comp_free(cs);
    if (mm <= 1) return 0;
    std::vector<int> parent(mm);
    for (int j = 0; j < mm; ++j) parent[j] = j;
    auto find = [&](int a) { while (parent[a] != a) { parent[a] = parent[parent[a]]; a = parent[a]; } return a; };
    for (size_t b = 0; b < bj.size(); ++b) {
        const int ra = find(bj[b] - mcon), rb = find(bk[b] - mcon);
        if (ra != rb) parent[std::max(ra, rb)] = std::min(ra, rb);
    }
    std::vector<int> label(mm, -1), count;
    for (int j = 0; j < mm; ++j) {                  // groups numbered by their first camera, members ascending
        const int r = find(j);
        if (label[r] < 0) { label[r] = (int)count.size(); count.push_back(0); }
        ++count[label[r]];
    }
    const int nc = (int)count.size();
    if (nc <= 1) return 0;
    int maxc = 0;
    for (int c : count) maxc = std::max(maxc, c);
    if (maxc * cnp > COMP_MAX_DIM) return 0;
    std::vector<int> ptr(nc + 1, 0), cams(mm);
    for (int c = 0; c < nc; ++c) ptr[c + 1] = ptr[c] + count[c];
    std::vector<int> cur(ptr.begin(), ptr.end() - 1);
    for (int j = 0; j < mm; ++j) cams[cur[label[find(j)]]++] = j;
    if (bsfm::dev_alloc((void**)&cs.d_ptr, (size_t)(nc + 1) * sizeof(int)) != hipSuccess || bsfm::dev_alloc((void**)&cs.d_cams, (size_t)mm * sizeof(int)) != hipSuccess) return -1;
    if (hipMemcpy(cs.d_ptr, ptr.data(), (size_t)(nc + 1) * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(cs.d_cams, cams.data(), (size_t)mm * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return -1;
    cs.ncomp = nc; cs.maxdim = maxc * cnp; cs.stride = cs.maxdim | 1;
    cs.lds = ((size_t)cs.maxdim * cs.stride + cs.maxdim) * sizeof(double) + (size_t)cs.maxdim * sizeof(int);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_comp_solve), hipFuncAttributeMaxDynamicSharedMemorySize, (int)cs.lds) != hipSuccess) return -1;
    cs.active = true;
    return 0;
}

// Same contract as potrf_solve: S x = E for the n = (m - mcon) cnp free unknowns, info = 0 or a failing row (1-based).
inline int comp_solve(const CompSolver& cs, PotrfWorkspace& w, int cnp, const double* S, int ld, const double* E, double* x_out,
                      int* d_info, hipStream_t st)
{
    if (w.ev0) (void)hipEventRecord(w.ev0, st);
    (void)hipGetLastError();                          // drop stale state (e.g. hipErrorNotReady of an earlier event query)
    hipLaunchKernelGGL(k_comp_solve, dim3(cs.ncomp), dim3(256), cs.lds, st, cnp, cs.maxdim, cs.stride, cs.d_ptr, cs.d_cams,
                       S, ld, E, x_out, d_info);
    if (w.ev1) (void)hipEventRecord(w.ev1, st);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace bsfm
