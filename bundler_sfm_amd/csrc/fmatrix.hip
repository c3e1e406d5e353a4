// fmatrix.hip -- batched epipolar geometry, gfx950.
//
// Replaces, over a batch of image pairs, the reference's
//   estimate_fmatrix_ransac_matches   lib/imagelib/fmatrix.c:293-475   8-point RANSAC, inlier = Sampson-type residual < thr
//   estimate_fmatrix_linear           lib/imagelib/fmatrix.c:729-890   (8 points: normalise, dgesv, un-normalise, rank 2)
//   fmatrix_compute_residual          lib/imagelib/fmatrix.c:63-87
//   refine_fmatrix_nonlinear_matches  lib/imagelib/fmatrix.c:637-659   lmdif on 8 entries of F, residual = sqrt of the above
//                                                                      on the rank-2 projection (fmatrix.c:494-524), tol 1e-12
// which BundlerApp::ComputeEpipolarGeometry (src/BundlerGeometry.cpp:330-386) reaches through EstimateFMatrix
// (src/Epipolar.cpp:118-237) once per image pair, 2048 trials each (BundlerApp.h:63-64).
//
// The reference draws its samples with rand(); so that results can be compared trial for trial, the generator is restated
// on the host (glibc random(), TYPE_3: r[i] = r[i-3] + r[i-31], output >> 1; seeding by the 16807 Lehmer step, 310 outputs
// discarded) and the sample loop of fmatrix.c:343-375 with it -- including the re-draws on repeated indices / coordinates and
// the early exit once a trial's inlier ratio exceeds success_ratio, after which the stream continues with the next pair
// exactly where the reference's would.
//
// MI355X design: one thread per (pair, trial): the 8 x 8 system is solved in registers (LU with partial pivoting, the order of
// operations of dgetf2 / dgetrs), the rank-2 projection is a one-sided Jacobi SVD of the 3 x 3 matrix, and the thread then
// walks over all matches of its pair (the 64 lanes of a wave read the same match: broadcast loads) counting residuals below
// the threshold.  Trials are independent; the host picks, per pair, the first trial with the largest count among the trials
// the reference would have run.  The non-linear refinement is one WAVE per pair on lmdif.hip.h (8 unknowns): each Jacobian
// pass projects the 9 perturbed matrices to rank 2 once, the lanes share the pair's inliers and merge their triangular
// factors with a butterfly TSQR (FmFcnWave).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/bsfm.h"
#include "lmdif.hip.h"

// ------------------------------------------------------------------------------------------------ host: rand() restated
extern "C" void bsfm_rand_seed(bsfm_rand_t* st, unsigned int seed)
{
    // glibc srandom_r for TYPE_3 (degree 31, separation 3)
    if (seed == 0) seed = 1;
    int32_t word = (int32_t)seed;
    st->s[0] = (unsigned int)word;
    for (int i = 1; i < 31; ++i) {
        const long hi = word / 127773, lo = word % 127773;
        long w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        word = (int32_t)w;
        st->s[i] = (unsigned int)word;
    }
    st->fi = 3; st->ri = 0;
    for (int i = 0; i < 310; ++i) (void)bsfm_rand_next(st);
}

extern "C" int bsfm_rand_next(bsfm_rand_t* st)
{
    st->s[st->fi] += st->s[st->ri];
    const int r = (int)((st->s[st->fi] >> 1) & 0x7fffffff);
    if (++st->fi >= 31) st->fi = 0;
    if (++st->ri >= 31) st->ri = 0;
    return r;
}

namespace {

// ------------------------------------------------------------------------------------------------ device
// residual of fmatrix.c:63-87 for r = (rx, ry, 1), l = (lx, ly, 1)
__device__ __forceinline__ double fm_residual(const double* F, double rx, double ry, double lx, double ly)
{
#pragma clang fp contract(off)
    const double Fl0 = F[0] * lx + F[1] * ly + F[2] * 1.0;
    const double Fl1 = F[3] * lx + F[4] * ly + F[5] * 1.0;
    const double Fl2 = F[6] * lx + F[7] * ly + F[8] * 1.0;
    const double Fr0 = F[0] * rx + F[3] * ry + F[6] * 1.0;
    const double Fr1 = F[1] * rx + F[4] * ry + F[7] * 1.0;
    const double pt = rx * Fl0 + ry * Fl1 + 1.0 * Fl2;
    return (1.0 / (Fl0 * Fl0 + Fl1 * Fl1) + 1.0 / (Fr0 * Fr0 + Fr1 * Fr1)) * (pt * pt);
}

// closest_rank2_matrix (fmatrix.c:687-705): drop the smallest singular value.  One-sided Jacobi: A V = B with orthogonal
// columns b_i = sigma_i u_i, so A = sum_i b_i v_i^T and the projection is A - b_min v_min^T.
__device__ void fm_rank2(const double* Fin, double* Fout)
{
    double A[3][3], V[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { A[i][j] = Fin[3 * i + j]; V[i][j] = (i == j) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            double al = 0.0, be = 0.0, ga = 0.0;
            for (int i = 0; i < 3; ++i) { al += A[i][p] * A[i][p]; be += A[i][q] * A[i][q]; ga += A[i][p] * A[i][q]; }
            if (ga == 0.0 || fabs(ga) <= 1e-17 * sqrt(al * be)) continue;
            rotated = true;
            const double zeta = (be - al) / (2.0 * ga);
            const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
            const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
            for (int i = 0; i < 3; ++i) {
                const double ap = A[i][p], aq = A[i][q];
                A[i][p] = c * ap - s * aq; A[i][q] = s * ap + c * aq;
                const double vp = V[i][p], vq = V[i][q];
                V[i][p] = c * vp - s * vq; V[i][q] = s * vp + c * vq;
            }
        }
        if (!rotated) break;
    }
    double n0 = 0.0, n1 = 0.0, n2 = 0.0;
    for (int i = 0; i < 3; ++i) { n0 += A[i][0] * A[i][0]; n1 += A[i][1] * A[i][1]; n2 += A[i][2] * A[i][2]; }
    const int k = (n0 <= n1 && n0 <= n2) ? 0 : (n1 <= n2 ? 1 : 2);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double v = 0.0;
            for (int c = 0; c < 3; ++c) if (c != k) v += A[i][c] * V[j][c];
            Fout[3 * i + j] = v;
        }
}

// estimate_fmatrix_linear for exactly 8 correspondences (fmatrix.c:729-890).  r = first, l = second point set.
__device__ void fm_linear8(const double* rx, const double* ry, const double* lx, const double* ly, double* Fout)
{
#pragma clang fp contract(off)
    double rcx = 0.0, rcy = 0.0, lcx = 0.0, lcy = 0.0;
    for (int i = 0; i < 8; ++i) { rcx += rx[i]; rcy += ry[i]; lcx += lx[i]; lcy += ly[i]; }
    rcx *= 1.0 / 8; rcy *= 1.0 / 8; lcx *= 1.0 / 8; lcy *= 1.0 / 8;        // the z components stay 1
    double rd = 0.0, ldist = 0.0;
    for (int i = 0; i < 8; ++i) {
        const double ax = rcx - rx[i], ay = rcy - ry[i], bx = lcx - lx[i], by = lcy - ly[i];
        rd += sqrt(ax * ax + ay * ay + 0.0); ldist += sqrt(bx * bx + by * by + 0.0);
    }
    rd /= 8; ldist /= 8;
    rd /= sqrt(2.0); ldist /= sqrt(2.0);
    const double rs = 1.0 / rd, ls = 1.0 / ldist;
    double a[8][8], b[8];
    for (int i = 0; i < 8; ++i) {
        const double up = rs * (rx[i] - rcx), vp = rs * (ry[i] - rcy), u = ls * (lx[i] - lcx), v = ls * (ly[i] - lcy);
        a[i][0] = u * up; a[i][1] = v * up; a[i][2] = up; a[i][3] = u * vp; a[i][4] = v * vp; a[i][5] = vp; a[i][6] = u; a[i][7] = v;
        b[i] = -1.0;
    }
    // dgesv: LU with partial pivoting (dgetf2: first largest |a_ij| of the column, reciprocal scaling), then dgetrs
    bool singular = false;
    for (int j = 0; j < 8; ++j) {
        int piv = j; double big = fabs(a[j][j]);
        for (int i = j + 1; i < 8; ++i) if (fabs(a[i][j]) > big) { big = fabs(a[i][j]); piv = i; }
        if (a[piv][j] != 0.0) {
            if (piv != j) {
                for (int c = 0; c < 8; ++c) { const double t = a[j][c]; a[j][c] = a[piv][c]; a[piv][c] = t; }
                const double t = b[j]; b[j] = b[piv]; b[piv] = t;
            }
            const double rcp = 1.0 / a[j][j];
            for (int i = j + 1; i < 8; ++i) a[i][j] *= rcp;
        } else singular = true;
        for (int i = j + 1; i < 8; ++i)
            for (int c = j + 1; c < 8; ++c) a[i][c] -= a[i][j] * a[j][c];
    }
    double X[8];
    if (singular) {
        for (int i = 0; i < 8; ++i) X[i] = -1.0;                               // dgesv leaves b untouched (info > 0), matrix.c:913-921
    } else {
        for (int j = 0; j < 8; ++j)                                            // L y = P b (unit lower, column sweep)
            for (int i = j + 1; i < 8; ++i) b[i] -= b[j] * a[i][j];
        for (int j = 7; j >= 0; --j) {                                         // U x = y
            b[j] /= a[j][j];
            for (int i = 0; i < j; ++i) b[i] -= b[j] * a[i][j];
        }
        for (int i = 0; i < 8; ++i) X[i] = b[i];
    }
    // un-normalise: F_new = H_p F H (matrix_product: sums in index order)
    const double F[9] = { X[0], X[1], X[2], X[3], X[4], X[5], X[6], X[7], 1.0 };
    const double H[9] = { ls, 0.0, -ls * lcx, 0.0, ls, -ls * lcy, 0.0, 0.0, 1.0 };
    const double Hp[9] = { rs, 0.0, 0.0, 0.0, rs, 0.0, -rs * rcx, -rs * rcy, 1.0 };
    double tmp[9], Fn[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += Hp[3 * i + k] * F[3 * k + j];
            tmp[3 * i + j] = s;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += tmp[3 * i + k] * H[3 * k + j];
            Fn[3 * i + j] = s;
        }
    fm_rank2(Fn, Fout);
}

// one thread per (pair, trial); samples: 8 match indices (relative to the pair) per trial, -1 = trial not drawn
__global__ __launch_bounds__(64) void k_fm_ransac(int npairs, int ntrials, const int* __restrict__ match_ptr,
        const double* __restrict__ a_xy, const double* __restrict__ b_xy, const int* __restrict__ samples, double threshold,
        int* __restrict__ counts, double* __restrict__ Fall)
{
    const int blocks_per_pair = (ntrials + 63) / 64;
    const int pair = blockIdx.x / blocks_per_pair, trial = (blockIdx.x % blocks_per_pair) * 64 + threadIdx.x;
    if (pair >= npairs || trial >= ntrials) return;
    const size_t slot = (size_t)pair * ntrials + trial;
    const int* sm = samples + slot * 8;
    const int m0 = match_ptr[pair], m1 = match_ptr[pair + 1];
    if (sm[0] < 0) { counts[slot] = -1; return; }
    double rx[8], ry[8], lx[8], ly[8];
    for (int j = 0; j < 8; ++j) {
        const size_t q = (size_t)(m0 + sm[j]);
        rx[j] = a_xy[2 * q]; ry[j] = a_xy[2 * q + 1]; lx[j] = b_xy[2 * q]; ly[j] = b_xy[2 * q + 1];
    }
    double F[9];
    fm_linear8(rx, ry, lx, ly, F);
    bool nan = false;
    for (int j = 0; j < 9; ++j) nan = nan || (F[j] != F[j]);
    int cnt = 0;
    if (!nan)
        for (int q = m0; q < m1; ++q)
            cnt += fm_residual(F, a_xy[2 * (size_t)q], a_xy[2 * (size_t)q + 1], b_xy[2 * (size_t)q], b_xy[2 * (size_t)q + 1]) < threshold ? 1 : 0;
    counts[slot] = cnt;
    for (int j = 0; j < 9; ++j) Fall[slot * 9 + j] = F[j];
}

// fmatrix_residuals (fmatrix.c:494-524) over the matches whose residual under F0 is below the threshold -- the inlier list
// EstimateFMatrix builds (src/Epipolar.cpp:153-169) -- without materialising that list.
struct FmFcn {
    const double* a;        // first point set (r), 2 per match
    const double* b;        // second (l)
    int n;                  // matches of the pair
    double F0[9], thr, scale;
    int m;                  // inliers of F0
    __device__ bool in(int q) const { return fm_residual(F0, a[2 * q], a[2 * q + 1], b[2 * q], b[2 * q + 1]) < thr; }
    __device__ int rows() const { return m; }
    __device__ void project(const double* x, double* F2) const
    {
        double F[9];
        for (int j = 0; j < 8; ++j) F[j] = x[j];
        F[8] = scale;
        fm_rank2(F, F2);
    }
    __device__ double fnorm(const double* x) const
    {
        double F2[9];
        project(x, F2);
        double s = 0.0;
        for (int q = 0; q < n; ++q)
            if (in(q)) { const double f = sqrt(fm_residual(F2, a[2 * q], a[2 * q + 1], b[2 * q], b[2 * q + 1])); s += f * f; }
        return sqrt(s);
    }
    __device__ void jac_qr(const double* x, bsfm_lm::QrN<8>& Q) const
    {
        const double eps = sqrt(bsfm_lm::LM_EPSMCH);
        double h[8], F2[9][9], xp[8];
        project(x, F2[8]);
        for (int j = 0; j < 8; ++j) {
            h[j] = eps * fabs(x[j]); if (h[j] == 0.0) h[j] = eps;
            for (int k = 0; k < 8; ++k) xp[k] = x[k];
            xp[j] = x[j] + h[j];
            project(xp, F2[j]);
        }
        Q.clear();
        for (int q = 0; q < n; ++q) {
            if (!in(q)) continue;
            const double rx = a[2 * q], ry = a[2 * q + 1], lx = b[2 * q], ly = b[2 * q + 1];
            const double f0 = sqrt(fm_residual(F2[8], rx, ry, lx, ly));
            double row[8];
            for (int j = 0; j < 8; ++j) row[j] = (sqrt(fm_residual(F2[j], rx, ry, lx, ly)) - f0) / h[j];
            Q.add_row(row, f0);
        }
    }
};

// The same functor for one WAVE per pair: every lane runs lm_lmdif<8> on identical data (uniform control flow); only the two
// passes over the matches are shared out -- lane l takes the matches l, l + 64, ... -- and combined with xor-butterflies in
// which both partners of a step compute the SAME expression (sum a + b resp. fold the factor of the higher lane into the one
// of the lower lane), so that all lanes hold bit-identical results afterwards and never diverge.  The butterfly on the
// triangular factors is a TSQR: R of the stacked [R_lo; R_hi] by folding the 8 rows of R_hi into R_lo with Givens rotations.
struct FmFcnWave : FmFcn {
    int lane;
    __device__ static double wsum(double v)
    {
        for (int s = 1; s < 64; s <<= 1) v += __shfl_xor(v, s, 64);
        return v;
    }
    __device__ double fnorm(const double* x) const
    {
        double F2[9];
        project(x, F2);
        double s = 0.0;
        for (int q = lane; q < n; q += 64)
            if (in(q)) { const double f = sqrt(fm_residual(F2, a[2 * q], a[2 * q + 1], b[2 * q], b[2 * q + 1])); s += f * f; }
        return sqrt(wsum(s));
    }
    __device__ void jac_qr(const double* x, bsfm_lm::QrN<8>& Q) const
    {
        const double eps = sqrt(bsfm_lm::LM_EPSMCH);
        double h[8], F2[9][9], xp[8];
        project(x, F2[8]);
        for (int j = 0; j < 8; ++j) {
            h[j] = eps * fabs(x[j]); if (h[j] == 0.0) h[j] = eps;
            for (int k = 0; k < 8; ++k) xp[k] = x[k];
            xp[j] = x[j] + h[j];
            project(xp, F2[j]);
        }
        Q.clear();
        for (int q = lane; q < n; q += 64) {
            if (!in(q)) continue;
            const double rx = a[2 * q], ry = a[2 * q + 1], lx = b[2 * q], ly = b[2 * q + 1];
            const double f0 = sqrt(fm_residual(F2[8], rx, ry, lx, ly));
            double row[8];
            for (int j = 0; j < 8; ++j) row[j] = (sqrt(fm_residual(F2[j], rx, ry, lx, ly)) - f0) / h[j];
            Q.add_row(row, f0);
        }
        for (int s = 1; s < 64; s <<= 1) {
            bsfm_lm::QrN<8> O;                                  // the partner's factor
            for (int i = 0; i < 8; ++i) {
                O.q[i] = __shfl_xor(Q.q[i], s, 64);
                for (int j = 0; j < 8; ++j) O.r[i][j] = j >= i ? __shfl_xor(Q.r[i][j], s, 64) : 0.0;
            }
            const bool low = (lane & s) == 0;                   // this lane is the lower one of the pair
            bsfm_lm::QrN<8> base, top;
            for (int i = 0; i < 8; ++i) {
                base.q[i] = low ? Q.q[i] : O.q[i]; top.q[i] = low ? O.q[i] : Q.q[i];
                for (int j = 0; j < 8; ++j) { base.r[i][j] = low ? Q.r[i][j] : O.r[i][j]; top.r[i][j] = low ? O.r[i][j] : Q.r[i][j]; }
            }
            for (int i = 0; i < 8; ++i) {
                double row[8];
                for (int j = 0; j < 8; ++j) row[j] = top.r[i][j];
                base.add_row(row, top.q[i]);
            }
            Q = base;
        }
    }
};

// One wave per pair (see FmFcnWave); count < 0 on entry marks a pair to skip.
__global__ __launch_bounds__(64) void k_fm_refine_wave(int npairs, const int* __restrict__ match_ptr, const double* __restrict__ a_xy,
        const double* __restrict__ b_xy, double threshold, double* __restrict__ F, int* __restrict__ count,
        unsigned char* __restrict__ inlier, int* __restrict__ info_out)
{
    const int p = blockIdx.x, lane = threadIdx.x;
    if (p >= npairs) return;
    const int m0 = match_ptr[p], n = match_ptr[p + 1] - m0;
    if (count[p] < 0) {                                         // uniform: every lane reads the same word
        for (int q = lane; q < n; q += 64) inlier[m0 + q] = 0;
        if (lane == 0) count[p] = 0;
        return;
    }
    FmFcnWave fcn;
    fcn.lane = lane;
    fcn.a = a_xy + 2 * (size_t)m0; fcn.b = b_xy + 2 * (size_t)m0; fcn.n = n; fcn.thr = threshold;
    for (int j = 0; j < 9; ++j) fcn.F0[j] = F[9 * (size_t)p + j];
    fcn.scale = fcn.F0[8];
    double mm = 0.0;
    for (int q = lane; q < n; q += 64) mm += fcn.in(q) ? 1.0 : 0.0;
    fcn.m = (int)FmFcnWave::wsum(mm);
    double x[8];
    for (int j = 0; j < 8; ++j) x[j] = fcn.F0[j];
    const int info = bsfm_lm::lm_lmdif<8>(fcn, x, 1.0e-12);
    double Fo[9];
    fcn.project(x, Fo);
    double c = 0.0;
    for (int q = lane; q < n; q += 64) {
        const bool in = fm_residual(Fo, fcn.a[2 * q], fcn.a[2 * q + 1], fcn.b[2 * q], fcn.b[2 * q + 1]) < threshold;
        inlier[m0 + q] = in ? 1 : 0; c += in ? 1.0 : 0.0;
    }
    c = FmFcnWave::wsum(c);
    __syncthreads();                                            // all lanes are done reading F0 / count before they are overwritten
    if (lane == 0) {
        for (int j = 0; j < 9; ++j) F[9 * (size_t)p + j] = Fo[j];
        count[p] = (int)c;
        if (info_out) info_out[p] = info;
    }
}

// EstimateFMatrix after the RANSAC (src/Epipolar.cpp:153-231): inliers of F, refinement on them, inliers of the result.
// One thread per pair (kept as a cross-check, BSFM_FM_REFINE=thread); count < 0 on entry marks a pair to skip.
__global__ __launch_bounds__(64) void k_fm_refine(int npairs, const int* __restrict__ match_ptr, const double* __restrict__ a_xy,
        const double* __restrict__ b_xy, double threshold, double* __restrict__ F, int* __restrict__ count,
        unsigned char* __restrict__ inlier, int* __restrict__ info_out)
{
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= npairs) return;
    const int m0 = match_ptr[p], n = match_ptr[p + 1] - m0;
    if (count[p] < 0) { count[p] = 0; for (int q = 0; q < n; ++q) inlier[m0 + q] = 0; return; }
    FmFcn fcn;
    fcn.a = a_xy + 2 * (size_t)m0; fcn.b = b_xy + 2 * (size_t)m0; fcn.n = n; fcn.thr = threshold;
    for (int j = 0; j < 9; ++j) fcn.F0[j] = F[9 * (size_t)p + j];
    fcn.scale = fcn.F0[8];
    int m = 0;
    for (int q = 0; q < n; ++q) m += fcn.in(q) ? 1 : 0;
    fcn.m = m;
    double x[8];
    for (int j = 0; j < 8; ++j) x[j] = fcn.F0[j];
    const int info = bsfm_lm::lm_lmdif<8>(fcn, x, 1.0e-12);
    double Fo[9];
    fcn.project(x, Fo);                                   // Ftmp[8] = global_scale; closest_rank2_matrix (fmatrix.c:652-655)
    int c = 0;
    for (int q = 0; q < n; ++q) {
        const bool in = fm_residual(Fo, fcn.a[2 * q], fcn.a[2 * q + 1], fcn.b[2 * q], fcn.b[2 * q + 1]) < threshold;
        inlier[m0 + q] = in ? 1 : 0; c += in ? 1 : 0;
    }
    for (int j = 0; j < 9; ++j) F[9 * (size_t)p + j] = Fo[j];
    count[p] = c;
    if (info_out) info_out[p] = info;
}

template <typename T> struct DevBuf {
    T* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    bool alloc(size_t n) { if (p) { (void)hipFree(p); p = nullptr; } return hipMalloc(&p, (n ? n : 1) * sizeof(T)) == hipSuccess; }
    bool up(const T* src, size_t n) { return n == 0 || hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice) == hipSuccess; }
    bool down(T* dst, size_t n, size_t off = 0) { return n == 0 || hipMemcpy(dst, p + off, n * sizeof(T), hipMemcpyDeviceToHost) == hipSuccess; }
};

// The sample loop of fmatrix.c:343-375 for all trials of one pair.  Returns false when the reference would have given up
// (1000 re-draws inside one trial: estimate_fmatrix_ransac_matches returns 0 at once); `after[t]` = rand() calls consumed up to and including
// trial t.
bool draw_samples(bsfm_rand_t& rng, int n, const double* a, const double* b, int ntrials, int* out, std::vector<int>& after,
                  int* gave_up_at)
{
    after.resize(ntrials);
    int drawn = 0;
    for (int t = 0; t < ntrials; ++t) {
        int* idxs = out + (size_t)t * 8;
        int round = 0;
        for (int j = 0; j < 8; ++j) {
            if (round == 1000) { for (int q = t; q < ntrials; ++q) out[(size_t)q * 8] = -1; *gave_up_at = t; return false; }
            const int idx = bsfm_rand_next(&rng) % n; ++drawn;
            bool reselect = false;
            for (int k = 0; k < j; ++k) {
                const int o = idxs[k];
                if (idx == o || (a[2 * idx] == a[2 * o] && a[2 * idx + 1] == a[2 * o + 1]) || (b[2 * idx] == b[2 * o] && b[2 * idx + 1] == b[2 * o + 1])) { reselect = true; break; }
            }
            if (reselect) { ++round; --j; continue; }
            idxs[j] = idx;
        }
        after[t] = drawn;
    }
    *gave_up_at = -1;
    return true;
}

}  // namespace

extern "C" int bsfm_fmatrix_ransac_batch(int npairs, const int* match_ptr, const double* a_xy, const double* b_xy, int num_trials,
                                         double threshold, double success_ratio, bsfm_rand_t* rng, double* F, int* inliers_max)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fprintf(stderr, "[bsfm] FATAL: no usable HIP device; the MI355X path has no CPU fallback\n");
        return BSFM_ERROR;
    }
    if (npairs < 0 || num_trials <= 0 || !match_ptr || !rng || (npairs > 0 && (!a_xy || !b_xy || !F || !inliers_max))) {
        fprintf(stderr, "[bsfm] fmatrix ransac: bad arguments\n");
        return BSFM_ERROR;
    }
    if (npairs == 0) return 0;
    const int nm = match_ptr[npairs];
    DevBuf<int> dptr, dsm, dcnt; DevBuf<double> da, db, dF;
    if (!dptr.alloc(npairs + 1) || !da.alloc(2 * (size_t)nm) || !db.alloc(2 * (size_t)nm) || !dptr.up(match_ptr, npairs + 1) ||
        !da.up(a_xy, 2 * (size_t)nm) || !db.up(b_xy, 2 * (size_t)nm)) { fprintf(stderr, "[bsfm] fmatrix ransac: device allocation failed\n"); return BSFM_ERROR; }
    constexpr int CHUNK = 128;                       // pairs per launch (128 x 2048 trials x 9 doubles = 19 MB of candidates)
    const size_t per = (size_t)num_trials;
    if (!dsm.alloc(CHUNK * per * 8) || !dcnt.alloc(CHUNK * per) || !dF.alloc(CHUNK * per * 9)) { fprintf(stderr, "[bsfm] fmatrix ransac: device allocation failed\n"); return BSFM_ERROR; }
    std::vector<int> hs(CHUNK * per * 8), hc(CHUNK * per);
    std::vector<std::vector<int>> after(CHUNK);
    std::vector<bsfm_rand_t> start(CHUNK + 1);       // generator state at the start of each pair of the chunk (and behind the last)
    auto state_after = [&](int q, int t) { bsfm_rand_t st = start[q]; for (int i = 0; i < after[q][t]; ++i) (void)bsfm_rand_next(&st); return st; };
    std::vector<int> gave_up(CHUNK);
    std::vector<int> cptr(CHUNK + 1);
    int p0 = 0;
    while (p0 < npairs) {
        // speculative: draw for the pairs p0.. as if nobody left its trial loop early, evaluate, then accept pairs in order up
        // to and including the first one that did leave early (its successors restart from the corrected generator state)
        const int np = std::min(CHUNK, npairs - p0);
        bsfm_rand_t spec = *rng;
        for (int q = 0; q < np; ++q) {
            const int m0 = match_ptr[p0 + q], n = match_ptr[p0 + q + 1] - m0;
            start[q] = spec;
            if (n < 8) {                             // fmatrix.c:313-317: fails before any draw
                for (size_t t = 0; t < per; ++t) hs[((size_t)q * per + t) * 8] = -1;
                after[q].clear(); gave_up[q] = -2;
                continue;
            }
            draw_samples(spec, n, a_xy + 2 * (size_t)m0, b_xy + 2 * (size_t)m0, num_trials, hs.data() + (size_t)q * per * 8, after[q], &gave_up[q]);
        }
        start[np] = spec;
        for (int q = 0; q <= np; ++q) cptr[q] = match_ptr[p0 + q];
        if (!dsm.up(hs.data(), (size_t)np * per * 8) || hipMemcpy(dptr.p, cptr.data(), (np + 1) * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return BSFM_ERROR;
        const int bpp = (num_trials + 63) / 64;
        hipLaunchKernelGGL(k_fm_ransac, dim3(np * bpp), dim3(64), 0, 0, np, num_trials, dptr.p, da.p, db.p, dsm.p, threshold, dcnt.p, dF.p);
        if (hipDeviceSynchronize() != hipSuccess || !dcnt.down(hc.data(), (size_t)np * per)) { fprintf(stderr, "[bsfm] fmatrix ransac: kernel failed\n"); return BSFM_ERROR; }
        int accepted = 0;
        for (int q = 0; q < np; ++q) {
            const int n = match_ptr[p0 + q + 1] - match_ptr[p0 + q];
            const int* c = hc.data() + (size_t)q * per;
            ++accepted;
            if (gave_up[q] == -2) { inliers_max[p0 + q] = 0; continue; }          // fewer than 8 matches: F untouched, nothing drawn
            int last = num_trials - 1;
            bool early = false;
            if (gave_up[q] >= 0) last = gave_up[q] - 1;                            // the reference returned 0 inside trial gave_up[q]
            int best = -1, best_cnt = 0;
            for (int t = 0; t <= last; ++t) {
                if (c[t] > best_cnt) { best_cnt = c[t]; best = t; }
                if ((double)c[t] / n > success_ratio) { last = t; early = true; break; }
            }
            if (gave_up[q] >= 0 && !early) {       // (an early exit BEFORE the aborted trial wins: the reference never reached that trial)
                // `return 0` from inside the sample loop (fmatrix.c:349-350): no F is copied out, and the generator has
                // consumed the draws of the aborted trial too -- replay that trial from the state before it
                inliers_max[p0 + q] = 0;
                bsfm_rand_t st = gave_up[q] > 0 ? state_after(q, gave_up[q] - 1) : *rng;
                const int m0 = match_ptr[p0 + q];
                const double* a = a_xy + 2 * (size_t)m0; const double* b = b_xy + 2 * (size_t)m0;
                int idxs[8], round = 0;
                for (int j = 0; j < 8 && round < 1000; ++j) {
                    const int idx = bsfm_rand_next(&st) % n;
                    bool res = false;
                    for (int k = 0; k < j; ++k) {
                        const int o = idxs[k];
                        if (idx == o || (a[2 * idx] == a[2 * o] && a[2 * idx + 1] == a[2 * o + 1]) || (b[2 * idx] == b[2 * o] && b[2 * idx + 1] == b[2 * o + 1])) { res = true; break; }
                    }
                    if (res) { ++round; --j; continue; }
                    idxs[j] = idx;
                }
                *rng = st;
                break;                                                             // successors restart from here
            }
            inliers_max[p0 + q] = best_cnt;
            if (best >= 0 && !dF.down(F + 9 * (size_t)(p0 + q), 9, ((size_t)q * per + best) * 9)) return BSFM_ERROR;
            *rng = (last == num_trials - 1 && gave_up[q] < 0) ? start[q + 1] : state_after(q, last);   // replay only after an early exit
            if (early && last < num_trials - 1) break;                             // left early: the speculation behind it is void
        }
        p0 += accepted;
    }
    return 0;
}

extern "C" int bsfm_estimate_fmatrix_batch(int npairs, const int* match_ptr, const double* k1_xy, const double* k2_xy,
                                           int num_trials, double threshold, bsfm_rand_t* rng, double* F, int* num_inliers,
                                           unsigned char* inlier, int* lm_info)
{
    if (npairs < 0 || !match_ptr || !rng || (npairs > 0 && (!k1_xy || !k2_xy || !F || !num_inliers || !inlier))) {
        fprintf(stderr, "[bsfm] estimate fmatrix: bad arguments\n");
        return BSFM_ERROR;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fprintf(stderr, "[bsfm] FATAL: no usable HIP device; the MI355X path has no CPU fallback\n");
        return BSFM_ERROR;
    }
    if (npairs == 0) return 0;
    // pairs with fewer than 20 matches are turned away before anything is drawn (src/Epipolar.cpp:127-130)
    std::vector<int> sel, sptr(1, 0);
    std::vector<double> sa, sb;
    for (int p = 0; p < npairs; ++p) {
        const int m0 = match_ptr[p], n = match_ptr[p + 1] - m0;
        if (n < 20) continue;
        sel.push_back(p); sptr.push_back(sptr.back() + n);
        sa.insert(sa.end(), k2_xy + 2 * (size_t)m0, k2_xy + 2 * (size_t)(m0 + n));    // k2 is the FIRST point argument (Epipolar.cpp:149)
        sb.insert(sb.end(), k1_xy + 2 * (size_t)m0, k1_xy + 2 * (size_t)(m0 + n));
    }
    for (int p = 0; p < npairs; ++p) num_inliers[p] = 0;
    memset(inlier, 0, (size_t)match_ptr[npairs]);
    const int ns = (int)sel.size();
    if (ns == 0) return 0;
    std::vector<double> Fs(9 * (size_t)ns, 0.0);
    std::vector<int> cnt(ns, 0);
    if (bsfm_fmatrix_ransac_batch(ns, sptr.data(), sa.data(), sb.data(), num_trials, threshold, 0.95, rng, Fs.data(), cnt.data()) != 0) return BSFM_ERROR;
    std::vector<char> found(ns);
    for (int q = 0; q < ns; ++q) { found[q] = cnt[q] > 0; if (!found[q]) cnt[q] = -1; }   // no trial found an inlier: nothing to refine, F untouched
    DevBuf<int> dptr, dcnt, dinfo; DevBuf<double> da, db, dF; DevBuf<unsigned char> din;
    const size_t nm = (size_t)sptr.back();
    if (!dptr.alloc(ns + 1) || !dcnt.alloc(ns) || !dinfo.alloc(ns) || !da.alloc(2 * nm) || !db.alloc(2 * nm) || !dF.alloc(9 * (size_t)ns) ||
        !din.alloc(nm) || !dptr.up(sptr.data(), ns + 1) || !dcnt.up(cnt.data(), ns) || !da.up(sa.data(), 2 * nm) || !db.up(sb.data(), 2 * nm) ||
        !dF.up(Fs.data(), 9 * (size_t)ns)) { fprintf(stderr, "[bsfm] estimate fmatrix: device allocation failed\n"); return BSFM_ERROR; }
    (void)hipMemset(dinfo.p, 0, ns * sizeof(int));
    const char* er = getenv("BSFM_FM_REFINE");
    if (er && !strcmp(er, "thread"))
        hipLaunchKernelGGL(k_fm_refine, dim3((ns + 63) / 64), dim3(64), 0, 0, ns, dptr.p, da.p, db.p, threshold, dF.p, dcnt.p, din.p, dinfo.p);
    else
        hipLaunchKernelGGL(k_fm_refine_wave, dim3(ns), dim3(64), 0, 0, ns, dptr.p, da.p, db.p, threshold, dF.p, dcnt.p, din.p, dinfo.p);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "[bsfm] estimate fmatrix: kernel failed\n"); return BSFM_ERROR; }
    std::vector<unsigned char> hin(nm);
    std::vector<int> hinfo(ns);
    std::vector<double> Fr(9 * (size_t)ns);
    if (!dF.down(Fr.data(), 9 * (size_t)ns) || !dcnt.down(cnt.data(), ns) || !din.down(hin.data(), nm) || !dinfo.down(hinfo.data(), ns)) return BSFM_ERROR;
    for (int q = 0; q < ns; ++q) {
        const int p = sel[q];
        num_inliers[p] = cnt[q];
        if (lm_info) lm_info[p] = hinfo[q];
        if (found[q]) memcpy(F + 9 * (size_t)p, Fr.data() + 9 * (size_t)q, 9 * sizeof(double));
        memcpy(inlier + match_ptr[p], hin.data() + sptr[q], (size_t)(sptr[q + 1] - sptr[q]));
    }
    if (lm_info) for (int p = 0, q = 0; p < npairs; ++p) { if (q < ns && sel[q] == p) ++q; else lm_info[p] = 0; }
    return 0;
}
