// synth.cpp -- deterministic synthetic inputs (host only): the BA scene of SURVEY.md section 8(d) and
// SIFT-like descriptors for the matcher.  Used by tests/ and bench.py; no reference code involved.
#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../../include/bsfm.h"

namespace {
struct Rng {
    unsigned long long s;
    explicit Rng(unsigned long long seed) : s(seed ? seed : 88172645463325252ULL) {}
    unsigned long long next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
    double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    double normal() {
        double u1 = uni(), u2 = uni();
        if (u1 < 1e-300) u1 = 1e-300;
        return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2);
    }
    int below(int n) { return (int)(uni() * n) % n; }
};

void project_exact(const bsfm_camera_params_t& c, const double* b, double* x)
{   // model of sfm_project_final (lib/sfm-driver/sfm.c:118-190), explicit centres, with distortion
    const double d[3] = { b[0] - c.t[0], b[1] - c.t[1], b[2] - c.t[2] };
    const double P0 = c.R[0] * d[0] + c.R[1] * d[1] + c.R[2] * d[2];
    const double P1 = c.R[3] * d[0] + c.R[4] * d[1] + c.R[5] * d[2];
    const double P2 = c.R[6] * d[0] + c.R[7] * d[1] + c.R[8] * d[2];
    double p0 = P0 * c.f / -P2, p1 = P1 * c.f / -P2;
    const double rsq = (p0 * p0 + p1 * p1) / (c.f * c.f);
    const double factor = 1.0 + c.k[0] * rsq + c.k[1] * rsq * rsq;
    x[0] = p0 * factor; x[1] = p1 * factor;
}
}  // namespace

extern "C" int bsfm_synth_ba(int m, int n, int deg, unsigned long long seed, int banded,
                             int* rowptr, int* colidx, double* projections,
                             bsfm_camera_params_t* cams, double* points)
{
    if (m < deg || deg < 1 || n < 0) return BSFM_ERROR;
    if (banded && m < 50 && m < deg) return BSFM_ERROR;
    Rng rng(seed);
    for (int i = 0; i < 3 * n; ++i) points[i] = 2.0 * rng.uni() - 1.0;
    for (int j = 0; j < m; ++j) {
        bsfm_camera_params_t& c = cams[j];
        memset(&c, 0, sizeof(c));
        const double a = 6.283185307179586476925 * j / m;
        const double ctr[3] = { 6.0 * cos(a), 6.0 * sin(a), 0.3 * sin(3.0 * a) };
        double z[3] = { ctr[0], ctr[1], ctr[2] };            // camera looks down -z at the origin
        const double zn = sqrt(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]);
        for (double& v : z) v /= zn;
        const double up[3] = { 0.0, 0.0, 1.0 };
        double xax[3] = { up[1] * z[2] - up[2] * z[1], up[2] * z[0] - up[0] * z[2], up[0] * z[1] - up[1] * z[0] };
        const double xn = sqrt(xax[0] * xax[0] + xax[1] * xax[1] + xax[2] * xax[2]);
        for (double& v : xax) v /= xn;
        const double yax[3] = { z[1] * xax[2] - z[2] * xax[1], z[2] * xax[0] - z[0] * xax[2], z[0] * xax[1] - z[1] * xax[0] };
        for (int q = 0; q < 3; ++q) { c.R[q] = xax[q]; c.R[3 + q] = yax[q]; c.R[6 + q] = z[q]; c.t[q] = ctr[q]; }
        c.f = 1000.0 + 100.0 * rng.uni();
        c.k[0] = -0.05 * rng.uni();
        c.k[1] = 0.01 * rng.uni();
        c.f_scale = 1.0; c.k_scale = 1.0;
    }
    std::vector<int> cs(deg);
    const int window = std::min(50, m);
    for (int i = 0; i < n; ++i) {
        rowptr[i] = i * deg;
        const int j0 = rng.below(m);
        if (!banded) {
            for (int d = 0; d < deg; ++d) cs[d] = (int)((j0 + ((long long)d * m) / deg) % m);
        } else {
            // deg distinct cameras out of the `window` neighbours starting at j0
            std::vector<int> pool(window);
            for (int q = 0; q < window; ++q) pool[q] = q;
            for (int d = 0; d < deg; ++d) { const int r = d + rng.below(window - d); std::swap(pool[d], pool[r]); cs[d] = (j0 + pool[d]) % m; }
        }
        std::sort(cs.begin(), cs.end());
        for (int d = 0; d < deg; ++d) {
            const int k = i * deg + d;
            colidx[k] = cs[d];
            double x[2];
            project_exact(cams[cs[d]], points + 3 * (size_t)i, x);
            projections[2 * (size_t)k] = x[0] + 0.5 * rng.normal();
            projections[2 * (size_t)k + 1] = x[1] + 0.5 * rng.normal();
        }
    }
    rowptr[n] = n * deg;
    for (int i = 0; i < 3 * n; ++i) points[i] += 0.01 * rng.normal();
    for (int j = 0; j < m; ++j) {
        for (int q = 0; q < 3; ++q) cams[j].t[q] += 0.01 * rng.normal();
        cams[j].f *= 1.0 + 0.01 * rng.normal();
    }
    return 0;
}

extern "C" int bsfm_synth_keys(int num, unsigned long long seed, const unsigned char* dup_from, int n_from,
                               unsigned char* keys_out)
{
    Rng rng(seed ^ 0x9E3779B97F4A7C15ULL);
    for (int i = 0; i < num; ++i) {
        unsigned char* k = keys_out + 128 * (size_t)i;
        if (dup_from && n_from > 0 && rng.uni() < 0.2) {
            const unsigned char* src = dup_from + 128 * (size_t)rng.below(n_from);
            for (int q = 0; q < 128; ++q) {
                int v = (int)src[q] + (rng.below(17) - 8);
                k[q] = (unsigned char)std::min(255, std::max(0, v));
            }
            continue;
        }
        double v[128], nrm = 0.0;
        for (int q = 0; q < 128; ++q) { v[q] = fabs(40.0 * rng.normal()); nrm += v[q] * v[q]; }
        nrm = sqrt(nrm);
        for (int q = 0; q < 128; ++q) {
            const double s = nrm > 0 ? v[q] * 512.0 / nrm : 0.0;
            k[q] = (unsigned char)std::min(255.0, std::max(0.0, floor(s + 0.5)));
        }
    }
    return 0;
}
