// boundary.hip -- the drop-in C-ABI boundary: run_sfm with the reference's exact signature
// (lib/sfm-driver/sfm.h:68-86, body lib/sfm-driver/sfm.c:592-1003) on top of the resident-problem API.
//
// What happens here, in the reference's order:
//   1. dense vmask -> CRS (the ordering contract of lib/sba-1.5/sba_levmar.c:653-663: k-th set bit of
//      vmask in row-major order == k-th measurement) -- integer bookkeeping, bit-exact;
//   2. cnp selection (sfm.c:637-643), parameter packing / scaling and constraints (inside
//      bsfm_problem_create, sfm.c:649-781);
//   3. LM on the GPU (bsfm_lm_*), itmax = 150, verbose summary lines as sfm.c:872-873;
//   4. unpack cameras / points in place (sfm.c:876-929);
//   5. optional Vout/Sout/Uout/Wout export at the solution (lib/sba-1.5/sba_levmar.c:1633-2026).
// fix_points != 0 selects the camera-only refinement (sba_mot_levmar, sfm.c:839-846); cameras with known intrinsics
// (sfm.c:339-358) are projected through their own K and Brown distortion; optimize_for_fisheye != 0 selects the fisheye
// projection (sfm.c:448-492).  Anything the GPU core cannot run fails loudly: there is deliberately no CPU fallback here.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <thread>
#include <algorithm>
#include <chrono>
#include <hip/hip_runtime_api.h>
#include "../../include/bsfm.h"
#include "index_build.h"
#include "devcache.h"

namespace {

// wall milliseconds of the phases of the last bsfm_run_sfm_ex call of this thread (bsfm_run_sfm_last_ms)
thread_local double g_run_ms[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
thread_local int g_create_failed = 0;            // the last bsfm_run_sfm_ex could not even build its problem
enum { RM_TOTAL = 0, RM_CRS, RM_CRS_UPLOAD, RM_CRS_KERNELS, RM_CREATE, RM_LM, RM_DOWNLOAD, RM_CRS_ON_DEVICE };
inline double ms_since(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// masks of at least this many bytes are turned into the CRS on the device (index_build.hip:crs_from_vmask_device); smaller ones by
// the host loop below (a 14-camera call of incremental Bundler has a few KB of mask: one upload + three launches would cost more)
size_t vmask_device_min()
{
    static const size_t v = [] { const char* e = getenv("BSFM_VMASK_DEVICE_MIN"); return e ? (size_t)atoll(e) : (size_t)1 << 20; }();
    return v;
}

// dense vmask -> CRS: the k-th set bit in row-major order is the k-th measurement (sba_levmar.c:653-663)
void vmask_to_crs(int n, int m, const char* vmask, std::vector<int>& rowptr, std::vector<int>& colidx)
{
    rowptr.assign((size_t)n + 1, 0);
    size_t nvis = 0;
    const size_t tot = (size_t)n * m;
    for (size_t q = 0; q < tot; ++q) nvis += (vmask[q] != 0);
    colidx.resize(nvis);
    size_t k = 0;
    for (int i = 0; i < n; ++i) {
        rowptr[i] = (int)k;
        const char* row = vmask + (size_t)i * m;
        for (int j = 0; j < m; ++j) if (row[j]) colidx[k++] = j;
    }
    rowptr[n] = (int)k;
}

// optional covariance export at the current parameters, undamped (sba_levmar.c:1633-2026)
void export_blocks(bsfm_problem_t* pb, int n, int mcon, int cnp, const std::vector<int>& rowptr, const std::vector<int>& colidx,
                   double* Vout, double* Sout, double* Uout, double* Wout)
{
    // The reference exports only when Sout != NULL (sba_levmar.c:1633); without it Vout keeps the V of the last iteration
    // (sba_levmar.c:1039-1051), Uout / Wout stay untouched.  Bundler passes all four or none (BundleTwo.cpp:1768,1902).
    if (!Sout) {
        if (Vout) bsfm_eval_normal_equations(pb, 0.0, nullptr, nullptr, Vout, nullptr, nullptr, nullptr, nullptr);
        return;
    }
    const size_t nvis = colidx.size();
    const int m = (int)bsfm_problem_num_cameras(pb);
    std::vector<double> J;
    if (Wout) J.resize(nvis * (size_t)(2 * cnp + 6));
    std::vector<double> Sred;                          // the library's S covers the free cameras only, row stride (m - mcon) cnp
    const size_t sd = (size_t)(m - mcon) * cnp, sm = (size_t)m * cnp;
    if (mcon > 0) Sred.resize(sd * sd);
    bsfm_eval_normal_equations(pb, 0.0, Uout, nullptr, Vout, nullptr, Wout ? J.data() : nullptr, mcon > 0 ? Sred.data() : Sout, nullptr);
    if (mcon > 0) {
        // Sout is an (m cnp)^2 buffer (sba.h:137).  The reference copies its (m - mcon) cnp wide S into it with stride m cnp
        // (sba_levmar.c:2017-2022), which scrambles the blocks when mcon > 0 -- Bundler never exports with fixed cameras.  Here the
        // reduced system of the free cameras is embedded at its natural place, the blocks of the fixed cameras are zero, and U_j
        // of a fixed camera is zero as in the reference (cleared and skipped for j < mcon, :1646-1648).
        memset(Sout, 0, sm * sm * sizeof(double));
        for (size_t r = 0; r < sd; ++r) memcpy(Sout + ((size_t)mcon * cnp + r) * sm + (size_t)mcon * cnp, Sred.data() + r * sd, sd * sizeof(double));
        if (Uout) memset(Uout, 0, (size_t)mcon * cnp * cnp * sizeof(double));
    }
    if (Wout) {   // Wout[(j*cnp+ii)*3*n + 3*i + jj] = (A_ij^T B_ij)[ii][jj]  (sba_levmar.c:1836-1846)
        const int js = 2 * cnp + 6;
        for (int i = 0; i < n; ++i)
            for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
                const int j = colidx[k];
                const double* A = &J[(size_t)k * js]; const double* B = A + 2 * cnp;
                for (int ii = 0; ii < cnp; ++ii)
                    for (int jj = 0; jj < 3; ++jj)
                        Wout[((size_t)j * cnp + ii) * 3 * n + (size_t)3 * i + jj] =
                            (j < mcon) ? 0.0 : A[ii] * B[jj] + A[cnp + ii] * B[3 + jj];
            }
    }
}

// ---- run_sfm on several GPUs of this node, inside one process -------------------------------------------------
// Points -- with all their observations -- are dealt to the ranks in contiguous ranges balanced by sum d_i^2 (the Schur work);
// the cameras are replicated; one host thread per GPU drives its shard's LM loop; U || ea, the packed union of the
// reduced-camera blocks || E and a few scalars are summed with RCCL over xGMI (comm.hip) on the compute streams; every rank
// factors the same reduced camera system, so the replicas stay bit-identical and no broadcast is needed.
struct MultiArgs {
    int n, m, ncons; const std::vector<int>* rowptr; const std::vector<int>* colidx; const double* projections;
    int est_focal_length, undistort, explicit_camera_centers;
    bsfm_camera_params_t* cams; double* pts;
    int use_constraints, use_point_constraints; const double* point_constraints; double point_constraint_weight;
    int optimize_for_fisheye;
};

constexpr int RUN_MULTI_NO_COMM = -1000;      // the communicator could not be created: nothing was touched

int run_multi(int G, const std::vector<int>& devs, const MultiArgs& a, const bsfm_options_t& opt, double info[BSFM_INFOSZ])
{
    const int n = a.n, m = a.m;
    const std::vector<int>& rp = *a.rowptr; const std::vector<int>& ci = *a.colidx;
    const int cnp = (a.est_focal_length ? 7 : 6) + (a.undistort ? 2 : 0);
    std::vector<int> bounds((size_t)G + 1, n);
    {
        std::vector<double> w((size_t)n + 1, 0.0);
        for (int i = 0; i < n; ++i) { const double d = rp[i + 1] - rp[i]; w[i + 1] = w[i] + d * d; }
        bounds[0] = 0;
        for (int g = 1; g < G; ++g)
            bounds[g] = (int)(std::lower_bound(w.begin() + 1, w.end(), w[n] * g / G) - (w.begin() + 1));
        for (int g = 1; g <= G; ++g) bounds[g] = std::max(bounds[g], bounds[g - 1]);
    }
    std::vector<bsfm_comm_t*> comms((size_t)G, nullptr);
    if (bsfm_comm_create_all(G, devs.data(), comms.data()) != 0) return RUN_MULTI_NO_COMM;      // the caller falls back to one GPU
    std::vector<int> rcs((size_t)G, BSFM_ERROR);
    std::vector<std::vector<double>> infos((size_t)G, std::vector<double>(BSFM_INFOSZ, 0.0));
    std::vector<bsfm_camera_params_t> cam_out((size_t)m);
    auto worker = [&](int g) {
        if (hipSetDevice(devs[g]) != hipSuccess) {
            // still take part in the "everybody has a problem" exchange, or the other ranks would wait for this one forever
            double bad = 1.0;
            (void)bsfm_comm_allreduce_host(comms[g], &bad, 1, 1);
            return;
        }
        const int lo = bounds[g], hi = bounds[g + 1], k0 = rp[lo];
        std::vector<int> lrp((size_t)(hi - lo) + 1);
        for (int i = lo; i <= hi; ++i) lrp[i - lo] = rp[i] - k0;
        bsfm_problem_desc_t d;
        memset(&d, 0, sizeof(d));
        d.n = hi - lo; d.m = m; d.mcon = a.ncons;
        d.rowptr = lrp.data(); d.colidx = ci.data() + k0; d.projections = a.projections + 2 * (size_t)k0;
        d.est_focal_length = a.est_focal_length; d.undistort = a.undistort; d.explicit_camera_centers = a.explicit_camera_centers;
        d.cameras = a.cams; d.points = a.pts + 3 * (size_t)lo;
        d.use_constraints = a.use_constraints; d.use_point_constraints = a.use_point_constraints;
        d.point_constraints = a.point_constraints ? a.point_constraints + 3 * (size_t)lo : nullptr;
        d.point_constraint_weight = a.point_constraint_weight;
        d.optimize_for_fisheye = a.optimize_for_fisheye;
        d.world_size = G; d.rank = g; d.nvis_global = (long long)ci.size(); d.nvars_global = (long long)m * cnp + 3LL * n;
        bsfm_options_t o = opt;
        if (g != 0) o.verbose = 0;
        bsfm_problem_t* pb = bsfm_problem_create(&d, &o);
        double okflag = pb ? 0.0 : 1.0;                                   // every rank must have its problem, or nobody starts
        if (bsfm_comm_allreduce_host(comms[g], &okflag, 1, 1) != 0) okflag = 1.0;
        if (okflag == 0.0) {
            bsfm_problem_set_comm(pb, comms[g]);
            int rc = bsfm_lm_begin(pb);
            if (rc == 0) bsfm_lm_iterate(pb, o.itmax);
            rc = bsfm_lm_finish(pb, infos[g].data());
            if (rc != BSFM_ERROR || infos[g][5] > 0) {
                if (g == 0) { memcpy(cam_out.data(), a.cams, (size_t)m * sizeof(bsfm_camera_params_t)); bsfm_problem_download(pb, nullptr, cam_out.data(), nullptr); }
                bsfm_problem_download(pb, nullptr, nullptr, d.n ? a.pts + 3 * (size_t)lo : nullptr);      // disjoint slices of init_pts
            }
            rcs[g] = rc;
        }
        if (pb) bsfm_problem_destroy(pb);
    };
    std::vector<std::thread> th;
    for (int g = 1; g < G; ++g) th.emplace_back(worker, g);
    int saved = 0; (void)hipGetDevice(&saved);
    worker(0);
    for (auto& t : th) t.join();
    (void)hipSetDevice(saved);
    for (int g = 0; g < G; ++g) bsfm_comm_destroy(comms[g]);
    for (int g = 0; g < G; ++g) if (rcs[g] == BSFM_ERROR && infos[g][5] <= 0) { for (int q = 0; q < BSFM_INFOSZ; ++q) info[q] = infos[0][q]; return BSFM_ERROR; }
    memcpy(a.cams, cam_out.data(), (size_t)m * sizeof(bsfm_camera_params_t));     // only after every rank has finished reading the inputs
    for (int q = 0; q < BSFM_INFOSZ; ++q) info[q] = infos[0][q];
    return rcs[0];
}

}  // namespace

extern "C" int bsfm_last_call_infra_failure(void);
extern "C" void bsfm_clear_infra_failure(void);

extern "C" {

// dense vmask -> CRS exactly as run_sfm does it internally (test entry for the ordering contract).
// rowptr (n + 1) / colidx (nvis) may be NULL to query the count only.  Host-only, needs no device.
int bsfm_crs_from_vmask(int n, int m, const char* vmask, int* rowptr, int* colidx)
{
    if (n < 0 || m <= 0 || !vmask) return BSFM_ERROR;
    std::vector<int> rp, ci;
    vmask_to_crs(n, m, vmask, rp, ci);
    if (rowptr) memcpy(rowptr, rp.data(), rp.size() * sizeof(int));
    if (colidx && !ci.empty()) memcpy(colidx, ci.data(), ci.size() * sizeof(int));
    return (int)ci.size();
}

// The same CRS built ON THE DEVICE (index_build.hip:crs_from_vmask_device: what run_sfm uses for masks of scale) and copied back:
// the test entry that pins the device path to the reference's ordering contract bit for bit.  Needs a HIP device.
int bsfm_crs_from_vmask_device(int n, int m, const char* vmask, int* rowptr, int* colidx, double* ms_out)
{
    if (n < 0 || m <= 0 || !vmask) return BSFM_ERROR;
    if (bsfm_device_count() <= 0) { fprintf(stderr, "[bsfm] bsfm_crs_from_vmask_device: no HIP device\n"); return BSFM_ERROR; }
    int *d_rp = nullptr, *d_ci = nullptr, nvis = 0;
    if (bsfm::crs_from_vmask_device(n, m, vmask, &d_rp, &d_ci, &nvis, ms_out, nullptr) != 0) return BSFM_ERROR;
    bool ok = true;
    if (rowptr) ok = hipMemcpy(rowptr, d_rp, ((size_t)n + 1) * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
    if (ok && colidx && nvis) ok = hipMemcpy(colidx, d_ci, (size_t)nvis * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
    bsfm::dev_free(d_rp); bsfm::dev_free(d_ci);
    return ok ? nvis : BSFM_ERROR;
}

// wall milliseconds of the last bsfm_run_sfm_ex / run_sfm call of the calling thread: "total", "crs" (vmask -> CRS, of which
// "crs_upload" / "crs_kernels" when it ran on the device: "crs_on_device" = 1), "create" (bsfm_problem_create: uploads, index
// construction, allocation), "lm" (all LM iterations), "download" (parameters back + optional export + teardown)
double bsfm_run_sfm_last_ms(const char* phase)
{
    static const char* names[8] = { "total", "crs", "crs_upload", "crs_kernels", "create", "lm", "download", "crs_on_device" };
    for (int q = 0; q < 8; ++q) if (phase && !strcmp(phase, names[q])) return g_run_ms[q];
    return -1.0;
}

int bsfm_sba_motstr_levmar(int n, int m, int mcon, char* vmask, double* p, int cnp, int pnp,
                           double* x, double* covx, int mnp, int camera_model, const void* model_data,
                           int itmax, int verbose, double opts[6], double info[BSFM_INFOSZ],
                           int use_constraints, bsfm_camera_constraints_t* constraints,
                           int use_point_constraints, bsfm_point_constraints_t* point_constraints,
                           double* Vout, double* Sout, double* Uout, double* Wout)
{
    double linfo[BSFM_INFOSZ];
    for (int i = 0; i < BSFM_INFOSZ; ++i) linfo[i] = 0.0;
    if (!info) info = linfo;
    auto refuse = [](const char* why) { fprintf(stderr, "[bsfm] bsfm_sba_motstr_levmar: %s; nothing was changed\n", why); return BSFM_ERROR; };
    if (camera_model != BSFM_MODEL_SNAVELY || !model_data) return refuse("only BSFM_MODEL_SNAVELY with its data block is implemented");
    const bsfm_snavely_model_t* md = static_cast<const bsfm_snavely_model_t*>(model_data);
    if (pnp != 3 || mnp != 2) return refuse("pnp must be 3 and mnp 2");
    if (covx) return refuse("measurement covariances (covx) are not implemented");
    if (cnp != 6 + (md->est_focal_length ? 1 : 0) + (md->undistort ? 2 : 0)) return refuse("cnp does not match the model flags");
    if (!md->R_init || (!md->est_focal_length && !md->f_init)) return refuse("model data incomplete (R_init / f_init)");
    for (int j = 0; j < m; ++j)
        if (p[(size_t)j * cnp + 3] != 0.0 || p[(size_t)j * cnp + 4] != 0.0 || p[(size_t)j * cnp + 5] != 0.0)
            return refuse("the rotation increments p[j*cnp+3..5] must be zero on entry (fold them into R_init)");

    std::vector<bsfm_camera_params_t> cams((size_t)m);
    memset(cams.data(), 0, cams.size() * sizeof(bsfm_camera_params_t));
    for (int j = 0; j < m; ++j) {
        memcpy(cams[j].R, md->R_init + 9 * (size_t)j, 9 * sizeof(double));
        cams[j].f = md->est_focal_length ? 1.0 : md->f_init[j];
        if (use_constraints && constraints)
            for (int q = 0; q < cnp; ++q) {
                cams[j].constrained[q] = constraints[j].constrained[q];
                cams[j].constraints[q] = constraints[j].constraints[q];
                cams[j].weights[q] = constraints[j].weights[q];
            }
    }
    std::vector<double> pcon;
    double pweight = 0.0;
    if (use_point_constraints && point_constraints) {
        pcon.assign((size_t)3 * n, 0.0);
        bool have = false;
        for (int i = 0; i < n; ++i) {
            if (!point_constraints[i].constrained) continue;
            const double* c = point_constraints[i].constraints;
            if (c[0] == 0.0 && c[1] == 0.0 && c[2] == 0.0) return refuse("a point constrained exactly to the origin cannot be expressed (zero vector = unconstrained, sfm.c:757-781)");
            if (have && point_constraints[i].weight != pweight) return refuse("point-constraint weights must be uniform");
            pweight = point_constraints[i].weight; have = true;
            pcon[3 * (size_t)i] = c[0]; pcon[3 * (size_t)i + 1] = c[1]; pcon[3 * (size_t)i + 2] = c[2];
        }
    }

    std::vector<int> rowptr, colidx;
    vmask_to_crs(n, m, vmask, rowptr, colidx);
    bsfm_options_t opt;
    bsfm_default_options(&opt);
    opt.itmax = itmax; opt.verbose = verbose;
    if (opts) for (int q = 0; q < 6; ++q) opt.opts[q] = opts[q];
    bsfm_problem_desc_t d;
    memset(&d, 0, sizeof(d));
    d.n = n; d.m = m; d.mcon = mcon;
    d.rowptr = rowptr.data(); d.colidx = colidx.data(); d.projections = x;
    d.est_focal_length = md->est_focal_length; d.undistort = md->undistort; d.explicit_camera_centers = md->explicit_camera_centers;
    d.cameras = cams.data(); d.points = p + (size_t)m * cnp;
    d.use_constraints = use_constraints && constraints; d.constraints_prescaled = 1;
    d.use_point_constraints = !pcon.empty(); d.point_constraints = pcon.empty() ? nullptr : pcon.data();
    d.point_constraint_weight = pweight;
    d.p_packed = p;
    d.world_size = 1;
    bsfm_problem_t* pb = bsfm_problem_create(&d, &opt);
    if (!pb) return BSFM_ERROR;
    int rc = bsfm_lm_begin(pb);
    if (rc == 0) bsfm_lm_iterate(pb, opt.itmax);
    rc = bsfm_lm_finish(pb, info);
    if (rc != BSFM_ERROR || info[5] > 0) bsfm_problem_download(pb, p, nullptr, nullptr);
    export_blocks(pb, n, mcon, cnp, rowptr, colidx, Vout, Sout, Uout, Wout);
    bsfm_problem_destroy(pb);
    return rc == BSFM_ERROR ? BSFM_ERROR : (int)info[5];
}

int bsfm_sba_mot_levmar(int n, int m, int mcon, char* vmask, double* p, int cnp, double* x, double* covx, int mnp,
                        int camera_model, const void* model_data, int itmax, int verbose, double opts[6],
                        double info[BSFM_INFOSZ], int use_constraints, bsfm_camera_constraints_t* constraints)
{
    double linfo[BSFM_INFOSZ];
    for (int i = 0; i < BSFM_INFOSZ; ++i) linfo[i] = 0.0;
    if (!info) info = linfo;
    auto refuse = [](const char* why) { fprintf(stderr, "[bsfm] bsfm_sba_mot_levmar: %s; nothing was changed\n", why); return BSFM_ERROR; };
    if (camera_model != BSFM_MODEL_SNAVELY || !model_data) return refuse("only BSFM_MODEL_SNAVELY with its data block is implemented");
    const bsfm_snavely_model_t* md = static_cast<const bsfm_snavely_model_t*>(model_data);
    if (mnp != 2) return refuse("mnp must be 2");
    if (covx) return refuse("measurement covariances (covx) are not implemented");
    if (cnp != 6 + (md->est_focal_length ? 1 : 0) + (md->undistort ? 2 : 0)) return refuse("cnp does not match the model flags");
    if (!md->R_init || !md->points || (!md->est_focal_length && !md->f_init)) return refuse("model data incomplete (R_init / points / f_init)");
    for (int j = 0; j < m; ++j)
        if (p[(size_t)j * cnp + 3] != 0.0 || p[(size_t)j * cnp + 4] != 0.0 || p[(size_t)j * cnp + 5] != 0.0)
            return refuse("the rotation increments p[j*cnp+3..5] must be zero on entry (fold them into R_init)");
    std::vector<bsfm_camera_params_t> cams((size_t)m);
    memset(cams.data(), 0, cams.size() * sizeof(bsfm_camera_params_t));
    for (int j = 0; j < m; ++j) {
        memcpy(cams[j].R, md->R_init + 9 * (size_t)j, 9 * sizeof(double));
        cams[j].f = md->est_focal_length ? 1.0 : md->f_init[j];
        if (use_constraints && constraints)
            for (int q = 0; q < cnp; ++q) {
                cams[j].constrained[q] = constraints[j].constrained[q];
                cams[j].constraints[q] = constraints[j].constraints[q];
                cams[j].weights[q] = constraints[j].weights[q];
            }
    }
    std::vector<double> packed((size_t)m * cnp + (size_t)3 * n);
    memcpy(packed.data(), p, (size_t)m * cnp * sizeof(double));
    memcpy(packed.data() + (size_t)m * cnp, md->points, (size_t)3 * n * sizeof(double));
    std::vector<int> rowptr, colidx;
    vmask_to_crs(n, m, vmask, rowptr, colidx);
    bsfm_options_t opt;
    bsfm_default_options(&opt);
    opt.itmax = itmax; opt.verbose = verbose;
    if (opts) for (int q = 0; q < 6; ++q) opt.opts[q] = opts[q];
    bsfm_problem_desc_t d;
    memset(&d, 0, sizeof(d));
    d.n = n; d.m = m; d.mcon = mcon;
    d.rowptr = rowptr.data(); d.colidx = colidx.data(); d.projections = x;
    d.est_focal_length = md->est_focal_length; d.undistort = md->undistort; d.explicit_camera_centers = md->explicit_camera_centers;
    d.cameras = cams.data(); d.points = md->points;
    d.use_constraints = use_constraints && constraints; d.constraints_prescaled = 1;
    d.p_packed = packed.data(); d.fix_points = 1; d.world_size = 1;
    bsfm_problem_t* pb = bsfm_problem_create(&d, &opt);
    if (!pb) return BSFM_ERROR;
    int rc = bsfm_lm_begin(pb);
    if (rc == 0) bsfm_lm_iterate(pb, opt.itmax);
    rc = bsfm_lm_finish(pb, info);
    if (rc != BSFM_ERROR || info[5] > 0) {
        bsfm_problem_download(pb, packed.data(), nullptr, nullptr);
        memcpy(p, packed.data(), (size_t)m * cnp * sizeof(double));
    }
    bsfm_problem_destroy(pb);
    return rc == BSFM_ERROR ? BSFM_ERROR : (int)info[5];
}

int bsfm_run_sfm_ex(int num_pts, int num_cameras, int ncons, char* vmask, double* projections,
                    int est_focal_length, int const_focal_length, int undistort, int explicit_camera_centers,
                    bsfm_camera_params_t* init_camera_params, bsfm_v3_t* init_pts,
                    int use_constraints, int use_point_constraints,
                    bsfm_v3_t* points_constraints, double point_constraint_weight,
                    int fix_points, int optimize_for_fisheye, double eps2,
                    double* Vout, double* Sout, double* Uout, double* Wout,
                    const bsfm_options_t* opt_in, double info[BSFM_INFOSZ])
{
    bsfm_options_t opt;
    if (opt_in) opt = *opt_in; else bsfm_default_options(&opt);
    opt.opts[2] = eps2;   // sfm.c:707
    double linfo[BSFM_INFOSZ];
    for (int i = 0; i < BSFM_INFOSZ; ++i) linfo[i] = 0.0;
    if (!info) info = linfo;

    if (est_focal_length && const_focal_length)
        printf("Error: case of constant focal length has not been implemented.\n");   // sfm.c:521-523
    // 1. vmask -> CRS: on the device for masks of scale (the 500 MB mask of 1 000 cameras x 500 000 points is uploaded once and
    //    counted / scanned / compacted there), by the host loop for the small calls of incremental reconstruction
    const auto t_all = std::chrono::steady_clock::now();
    for (double& v : g_run_ms) v = 0.0;
    std::vector<int> rowptr, colidx;
    int *d_rp = nullptr, *d_ci = nullptr;                     // device CRS (owned here)
    struct DevCrsGuard { int*& a; int*& b; ~DevCrsGuard() { bsfm::dev_free(a); bsfm::dev_free(b); } } crs_guard{ d_rp, d_ci };
    bool host_crs = true;
    {
        const auto t0 = std::chrono::steady_clock::now();
        const size_t mask_bytes = (size_t)std::max(num_pts, 0) * (size_t)std::max(num_cameras, 0);
        if (vmask && num_pts > 0 && num_cameras > 0 && mask_bytes >= vmask_device_min() && bsfm_device_count() > 0) {
            int nvis = 0; double ms3[3] = { 0, 0, 0 };
            if (bsfm::crs_from_vmask_device(num_pts, num_cameras, vmask, &d_rp, &d_ci, &nvis, ms3, nullptr) == 0) {
                host_crs = false;
                g_run_ms[RM_CRS_UPLOAD] = ms3[0]; g_run_ms[RM_CRS_KERNELS] = ms3[1]; g_run_ms[RM_CRS_ON_DEVICE] = 1.0;
            } else fprintf(stderr, "[bsfm] run_sfm: device-side vmask -> CRS failed, using the host loop\n");
        }
        if (host_crs) vmask_to_crs(num_pts, num_cameras, vmask, rowptr, colidx);
        g_run_ms[RM_CRS] = ms_since(t0);
    }
    // host copies of a device-built CRS, only for the paths that index with them on the host (multi-GPU sharding, W export)
    auto need_host_crs = [&]() -> bool {
        if (host_crs) return true;
        int nvis = 0;
        rowptr.resize((size_t)num_pts + 1);
        if (hipMemcpy(rowptr.data(), d_rp, rowptr.size() * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return false;
        nvis = rowptr[num_pts];
        colidx.resize((size_t)nvis);
        if (nvis && hipMemcpy(colidx.data(), d_ci, (size_t)nvis * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return false;
        host_crs = true;
        return true;
    };

    // more than one GPU asked for (opt.num_gpus / BSFM_NUM_GPUS; -1 = all visible): shard the points inside this process.  The
    // camera-only refinement and the covariance export stay on one GPU.
    {
        int G = opt.num_gpus, ndev = bsfm_device_count();
        if (G < 0) G = ndev;
        if (G > 1 && !fix_points && !(Vout || Sout || Uout || Wout) && ndev > 0 && num_pts >= G) {
            std::vector<int> devs((size_t)G);
            const bool share = getenv("BSFM_ALLOW_SHARED_DEVICE") != nullptr;      // test rigs: several ranks on one GPU (loopback transport)
            if (G > ndev && !share) { fprintf(stderr, "[bsfm] run_sfm: %d GPUs asked for, %d visible: using %d\n", G, ndev, ndev); G = ndev; devs.resize((size_t)G); }
            for (int g = 0; g < G; ++g) devs[g] = g % ndev;
            if (G > 1) {
                if (!need_host_crs()) return BSFM_ERROR;
                MultiArgs a = { num_pts, num_cameras, ncons, &rowptr, &colidx, projections, est_focal_length, undistort, explicit_camera_centers,
                                init_camera_params, reinterpret_cast<double*>(init_pts), use_constraints, use_point_constraints,
                                reinterpret_cast<const double*>(points_constraints), point_constraint_weight, optimize_for_fisheye ? 1 : 0 };
                const int rc = run_multi(G, devs, a, opt, info);
                if (rc != RUN_MULTI_NO_COMM) {
                    if (opt.verbose >= 1) {
                        printf("[run_sfm] Number of iterations: %d\n", (int)info[5]);   // sfm.c:872-873
                        printf("info[6] = %0.3f\n", info[6]);
                    }
                    return rc;
                }
                // librccl missing / ncclCommInitAll failed: the job still runs, on one GPU (it used to return an error
                // that the void run_sfm could not report, and Bundler went on with unoptimised parameters)
                fprintf(stderr, "[bsfm] run_sfm: the %d-GPU communicator could not be set up -- continuing on ONE GPU\n", G);
            }
        }
    }
    bsfm_problem_desc_t d;
    memset(&d, 0, sizeof(d));
    d.n = num_pts; d.m = num_cameras; d.mcon = ncons;
    if (host_crs) { d.rowptr = rowptr.data(); d.colidx = colidx.data(); }
    else { d.rowptr = d_rp; d.colidx = d_ci; d.arrays_on_device = BSFM_INDEX_ON_DEVICE; }
    d.projections = projections;
    d.est_focal_length = est_focal_length; d.undistort = undistort; d.explicit_camera_centers = explicit_camera_centers;
    d.cameras = init_camera_params; d.points = reinterpret_cast<const double*>(init_pts);
    d.use_constraints = use_constraints; d.use_point_constraints = use_point_constraints;
    d.point_constraints = reinterpret_cast<const double*>(points_constraints);
    d.point_constraint_weight = point_constraint_weight;
    d.fix_points = fix_points ? 1 : 0;              // sba_mot_levmar: cameras only, no point constraints, no V/S/W
    d.optimize_for_fisheye = optimize_for_fisheye ? 1 : 0;   // sfm.c:819-851
    d.world_size = 1; d.rank = 0;

    const auto t_cr = std::chrono::steady_clock::now();
    bsfm_problem_t* pb = bsfm_problem_create(&d, &opt);
    if (!pb) { g_create_failed = 1; return BSFM_ERROR; }
    g_run_ms[RM_CREATE] = ms_since(t_cr);
    const auto t_lm = std::chrono::steady_clock::now();
    int rc = bsfm_lm_begin(pb);
    if (rc == 0) bsfm_lm_iterate(pb, opt.itmax);
    rc = bsfm_lm_finish(pb, info);
    g_run_ms[RM_LM] = ms_since(t_lm);
    if (opt.verbose >= 1) {
        printf("[run_sfm] Number of iterations: %d\n", (int)info[5]);   // sfm.c:872-873
        printf("info[6] = %0.3f\n", info[6]);
    }
    const int cnp = bsfm_problem_cnp(pb);
    const auto t_dl = std::chrono::steady_clock::now();
    if (rc != BSFM_ERROR || info[5] > 0) {
        // the reference copies the parameter vector back unconditionally (sfm.c:876-929)
        bsfm_problem_download(pb, nullptr, init_camera_params, fix_points ? nullptr : reinterpret_cast<double*>(init_pts));
    }
    if (!fix_points && (Vout || Sout || Uout || Wout)) {
        if (Sout && Wout && !need_host_crs()) { bsfm_problem_destroy(pb); return BSFM_ERROR; }
        export_blocks(pb, num_pts, ncons, cnp, rowptr, colidx, Vout, Sout, Uout, Wout);
    }
    bsfm_problem_destroy(pb);
    g_run_ms[RM_DOWNLOAD] = ms_since(t_dl);
    g_run_ms[RM_TOTAL] = ms_since(t_all);
    return rc;
}

void run_sfm(int num_pts, int num_cameras, int ncons, char* vmask, double* projections,
             int est_focal_length, int const_focal_length, int undistort, int explicit_camera_centers,
             bsfm_camera_params_t* init_camera_params, bsfm_v3_t* init_pts,
             int use_constraints, int use_point_constraints,
             bsfm_v3_t* points_constraints, double point_constraint_weight,
             int fix_points, int optimize_for_fisheye, double eps2,
             double* Vout, double* Sout, double* Uout, double* Wout)
{
    g_create_failed = 0;
    bsfm_clear_infra_failure();
    const int rc = bsfm_run_sfm_ex(num_pts, num_cameras, ncons, vmask, projections, est_focal_length,
                                   const_focal_length, undistort, explicit_camera_centers, init_camera_params,
                                   init_pts, use_constraints, use_point_constraints, points_constraints,
                                   point_constraint_weight, fix_points, optimize_for_fisheye, eps2,
                                   Vout, Sout, Uout, Wout, nullptr, nullptr);
    // run_sfm returns void: a failure of the machinery (no device, a HIP error, a problem that could not be built, a hand-off
    // that timed out) must not look like a finished adjustment to the caller -- the reference's fatal paths exit(1) as well
    // (lib/sba-1.5/sba_levmar.c:72-83).  SBA's own numerical exits (SBA_ERROR: stop 7, "almost singular", too few measurements)
    // return to the caller as they do in the reference.
    if (rc == BSFM_ERROR && (bsfm_device_count() <= 0 || g_create_failed || bsfm_last_call_infra_failure())) {
        fprintf(stderr, "[bsfm] run_sfm: %s -- aborting (there is no CPU fallback)\n",
                bsfm_device_count() <= 0 ? "no HIP device" : (g_create_failed ? "the problem could not be set up on the device" : "a device-side failure"));
        exit(1);
    }
}

}  // extern "C"
