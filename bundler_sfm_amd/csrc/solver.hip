// solver.hip -- host side of the MI355X bundle-adjustment core: resident problem object, index
// construction, and the Levenberg-Marquardt driver.
//
// The LM control flow restates lib/sba-1.5/sba_levmar.c:457-2081 (sba_motstr_levmar_x) statement by
// statement where it decides something (stop rules 1-8, damping update, the "constraint cost at the trial
// point uses the OLD p" quirk :1488-1522, stop-8 leaving p un-updated :1569-1572, itmax overriding the stop
// code :1617), and sba_mot_levmar_x (:2090-2690) for the camera-only mode; all O(nvis) work is in the HIP kernels of
// kernels.hip.h / schur.hip.h and the dense solve in potrf.hip.h (which adds two internal streams of its own).
// The host reads ONE small scalar block per decision point.  Multi-GPU: point-sharded ranks exchange U||ea, the packed
// union of the reduced-camera blocks ||E, and small scalar blocks through the caller's all-reduce hook.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <cfloat>
#include <vector>
#include <algorithm>
#include <string>
#include <mutex>
#include <unordered_map>
#include <deque>
#include <utility>
#include <dlfcn.h>
#include <chrono>

#include "../../include/bsfm.h"
#include "devcache.h"
#include "kernels.hip.h"
#include "schur.hip.h"
#include "potrf.hip.h"
#include "chol_dyn.hip.h"
#include "compsolve.hip.h"

using namespace bsfm;

// Infrastructure failures (a HIP call that failed, a hand-off that timed out, a collective that broke) as opposed to the numerical
// exits SBA itself knows (stop 7, "almost singular", too few measurements): the drop-in run_sfm aborts the process on the former, as
// the reference's fatal paths do (exit(1), lib/sba-1.5/sba_levmar.c:72-83), and carries on after the latter, as the reference does.
static thread_local int g_infra_failure = 0;
extern "C" int bsfm_last_call_infra_failure(void) { return g_infra_failure; }
extern "C" void bsfm_clear_infra_failure(void) { g_infra_failure = 0; }

#define HIP_OK(call)                                                                               \
    do {                                                                                           \
        hipError_t _e = (call);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            fprintf(stderr, "[bsfm] HIP error %s at %s:%d: %s\n", hipGetErrorName(_e), __FILE__,   \
                    __LINE__, #call);                                                              \
            g_infra_failure = 1;                                                                   \
            return BSFM_ERROR;                                                                     \
        }                                                                                          \
    } while (0)

namespace {

constexpr double SBA_EPSILON_SQ = 1E-12 * 1E-12;   // lib/sba-1.5/sba_levmar.c:35-36
constexpr double SBA_ONE_THIRD = 0.3333333334;     // lib/sba-1.5/sba_levmar.c:38

enum Phase { PH_JAC = 0, PH_CAMBLK, PH_PTBLK, PH_INVERT, PH_SCHUR, PH_SOLVE, PH_BACKSUB, PH_RESID, PH_SCHUR_PREP, PH_SCHUR_TASKS, PH_COUNT };
const char* kPhaseNames[PH_COUNT] = { "jacobian", "cam_blocks", "point_blocks", "point_invert", "schur",
                                      "solve", "backsub", "residual", "schur_prep", "schur_tasks" };

// scalar block layout (device doubles)
enum Scal { SC_COST = 0, SC_COST_TRIAL, SC_PCT, SC_CAM3 /*3*/, SC_PT_DP = 6, SC_PT_P, SC_PT_DL,
            SC_EABINF_A, SC_EABINF_B, SC_MAXDIAG_U, SC_MAXDIAG_V, SC_PL2_A, SC_PL2_B, SC_CCOST, SC_COUNT = 24 };

template <typename T> hipError_t dmalloc(T** p, size_t count)
{
    return bsfm::dev_alloc(reinterpret_cast<void**>(p), std::max<size_t>(count, 1) * sizeof(T));
}

}  // namespace

// ---- device block cache (devcache.h) --------------------------------------------------------------------------------------
namespace bsfm {
namespace {
struct DevCache {
    // idle blocks by (device, size class) -- constant-time hand-out however many classes an incremental reconstruction leaves
    // behind -- plus their order of arrival for the eviction (entries of blocks that went out again are skipped lazily)
    std::mutex mu;
    std::unordered_map<unsigned long long, std::vector<void*>> bins;
    std::deque<std::pair<unsigned long long, void*>> fifo;
    std::unordered_map<void*, unsigned long long> idle_key;     // idle block -> its bin
    struct Live { size_t cls; int dev; };
    std::unordered_map<void*, Live> live;                        // class size and ALLOCATING device of every block handed out
    size_t idle_bytes = 0, cap = 0;
    DevCache()
    {
        long mb = 6144;
        if (const char* e = getenv("BSFM_DEVCACHE_MB")) mb = atol(e);
        cap = mb > 0 ? (size_t)mb << 20 : 0;
    }
    static size_t size_class(size_t bytes)
    {
        if (bytes <= 4096) return 4096;
        int lg = 63 - __builtin_clzll((unsigned long long)bytes);
        const size_t step = (size_t)1 << (lg - 3);               // eight classes per octave
        return (bytes + step - 1) / step * step;
    }
    static unsigned long long key_of(int dev, size_t cls) { return ((unsigned long long)(unsigned)dev << 48) ^ (unsigned long long)cls; }
    static size_t class_of(unsigned long long key) { return (size_t)(key & ((1ull << 48) - 1)); }
    void drop_locked(unsigned long long key, void* p)             // an idle block back to the driver
    {
        std::vector<void*>& v = bins[key];
        for (size_t i = 0; i < v.size(); ++i) if (v[i] == p) { v[i] = v.back(); v.pop_back(); break; }
        idle_key.erase(p);
        idle_bytes -= class_of(key);
        (void)hipFree(p);
    }
    void trim_locked()
    {
        for (auto& kv : bins) for (void* p : kv.second) (void)hipFree(p);
        bins.clear(); fifo.clear(); idle_key.clear(); idle_bytes = 0;
    }
};
DevCache& dev_cache() { static DevCache* c = new DevCache(); return *c; }   // leaked on purpose (no teardown order issues at exit)
}  // namespace

hipError_t dev_alloc(void** p, size_t bytes)
{
    DevCache& c = dev_cache();
    const size_t cls = DevCache::size_class(bytes);
    int dev = 0; (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(c.mu);
    auto it = c.bins.find(DevCache::key_of(dev, cls));
    if (it != c.bins.end() && !it->second.empty()) {
        *p = it->second.back(); it->second.pop_back();
        c.idle_key.erase(*p); c.idle_bytes -= cls;
        c.live[*p] = DevCache::Live{ cls, dev };
        return hipSuccess;
    }
    hipError_t e = hipMalloc(p, cls);
    if (e != hipSuccess && c.idle_bytes) {           // the free list may be what is in the way: give it back and try again
        (void)hipGetLastError();
        (void)hipDeviceSynchronize();
        c.trim_locked();
        e = hipMalloc(p, cls);
    }
    if (e == hipSuccess) c.live[*p] = DevCache::Live{ cls, dev };
    return e;
}

void dev_free(void* p, bool synced)
{
    if (!p) return;
    DevCache& c = dev_cache();
    std::unique_lock<std::mutex> lk(c.mu);
    auto it = c.live.find(p);
    if (it == c.live.end()) { lk.unlock(); (void)hipFree(p); return; }     // not one of ours
    const size_t cls = it->second.cls;
    const int dev = it->second.dev;              // the device the block lives on, NOT the one that happens to be current (ADVICE r3:
    c.live.erase(it);                            // a problem created on GPU 1 and destroyed with GPU 0 current used to land in GPU 0's bins)
    if (c.cap == 0 || cls > c.cap) { lk.unlock(); (void)hipFree(p); return; }
    if (!synced) {
        // nothing may still be using the block: synchronise ITS device
        lk.unlock();
        int cur = 0; (void)hipGetDevice(&cur);
        if (cur != dev) (void)hipSetDevice(dev);
        (void)hipDeviceSynchronize();
        if (cur != dev) (void)hipSetDevice(cur);
        lk.lock();
    }
    const unsigned long long key = DevCache::key_of(dev, cls);
    c.bins[key].push_back(p); c.idle_key[p] = key; c.fifo.emplace_back(key, p); c.idle_bytes += cls;
    size_t guard = c.fifo.size();
    while (c.idle_bytes > c.cap && guard-- > 0) {                           // the oldest idle blocks make room
        const auto old = c.fifo.front(); c.fifo.pop_front();
        auto ik = c.idle_key.find(old.second);
        if (ik == c.idle_key.end() || ik->second != old.first) continue;    // stale: that block went out again since
        if (old.second == p) { c.fifo.push_back(old); continue; }           // the newcomer stays (its turn comes later)
        c.drop_locked(old.first, old.second);
    }
    if (c.fifo.size() > 4 * (c.idle_key.size() + 16)) {                     // compact the arrival list: one entry per idle block
        std::deque<std::pair<unsigned long long, void*>> keep;
        std::unordered_map<void*, int> seen;
        for (auto e = c.fifo.rbegin(); e != c.fifo.rend(); ++e) {           // (the newest entry of a block is the one that counts)
            auto ik = c.idle_key.find(e->second);
            if (ik != c.idle_key.end() && ik->second == e->first && !seen[e->second]++) keep.push_front(*e);
        }
        c.fifo.swap(keep);
    }
}
}  // namespace bsfm

extern "C" void bsfm_device_cache_trim(void)
{
    (void)hipDeviceSynchronize();
    {
        bsfm::DevCache& c = bsfm::dev_cache();
        std::lock_guard<std::mutex> lk(c.mu);
        c.trim_locked();
    }
    // the stream-ordered pools the index construction draws its sort scratch from keep their pages too
    int dev = 0;
    hipMemPool_t pool = nullptr;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess && pool) (void)hipMemPoolTrimTo(pool, 0);
    bsfm::index_pool_trim();
    (void)hipGetLastError();
}

struct bsfm_problem {
    bsfm_options_t opt;
    DevProblem P{};
    int cnp = 0, nvars_local = 0;
    int world = 1, rank = 0;
    int mot = 0;                        // camera-only refinement (desc.fix_points)
    long long nvis_global = 0, nvars_global = 0;
    int Sdim = 0, ld = 0;
    // device
    double *d_x = nullptr, *d_xc = nullptr, *d_Rinit = nullptr, *d_finit = nullptr;
    double *d_known = nullptr;          // m x CT_EXT: known-intrinsics / fisheye block of the camera table (model.hip.h); null when unused
    int fisheye_mode = 0;               // run_sfm(optimize_for_fisheye): sfm_project_point2_fisheye instead of sfm_project_point3
    int *d_obs_cam = nullptr, *d_obs_pt = nullptr, *d_rowptr = nullptr, *d_camptr = nullptr, *d_camobs = nullptr;
    int *d_campos = nullptr, *d_cam_pt = nullptr, *d_cam_cam = nullptr;
    unsigned char *d_ccon = nullptr, *d_pcon = nullptr;
    double *d_cval = nullptr, *d_cw = nullptr, *d_pval = nullptr;
    double *d_p = nullptr, *d_pdp = nullptr, *d_dp = nullptr;
    double *d_camtab = nullptr, *d_camtab_trial = nullptr;
    double *d_e = nullptr, *d_hx = nullptr;
    // camera-major mirrors of the point part of parameter vectors (32 bytes per observation; kernels.hip.h: k_point_mirror): slot s mirrors
    // the vector ptc_tag[s] points at (d_p or d_pdp -- the tags follow the pointers through the accept swap), null = stale
    double* d_ptc[2] = { nullptr, nullptr }; const double* ptc_tag[2] = { nullptr, nullptr };
    double *d_Ac = nullptr, *d_Bc = nullptr, *d_Cc = nullptr, *d_U = nullptr, *d_ea = nullptr, *d_V = nullptr, *d_Vinv = nullptr, *d_eb = nullptr;
    double *d_S = nullptr, *d_E = nullptr;
    int export_full_s = 0;              // bsfm_eval_normal_equations hands S out as a full symmetric matrix: clear all of it
    double *d_partials = nullptr;       // schur task partials
    double *d_epart = nullptr;          // ... and their parts of the reduced right-hand side (diagonal-block tasks)
    double *d_campart = nullptr;        // per-camera slice partials of k_cam_blocks / k_schur_rhs (m x CAM_SPLIT x 54)
    double *d_red = nullptr;            // block partials for reductions
    double *d_scal = nullptr;
    double *d_mixed = nullptr;          // staging of allreduce_mixed: a few sums + world slots per maximum
    int *d_flags = nullptr;             // [0] singular V, [1] potrf info
    bool backsub_two_pass = false;      // k_backsub_obs + gather (nvis >= 400 000, or BSFM_BACKSUB_TWO_PASS=0|1); below: four lanes per point in k_backsub and k_point_blocks
    size_t tick_res = 0, tick_back = 0; // word offsets of the group tickets in d_tickets
    size_t tick_words = 0;
    unsigned* d_tickets = nullptr;      // "last workgroup finishes the job" tickets (kernels.hip.h): [0] residual, [1] iteration scalars, [2] back-substitution, [8 ..) one per camera
    // schur structure
    int ntriples = 0, ntasks = 0, nblk = 0;
    int2* d_triples = nullptr; SchurTask* d_tasks = nullptr; int* d_tri_pt = nullptr;
    int nslots = 0;                     // entries of d_tasks (launch order, padded to whole workgroups)
    int *d_blk_j = nullptr, *d_blk_k = nullptr, *d_blk_task0 = nullptr;
    // d_blk_range = partial-sum slots per block (its tasks', in task order); d_tasks_launch = the task kernel's launch list (== d_tasks)
    int2* d_blk_range = nullptr;
    SchurTask* d_tasks_launch = nullptr;
    // multi-GPU exchange of the reduced camera system: the UNION over ranks of the non-empty blocks S_jk (j <= k),
    // one cnp x cnp sum per block, is what crosses xGMI -- not the dense (9m)^2 matrix
    std::vector<int> h_blk_j, h_blk_k;
    int ngblk = -1;                       // -1: union not exchanged yet
    int *d_gidx = nullptr, *d_gblk_j = nullptr, *d_gblk_k = nullptr;
    double* d_G = nullptr;
    // host
    double* h_scal = nullptr; int* h_flags = nullptr;   // pinned
    std::vector<double> h_Rinit;
    std::vector<bsfm_camera_params_t> h_cams;      // the caller's camera structs (constraints, known intrinsics ...): template for bsfm_problem_append
    bsfm_problem_desc_t desc0{};                  // scalar fields of the description the problem was created from
    hipStream_t stream = nullptr; bool own_stream = false;
    int flow_fallbacks = 0;             // solves repeated on the stream-ordered Cholesky schedule after a hand-off time-out of the dataflow launch
    bool empty_rows = false;           // some point has no observation (index build)
    bool fuse_invert = false;          // V*^-1 is computed inside k_schur_prep (every point has an observation; BSFM_FUSE_INVERT=0|1)
    bool speculate = false;            // launch the first damping attempt of an iteration before its gradient test is read (small problems; BSFM_SPECULATE=0|1)
    bsfm_allreduce_fn allreduce = nullptr; void* allreduce_ctx = nullptr;
    bsfm_comm_t* comm = nullptr;        // library-side collective (comm.hip: RCCL over xGMI); takes precedence over the hook
    PotrfWorkspace potrf;
    CompSolver comps;                   // opt-in: independent camera groups solved one workgroup each (compsolve.hip.h)
    // opt-in envelope solver: free camera c sits at position h_spos[c] of the reordered reduced system (reverse Cuthill-McKee);
    // the tile envelope itself lives in potrf.env_rows / potrf.d_last
    std::vector<int> h_spos; int* d_spos = nullptr; double* d_xperm = nullptr; bool envelope = false;
    // LM state (names follow sba_levmar.c)
    int itno = 0, stop = 0, nu = 2, nfev = 0, njev = 0, nlss = 0, began = 0, error = 0;
    double mu = 0.0, p_eL2 = 0.0, init_p_eL2 = 0.0, eab_inf = 0.0, dp_L2 = DBL_MAX, p_L2 = 0.0, maxdiag = DBL_MIN;
    // timing
    hipEvent_t ev[PH_COUNT][2] = {}; bool ev_ok = false, ev_created = false;
    double ph_ms[PH_COUNT]; int ph_cnt[PH_COUNT];
    int red_blocks = 0;
    double index_build_ms = 0.0;        // device time of the index construction (index_build.hip)
    double create_ms[4] = { 0, 0, 0, 0 };   // host wall time of problem_create: total, upload, index, allocation
};

namespace {

void free_all(bsfm_problem* pb)
{
    (void)hipDeviceSynchronize();          // nothing may still be using the blocks that go back to the cache
    void* ptrs[] = { pb->d_x, pb->d_xc, pb->d_Rinit, pb->d_finit, pb->d_known, pb->d_obs_cam, pb->d_obs_pt, pb->d_rowptr, pb->d_camptr,
                     pb->d_camobs, pb->d_campos, pb->d_cam_pt, pb->d_cam_cam, pb->d_Ac, pb->d_Bc, pb->d_Cc, pb->d_ccon, pb->d_pcon, pb->d_cval, pb->d_cw, pb->d_pval, pb->d_p, pb->d_pdp,
                     pb->d_dp, pb->d_camtab, pb->d_camtab_trial, pb->d_e, pb->d_hx, pb->d_ptc[0], pb->d_ptc[1], pb->d_U,
                     pb->d_V, pb->d_Vinv, pb->d_eb, pb->d_S, pb->d_E, pb->d_partials, pb->d_epart, pb->d_campart, pb->d_red, pb->d_scal, pb->d_mixed, pb->d_tickets,
                     pb->d_triples, pb->d_tri_pt, pb->d_tasks, pb->d_blk_j, pb->d_blk_k, pb->d_blk_task0,
                     pb->d_blk_range,
                     pb->d_gidx, pb->d_gblk_j, pb->d_gblk_k, pb->d_G, pb->d_spos, pb->d_xperm };
    for (void* p : ptrs) bsfm::dev_free(p, true);          // bsfm_problem_destroy has synchronised the device
    if (pb->h_scal) (void)hipHostFree(pb->h_scal);
    potrf_free(pb->potrf);
    comp_free(pb->comps);
    if (pb->ev_created) for (int i = 0; i < PH_COUNT; ++i) { if (pb->ev[i][0]) (void)hipEventDestroy(pb->ev[i][0]); if (pb->ev[i][1]) (void)hipEventDestroy(pb->ev[i][1]); }
    if (pb->own_stream && pb->stream) stream_pool().release(pb->stream);
}

int setup_components(bsfm_problem* pb, const std::vector<int>& bj, const std::vector<int>& bk);

// Takes over the arrays of the device-side index construction (index_build.hip): camera-major maps and, unless the problem is
// camera-only, the co-visibility triples bucketed by reduced-camera block (j <= k) in (j,k) order and, inside a block, in
// point order -- the order the reference visits them (sba_levmar.c:1218-1268) -- cut into tasks of <= schur_chunk() triples.
int adopt_index(bsfm_problem* pb, DeviceIndex& ix)
{
    pb->d_obs_pt = ix.obs_pt; pb->d_camptr = ix.camptr; pb->d_camobs = ix.camobs; pb->d_campos = ix.campos;
    pb->d_cam_pt = ix.cam_pt; pb->d_cam_cam = ix.cam_cam;
    pb->d_triples = ix.triples; pb->d_tri_pt = ix.tri_pt; pb->d_tasks = ix.tasks;
    pb->d_blk_j = ix.blk_j; pb->d_blk_k = ix.blk_k; pb->d_blk_task0 = ix.blk_task0;
    pb->ntriples = ix.ntriples; pb->ntasks = ix.ntasks; pb->nblk = ix.nblk; pb->nslots = ix.nslots;
    pb->d_blk_range = ix.blk_range;
    pb->d_tasks_launch = ix.tasks_launch;
    pb->h_blk_j.swap(ix.h_blk_j); pb->h_blk_k.swap(ix.h_blk_k);
    pb->index_build_ms = ix.build_ms;
    pb->empty_rows = ix.empty_rows;
    pb->fuse_invert = pb->fuse_invert && !pb->empty_rows;
    ix = DeviceIndex();                                  // ownership moved: free_all releases the arrays
    if (pb->mot) return 0;
    if (pb->world == 1 && setup_components(pb, pb->h_blk_j, pb->h_blk_k)) return BSFM_ERROR;   // world > 1: after the block-union exchange
    HIP_OK(dmalloc(&pb->d_partials, (size_t)pb->ntasks * pb->cnp * pb->cnp));      // one slot per task
    HIP_OK(dmalloc(&pb->d_epart, (size_t)pb->ntasks * pb->cnp));
    return 0;
}

// Extended camera-model block (model.hip.h, CT_KN): known intrinsics (sfm.c:339-358) and the fisheye projection
// (sfm.c:426-492).  Returns false (and leaves `kn` empty) when no camera needs it.
bool fill_ext_block(const bsfm_camera_params_t* cams, int m, int fisheye_mode, std::vector<double>& kn)
{
    bool any = fisheye_mode != 0;
    for (int j = 0; j < m && !any; ++j) any = cams[j].known_intrinsics != 0;
    if (!any) return false;
    kn.assign((size_t)m * CT_EXT, 0.0);
    for (int j = 0; j < m; ++j) {
        double* q = &kn[(size_t)j * CT_EXT];
        if (cams[j].known_intrinsics) {
            q[0] = 1.0;
            for (int t = 0; t < 5; ++t) q[1 + t] = cams[j].k_known[t];
            q[6] = cams[j].K_known[0]; q[7] = cams[j].K_known[1]; q[8] = cams[j].K_known[2];
            q[9] = cams[j].K_known[4]; q[10] = cams[j].K_known[5];
        }
        if (fisheye_mode) {
            q[11] = 1.0;
            q[12] = cams[j].fisheye ? 1.0 : 0.0;
            q[13] = cams[j].f_cx; q[14] = cams[j].f_cy; q[15] = cams[j].f_rad; q[16] = cams[j].f_angle; q[17] = cams[j].f_focal;
        }
    }
    return true;
}

void pack_params(const bsfm_problem* pb, const bsfm_camera_params_t* cams, const double* pts, int n,
                 std::vector<double>& p)
{
    // lib/sfm-driver/sfm.c:652-703
    const int cnp = pb->cnp, m = pb->P.m;
    const ModelCfg& c = pb->P.cfg;
    p.assign((size_t)m * cnp + (size_t)3 * n, 0.0);
    for (int j = 0; j < m; ++j) {
        double* a = &p[(size_t)j * cnp];
        a[0] = cams[j].t[0]; a[1] = cams[j].t[1]; a[2] = cams[j].t[2];
        int col = 6;
        if (c.est_focal) { a[6] = cams[j].f * c.f_scale; col = 7; }
        if (c.undistort) { a[col] = cams[j].k[0] * c.k_scale; a[col + 1] = cams[j].k[1] * c.k_scale; }
    }
    if (n && pts) memcpy(&p[(size_t)m * cnp], pts, sizeof(double) * 3 * n);
}

inline void ph_begin(bsfm_problem* pb, int ph) { if (pb->ev_ok) (void)hipEventRecord(pb->ev[ph][0], pb->stream); }
inline void ph_end(bsfm_problem* pb, int ph) { if (pb->ev_ok) (void)hipEventRecord(pb->ev[ph][1], pb->stream); }

#define DISPATCH_CNP(cnp, EXPR)                   \
    switch (cnp) {                                \
        case 6: { constexpr int C = 6; EXPR; } break; \
        case 7: { constexpr int C = 7; EXPR; } break; \
        case 8: { constexpr int C = 8; EXPR; } break; \
        default: { constexpr int C = 9; EXPR; } break; \
    }

inline int grid_for(size_t count, int block) { return (int)std::max<size_t>(1, (count + block - 1) / block); }

void launch_cam_table(bsfm_problem* pb, const double* p, double* camtab)
{
    hipLaunchKernelGGL(k_cam_table, dim3(grid_for((size_t)pb->P.m * CT_PARTS, 64)), dim3(64), 0, pb->stream, pb->P.cfg, pb->P.m, p,
                       pb->d_Rinit, pb->d_finit, pb->d_known, pb->opt.jacobian == BSFM_JAC_FD ? 1 : 0, camtab);
}

// The camera-major mirror of the points of parameter vector `p` (d_p or d_pdp): the slot tagged with it, else built now by a gather
// into the slot that does not hold the mirror of the accepted parameters.  k_backsub writes the mirror of every trial vector itself
// (lm_iterate), so inside the LM loop this only ever finds a tag.
void invalidate_point_mirrors(bsfm_problem* pb) { pb->ptc_tag[0] = pb->ptc_tag[1] = nullptr; }
int trial_mirror_slot(const bsfm_problem* pb) { return pb->ptc_tag[0] == pb->d_p ? 1 : 0; }
const double* points_mirror(bsfm_problem* pb, const double* p)
{
    for (int s = 0; s < 2; ++s) if (pb->ptc_tag[s] == p) return pb->d_ptc[s];
    const int s = trial_mirror_slot(pb);
    if (pb->P.nvis > 0)
        hipLaunchKernelGGL(k_point_mirror, dim3(grid_for(pb->P.nvis, 256)), dim3(256), 0, pb->stream, pb->P.nvis, pb->d_cam_pt,
                           p + (size_t)pb->P.m * pb->cnp, pb->d_ptc[s]);
    pb->ptc_tag[s] = p;
    return pb->d_ptc[s];
}

// e_out = x - proj(p) (camera-major order); SC slot gets sum e^2 ; optional pct-change vs e_prev into SC_PCT.
// p_points: the parameter vector whose POINTS are projected (the cameras come through camtab) -- the camera-only refinement evaluates
// trial cameras against the points of the accepted vector.
void launch_residual(bsfm_problem* pb, const double* camtab, const double* p_points, double* e_out,
                     const double* e_prev, int cost_slot)
{
    const int nb = grid_for(pb->P.nvis, RES_BLOCK);
    double* pc = pb->d_red, *pp = pb->d_red + pb->red_blocks;
    const double* ptc = points_mirror(pb, p_points);
    if (pb->P.nvis > 0 && pb->d_known)
        hipLaunchKernelGGL(k_residual<true>, dim3(nb), dim3(RES_BLOCK), 0, pb->stream, pb->P.cfg, pb->P.nvis, pb->d_xc,
                           pb->d_cam_cam, ptc, camtab, e_out, e_prev, pb->opt.opts[5], pc,
                           e_prev ? pp : nullptr, pb->d_tickets + 0, pb->d_tickets + pb->tick_res, pb->d_scal + cost_slot, pb->d_scal + SC_PCT);
    if (pb->P.nvis > 0 && !pb->d_known)
        hipLaunchKernelGGL(k_residual<false>, dim3(nb), dim3(RES_BLOCK), 0, pb->stream, pb->P.cfg, pb->P.nvis, pb->d_xc,
                           pb->d_cam_cam, ptc, camtab, e_out, e_prev, pb->opt.opts[5], pc,
                           e_prev ? pp : nullptr, pb->d_tickets + 0, pb->d_tickets + pb->tick_res, pb->d_scal + cost_slot, pb->d_scal + SC_PCT);
    // (the sum over the workgroups' partial costs -- k_reduce_sum_max -- is done by the workgroup of k_residual that arrives last)
    if (pb->P.nvis <= 0)
        hipLaunchKernelGGL(k_reduce_sum_max, dim3(1), dim3(256), 0, pb->stream, pc, e_prev ? pp : (const double*)nullptr, 0,
                           pb->d_scal + cost_slot, pb->d_scal + SC_PCT);
}

int read_scalars(bsfm_problem* pb)
{
    // scalars and the two flag words in ONE copy (the flags sit right behind the scalar block on both sides)
    HIP_OK(hipMemcpyAsync(pb->h_scal, pb->d_scal, (SC_COUNT + 16 + 2) * sizeof(double), hipMemcpyDeviceToHost, pb->stream));
    HIP_OK(hipStreamSynchronize(pb->stream));
    return 0;
}

inline bool has_collective(const bsfm_problem* pb) { return pb->world > 1 && (pb->comm || pb->allreduce); }

// cross-rank reductions of host scalars go through the same device path (tiny device buffer)
int allreduce_host(bsfm_problem* pb, double* vals, int count, int op)
{
    if (!has_collective(pb)) return 0;
    double* tmp = pb->d_scal + SC_COUNT;   // spare slots
    HIP_OK(hipMemcpyAsync(tmp, vals, count * sizeof(double), hipMemcpyHostToDevice, pb->stream));
    if (pb->comm) {
        if (bsfm_comm_allreduce(pb->comm, tmp, (size_t)count, op, pb->stream) != 0) { g_infra_failure = 1; return BSFM_ERROR; }
        HIP_OK(hipMemcpyAsync(vals, tmp, count * sizeof(double), hipMemcpyDeviceToHost, pb->stream));
        HIP_OK(hipStreamSynchronize(pb->stream));
        return 0;
    }
    HIP_OK(hipStreamSynchronize(pb->stream));
    if (pb->allreduce(tmp, (size_t)count, op, pb->allreduce_ctx) != 0) { g_infra_failure = 1; return BSFM_ERROR; }
    HIP_OK(hipMemcpy(vals, tmp, count * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}
// One exchange for a handful of host scalars of both kinds: `ns` values to be summed and `nm` values to be maximised.
// The maxima ride the same SUM as an all-gather: value q of rank r sits in slot q*world + r, zeros elsewhere.
int allreduce_mixed(bsfm_problem* pb, double* sums, int ns, double* maxs, int nm)
{
    if (!has_collective(pb)) return 0;
    const int count = ns + nm * pb->world;
    std::vector<double> h((size_t)count, 0.0);
    for (int q = 0; q < ns; ++q) h[q] = sums[q];
    for (int q = 0; q < nm; ++q) h[ns + q * pb->world + pb->rank] = maxs[q];
    double* tmp = pb->d_mixed;
    HIP_OK(hipMemcpyAsync(tmp, h.data(), (size_t)count * sizeof(double), hipMemcpyHostToDevice, pb->stream));
    if (pb->comm) {
        if (bsfm_comm_allreduce(pb->comm, tmp, (size_t)count, 0, pb->stream) != 0) { g_infra_failure = 1; return BSFM_ERROR; }
        HIP_OK(hipMemcpyAsync(h.data(), tmp, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, pb->stream));
        HIP_OK(hipStreamSynchronize(pb->stream));
    } else {
        HIP_OK(hipStreamSynchronize(pb->stream));
        if (pb->allreduce(tmp, (size_t)count, 0, pb->allreduce_ctx) != 0) { g_infra_failure = 1; return BSFM_ERROR; }
        HIP_OK(hipMemcpy(h.data(), tmp, (size_t)count * sizeof(double), hipMemcpyDeviceToHost));
    }
    for (int q = 0; q < ns; ++q) sums[q] = h[q];
    for (int q = 0; q < nm; ++q) {
        double m = h[ns + q * pb->world];
        for (int r = 1; r < pb->world; ++r) m = std::max(m, h[ns + q * pb->world + r]);
        maxs[q] = m;
    }
    return 0;
}
int allreduce_dev(bsfm_problem* pb, double* dbuf, size_t count, int op)
{
    if (!has_collective(pb)) return 0;
    if (pb->comm) { const int rc = bsfm_comm_allreduce(pb->comm, dbuf, count, op, pb->stream); if (rc) g_infra_failure = 1; return rc; }      // enqueued on the compute stream: no host hop
    HIP_OK(hipStreamSynchronize(pb->stream));
    if (pb->allreduce(dbuf, count, op, pb->allreduce_ctx) != 0) { fprintf(stderr, "[bsfm] allreduce hook failed\n"); g_infra_failure = 1; return BSFM_ERROR; }
    return 0;
}

void collect_phase_times(bsfm_problem* pb)
{
    if (!pb->ev_ok) return;
    for (int i = 0; i < PH_COUNT; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pb->ev[i][0], pb->ev[i][1]) == hipSuccess && ms >= 0.f) { pb->ph_ms[i] += ms; pb->ph_cnt[i]++; }
    }
    potrf_collect_time(pb->potrf);
    if (pb->potrf.flow) flow_collect_time(*pb->potrf.flow);
    (void)hipGetLastError();      // phases that did not run this iteration leave hipErrorInvalidHandle behind: do not let it stick
}

// J, U/ea, V/eb at the current p
// with_iter_scalars: the point-block kernel also leaves the iteration's scalars (sba_levmar.c:1085-1128) in d_scal and clears the flags of the
// first solve attempt (round 6: was k_iter_partials + a memset); *did_scalars says whether it did (no points: the caller runs k_iter_final)
int compute_normal_blocks(bsfm_problem* pb, bool with_iter_scalars = false, bool* did_scalars = nullptr)
{
    if (did_scalars) *did_scalars = false;
    const int cnp = pb->cnp;
    DevProblem& P = pb->P;
    const double* pbpts = pb->d_p + (size_t)P.m * cnp;
    const double* ptc = points_mirror(pb, pb->d_p);          // (a tag hit inside the LM loop: lm_begin / the accepted trial wrote it)
    ph_begin(pb, PH_JAC);
    if (P.nvis > 0) {
        if (pb->opt.jacobian == BSFM_JAC_FD) {
            if (pb->d_known) {
                DISPATCH_CNP(cnp, hipLaunchKernelGGL((k_jacobian<C, true, true>), dim3(grid_for(P.nvis, 256)), dim3(256), 0, pb->stream,
                                                      P.cfg, P.nvis, pb->d_cam_cam, ptc, pb->d_camtab, pb->d_e, pb->d_Ac, pb->d_Bc));
            } else
            DISPATCH_CNP(cnp, hipLaunchKernelGGL((k_jacobian<C, true, false>), dim3(grid_for(P.nvis, 256)), dim3(256), 0, pb->stream,
                                                  P.cfg, P.nvis, pb->d_cam_cam, ptc, pb->d_camtab, pb->d_e, pb->d_Ac, pb->d_Bc));
        } else {
            DISPATCH_CNP(cnp, hipLaunchKernelGGL((k_jacobian<C, false, false>), dim3(grid_for(P.nvis, 256)), dim3(256), 0, pb->stream,
                                                  P.cfg, P.nvis, pb->d_cam_cam, ptc, pb->d_camtab, pb->d_e, pb->d_Ac, pb->d_Bc));
        }
    }
    ph_end(pb, PH_JAC);
    ph_begin(pb, PH_CAMBLK);
    // (k_cam_blocks_fin's sums over the CAM_SPLIT slices of a camera are done by the slice workgroup that arrives last)
    // single GPU: the camera constraints (sba_levmar.c:953-962) are added by the workgroup that finishes a camera's sums -- one launch fewer, and
    // Bundler always constrains (focal length, distortion: src/Bundle.cpp:942-974); several ranks: after the reduction of U and ea, below
    const double* pa_con = (P.ccon && pb->world == 1) ? (const double*)pb->d_p : nullptr;
    DISPATCH_CNP(cnp, hipLaunchKernelGGL((k_cam_blocks<C>), dim3(P.m * CAM_SPLIT), dim3(256), 0, pb->stream, P, pb->d_e, pb->d_campart, pb->d_tickets + 8, pa_con));
    ph_end(pb, PH_CAMBLK);
    if (pb->world > 1) {   // U and ea are sums over ALL points: exchange step 1, 90*m doubles
        if (allreduce_dev(pb, pb->d_U, (size_t)P.m * cnp * cnp + (size_t)P.m * cnp, 0)) return BSFM_ERROR;   // ea follows U
    }
    if (P.ccon && !pa_con)
        hipLaunchKernelGGL(k_cam_constraints, dim3(grid_for((size_t)P.m * cnp, 256)), dim3(256), 0, pb->stream, P, pb->d_p);
    ph_begin(pb, PH_PTBLK);
    if (P.n > 0 && !pb->mot) {
        if (with_iter_scalars) {
            IterFinalArgs fa; fa.pa = pb->d_p; fa.have_points = 1; fa.point_part_slot = pb->world > 1 ? SC_COUNT + 8 : -1;
            fa.s_eabinf_a = SC_EABINF_A; fa.s_eabinf_b = SC_EABINF_B; fa.s_maxdiag_u = SC_MAXDIAG_U; fa.s_maxdiag_v = SC_MAXDIAG_V;
            fa.s_pl2_a = SC_PL2_A; fa.s_pl2_b = SC_PL2_B; fa.s_ccost = SC_CCOST; fa.scal = pb->d_scal;
            // (small problems: four lanes per point; the camera side of the scalars is one more workgroup behind the points')
            if (pb->backsub_two_pass) {
                fa.point_blocks = grid_for(P.n, 256);
                DISPATCH_CNP(cnp, hipLaunchKernelGGL((k_point_blocks<C, 1>), dim3(fa.point_blocks + 1), dim3(256), 0, pb->stream, P, pbpts, pb->d_red, pb->d_tickets + 1,
                                                      pb->d_tickets + pb->tick_back, fa, pb->d_flags));
            } else {
                fa.point_blocks = grid_for((size_t)BS_LANES * (size_t)P.n, 256);
                DISPATCH_CNP(cnp, hipLaunchKernelGGL((k_point_blocks<C, 4>), dim3(fa.point_blocks + 1), dim3(256), 0, pb->stream, P, pbpts, pb->d_red, pb->d_tickets + 1,
                                                      pb->d_tickets + pb->tick_back, fa, pb->d_flags));
            }
            if (did_scalars) *did_scalars = true;
        } else DISPATCH_CNP(cnp, hipLaunchKernelGGL((k_point_blocks<C, 1>), dim3(grid_for(P.n, 256)), dim3(256), 0, pb->stream, P, pbpts));
    }
    ph_end(pb, PH_PTBLK);
    return 0;
}

// Opt-in envelope solver (BSFM_SOLVER_ENVELOPE, or BSFM_SOLVER_AUTO on a scene that does not fall into small groups).
// The reduced camera system of a real reconstruction is sparse in 9 x 9 blocks (S_jk != 0 only when cameras j and k share a point)
// but the reference factors it densely (sba_Axb_Chol).  Cholesky without pivoting creates no fill outside the ENVELOPE of the matrix
// (row i: from its first non-zero column to the diagonal), so after a bandwidth-reducing renumbering of the cameras -- reverse
// Cuthill-McKee on the co-visibility graph, the blocks list built once per problem -- whole 128 x 128 tiles of the factor are
// structurally zero and the tiled factorisation of potrf.hip.h simply skips them: step k works on env_rows[k] tile rows instead of
// all rows below k.  Exact; only the summation order inside the skipped (all-zero) products differs from the dense path.
int setup_envelope(bsfm_problem* pb, const std::vector<int>& bj, const std::vector<int>& bk)
{
    const int mm = pb->P.m - pb->P.mcon, mcon = pb->P.mcon, cnp = pb->cnp;
    pb->envelope = false;
    if (mm <= 0 || pb->opt.potrf_backend != 0) return 0;
    // ---- reverse Cuthill-McKee
    std::vector<std::vector<int>> adj((size_t)mm);
    for (size_t b = 0; b < bj.size(); ++b) {
        const int a = bj[b] - mcon, c = bk[b] - mcon;
        if (a != c) { adj[a].push_back(c); adj[c].push_back(a); }
    }
    std::vector<int> deg((size_t)mm);
    for (int j = 0; j < mm; ++j) deg[j] = (int)adj[j].size();
    for (int j = 0; j < mm; ++j) std::sort(adj[j].begin(), adj[j].end(), [&](int x, int y) { return deg[x] != deg[y] ? deg[x] < deg[y] : x < y; });
    std::vector<int> order; order.reserve((size_t)mm);
    std::vector<char> seen((size_t)mm, 0);
    std::vector<int> byDeg((size_t)mm);
    for (int j = 0; j < mm; ++j) byDeg[j] = j;
    std::sort(byDeg.begin(), byDeg.end(), [&](int x, int y) { return deg[x] != deg[y] ? deg[x] < deg[y] : x < y; });
    auto bfs_far = [&](int start, std::vector<int>& visited_out) {     // last node of a breadth-first sweep (pseudo-peripheral search)
        std::vector<int> q{ start }; std::vector<char> mark((size_t)mm, 0); mark[start] = 1;
        for (size_t h = 0; h < q.size(); ++h) for (int v : adj[q[h]]) if (!mark[v] && !seen[v]) { mark[v] = 1; q.push_back(v); }
        visited_out = q;
        return q.back();
    };
    for (int s0 : byDeg) {
        if (seen[s0]) continue;
        std::vector<int> comp;
        int start = bfs_far(s0, comp);
        start = bfs_far(start, comp);                                  // two sweeps: a node far from a node far from the seed
        std::vector<int> q{ start }; seen[start] = 1;
        for (size_t h = 0; h < q.size(); ++h) for (int v : adj[q[h]]) if (!seen[v]) { seen[v] = 1; q.push_back(v); }
        order.insert(order.end(), q.begin(), q.end());
    }
    std::reverse(order.begin(), order.end());
    pb->h_spos.assign((size_t)mm, 0);
    for (int p = 0; p < mm; ++p) pb->h_spos[order[p]] = p;
    // ---- tile envelope of the reordered matrix (lower triangle: row tile I, first non-zero column tile first[I])
    const int nt = pb->ld / POTRF_NB;
    std::vector<int> first((size_t)nt);
    for (int I = 0; I < nt; ++I) first[I] = I;
    for (size_t b = 0; b < bj.size(); ++b) {
        const int pa = pb->h_spos[bj[b] - mcon], pc = pb->h_spos[bk[b] - mcon];
        const int hi = std::max(pa, pc), lo = std::min(pa, pc);
        const int c0 = lo * cnp / POTRF_NB;
        for (int I = hi * cnp / POTRF_NB; I <= (hi * cnp + cnp - 1) / POTRF_NB; ++I) first[I] = std::min(first[I], c0);
    }
    std::vector<int> last((size_t)nt);
    for (int K = 0; K < nt; ++K) last[K] = K;
    for (int I = 0; I < nt; ++I) for (int K = first[I]; K <= I; ++K) last[K] = std::max(last[K], I);
    pb->potrf.env_rows.assign((size_t)nt, 0);
    long long tiles_env = 0, tiles_dense = 0;
    for (int K = 0; K < nt; ++K) { pb->potrf.env_rows[K] = last[K] - K; tiles_env += (long long)(last[K] - K) * (last[K] - K + 1) / 2; tiles_dense += (long long)(nt - K - 1) * (nt - K) / 2; }
    if (!pb->d_spos) HIP_OK(dmalloc(&pb->d_spos, (size_t)mm));
    if (!pb->d_xperm) HIP_OK(dmalloc(&pb->d_xperm, (size_t)pb->ld));
    if (!pb->potrf.d_last) HIP_OK(dmalloc(&pb->potrf.d_last, (size_t)nt));
    HIP_OK(hipMemcpy(pb->d_spos, pb->h_spos.data(), (size_t)mm * sizeof(int), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(pb->potrf.d_last, last.data(), (size_t)nt * sizeof(int), hipMemcpyHostToDevice));
    pb->envelope = true;
    if (pb->opt.verbose >= 2)
        printf("[bsfm] reduced camera system: envelope solver, %lld of %lld tile products (%.1f %%)\n", tiles_env, tiles_dense,
               tiles_dense ? 100.0 * (double)tiles_env / (double)tiles_dense : 0.0);
    return 0;
}

// Opt-in structure-aware reduced solves: groups of cameras that share no point with the rest (compsolve.hip.h), else / or the envelope
int setup_components(bsfm_problem* pb, const std::vector<int>& bj, const std::vector<int>& bk)
{
    if (pb->opt.potrf_backend != 0) return 0;
    if (pb->opt.reduced_solver == BSFM_SOLVER_AUTO) {
        if (comp_setup(pb->comps, pb->P.m - pb->P.mcon, pb->cnp, bj, bk, pb->P.mcon)) return BSFM_ERROR;
        if (pb->opt.verbose >= 2) {
            if (pb->comps.active) printf("[bsfm] reduced camera system: %d independent camera groups (largest %d unknowns), solved group by group\n",
                                         pb->comps.ncomp, pb->comps.maxdim);
            else printf("[bsfm] reduced camera system: connected (or a group too large for LDS): envelope solver\n");
        }
        if (pb->comps.active) return 0;
    }
    if (pb->opt.reduced_solver == BSFM_SOLVER_AUTO || pb->opt.reduced_solver == BSFM_SOLVER_ENVELOPE) {
        // multi-rank jobs come here with the job-wide UNION of the block lists (exchange_block_union): every rank derives the same
        // numbering and the same envelope, the exchange buffer itself stays in the natural numbering
        return setup_envelope(pb, bj, bk);
    }
    return 0;
}

// Union over ranks of the non-empty reduced-camera blocks.  The hook only sums, so the all-gather is a sum of
// disjoint segments: rank r writes key+1 of its blocks into segment r of a world x maxcount buffer (0 = empty slot).
int exchange_block_union(bsfm_problem* pb)
{
    const int m = pb->P.m;
    double cnt = (double)pb->nblk;
    if (allreduce_host(pb, &cnt, 1, 1)) return BSFM_ERROR;
    const size_t maxcnt = (size_t)cnt, total = maxcnt * (size_t)pb->world;
    std::vector<double> seg(std::max<size_t>(total, 1), 0.0);
    for (int b = 0; b < pb->nblk; ++b)
        seg[(size_t)pb->rank * maxcnt + b] = (double)((long long)pb->h_blk_j[b] * m + pb->h_blk_k[b] + 1);
    double* dseg = nullptr;
    HIP_OK(dmalloc(&dseg, total));
    // upload, exchange and download all ride the compute stream (one synchronisation at the end; the host needs the union to build
    // the packed layout, once per problem)
    HIP_OK(hipMemcpyAsync(dseg, seg.data(), total * sizeof(double), hipMemcpyHostToDevice, pb->stream));
    if (total && allreduce_dev(pb, dseg, total, 0)) { bsfm::dev_free(dseg); return BSFM_ERROR; }
    HIP_OK(hipMemcpyAsync(seg.data(), dseg, total * sizeof(double), hipMemcpyDeviceToHost, pb->stream));
    HIP_OK(hipStreamSynchronize(pb->stream));
    bsfm::dev_free(dseg);
    std::vector<long long> keys;
    keys.reserve(total);
    for (size_t q = 0; q < total; ++q) if (seg[q] > 0.5) keys.push_back((long long)(seg[q] + 0.5) - 1);
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    const int ng = (int)keys.size();
    std::vector<int> gj(ng), gk(ng), gidx(pb->nblk);
    for (int g = 0; g < ng; ++g) { gj[g] = (int)(keys[g] / m); gk[g] = (int)(keys[g] % m); }
    for (int b = 0; b < pb->nblk; ++b) {
        const long long key = (long long)pb->h_blk_j[b] * m + pb->h_blk_k[b];
        gidx[b] = (int)(std::lower_bound(keys.begin(), keys.end(), key) - keys.begin());
    }
    HIP_OK(dmalloc(&pb->d_gidx, (size_t)pb->nblk)); HIP_OK(dmalloc(&pb->d_gblk_j, (size_t)ng)); HIP_OK(dmalloc(&pb->d_gblk_k, (size_t)ng));
    HIP_OK(dmalloc(&pb->d_G, (size_t)ng * pb->cnp * pb->cnp + (size_t)pb->ld));     // + tail: this rank's part of E
    if (pb->nblk) HIP_OK(hipMemcpy(pb->d_gidx, gidx.data(), (size_t)pb->nblk * sizeof(int), hipMemcpyHostToDevice));
    if (ng) {
        HIP_OK(hipMemcpy(pb->d_gblk_j, gj.data(), (size_t)ng * sizeof(int), hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(pb->d_gblk_k, gk.data(), (size_t)ng * sizeof(int), hipMemcpyHostToDevice));
    }
    pb->ngblk = ng;
    if (setup_components(pb, gj, gk)) return BSFM_ERROR;          // same structure on every rank
    if (pb->opt.verbose >= 2)
        printf("[bsfm] rank %d: %d local / %d global reduced-camera blocks, %.1f MB per exchange (dense S: %.1f MB)\n", pb->rank,
               pb->nblk, ng, ng * pb->cnp * pb->cnp * 8e-6, (double)pb->ld * pb->ld * 8e-6);
    return 0;
}

// S, E for damping mu (Vinv must be current)
int compute_schur(bsfm_problem* pb, double mu)
{
    const int cnp = pb->cnp;
    DevProblem& P = pb->P;
    const int mm = P.m - P.mcon;
    const int lead = pb->rank == 0 ? 1 : 0;
    const bool packed = has_collective(pb);
    if (packed && pb->ngblk < 0 && exchange_block_union(pb)) return BSFM_ERROR;      // (first attempt of a multi-rank job: also decides on the envelope)
    // envelope solver: S, E are assembled in the reordered camera numbering (exports always use the natural one)
    const int* spos = (pb->envelope && !pb->export_full_s) ? pb->d_spos : nullptr;
    if (pb->export_full_s) (void)hipMemsetAsync(pb->d_S, 0, (size_t)pb->ld * pb->ld * sizeof(double), pb->stream);
    bool rhs_done = false, diag_done = false;
    double* Edst = packed ? pb->d_G + (size_t)pb->ngblk * cnp * cnp : pb->d_E;     // packed: E rides behind the blocks
    ZeroTilesArgs zt; memset(&zt, 0, sizeof zt);      // != 0 first_block: k_schur_prep clears the tiles / initialises E in workgroups appended to its grid
    const int nprep = grid_for(4 * (size_t)P.nvis, 256);
    if (!pb->export_full_s && !pb->comps.active && !packed && mm > 0 && pb->ntasks > 0) {
        const int nt = pb->ld / POTRF_NB, ntl = nt * (nt + 1) / 2;
        zt.S = pb->d_S; zt.ld = pb->ld; zt.ntl = ntl; zt.count = mm * cnp; zt.off = P.mcon * cnp; zt.add_ea = lead; zt.ea = pb->d_ea; zt.E = Edst; zt.spos = spos; zt.cnp = cnp;
        zt.first_block = nprep;
        rhs_done = true;
    } else
    if (!pb->export_full_s && !pb->comps.active) {   // (the group-by-group solve never writes S: blocks that are structurally empty stay zero)
        const int nt = pb->ld / POTRF_NB, ntl = nt * (nt + 1) / 2;
        if (!packed && mm > 0) {      // one launch: the lower tiles of S cleared, E = ea
            hipLaunchKernelGGL(k_zero_lower_tiles, dim3(ntl + grid_for((size_t)mm * cnp, 256)), dim3(256), 0, pb->stream, pb->d_S, pb->ld, ntl,
                               mm * cnp, P.mcon * cnp, lead, (const double*)pb->d_ea, Edst, spos, cnp);
            rhs_done = true;
        } else hipLaunchKernelGGL(k_zero_lower_tiles, dim3(ntl), dim3(256), 0, pb->stream, pb->d_S, pb->ld);
    }
    if (packed) (void)hipMemsetAsync(pb->d_G, 0, ((size_t)pb->ngblk * cnp * cnp + (size_t)pb->ld) * sizeof(double), pb->stream);
    if (mm > 0 && !rhs_done)
        hipLaunchKernelGGL(k_rhs_init, dim3(grid_for((size_t)mm * cnp, 256)), dim3(256), 0, pb->stream, mm * cnp, P.mcon * cnp,
                           lead, pb->d_ea, Edst, packed ? (const int*)nullptr : spos, cnp);     // (packed: E travels in the natural numbering)
    if (pb->ntasks > 0) {
        // C_ij = B_ij V*_i^-1 || C_ij eb_i for this attempt's mu, then the task kernel (schur.hip.h)
        ph_begin(pb, PH_SCHUR_PREP);
        const int nzero = zt.first_block > 0 ? zt.ntl + grid_for((size_t)zt.count, 256) : 0;
        if (pb->fuse_invert)
            hipLaunchKernelGGL(k_schur_prep, dim3(nprep + nzero), dim3(256), 0, pb->stream, P.nvis, pb->d_obs_pt, pb->d_campos, pb->d_Bc, pb->d_Vinv, pb->d_eb, pb->d_Cc,
                               (const double*)pb->d_V, mu, (const int*)pb->d_rowptr, pb->d_Vinv, pb->d_flags, zt);
        else
            hipLaunchKernelGGL(k_schur_prep, dim3(nprep + nzero), dim3(256), 0, pb->stream, P.nvis, pb->d_obs_pt, pb->d_campos, pb->d_Bc, pb->d_Vinv, pb->d_eb, pb->d_Cc,
                               (const double*)nullptr, 0.0, (const int*)nullptr, (double*)nullptr, (int*)nullptr, zt);
        ph_end(pb, PH_SCHUR_PREP);
        static const int wps = [] { const char* e = getenv("BSFM_SCHUR_WPS"); const int v = e ? atoi(e) : 3; return v < 2 ? 2 : (v > 4 ? 4 : v); }();
        ph_begin(pb, PH_SCHUR_TASKS);
        if (pb->d_tasks_launch) {     // one wave per task, both sides gathered
            const SchurTask* tl = pb->d_tasks_launch;
            const dim3 sg((pb->nslots + 3) / 4);
            // block sums on v_mfma_f64_4x4x4_4b (round 5: 0.847 against 0.875 ms at the headline size, three workgroups per CU);
            // BSFM_SCHUR_MFMA=16 = the 16 x 16 x 4 tiles of rounds 3-4 (A/B)
            static const bool m4 = [] { const char* e = getenv("BSFM_SCHUR_MFMA"); return !(e && atoi(e) == 16); }();
#define BSFM_LAUNCH_TASKS(W_, M_) DISPATCH_CNP(cnp, hipLaunchKernelGGL((k_schur_tasks<C, W_, M_>), sg, dim3(256), 0, pb->stream, P, tl, pb->nslots, pb->d_triples, pb->d_partials, pb->d_epart))
            if (m4) { if (wps == 2) { BSFM_LAUNCH_TASKS(2, true); } else if (wps == 4) { BSFM_LAUNCH_TASKS(4, true); } else { BSFM_LAUNCH_TASKS(3, true); } }
            else { if (wps == 2) { BSFM_LAUNCH_TASKS(2, false); } else if (wps == 4) { BSFM_LAUNCH_TASKS(4, false); } else { BSFM_LAUNCH_TASKS(3, false); } }
#undef BSFM_LAUNCH_TASKS
        }
        ph_end(pb, PH_SCHUR_TASKS);
        if (packed) {
            DISPATCH_CNP(cnp, hipLaunchKernelGGL((k_schur_pack<C>), dim3(pb->nblk), dim3(128), 0, pb->stream, pb->nblk,
                                                  pb->d_blk_j, pb->d_blk_k, pb->d_blk_range, pb->d_partials, pb->d_epart,
                                                  pb->d_gidx, pb->d_G, P.mcon, Edst));
        } else {
            // (+ mm blocks: the diagonal blocks of cameras without observations, k_schur_diag_fill's job)
            DISPATCH_CNP(cnp, hipLaunchKernelGGL((k_schur_assemble<C>), dim3(pb->nblk + std::max(mm, 0)), dim3(128), 0, pb->stream, pb->nblk,
                                                  pb->d_blk_j, pb->d_blk_k, pb->d_blk_range, pb->d_partials, pb->d_epart,
                                                  pb->d_U, mu, P.mcon, pb->d_S, pb->ld, Edst, spos, P.m, (const int*)pb->d_camptr));
            diag_done = true;
        }
    }
    if (packed) {
        // exchange step 2: the block sums of the union structure and E in ONE buffer; U was already summed over
        // ranks, so every rank then assembles the SAME S = [j==k](U_j + mu I) - G_jk and solves it redundantly
        if (allreduce_dev(pb, pb->d_G, (size_t)pb->ngblk * cnp * cnp + (size_t)pb->Sdim, 0)) return BSFM_ERROR;
        if (spos && mm > 0)      // envelope solver: into the reordered numbering on the way out of the exchange buffer
            hipLaunchKernelGGL(k_rhs_init, dim3(grid_for((size_t)mm * cnp, 256)), dim3(256), 0, pb->stream, mm * cnp, 0, 1, (const double*)Edst, pb->d_E, spos, cnp);
        else (void)hipMemcpyAsync(pb->d_E, Edst, (size_t)pb->Sdim * sizeof(double), hipMemcpyDeviceToDevice, pb->stream);
        if (mm > 0)
            DISPATCH_CNP(cnp, hipLaunchKernelGGL((k_schur_diag_fill<C>), dim3(mm), dim3(128), 0, pb->stream, P.m, P.mcon,
                                                  (const int*)nullptr, pb->d_U, mu, pb->d_S, pb->ld, spos));
        if (pb->ngblk > 0)
            DISPATCH_CNP(cnp, hipLaunchKernelGGL((k_schur_unpack<C>), dim3(pb->ngblk), dim3(128), 0, pb->stream, pb->ngblk,
                                                  pb->d_gblk_j, pb->d_gblk_k, pb->d_G, pb->d_U, mu, P.mcon, pb->d_S, pb->ld, spos));
    } else if (mm > 0 && !diag_done) {
        DISPATCH_CNP(cnp, hipLaunchKernelGGL((k_schur_diag_fill<C>), dim3(mm), dim3(128), 0, pb->stream, P.m, P.mcon,
                                              pb->d_camptr, pb->d_U, mu, pb->d_S, pb->ld, spos));
    }
    return 0;
}

}  // namespace

// ================================================================================================ API
extern "C" {

void bsfm_default_options(bsfm_options_t* opt)
{
    memset(opt, 0, sizeof(*opt));
    opt->jacobian = BSFM_JAC_FD;
    if (const char* e = getenv("BSFM_JACOBIAN")) {
        if (!strcmp(e, "analytic")) opt->jacobian = BSFM_JAC_ANALYTIC;
        else if (!strcmp(e, "fd")) opt->jacobian = BSFM_JAC_FD;
    }
    opt->itmax = 150;   // MAX_ITERS, sfm.c:814
    opt->verbose = 1;
    opt->opts[0] = 1.0e-3; opt->opts[1] = 1.0e-10; opt->opts[2] = 0.0;
    opt->opts[3] = 1.0e-12; opt->opts[4] = 0.0; opt->opts[5] = 4.0e-2;   // sfm.c:705-714
    opt->potrf_backend = 0;
    if (const char* e = getenv("BSFM_POTRF")) if (!strcmp(e, "rocsolver")) opt->potrf_backend = 1;
    opt->reduced_solver = BSFM_SOLVER_DENSE;
    if (const char* e = getenv("BSFM_REDUCED_SOLVER")) {
        if (!strcmp(e, "auto")) opt->reduced_solver = BSFM_SOLVER_AUTO;
        else if (!strcmp(e, "envelope")) opt->reduced_solver = BSFM_SOLVER_ENVELOPE;
    }
    opt->num_gpus = 0;
    if (const char* e = getenv("BSFM_NUM_GPUS")) opt->num_gpus = atoi(e);
}

int bsfm_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* bsfm_version(void) { return "bundler_sfm_amd 0.1 (gfx950)"; }

int bsfm_device_synchronize(void) { return hipDeviceSynchronize() == hipSuccess ? 0 : BSFM_ERROR; }

bsfm_problem_t* bsfm_problem_create(const bsfm_problem_desc_t* d, const bsfm_options_t* opt_in)
{
    if (bsfm_device_count() <= 0) {
        fprintf(stderr, "[bsfm] FATAL: no usable HIP device; the MI355X path has no CPU fallback\n");
        return nullptr;
    }
    if (!d || d->n < 0 || d->m <= 0 || d->mcon < 0 || d->mcon > d->m) { fprintf(stderr, "[bsfm] bad problem description\n"); return nullptr; }
    if (!d->rowptr || !d->cameras || (d->n > 0 && (!d->points && !d->p_packed))) { fprintf(stderr, "[bsfm] bad problem description: rowptr / cameras / points missing\n"); return nullptr; }
    int rp_ends[2] = { 0, 0 };              // rowptr[0], rowptr[n] (the arrays may already live on the device)
    const bool index_dev = d->arrays_on_device != 0;                        // rowptr / colidx are device pointers
    const bool data_dev = d->arrays_on_device == BSFM_ARRAYS_ON_DEVICE;     // ... projections and points as well
    if (index_dev) {
        if (hipMemcpy(&rp_ends[0], d->rowptr, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(&rp_ends[1], d->rowptr + d->n, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "[bsfm] bad problem description: device rowptr unreadable\n"); return nullptr; }
        if (data_dev && d->use_point_constraints && d->point_constraints) { fprintf(stderr, "[bsfm] bad problem description: point constraints cannot be combined with arrays_on_device\n"); return nullptr; }
    } else { rp_ends[0] = d->rowptr[0]; rp_ends[1] = d->rowptr[d->n]; }
    if (rp_ends[0] != 0 || rp_ends[1] < 0) { fprintf(stderr, "[bsfm] bad problem description: rowptr[0] must be 0 and rowptr[n] >= 0\n"); return nullptr; }
    if (rp_ends[1] > 0 && (!d->colidx || !d->projections)) { fprintf(stderr, "[bsfm] bad problem description: colidx / projections missing\n"); return nullptr; }
    {   // int32 indexing everywhere ("all indexing is int32"): refuse sizes that would overflow it
        const long long nv = 9LL * d->m + 3LL * d->n, no = 2LL * rp_ends[1];
        if (nv > 0x7fffffffLL || no > 0x7fffffffLL) { fprintf(stderr, "[bsfm] problem too large for 32-bit indexing (%lld unknowns, %lld measurements)\n", nv, no); return nullptr; }
    }
    const auto t_create0 = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    bsfm_problem* pb = new bsfm_problem();
    if (opt_in) pb->opt = *opt_in; else bsfm_default_options(&pb->opt);
    auto fail = [&](const char* why) -> bsfm_problem_t* {
        fprintf(stderr, "[bsfm] problem_create failed: %s\n", why);
        free_all(pb); delete pb; return nullptr;
    };
    const int n = d->n, m = d->m;
    const int nvis = rp_ends[1];
    const int cnp = (d->est_focal_length ? 7 : 6) + (d->undistort ? 2 : 0);
    pb->cnp = cnp;
    DevProblem& P = pb->P;
    P.cfg.cnp = cnp; P.cfg.est_focal = d->est_focal_length ? 1 : 0; P.cfg.undistort = d->undistort ? 1 : 0;
    P.cfg.explicit_centers = d->explicit_camera_centers ? 1 : 0;
    P.cfg.f_scale = 0.001; P.cfg.k_scale = 5.0;                    // sfm.c:634-635
    P.n = n; P.m = m; P.mcon = d->mcon; P.nvis = nvis;
    pb->world = d->world_size > 1 ? d->world_size : 1; pb->rank = d->world_size > 1 ? d->rank : 0;
    pb->nvis_global = d->nvis_global > 0 ? d->nvis_global : nvis;
    pb->mot = d->fix_points ? 1 : 0;
    pb->nvars_global = d->nvars_global > 0 ? d->nvars_global : (pb->mot ? (long long)m * cnp : (long long)m * cnp + 3LL * n);
    pb->nvars_local = m * cnp + 3 * n;
    P.nvis_global = (double)pb->nvis_global;
    pb->Sdim = (m - d->mcon) * cnp;
    pb->ld = std::max(POTRF_NB, (pb->Sdim + POTRF_NB - 1) / POTRF_NB * POTRF_NB);

    if (!(pb->stream = stream_pool().acquire())) return fail("stream");
    pb->own_stream = true;
#define DM(ptr, cnt) if (dmalloc(&ptr, (size_t)(cnt)) != hipSuccess) return fail("hipMalloc " #ptr)
    auto up = [&](void* dst, const void* src, size_t bytes) { return bytes == 0 || hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess; };
    auto up_any = [&](void* dst, const void* src, size_t bytes) { return bytes == 0 || hipMemcpy(dst, src, bytes, data_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice) == hipSuccess; };
    auto up_idx = [&](void* dst, const void* src, size_t bytes) { return bytes == 0 || hipMemcpy(dst, src, bytes, index_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice) == hipSuccess; };
    // ---- index bookkeeping (bit-exact integer work, built on the device from the caller's CRS; index_build.hip)
    const auto t_up0 = std::chrono::steady_clock::now();
    DM(pb->d_rowptr, n + 1); DM(pb->d_obs_cam, nvis); DM(pb->d_x, 2 * (size_t)nvis);
    if (!(up_idx(pb->d_rowptr, d->rowptr, ((size_t)n + 1) * sizeof(int)) && up_idx(pb->d_obs_cam, d->colidx, (size_t)nvis * sizeof(int)) &&
          up_any(pb->d_x, d->projections, 2 * (size_t)nvis * sizeof(double)))) return fail("upload of the visibility index");
    pb->create_ms[1] = ms_since(t_up0);
    {
        const auto t_ix0 = std::chrono::steady_clock::now();
        const char* eo = getenv("BSFM_SCHUR_ORDER");
        DeviceIndex ix;
        if (build_index_device(n, m, d->mcon, nvis, pb->d_rowptr, pb->d_obs_cam, !pb->mot, (eo && !strcmp(eo, "block")) ? SCHUR_ORDER_BLOCK : (eo && !strcmp(eo, "point")) ? SCHUR_ORDER_POINT : SCHUR_ORDER_CLUSTERED, ix, pb->stream) != 0) {
            free_index_device(ix);
            return fail("index construction");
        }
        if (adopt_index(pb, ix) != 0) return fail("schur structure");
        pb->create_ms[2] = ms_since(t_ix0);
    }
    const auto t_al0 = std::chrono::steady_clock::now();
    DM(pb->d_Rinit, 9 * (size_t)m); DM(pb->d_finit, m);
    DM(pb->d_p, pb->nvars_local); DM(pb->d_pdp, pb->nvars_local); DM(pb->d_dp, pb->nvars_local);
    DM(pb->d_camtab, (size_t)m * CT_STRIDE); DM(pb->d_camtab_trial, (size_t)m * CT_STRIDE);
    DM(pb->d_e, 2 * (size_t)nvis); DM(pb->d_hx, 2 * (size_t)nvis);
    DM(pb->d_ptc[0], 4 * (size_t)nvis); DM(pb->d_ptc[1], 4 * (size_t)nvis);
    DM(pb->d_Ac, (size_t)nvis * 2 * cnp); DM(pb->d_Bc, (size_t)nvis * 8); if (!pb->mot) { DM(pb->d_Cc, (size_t)nvis * 8); } DM(pb->d_xc, 2 * (size_t)nvis); DM(pb->d_campart, (size_t)m * CAM_SPLIT * (cnp * (cnp + 1) / 2 + cnp)); DM(pb->d_U, (size_t)m * cnp * cnp + (size_t)m * cnp); pb->d_ea = pb->d_U + (size_t)m * cnp * cnp;   /* one buffer: one exchange */
    DM(pb->d_V, 6 * (size_t)n); DM(pb->d_Vinv, 6 * (size_t)n); DM(pb->d_eb, 3 * (size_t)n);
    DM(pb->d_S, (size_t)pb->ld * pb->ld); DM(pb->d_E, pb->ld);
    pb->red_blocks = std::max(grid_for(nvis, RES_BLOCK), std::max(grid_for((size_t)BS_LANES * (size_t)n, 256), 1024));
    DM(pb->d_red, 4 * (size_t)pb->red_blocks); DM(pb->d_scal, SC_COUNT + 16 + 2); DM(pb->d_mixed, 8 + 4 * (size_t)std::max(1, d->world_size));
    pb->d_flags = reinterpret_cast<int*>(pb->d_scal + SC_COUNT + 16);      // 4 ints behind the scalars: both travel in one copy
    {   // [0 .. 8) top-level words, [8 .. 8 + m) the cameras', then the group words of k_residual's and k_backsub's grids (kernels.hip.h)
        pb->tick_res = (8 + (size_t)m + 31) / 32 * 32;
        pb->tick_back = pb->tick_res + ticket_group_words((size_t)grid_for(nvis, RES_BLOCK));
        const size_t words = pb->tick_back + ticket_group_words((size_t)grid_for((size_t)BS_LANES * (size_t)n, 256));      // (k_backsub's one-pass form: four lanes per point)
        DM(pb->d_tickets, words);
        pb->tick_words = words;
        if (hipMemsetAsync(pb->d_tickets, 0, words * sizeof(unsigned), pb->stream) != hipSuccess) return fail("tickets");
    }
#undef DM
    if (nvis > 0) hipLaunchKernelGGL(k_permute16, dim3(grid_for(nvis, 256)), dim3(256), 0, pb->stream, nvis, pb->d_camobs, pb->d_x, pb->d_xc);   // measurements in camera-major order
    if (hipHostMalloc((void**)&pb->h_scal, (SC_COUNT + 16 + 2) * sizeof(double)) != hipSuccess) return fail("pinned");
    pb->h_flags = reinterpret_cast<int*>(pb->h_scal + SC_COUNT + 16);
    (void)hipMemsetAsync(pb->d_scal, 0, (SC_COUNT + 16 + 2) * sizeof(double), pb->stream);
    (void)hipMemsetAsync(pb->d_S, 0, (size_t)pb->ld * pb->ld * sizeof(double), pb->stream); (void)hipMemsetAsync(pb->d_E, 0, pb->ld * sizeof(double), pb->stream);
    (void)hipMemsetAsync(pb->d_dp, 0, pb->nvars_local * sizeof(double), pb->stream);
    pb->create_ms[3] = ms_since(t_al0);
    bool ok = true;
    pb->h_Rinit.resize(9 * (size_t)m);
    std::vector<double> finit(m);
    for (int j = 0; j < m; ++j) { memcpy(&pb->h_Rinit[9 * (size_t)j], d->cameras[j].R, 9 * sizeof(double)); finit[j] = d->cameras[j].f; }
    ok = ok && up(pb->d_Rinit, pb->h_Rinit.data(), 9 * (size_t)m * sizeof(double)) && up(pb->d_finit, finit.data(), m * sizeof(double));
    {   // cameras with known intrinsics (sfm.c:339-358) and the fisheye projection (sfm.c:426-492): their block of the
        // camera table; the closed-form Jacobian covers neither, so such problems use the reference's own forward differences
        pb->fisheye_mode = d->optimize_for_fisheye ? 1 : 0;
        std::vector<double> kn;
        if (fill_ext_block(d->cameras, m, pb->fisheye_mode, kn)) {
            if (dmalloc(&pb->d_known, kn.size()) != hipSuccess) return fail("hipMalloc extended camera model");
            ok = ok && up(pb->d_known, kn.data(), kn.size() * sizeof(double));
            if (pb->opt.jacobian != BSFM_JAC_FD) {
                if (pb->opt.verbose >= 1) printf("[bsfm] known intrinsics / fisheye projection: using the forward-difference Jacobian\n");
                pb->opt.jacobian = BSFM_JAC_FD;
            }
        }
    }
    std::vector<double> p;
    if (d->p_packed) p.assign(d->p_packed, d->p_packed + (size_t)m * cnp + (size_t)3 * n);
    else pack_params(pb, d->cameras, data_dev ? nullptr : d->points, n, p);
    ok = ok && up(pb->d_p, p.data(), p.size() * sizeof(double));
    if (data_dev && !d->p_packed) ok = ok && up_any(pb->d_p + (size_t)m * cnp, d->points, 3 * (size_t)n * sizeof(double));
    pb->h_cams.assign(d->cameras, d->cameras + m);
    pb->desc0 = *d;
    pb->desc0.rowptr = nullptr; pb->desc0.colidx = nullptr; pb->desc0.projections = nullptr; pb->desc0.cameras = nullptr; pb->desc0.points = nullptr;
    pb->desc0.point_constraints = nullptr; pb->desc0.p_packed = nullptr; pb->desc0.arrays_on_device = 0;
    if (d->use_constraints) {   // sfm.c:721-754 (note the hard-wired indices 6,7,8 of the rescaling)
        std::vector<unsigned char> con((size_t)m * cnp); std::vector<double> val((size_t)m * cnp), w((size_t)m * cnp);
        for (int j = 0; j < m; ++j) {
            double cv[9], cw[9];
            for (int q = 0; q < 9; ++q) { cv[q] = d->cameras[j].constraints[q]; cw[q] = d->cameras[j].weights[q]; }
            if (d->est_focal_length && !d->constraints_prescaled) { cv[6] *= P.cfg.f_scale; cw[6] *= 1.0 / (P.cfg.f_scale * P.cfg.f_scale); }
            if (d->undistort && !d->constraints_prescaled) { cv[7] *= P.cfg.k_scale; cw[7] *= 1.0 / (P.cfg.k_scale * P.cfg.k_scale);
                                cv[8] *= P.cfg.k_scale; cw[8] *= 1.0 / (P.cfg.k_scale * P.cfg.k_scale); }
            for (int q = 0; q < cnp; ++q) { con[(size_t)j * cnp + q] = d->cameras[j].constrained[q] ? 1 : 0; val[(size_t)j * cnp + q] = cv[q]; w[(size_t)j * cnp + q] = cw[q]; }
        }
        if (dmalloc(&pb->d_ccon, con.size()) != hipSuccess || dmalloc(&pb->d_cval, val.size()) != hipSuccess || dmalloc(&pb->d_cw, w.size()) != hipSuccess) return fail("hipMalloc constraints");
        ok = ok && up(pb->d_ccon, con.data(), con.size()) && up(pb->d_cval, val.data(), val.size() * sizeof(double)) && up(pb->d_cw, w.data(), w.size() * sizeof(double));
    }
    if (d->use_point_constraints && d->point_constraints && !d->fix_points) {   // sfm.c:757-781 (sba_mot_levmar takes none)
        std::vector<unsigned char> con(std::max(n, 1));
        for (int i = 0; i < n; ++i) { const double* c = d->point_constraints + 3 * (size_t)i; con[i] = !(c[0] == 0.0 && c[1] == 0.0 && c[2] == 0.0); }
        if (dmalloc(&pb->d_pcon, (size_t)n) != hipSuccess || dmalloc(&pb->d_pval, 3 * (size_t)n) != hipSuccess) return fail("hipMalloc point constraints");
        ok = ok && up(pb->d_pcon, con.data(), n) && up(pb->d_pval, d->point_constraints, 3 * (size_t)n * sizeof(double));
        P.pweight = d->point_constraint_weight;
    }
    if (!ok) return fail("upload");
    P.x = pb->d_x; P.xc = pb->d_xc; P.obs_cam = pb->d_obs_cam; P.obs_pt = pb->d_obs_pt; P.rowptr = pb->d_rowptr;
    P.camptr = pb->d_camptr; P.camobs = pb->d_camobs; P.campos = pb->d_campos; P.cam_pt = pb->d_cam_pt; P.cam_cam = pb->d_cam_cam; P.Rinit = pb->d_Rinit; P.finit = pb->d_finit;
    P.ccon = pb->d_ccon; P.cval = pb->d_cval; P.cw = pb->d_cw; P.pcon = pb->d_pcon; P.pval = pb->d_pval;
    P.Ac = pb->d_Ac; P.Bc = pb->d_Bc; P.Cc = pb->d_Cc; P.U = pb->d_U; P.ea = pb->d_ea; P.V = pb->d_V; P.Vinv = pb->d_Vinv; P.eb = pb->d_eb;
    if (potrf_init(pb->potrf, pb->ld, pb->opt.potrf_backend) != 0) return fail("potrf workspace");
    // Phase timing (16 event records per solve attempt + their read-back) is on for problems where it is noise (>= 100 000
    // observations: bench.py's phases_ms) and off for the small problems of incremental reconstruction, where those host calls
    // were a tenth of an iteration; BSFM_PHASE_TIMING=1 / 0 forces it.
    pb->backsub_two_pass = nvis >= 400000;      // (round 6: the one-pass form with four lanes per point wins up to ~400 000 observations: 24 against 29 us at 200 000, 48 / 44 at 500 000)
    if (const char* e = getenv("BSFM_BACKSUB_TWO_PASS")) pb->backsub_two_pass = atoi(e) != 0;
    pb->ev_ok = nvis >= 2000000;      // (round 6: was 100 000 -- exactly the 50-camera problem of the latency table, whose iteration the event records made a third longer)
    if (const char* e = getenv("BSFM_PHASE_TIMING")) pb->ev_ok = atoi(e) != 0;
    // measured (profiles/r03_small_problem_latency_speculate.txt, profiles/r06_small_problem_latency.txt): 10 % of an iteration at 14 cameras
    pb->speculate = true;      // (round 6: every size -- 3 - 5 % at 50 / 100 cameras once the event records were gone, 1 % at 400; one host round trip per iteration at 1 000; was <= 30 000 observations)
    pb->fuse_invert = !pb->empty_rows;
    if (const char* e = getenv("BSFM_FUSE_INVERT")) pb->fuse_invert = atoi(e) != 0 && !pb->empty_rows;
    if (const char* e = getenv("BSFM_SPECULATE")) pb->speculate = atoi(e) != 0;
    pb->potrf.timing = pb->ev_ok ? 1 : 0;
    for (int i = 0; pb->ev_ok && i < PH_COUNT; ++i) {
        pb->ev_created = true;
        if (hipEventCreate(&pb->ev[i][0]) != hipSuccess || hipEventCreate(&pb->ev[i][1]) != hipSuccess) pb->ev_ok = false;
    }
    for (int i = 0; i < PH_COUNT; ++i) { pb->ph_ms[i] = 0.0; pb->ph_cnt[i] = 0; }
    (void)hipDeviceSynchronize();
    pb->create_ms[0] = ms_since(t_create0);
    return pb;
}

void bsfm_problem_destroy(bsfm_problem_t* pb)
{
    if (!pb) return;
    (void)hipDeviceSynchronize();
    free_all(pb);
    delete pb;
}

void bsfm_problem_set_allreduce(bsfm_problem_t* pb, bsfm_allreduce_fn fn, void* ctx) { pb->allreduce = fn; pb->allreduce_ctx = ctx; }
void bsfm_problem_set_comm(bsfm_problem_t* pb, bsfm_comm_t* comm)
{
    pb->comm = comm;
    // BSFM_DIST_CHOL=1: the ranks factor the (replicated) reduced camera system TOGETHER instead of each factoring all of it (chol_flow.hip.h: FlowDist).
    // Opt-in: it needs a transport that can map the peers' buffers ("ipc", "loopback"), and it has only run between processes sharing one device.
    pb->potrf.dist_comm = nullptr;
    if (const char* e = getenv("BSFM_DIST_CHOL"))
        if (atoi(e) != 0 && comm && bsfm_comm_world(comm) > 1 && strcmp(bsfm_comm_transport(comm), "rccl") != 0) pb->potrf.dist_comm = comm;
}

void bsfm_problem_set_stream(bsfm_problem_t* pb, void* s)
{
    (void)hipStreamSynchronize(pb->stream);
    if (pb->own_stream && pb->stream) stream_pool().release(pb->stream);
    if (s) { pb->stream = (hipStream_t)s; pb->own_stream = false; }
    else { pb->stream = stream_pool().acquire(); pb->own_stream = true; }
}

int bsfm_problem_append(bsfm_problem_t* pb, int num_new_cameras, const bsfm_camera_params_t* new_cameras,
                        int num_new_points, const double* new_points,
                        int nadd, const int* add_pt, const int* add_cam, const double* add_xy)
{
    if (!pb || num_new_cameras < 0 || num_new_points < 0 || nadd < 0 || (num_new_cameras && !new_cameras) || (num_new_points && !new_points) ||
        (nadd && (!add_pt || !add_cam || !add_xy))) { fprintf(stderr, "[bsfm] bsfm_problem_append: bad arguments\n"); return BSFM_ERROR; }
    if (pb->world > 1 || pb->mot) { fprintf(stderr, "[bsfm] bsfm_problem_append: single-rank motion + structure problems only\n"); return BSFM_ERROR; }
    const int m0 = pb->P.m, n0 = pb->P.n, nvis0 = pb->P.nvis, cnp = pb->cnp;
    const int m1 = m0 + num_new_cameras, n1 = n0 + num_new_points;
    HIP_OK(hipStreamSynchronize(pb->stream));
    // old cameras as run_sfm would hand them back (rotation increment folded into R on the host: 72 KB at 1 000 cameras)
    std::vector<bsfm_camera_params_t> cams((size_t)m1);
    memcpy(cams.data(), pb->h_cams.data(), (size_t)m0 * sizeof(bsfm_camera_params_t));
    if (bsfm_problem_download(pb, nullptr, cams.data(), nullptr) != 0) return BSFM_ERROR;
    for (int j = 0; j < num_new_cameras; ++j) cams[(size_t)m0 + j] = new_cameras[j];
    // new observations / points to the device; merge with the resident ones there
    int *d_apt = nullptr, *d_acam = nullptr, *d_rp = nullptr, *d_ci = nullptr; double *d_axy = nullptr, *d_x = nullptr, *d_pts = nullptr;
    auto cleanup = [&] { for (void* q : { (void*)d_apt, (void*)d_acam, (void*)d_axy, (void*)d_rp, (void*)d_ci, (void*)d_x, (void*)d_pts }) bsfm::dev_free(q); };
    bool ok = dmalloc(&d_apt, (size_t)nadd) == hipSuccess && dmalloc(&d_acam, (size_t)nadd) == hipSuccess && dmalloc(&d_axy, 2 * (size_t)nadd) == hipSuccess &&
              dmalloc(&d_pts, 3 * (size_t)n1) == hipSuccess;
    if (ok && nadd) ok = hipMemcpy(d_apt, add_pt, (size_t)nadd * sizeof(int), hipMemcpyHostToDevice) == hipSuccess &&
                         hipMemcpy(d_acam, add_cam, (size_t)nadd * sizeof(int), hipMemcpyHostToDevice) == hipSuccess &&
                         hipMemcpy(d_axy, add_xy, 2 * (size_t)nadd * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
    if (ok && n0) ok = hipMemcpy(d_pts, pb->d_p + (size_t)m0 * cnp, 3 * (size_t)n0 * sizeof(double), hipMemcpyDeviceToDevice) == hipSuccess;
    if (ok && num_new_points) ok = hipMemcpy(d_pts + 3 * (size_t)n0, new_points, 3 * (size_t)num_new_points * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { cleanup(); fprintf(stderr, "[bsfm] bsfm_problem_append: allocation / upload failed\n"); return BSFM_ERROR; }
    if (merge_observations_device(n1, m1, nvis0, pb->d_obs_pt, pb->d_obs_cam, pb->d_x, nadd, d_apt, d_acam, d_axy, &d_rp, &d_ci, &d_x, pb->stream) != 0) { cleanup(); return BSFM_ERROR; }
    bsfm_problem_desc_t d = pb->desc0;
    d.n = n1; d.m = m1;
    d.rowptr = d_rp; d.colidx = d_ci; d.projections = d_x; d.points = d_pts; d.cameras = cams.data();
    d.use_point_constraints = 0; d.point_constraints = nullptr; d.arrays_on_device = 1;
    d.nvis_global = 0; d.nvars_global = 0;
    bsfm_problem* grown = bsfm_problem_create(&d, &pb->opt);
    if (!grown) { cleanup(); return BSFM_ERROR; }
    if (pb->d_pcon) {   // point constraints of the old points carry over; new points are unconstrained
        bool okc = dmalloc(&grown->d_pcon, (size_t)n1) == hipSuccess && dmalloc(&grown->d_pval, 3 * (size_t)n1) == hipSuccess &&
                   hipMemset(grown->d_pcon, 0, (size_t)n1) == hipSuccess && hipMemset(grown->d_pval, 0, 3 * (size_t)n1 * sizeof(double)) == hipSuccess &&
                   hipMemcpy(grown->d_pcon, pb->d_pcon, (size_t)n0, hipMemcpyDeviceToDevice) == hipSuccess &&
                   hipMemcpy(grown->d_pval, pb->d_pval, 3 * (size_t)n0 * sizeof(double), hipMemcpyDeviceToDevice) == hipSuccess;
        if (!okc) { bsfm_problem_destroy(grown); cleanup(); return BSFM_ERROR; }
        grown->P.pcon = grown->d_pcon; grown->P.pval = grown->d_pval; grown->P.pweight = pb->P.pweight;
        grown->desc0.use_point_constraints = 1; grown->desc0.point_constraint_weight = pb->P.pweight;
    }
    grown->allreduce = pb->allreduce; grown->allreduce_ctx = pb->allreduce_ctx; grown->comm = pb->comm;
    std::swap(*pb, *grown);                 // the caller's handle now holds the grown problem ...
    bsfm_problem_destroy(grown);            // ... and the old arrays go
    cleanup();
    return 0;
}

int bsfm_problem_remove_points(bsfm_problem_t* pb, const unsigned char* remove, int* remap_out)
{
    if (!pb || !remove) { fprintf(stderr, "[bsfm] bsfm_problem_remove_points: bad arguments\n"); return BSFM_ERROR; }
    if (pb->world > 1 || pb->mot) { fprintf(stderr, "[bsfm] bsfm_problem_remove_points: single-rank motion + structure problems only\n"); return BSFM_ERROR; }
    const int m = pb->P.m, n0 = pb->P.n, nvis0 = pb->P.nvis, cnp = pb->cnp;
    HIP_OK(hipStreamSynchronize(pb->stream));
    int nrem = 0;
    for (int i = 0; i < n0; ++i) nrem += remove[i] != 0;
    if (nrem == 0) { if (remap_out) for (int i = 0; i < n0; ++i) remap_out[i] = i; return 0; }
    // cameras as run_sfm would hand them back (rotation increment folded into R on the host), as in bsfm_problem_append
    std::vector<bsfm_camera_params_t> cams(pb->h_cams);
    if (bsfm_problem_download(pb, nullptr, cams.data(), nullptr) != 0) return BSFM_ERROR;
    unsigned char* d_rm = nullptr; int *d_rp = nullptr, *d_ci = nullptr, *d_remap = nullptr; double *d_x = nullptr, *d_pts = nullptr;
    auto cleanup = [&] { for (void* q : { (void*)d_rm, (void*)d_rp, (void*)d_ci, (void*)d_remap, (void*)d_x, (void*)d_pts }) bsfm::dev_free(q); };
    if (dmalloc(&d_rm, (size_t)n0) != hipSuccess || hipMemcpy(d_rm, remove, (size_t)n0, hipMemcpyHostToDevice) != hipSuccess) { cleanup(); return BSFM_ERROR; }
    int n1 = 0, nvis1 = 0;
    if (compact_points_device(n0, nvis0, pb->d_rowptr, pb->d_obs_pt, pb->d_obs_cam, pb->d_x, d_rm, &d_rp, &d_ci, &d_x, &d_remap, &n1, &nvis1, pb->stream) != 0) { cleanup(); return BSFM_ERROR; }
    bool ok = dmalloc(&d_pts, 3 * (size_t)std::max(n1, 1)) == hipSuccess &&
              gather_kept_device(n0, d_remap, 3 * (int)sizeof(double), pb->d_p + (size_t)m * cnp, d_pts, pb->stream) == 0 &&
              hipStreamSynchronize(pb->stream) == hipSuccess;
    if (ok && remap_out) ok = hipMemcpy(remap_out, d_remap, (size_t)n0 * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
    if (!ok) { cleanup(); fprintf(stderr, "[bsfm] bsfm_problem_remove_points: allocation / copy failed\n"); return BSFM_ERROR; }
    bsfm_problem_desc_t d = pb->desc0;
    d.n = n1; d.m = m;
    d.rowptr = d_rp; d.colidx = d_ci; d.projections = d_x; d.points = d_pts; d.cameras = cams.data();
    d.use_point_constraints = 0; d.point_constraints = nullptr; d.arrays_on_device = 1;
    d.nvis_global = 0; d.nvars_global = 0;
    bsfm_problem* small = bsfm_problem_create(&d, &pb->opt);
    if (!small) { cleanup(); return BSFM_ERROR; }
    if (pb->d_pcon) {   // point constraints follow their points
        bool okc = dmalloc(&small->d_pcon, (size_t)std::max(n1, 1)) == hipSuccess && dmalloc(&small->d_pval, 3 * (size_t)std::max(n1, 1)) == hipSuccess &&
                   gather_kept_device(n0, d_remap, 1, pb->d_pcon, small->d_pcon, pb->stream) == 0 &&
                   gather_kept_device(n0, d_remap, 3 * (int)sizeof(double), pb->d_pval, small->d_pval, pb->stream) == 0 &&
                   hipStreamSynchronize(pb->stream) == hipSuccess;
        if (!okc) { bsfm_problem_destroy(small); cleanup(); return BSFM_ERROR; }
        small->P.pcon = small->d_pcon; small->P.pval = small->d_pval; small->P.pweight = pb->P.pweight;
        small->desc0.use_point_constraints = 1; small->desc0.point_constraint_weight = pb->P.pweight;
    }
    small->allreduce = pb->allreduce; small->allreduce_ctx = pb->allreduce_ctx; small->comm = pb->comm;
    std::swap(*pb, *small);
    bsfm_problem_destroy(small);
    cleanup();
    return nrem;
}

int bsfm_problem_reset_params(bsfm_problem_t* pb, const bsfm_camera_params_t* cams, const double* pts)
{
    std::vector<double> p;
    pack_params(pb, cams, pts, pb->P.n, p);
    std::vector<double> finit(pb->P.m);
    for (int j = 0; j < pb->P.m; ++j) { memcpy(&pb->h_Rinit[9 * (size_t)j], cams[j].R, 9 * sizeof(double)); finit[j] = cams[j].f; }
    HIP_OK(hipMemcpy(pb->d_p, p.data(), p.size() * sizeof(double), hipMemcpyHostToDevice));
    invalidate_point_mirrors(pb);
    HIP_OK(hipMemcpy(pb->d_Rinit, pb->h_Rinit.data(), pb->h_Rinit.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(pb->d_finit, finit.data(), finit.size() * sizeof(double), hipMemcpyHostToDevice));
    pb->h_cams.assign(cams, cams + pb->P.m);
    {   // extended-model block of the new cameras
        std::vector<double> kn;
        const bool any = fill_ext_block(cams, pb->P.m, pb->fisheye_mode, kn);
        if (any || pb->d_known) {
            if (!pb->d_known) HIP_OK(dmalloc(&pb->d_known, kn.size()));
            HIP_OK(hipMemcpy(pb->d_known, kn.data(), kn.size() * sizeof(double), hipMemcpyHostToDevice));
            if (any) pb->opt.jacobian = BSFM_JAC_FD;
        }
    }
    pb->began = 0;
    return 0;
}

// ---- test entries for the index bookkeeping: the arrays the kernels actually index with, straight from HBM
int bsfm_problem_export_index(bsfm_problem_t* pb, int* rowptr, int* colidx, int* obs_pt, int* camptr, int* camobs, int* campos,
                              int* cam_pt, int* cam_cam)
{
    const size_t nv = (size_t)pb->P.nvis;
    HIP_OK(hipStreamSynchronize(pb->stream));
    auto down = [](int* dst, const int* src, size_t cnt) { return !dst || cnt == 0 || hipMemcpy(dst, src, cnt * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess; };
    const bool ok = down(rowptr, pb->d_rowptr, (size_t)pb->P.n + 1) && down(colidx, pb->d_obs_cam, nv) && down(obs_pt, pb->d_obs_pt, nv) &&
                    down(camptr, pb->d_camptr, (size_t)pb->P.m + 1) && down(camobs, pb->d_camobs, nv) && down(campos, pb->d_campos, nv) &&
                    down(cam_pt, pb->d_cam_pt, nv) && down(cam_cam, pb->d_cam_cam, nv);
    return ok ? 0 : BSFM_ERROR;
}

int bsfm_problem_schur_sizes(const bsfm_problem_t* pb, int* ntriples, int* nblk, int* ntasks, int* nslots)
{
    if (ntriples) *ntriples = pb->ntriples;
    if (nblk) *nblk = pb->nblk;
    if (ntasks) *ntasks = pb->ntasks;
    if (nslots) *nslots = pb->nslots;
    return 0;
}

int bsfm_problem_export_schur(bsfm_problem_t* pb, int* triples, int* tri_pt, int* blk_j, int* blk_k, int* blk_task0, int* tasks)
{
    if (pb->mot) return BSFM_ERROR;
    HIP_OK(hipStreamSynchronize(pb->stream));
    auto down = [](void* dst, const void* src, size_t bytes) { return !dst || bytes == 0 || hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) == hipSuccess; };
    const bool ok = down(triples, pb->d_triples, (size_t)pb->ntriples * sizeof(int2)) && down(tri_pt, pb->d_tri_pt, (size_t)pb->ntriples * sizeof(int)) &&
                    down(blk_j, pb->d_blk_j, (size_t)pb->nblk * sizeof(int)) && down(blk_k, pb->d_blk_k, (size_t)pb->nblk * sizeof(int)) &&
                    down(blk_task0, pb->d_blk_task0, ((size_t)pb->nblk + 1) * sizeof(int)) && down(tasks, pb->d_tasks, (size_t)pb->nslots * sizeof(SchurTask));
    return ok ? 0 : BSFM_ERROR;
}

int bsfm_schur_chunk(void) { return schur_chunk(); }
int bsfm_problem_cnp(const bsfm_problem_t* pb) { return pb->cnp; }
int bsfm_problem_num_cameras(const bsfm_problem_t* pb) { return pb->P.m; }
int bsfm_problem_num_points(const bsfm_problem_t* pb) { return pb->P.n; }
long long bsfm_problem_nvis(const bsfm_problem_t* pb) { return pb->P.nvis; }
int bsfm_lm_solve_attempts(const bsfm_problem_t* pb) { return pb->nlss; }

double bsfm_lm_last_kernel_ms(const bsfm_problem_t* pb, const char* phase)
{
    for (int i = 0; i < PH_COUNT; ++i)
        if (!strcmp(phase, kPhaseNames[i])) return pb->ph_cnt[i] ? pb->ph_ms[i] / pb->ph_cnt[i] : -1.0;
    if (!strcmp(phase, "flow_fallbacks")) return (double)pb->flow_fallbacks;
    if (!strcmp(phase, "groups")) return pb->comps.active ? (double)pb->comps.ncomp : 0.0;   // group-by-group reduced solve in use?
    if (!strcmp(phase, "potrf")) return pb->potrf.cnt ? pb->potrf.ms / pb->potrf.cnt : -1.0;
    if (!strcmp(phase, "syrk")) return pb->potrf.syrk_cnt ? pb->potrf.syrk_ms / (double)pb->potrf.syrk_cnt : -1.0;
    if (!strcmp(phase, "syrk_gflop")) return pb->potrf.syrk_cnt ? pb->potrf.syrk_flops * 1e-9 / (double)pb->potrf.syrk_cnt : -1.0;
    if (!strcmp(phase, "syrk_launches")) return pb->potrf.cnt ? (double)pb->potrf.syrk_cnt / (double)pb->potrf.cnt : -1.0;
    // the tile-dataflow factorisation (chol_flow.hip.h): HIP-event time of a k_chol_flow launch, the flops it was scheduled to do
    // (UPD + TRSM tile products at 2 * 128^3, POTRF + inverse at 128^3 per diagonal tile), tasks per launch
    if (!strcmp(phase, "flow_kernel")) return pb->potrf.flow && pb->potrf.flow->kern_cnt ? pb->potrf.flow->kern_ms / (double)pb->potrf.flow->kern_cnt : -1.0;
    if (!strcmp(phase, "flow_gflop")) return pb->potrf.flow && pb->potrf.flow->kern_cnt ? (pb->potrf.flow->flops + (double)pb->potrf.flow->nblk * POTRF_NB * POTRF_NB * POTRF_NB) * 1e-9 : -1.0;
    if (!strcmp(phase, "flow_tasks")) return pb->potrf.flow && pb->potrf.flow->kern_cnt ? (double)pb->potrf.flow->sched.tasks.size() : -1.0;
    if (!strcmp(phase, "flow_dist")) return pb->potrf.flow && pb->potrf.flow->dist ? (double)pb->potrf.flow->dist->n : 0.0;      // ranks that factored the last system together (0: replicated)
    if (!strcmp(phase, "flow_dynamic")) return pb->potrf.flow ? (double)pb->potrf.flow->dynamic : -1.0;
    if (!strcmp(phase, "flow_sim_us")) return pb->potrf.flow && pb->potrf.flow->kern_cnt ? pb->potrf.flow->sched.sim_us : -1.0;
    // bsfm_problem_create: host wall time (total / upload of the visibility index / index construction / allocation) and the
    // device time of the index construction alone
    if (!strcmp(phase, "create_total")) return pb->create_ms[0];
    if (!strcmp(phase, "create_upload")) return pb->create_ms[1];
    if (!strcmp(phase, "create_index")) return pb->create_ms[2];
    if (!strcmp(phase, "create_alloc")) return pb->create_ms[3];
    if (!strcmp(phase, "index_build")) return pb->index_build_ms;
    return -1.0;
}

int bsfm_lm_begin(bsfm_problem_t* pb)
{
    const long long nobs = 2 * pb->nvis_global;
    // The "last workgroup finishes" tickets (kernels.hip.h) go back to zero by themselves at the end of every launch.  A run that ended in an error may
    // have left one half-drawn -- the next launch would then never see a "last" workgroup and keep stale sums -- so after an error they are cleared
    // here (not unconditionally: on a 14-camera problem every stream operation is 3 % of an iteration).
    if (pb->error && pb->d_tickets && pb->tick_words)
        (void)hipMemsetAsync(pb->d_tickets, 0, pb->tick_words * sizeof(unsigned), pb->stream);
    pb->itno = 0; pb->stop = 0; pb->nu = 2; pb->nfev = 0; pb->njev = 0; pb->nlss = 0; pb->error = 0;
    pb->mu = 0.0; pb->eab_inf = 0.0; pb->dp_L2 = DBL_MAX; pb->p_L2 = 0.0; pb->maxdiag = DBL_MIN;
    for (int i = 0; i < PH_COUNT; ++i) { pb->ph_ms[i] = 0.0; pb->ph_cnt[i] = 0; }
    pb->potrf.ms = 0.0; pb->potrf.cnt = 0; pb->potrf.syrk_ms = 0.0; pb->potrf.syrk_cnt = 0; pb->potrf.syrk_flops = 0.0;
    if (pb->potrf.flow) { pb->potrf.flow->kern_ms = 0.0; pb->potrf.flow->kern_cnt = 0; pb->potrf.flow->kern_pending = false; }
    if (nobs < pb->nvars_global) {   // sba_levmar.c:647-650
        fprintf(stderr, "SBA: sba_motstr_levmar_x() cannot solve a problem with fewer measurements [%lld] than unknowns [%lld]\n",
                nobs, pb->nvars_global);
        pb->error = 1; pb->began = 1;
        return BSFM_ERROR;
    }
    invalidate_point_mirrors(pb);                       // whatever happened to the parameters since the last run: gather once
    launch_cam_table(pb, pb->d_p, pb->d_camtab);
    launch_residual(pb, pb->d_camtab, pb->d_p, pb->d_e, nullptr, SC_COST);
    hipLaunchKernelGGL(k_constraint_cost, dim3(1), dim3(256), 0, pb->stream, pb->P, pb->d_p,
                       pb->d_p + (size_t)pb->P.m * pb->cnp, 1, pb->d_scal + SC_CCOST);
    if (read_scalars(pb)) return BSFM_ERROR;
    double v[2] = { pb->h_scal[SC_COST], 0.0 };
    // point-constraint part of SC_CCOST is per rank, camera part replicated: split them for the reduction
    if (pb->world > 1) {
        hipLaunchKernelGGL(k_constraint_cost, dim3(1), dim3(256), 0, pb->stream, pb->P, pb->d_p,
                           pb->d_p + (size_t)pb->P.m * pb->cnp, 0, pb->d_scal + SC_COUNT + 8);
        HIP_OK(hipMemcpyAsync(&pb->h_scal[SC_COUNT + 8], pb->d_scal + SC_COUNT + 8, sizeof(double), hipMemcpyDeviceToHost, pb->stream));
        HIP_OK(hipStreamSynchronize(pb->stream));
        const double ptpart = pb->h_scal[SC_COUNT + 8], campart = pb->h_scal[SC_CCOST] - ptpart;
        v[1] = ptpart;
        if (allreduce_host(pb, v, 2, 0)) return BSFM_ERROR;
        pb->p_eL2 = v[0] + campart + v[1];
    } else {
        pb->p_eL2 = v[0] + pb->h_scal[SC_CCOST];
    }
    pb->nfev = 1;
    if (pb->opt.verbose >= 2) printf("initial motstr-SBA error %g [%g]\n", pb->p_eL2, pb->p_eL2 / (double)pb->nvis_global);
    pb->init_p_eL2 = pb->p_eL2;
    if (!std::isfinite(pb->p_eL2)) pb->stop = 7;
    pb->began = 1;
    return 0;
}

// Camera-only LM (sba_mot_levmar_x, lib/sba-1.5/sba_levmar.c:2090-2690): same damping control as the full problem, but
// the step is one cnp x cnp solve per camera, there is no Snavely stop rule (stop 8) and nlss counts cameras.
static int lm_iterate_mot(bsfm_problem_t* pb, int iters)
{
    const int cnp = pb->cnp;
    DevProblem& P = pb->P;
    const int itmax = pb->opt.itmax;
    const double tau = fabs(pb->opt.opts[0]), eps1 = fabs(pb->opt.opts[1]), eps2 = fabs(pb->opt.opts[2]),
                 eps2_sq = pb->opt.opts[2] * pb->opt.opts[2], eps3_sq = pb->opt.opts[3] * pb->opt.opts[3],
                 eps4_sq = pb->opt.opts[4] * pb->opt.opts[4];
    const size_t npts3 = (size_t)3 * P.n;
    int done = 0;
    for (; pb->itno < itmax && !pb->stop && done < iters; ++pb->itno, ++done) {
        if (compute_normal_blocks(pb)) return BSFM_ERROR;      // J (A part is what matters), U, ea (+ constraints)
        ++pb->njev;
        double* d_pa = pb->d_p;
        hipLaunchKernelGGL(k_iter_final, dim3(1), dim3(256), 0, pb->stream, P, d_pa, d_pa + (size_t)P.m * cnp, pb->d_red, 0, -1,
                           (int)SC_EABINF_A, (int)SC_EABINF_B, (int)SC_MAXDIAG_U, (int)SC_MAXDIAG_V, (int)SC_PL2_A, (int)SC_PL2_B,
                           (int)SC_CCOST, pb->d_scal);
        if (read_scalars(pb)) return BSFM_ERROR;
        const double ccost = pb->h_scal[SC_CCOST];
        pb->eab_inf = pb->h_scal[SC_EABINF_A];
        pb->p_L2 = pb->h_scal[SC_PL2_A];
        pb->maxdiag = std::max(DBL_MIN, pb->h_scal[SC_MAXDIAG_U]);
        if (pb->eab_inf <= eps1) { pb->dp_L2 = 0.0; pb->stop = 1; break; }
        if (pb->itno == 0) pb->mu = tau * pb->maxdiag;
        while (1) {
            const double mu = pb->mu;
            (void)hipMemsetAsync(pb->d_flags, 0, 4 * sizeof(int), pb->stream);
            ph_begin(pb, PH_SOLVE);
            DISPATCH_CNP(cnp, hipLaunchKernelGGL((k_cam_solve<C>), dim3(grid_for(P.m, 64)), dim3(64), 0, pb->stream, P, mu, pb->d_dp, pb->d_flags));
            ph_end(pb, PH_SOLVE);
            pb->nlss += P.m - P.mcon;                       // one linear system per free camera (sba_levmar.c:2505-2513)
            hipLaunchKernelGGL(k_cam_step, dim3(1), dim3(256), 0, pb->stream, P.m * cnp, P.mcon * cnp, mu, d_pa, pb->d_dp, pb->d_ea,
                               pb->d_pdp, pb->d_scal + SC_CAM3);
            if (npts3) (void)hipMemcpyAsync(pb->d_pdp + (size_t)P.m * cnp, pb->d_p + (size_t)P.m * cnp, npts3 * sizeof(double), hipMemcpyDeviceToDevice, pb->stream);
            ph_begin(pb, PH_RESID);
            launch_cam_table(pb, pb->d_pdp, pb->d_camtab_trial);
            launch_residual(pb, pb->d_camtab_trial, pb->d_p, pb->d_hx, pb->d_e, SC_COST_TRIAL);      // (the points do not move: those of d_p)
            ph_end(pb, PH_RESID);
            if (read_scalars(pb)) return BSFM_ERROR;
            collect_phase_times(pb);
            double sums[1] = { pb->h_scal[SC_COST_TRIAL] };
            if (pb->world > 1 && allreduce_host(pb, sums, 1, 0)) return BSFM_ERROR;
            bool accepted = false;
            if (pb->h_flags[1] == 0) {                      // every U_j + mu I was positive definite (nsolved == m)
                pb->dp_L2 = pb->h_scal[SC_CAM3 + 0];
                const double dL = pb->h_scal[SC_CAM3 + 2];
                if (pb->dp_L2 <= eps2_sq * pb->p_L2) { pb->stop = 2; break; }
                if (pb->dp_L2 >= (pb->p_L2 + eps2) / SBA_EPSILON_SQ) {
                    fprintf(stderr, "SBA: the matrix of the augmented normal equations is almost singular in sba_mot_levmar_x(),\n"
                                    "     minimization should be restarted from the current solution with an increased damping term\n");
                    pb->error = 1;
                    return BSFM_ERROR;
                }
                ++pb->nfev;
                double pdp_eL2 = sums[0];
                if (!std::isfinite(pdp_eL2)) { pb->stop = 7; break; }
                pdp_eL2 += ccost;                            // constraint terms at the OLD p, as in the reference (sba_levmar.c:2592-2611)
                const double dF = pb->p_eL2 - pdp_eL2;
                if (pb->opt.verbose >= 2)
                    printf("\ndamping term %8g, gain ratio %8g, errors %8g / %8g = %g\n", mu, dL != 0.0 ? dF / dL : dF / DBL_EPSILON,
                           pb->p_eL2 / (double)pb->nvis_global, pdp_eL2 / (double)pb->nvis_global, pb->p_eL2 / pdp_eL2);
                if (dL > 0.0 && dF > 0.0) {
                    double tmp = (2.0 * dF / dL - 1.0);
                    tmp = 1.0 - tmp * tmp * tmp;
                    pb->mu = pb->mu * ((tmp >= SBA_ONE_THIRD) ? tmp : SBA_ONE_THIRD);
                    pb->nu = 2;
                    if (pdp_eL2 - 2.0 * sqrt(pb->p_eL2 * pdp_eL2) < (eps4_sq - 1.0) * pb->p_eL2) pb->stop = 4;
                    std::swap(pb->d_p, pb->d_pdp);
                    for (int ms = 0; ms < 2; ++ms) if (pb->ptc_tag[ms] == pb->d_pdp) pb->ptc_tag[ms] = pb->d_p;      // the points did not move: the mirror of the old vector is the mirror of the new one
                    std::swap(pb->d_e, pb->d_hx);
                    std::swap(pb->d_camtab, pb->d_camtab_trial);
                    pb->p_eL2 = pdp_eL2;
                    accepted = true;
                }
            }
            if (accepted) break;
            pb->mu *= pb->nu;
            const int nu2 = (int)((unsigned)pb->nu << 1);
            if (nu2 <= pb->nu) {
                fprintf(stderr, "SBA: too many failed attempts to increase the damping factor in sba_mot_levmar_x()! Singular Hessian matrix?\n");
                pb->stop = 6;
                break;
            }
            pb->nu = nu2;
        }
        if (pb->p_eL2 <= eps3_sq) pb->stop = 5;
    }
    if (pb->itno >= itmax) pb->stop = 3;
    return pb->stop;
}

static int lm_iterate_impl(bsfm_problem_t* pb, int iters);

int bsfm_lm_iterate(bsfm_problem_t* pb, int iters)
{
    const int before = g_infra_failure;
    const int rc = lm_iterate_impl(pb, iters);
    if (rc == BSFM_ERROR && g_infra_failure && !before && has_collective(pb)) {
        // This rank failed on its own (HIP error, broken collective) in the middle of an iteration: its peers are, or will shortly be,
        // blocked inside a collective that this rank will never join, and RCCL has no time-out.  Ending the process is the only way to
        // end the job (the launcher then takes the other ranks down); the reference's own fatal paths exit(1) as well.
        fprintf(stderr, "[bsfm] FATAL: rank %d of %d failed inside an LM iteration; aborting the process so that its peers are not left "
                        "waiting in a collective\n", pb->rank, pb->world);
        fflush(stderr);
        abort();
    }
    return rc;
}

static int lm_iterate_impl(bsfm_problem_t* pb, int iters)
{
    if (!pb->began) { if (bsfm_lm_begin(pb) != 0) return BSFM_ERROR; }
    if (pb->error) return BSFM_ERROR;
    if (pb->mot) return lm_iterate_mot(pb, iters);
    const int cnp = pb->cnp;
    DevProblem& P = pb->P;
    const int itmax = pb->opt.itmax;
    const double tau = fabs(pb->opt.opts[0]), eps1 = fabs(pb->opt.opts[1]), eps2 = fabs(pb->opt.opts[2]),
                 eps2_sq = pb->opt.opts[2] * pb->opt.opts[2], eps3_sq = pb->opt.opts[3] * pb->opt.opts[3],
                 eps4_sq = pb->opt.opts[4] * pb->opt.opts[4], eps5 = pb->opt.opts[5];
    double* d_pa = pb->d_p; double* d_pb = pb->d_p + (size_t)P.m * cnp;
    double* d_dpa = pb->d_dp; double* d_dpb = pb->d_dp + (size_t)P.m * cnp;
    double* d_pdpa = pb->d_pdp; double* d_pdpb = pb->d_pdp + (size_t)P.m * cnp;
    const int nbp = grid_for(P.n, 256);
    int done = 0;

    for (; pb->itno < itmax && !pb->stop && done < iters; ++pb->itno, ++done) {
        // J, U / ea, V / eb -- and, in the last workgroup of the point-block kernel, ||J^T e||_inf, ||p||^2, the largest diagonal entry
        // (sba_levmar.c:1085-1128) and the cleared flags of the first solve attempt
        bool flags_cleared = false;
        if (compute_normal_blocks(pb, true, &flags_cleared)) return BSFM_ERROR;
        ++pb->njev;
        if (!flags_cleared)      // a problem without points: the camera side alone (the point entries of d_scal stay at their initial zeros)
            hipLaunchKernelGGL(k_iter_final, dim3(1), dim3(256), 0, pb->stream, P, (const double*)d_pa, (const double*)d_pb, (const double*)pb->d_red, 0, -1,
                               SC_EABINF_A, SC_EABINF_B, SC_MAXDIAG_U, SC_MAXDIAG_V, SC_PL2_A, SC_PL2_B, SC_CCOST, pb->d_scal);
        // The gradient norm / parameter norm / largest diagonal of this iteration (sba_levmar.c:1085-1130).  They cost a round trip to the
        // host, and on a 14-camera problem a round trip is a tenth of the iteration -- so from the second iteration on (mu is known
        // then) the first damping attempt is launched BEFORE they are read and they come back with the attempt's own scalars.  If the
        // gradient test says stop, the attempt is dropped: it has only written trial buffers and no counter has moved.
        double ccost = 0.0;
        auto take_iteration_scalars = [&]() -> int {
            const double vmaxdiag = pb->h_scal[SC_MAXDIAG_V];   // diagonals are sums of squares (>= 0): a 0-based max is exact
            ccost = pb->h_scal[SC_CCOST];
            double mx[2] = { pb->h_scal[SC_EABINF_B], vmaxdiag };
            double sm[2] = { pb->h_scal[SC_PL2_B], 0.0 };
            if (pb->world > 1) {
                HIP_OK(hipMemcpy(&sm[1], pb->d_scal + SC_COUNT + 8, sizeof(double), hipMemcpyDeviceToHost));
                const double campart = ccost - sm[1];
                if (allreduce_mixed(pb, sm, 2, mx, 2)) return BSFM_ERROR;
                ccost = campart + sm[1];
            }
            pb->eab_inf = std::max(pb->h_scal[SC_EABINF_A], mx[0]);
            pb->p_L2 = pb->h_scal[SC_PL2_A] + sm[0];
            double md = DBL_MIN;
            if (pb->h_scal[SC_MAXDIAG_U] > md) md = pb->h_scal[SC_MAXDIAG_U];
            if (mx[1] > md) md = mx[1];
            pb->maxdiag = md;
            return 0;
        };
        bool deferred = pb->itno > 0 && pb->world == 1 && pb->speculate;
        bool grad_stop = false;        // the deferred gradient test fired: leave the iteration loop exactly where the reference does (sba_levmar.c:1125)
        if (!deferred) {
            if (read_scalars(pb)) return BSFM_ERROR;
            if (take_iteration_scalars()) return BSFM_ERROR;
            if (pb->eab_inf <= eps1) { pb->dp_L2 = 0.0; pb->stop = 1; break; }
            if (pb->itno == 0) pb->mu = tau * pb->maxdiag;
        }

        while (1) {   // determine increment using adaptive damping (sba_levmar.c:1131)
            const double mu = pb->mu;
            if (!flags_cleared) (void)hipMemsetAsync(pb->d_flags, 0, 4 * sizeof(int), pb->stream);      // (a repeated attempt, or no point-block kernel before it)
            flags_cleared = false;
            ph_begin(pb, PH_INVERT);
            // V*^-1: inside k_schur_prep when that kernel runs and every point has an observation (round 6), else here
            if (P.n > 0 && !(pb->fuse_invert && pb->ntasks > 0))
                hipLaunchKernelGGL(k_point_invert, dim3(nbp), dim3(256), 0, pb->stream, P.n, mu, pb->d_V, pb->d_Vinv, pb->d_flags);
            ph_end(pb, PH_INVERT);
            ph_begin(pb, PH_SCHUR);
            if (compute_schur(pb, mu)) return BSFM_ERROR;
            ph_end(pb, PH_SCHUR);
            ph_begin(pb, PH_SOLVE);
            // S dpa = E, Cholesky (sba_Axb_Chol, lib/sba-1.5/sba_lapack.c:374-485); info -> d_flags[1]
            if (pb->comps.active) {
                if (comp_solve(pb->comps, pb->potrf, cnp, pb->d_S, pb->ld, pb->d_E, d_dpa + (size_t)P.mcon * cnp, pb->d_flags + 1, pb->stream)) return BSFM_ERROR;
            } else if (pb->envelope) {
                if (potrf_solve(pb->potrf, pb->d_S, pb->ld, pb->Sdim, pb->d_E, pb->d_xperm, pb->d_flags + 1, pb->stream)) return BSFM_ERROR;
                hipLaunchKernelGGL(k_unpermute_step, dim3(grid_for((size_t)pb->Sdim, 256)), dim3(256), 0, pb->stream, pb->Sdim, cnp,
                                   (const int*)pb->d_spos, (const double*)pb->d_xperm, d_dpa + (size_t)P.mcon * cnp);
            } else if (potrf_solve(pb->potrf, pb->d_S, pb->ld, pb->Sdim, pb->d_E, d_dpa + (size_t)P.mcon * cnp, pb->d_flags + 1, pb->stream)) return BSFM_ERROR;
            ph_end(pb, PH_SOLVE);
            if (P.mcon > 0) (void)hipMemsetAsync(d_dpa, 0, (size_t)P.mcon * cnp * sizeof(double), pb->stream);
            ph_begin(pb, PH_BACKSUB);
            if (P.n > 0) {
                // back-substitution; its last workgroup also does the camera part of the step with the sums (k_step_sums) and the camera
                // table of the trial point (k_cam_table): three launches in one
                const int ms = trial_mirror_slot(pb);            // the trial points also go to the camera-major mirror the residual kernel streams
                StepFinalArgs fa; fa.count = P.m * cnp; fa.fixed = P.mcon * cnp; fa.pa = d_pa; fa.pdpa = d_pdpa; fa.out3 = pb->d_scal + SC_CAM3;
                const bool table_in_backsub = P.m < 256;      // (see k_backsub: the finishing workgroup builds the trial point's camera table only on small problems)
                fa.pt3 = pb->d_scal + SC_PT_DP; fa.known = pb->d_known; fa.with_fd = pb->opt.jacobian == BSFM_JAC_FD ? 1 : 0; fa.camtab_trial = table_in_backsub ? pb->d_camtab_trial : nullptr;
                fa.ticket_groups = pb->d_tickets + pb->tick_back;
                // big problems: the per-observation products W^T da in a streaming pass of their own (k_backsub_obs); they go where the Schur
                // phase kept its per-observation records (d_Cc: dead once S is assembled)
                const double* wobs = nullptr;
                if (pb->backsub_two_pass && pb->d_Cc) {
                    DISPATCH_CNP(cnp, hipLaunchKernelGGL((k_backsub_obs<C>), dim3(grid_for((size_t)P.nvis, 256)), dim3(256), 0, pb->stream, P.nvis, P.mcon,
                                                          (const int*)pb->d_cam_cam, (const double*)pb->d_Ac, (const double*)pb->d_Bc, (const double*)d_dpa, pb->d_Cc));
                    wobs = pb->d_Cc;
                }
                // (one-pass form: four lanes per point; the trial point's camera table, where this kernel builds it, is one more workgroup behind the points')
                fa.point_blocks = wobs ? nbp : grid_for((size_t)BS_LANES * (size_t)P.n, 256);
                const unsigned bs_grid = (unsigned)fa.point_blocks + (table_in_backsub ? 1u : 0u);
                if (wobs) {
                    DISPATCH_CNP(cnp, hipLaunchKernelGGL((k_backsub<C, true>), dim3(bs_grid), dim3(256), 0, pb->stream, P, mu, d_dpa, d_pb, d_dpb, d_pdpb, pb->d_red, pb->d_ptc[ms],
                                                          wobs, pb->d_tickets + 2, fa));
                } else {
                    DISPATCH_CNP(cnp, hipLaunchKernelGGL((k_backsub<C, false>), dim3(bs_grid), dim3(256), 0, pb->stream, P, mu, d_dpa, d_pb, d_dpb, d_pdpb, pb->d_red, pb->d_ptc[ms],
                                                          wobs, pb->d_tickets + 2, fa));
                }
                if (!table_in_backsub) launch_cam_table(pb, pb->d_pdp, pb->d_camtab_trial);
                pb->ptc_tag[ms] = pb->d_pdp;
            } else {
                hipLaunchKernelGGL(k_step_sums, dim3(1), dim3(256), 0, pb->stream, P.m * cnp, P.mcon * cnp, mu, d_pa, d_dpa, pb->d_ea, d_pdpa,
                                   pb->d_scal + SC_CAM3, pb->d_red, 0, pb->d_scal + SC_PT_DP);
                launch_cam_table(pb, pb->d_pdp, pb->d_camtab_trial);
            }
            ph_end(pb, PH_BACKSUB);
            ph_begin(pb, PH_RESID);
            launch_residual(pb, pb->d_camtab_trial, pb->d_pdp, pb->d_hx, pb->d_e, SC_COST_TRIAL);
            ph_end(pb, PH_RESID);
            if (read_scalars(pb)) return BSFM_ERROR;
            collect_phase_times(pb);
            if (deferred) {      // the iteration's own scalars arrived with this attempt's
                deferred = false;
                if (take_iteration_scalars()) return BSFM_ERROR;
                if (pb->eab_inf <= eps1) { pb->dp_L2 = 0.0; pb->stop = 1; grad_stop = true; break; }      // the attempt is dropped unseen
            }

            double flagsd[1] = { (double)pb->h_flags[0] };
            double sums[4] = { pb->h_scal[SC_PT_DP], pb->h_scal[SC_PT_DL], pb->h_scal[SC_COST_TRIAL], 0.0 };
            double maxs[1] = { pb->h_scal[SC_PCT] };
            int potrf_info = pb->h_flags[1];
            if (pb->world > 1) {
                // a hand-off time-out inside one rank's solve rides the exchange: every rank learns of it and they leave TOGETHER
                // (a rank that left alone would strand its peers inside the next collective)
                // (distributed factorisation: every rank holds dpotrf's word for its own tile columns; the system's is the smallest positive one)
                double mx3[4] = { flagsd[0], maxs[0], potrf_info < 0 ? 1.0 : 0.0, potrf_info > 0 ? 1e9 - (double)potrf_info : 0.0 };
                if (allreduce_mixed(pb, sums, 3, mx3, 4)) return BSFM_ERROR;
                flagsd[0] = mx3[0]; maxs[0] = mx3[1];
                if (mx3[2] != 0.0 && potrf_info >= 0) potrf_info = POTRF_INFO_TIMEOUT;
                if (potrf_info >= 0 && pb->potrf.dist_comm) potrf_info = mx3[3] > 0.0 ? (int)(1e9 - mx3[3] + 0.5) : 0;
            }
            const bool singularV = flagsd[0] != 0.0;
            if (potrf_info < 0) {         // POTRF_INFO_TIMEOUT: a hand-off inside the persistent Cholesky kernels never arrived
                // The tile-dataflow launch needs its workgroups co-resident to make progress at full speed; a GPU shared with another
                // process can starve it beyond the spin limit.  The three-stream schedule of rounds 1-3 (same binary, same arithmetic,
                // ordinary launches that wait for nothing but stream order) does not: this problem switches to it, warns once, and the
                // attempt is repeated from the point inversion on -- S is rebuilt from the block sums, no LM counter has moved yet.
                // Every rank of a multi-GPU job sees the same verdict (it rode the exchange above), so they switch together.
                const int nblk_s = (pb->Sdim + POTRF_NB - 1) / POTRF_NB;
                if (pb->potrf.use_flow && !pb->comps.active && nblk_s <= POTRF_MAX_TILES) {
                    static bool warned = false;
                    if (!warned) { warned = true;
                        fprintf(stderr, "[bsfm] WARNING: the tile-dataflow Cholesky launch timed out waiting for its own workgroups (GPU shared with another "
                                        "job?); falling back to the stream-ordered schedule for this problem\n"); }
                    pb->potrf.use_flow = 0;
                    ++pb->flow_fallbacks;
                    continue;
                }
                fprintf(stderr, "[bsfm] FATAL: the reduced camera solve timed out inside its persistent kernels (info %d)\n", potrf_info);
                pb->error = 1; g_infra_failure = 1;
                return BSFM_ERROR;
            }
            bool accepted = false;
            if (singularV) {
                fprintf(stderr, "SBA: singular matrix V*_i in sba_motstr_levmar_x(), increasing damping\n");
            } else {
                ++pb->nlss;
                const bool issolved = potrf_info == 0;
                if (!issolved)
                    fprintf(stderr, "LAPACK error: the leading minor of order %d is not positive definite,\nthe factorization could not be completed for dpotf2/dpotrf in sba_Axb_Chol()\n", potrf_info);
                if (issolved) {
                    pb->dp_L2 = pb->h_scal[SC_CAM3 + 0] + sums[0];
                    const double dL = pb->h_scal[SC_CAM3 + 2] + sums[1];
                    if (pb->dp_L2 <= eps2_sq * pb->p_L2) { pb->stop = 2; break; }
                    if (pb->dp_L2 >= (pb->p_L2 + eps2) / SBA_EPSILON_SQ) {
                        fprintf(stderr, "SBA: the matrix of the augmented normal equations is almost singular in sba_motstr_levmar_x(),\n"
                                        "     minimization should be restarted from the current solution with an increased damping term\n");
                        pb->error = 1;
                        return BSFM_ERROR;
                    }
                    ++pb->nfev;
                    double pdp_eL2 = sums[2];
                    if (!std::isfinite(pdp_eL2)) { pb->stop = 7; break; }
                    pdp_eL2 += ccost;   // constraint terms evaluated at the OLD p (sba_levmar.c:1488-1522)
                    const double dF = pb->p_eL2 - pdp_eL2;
                    if (pb->opt.verbose >= 2) {
                        printf("\ndamping term %8g, gain ratio %8g, errors %8g / %8g = %g\n", mu, dL != 0.0 ? dF / dL : dF / DBL_EPSILON,
                               pb->p_eL2 / (double)pb->nvis_global, pdp_eL2 / (double)pb->nvis_global, pb->p_eL2 / pdp_eL2);
                        printf("pdp_eL2: %0.3f, nvis: %lld\n", pdp_eL2, pb->nvis_global);
                    }
                    if (dL > 0.0 && dF > 0.0) {
                        double tmp = (2.0 * dF / dL - 1.0);
                        tmp = 1.0 - tmp * tmp * tmp;
                        pb->mu = pb->mu * ((tmp >= SBA_ONE_THIRD) ? tmp : SBA_ONE_THIRD);
                        pb->nu = 2;
                        const double max_pct_change = maxs[0];
                        if (pb->opt.verbose >= 2) printf("max_pct_change: %0.3e\n", max_pct_change);
                        if (pdp_eL2 - 2.0 * sqrt(pb->p_eL2 * pdp_eL2) < (eps4_sq - 1.0) * pb->p_eL2) pb->stop = 4;
                        if (max_pct_change < eps5 && pb->itno >= 4) { pb->stop = 8; break; }   // p NOT updated (sba_levmar.c:1569-1572)
                        std::swap(pb->d_p, pb->d_pdp);
                        std::swap(pb->d_e, pb->d_hx);
                        std::swap(pb->d_camtab, pb->d_camtab_trial);
                        d_pa = pb->d_p; d_pb = pb->d_p + (size_t)P.m * cnp;
                        d_pdpa = pb->d_pdp; d_pdpb = pb->d_pdp + (size_t)P.m * cnp;
                        pb->p_eL2 = pdp_eL2;
                        accepted = true;
                    }
                }
            }
            if (accepted) break;
            // moredamping (sba_levmar.c:1584-1611)
            pb->mu *= pb->nu;
            const int nu2 = (int)((unsigned)pb->nu << 1);
            if (nu2 <= pb->nu) {
                fprintf(stderr, "SBA: too many failed attempts to increase the damping factor in sba_motstr_levmar_x()! Singular Hessian matrix?\n");
                pb->stop = 6;
                break;
            }
            pb->nu = nu2;
        }
        if (grad_stop) break;
        if (pb->p_eL2 <= eps3_sq) pb->stop = 5;
    }
    if (pb->itno >= itmax) pb->stop = 3;
    return pb->stop;
}

int bsfm_lm_finish(bsfm_problem_t* pb, double info[BSFM_INFOSZ])
{
    if (!pb->began) return BSFM_ERROR;
    if (info) {   // sba_levmar.c:2028-2049 and the nvis scaling of sba_levmar_wrap.c:684-695
        info[0] = pb->init_p_eL2; info[1] = pb->p_eL2; info[2] = pb->eab_inf; info[3] = pb->dp_L2;
        info[4] = pb->mu / pb->maxdiag; info[5] = pb->itno; info[6] = pb->stop;
        info[7] = (double)pb->nfev * (double)pb->nvis_global; info[8] = (double)pb->njev * (double)pb->nvis_global; info[9] = pb->nlss;
    }
    if (pb->error) return BSFM_ERROR;
    return (pb->stop != 7) ? pb->itno : BSFM_ERROR;
}

int bsfm_problem_download(bsfm_problem_t* pb, double* p_out, bsfm_camera_params_t* cams, double* pts)
{
    const int cnp = pb->cnp, m = pb->P.m, n = pb->P.n;
    std::vector<double> p(pb->nvars_local);
    HIP_OK(hipStreamSynchronize(pb->stream));
    HIP_OK(hipMemcpy(p.data(), pb->d_p, p.size() * sizeof(double), hipMemcpyDeviceToHost));
    if (p_out) memcpy(p_out, p.data(), p.size() * sizeof(double));
    if (cams) {   // sfm.c:876-922
        const ModelCfg& c = pb->P.cfg;
        for (int j = 0; j < m; ++j) {
            const double* a = &p[(size_t)j * cnp];
            cams[j].t[0] = a[0]; cams[j].t[1] = a[1]; cams[j].t[2] = a[2];
            // rot_update on the host (sfm.c:77-116)
            const double* R0 = &pb->h_Rinit[9 * (size_t)j];
            const double w0 = a[3], w1 = a[4], w2 = a[5];
            const double th = sqrt(w0 * w0 + w1 * w1 + w2 * w2);
            double Rn[9];
            if (th == 0.0) memcpy(Rn, R0, sizeof(Rn));
            else {
                const double nn[3] = { w0 / th, w1 / th, w2 / th };
                const double nx[9] = { 0.0, -nn[2], nn[1], nn[2], 0.0, -nn[0], -nn[1], nn[0], 0.0 };
                double nxsq[9], dR[9];
                for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc)
                    nxsq[3 * r + cc] = nx[3 * r] * nx[cc] + nx[3 * r + 1] * nx[3 + cc] + nx[3 * r + 2] * nx[6 + cc];
                const double s = sin(th), c1 = 1.0 - cos(th);
                for (int k = 0; k < 9; ++k) dR[k] = ((k % 4 == 0) ? 1.0 : 0.0) + nx[k] * s + nxsq[k] * c1;
                for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc)
                    Rn[3 * r + cc] = dR[3 * r] * R0[cc] + dR[3 * r + 1] * R0[3 + cc] + dR[3 * r + 2] * R0[6 + cc];
            }
            memcpy(cams[j].R, Rn, sizeof(Rn));
            int col = 6;
            if (c.est_focal) { cams[j].f = a[6] / c.f_scale; col = 7; }
            if (c.undistort) { cams[j].k[0] = a[col] / c.k_scale; cams[j].k[1] = a[col + 1] / c.k_scale; }
            cams[j].f_scale = 1.0; cams[j].k_scale = 1.0;
        }
    }
    if (pts && n) memcpy(pts, &p[(size_t)m * cnp], sizeof(double) * 3 * n);
    return 0;
}

int bsfm_problem_outlier_stats(bsfm_problem_t* pb, double min_thr, double max_thr, int* cam_nobs, double* cam_mean,
                               double* cam_kth80, double* cam_kth50, double* cam_thresh,
                               unsigned char* point_outlier, double* point_err, double* global_mean)
{
    const int m = pb->P.m, n = pb->P.n, nvis = pb->P.nvis;
    // residuals at the current parameters (idempotent for the LM state: d_e always holds e(p))
    launch_cam_table(pb, pb->d_p, pb->d_camtab);
    launch_residual(pb, pb->d_camtab, pb->d_p, pb->d_e, nullptr, SC_COST);
    double *dist = nullptr, *dist_cm = nullptr, *stats = nullptr, *perr = nullptr; int* cnt = nullptr; unsigned char* pflag = nullptr;
    auto cleanup = [&] { for (void* q : { (void*)dist, (void*)dist_cm, (void*)stats, (void*)perr, (void*)cnt, (void*)pflag }) bsfm::dev_free(q); };
    if (dmalloc(&dist, (size_t)nvis) != hipSuccess || dmalloc(&dist_cm, (size_t)nvis) != hipSuccess ||
        dmalloc(&stats, (size_t)4 * m) != hipSuccess || dmalloc(&perr, (size_t)n) != hipSuccess ||
        dmalloc(&cnt, (size_t)m) != hipSuccess || dmalloc(&pflag, (size_t)n) != hipSuccess) { cleanup(); return BSFM_ERROR; }
    if (nvis > 0)
        hipLaunchKernelGGL(k_obs_dist, dim3(grid_for(nvis, 256)), dim3(256), 0, pb->stream, nvis, pb->d_e, pb->d_campos, dist, dist_cm);
    if (m > 0)
        hipLaunchKernelGGL(k_cam_dist_stats, dim3(m), dim3(256), 0, pb->stream, m, pb->d_camptr, dist_cm, min_thr, max_thr,
                           cnt, stats, stats + m, stats + 2 * (size_t)m, stats + 3 * (size_t)m);
    if (n > 0)
        hipLaunchKernelGGL(k_point_outliers, dim3(grid_for(n, 256)), dim3(256), 0, pb->stream, n, pb->d_rowptr, pb->d_obs_cam, dist,
                           stats + 3 * (size_t)m, pb->d_pcon ? pb->d_pval : (const double*)nullptr, pflag, perr);
    if (hipStreamSynchronize(pb->stream) != hipSuccess) { cleanup(); return BSFM_ERROR; }
    std::vector<double> h((size_t)4 * m); std::vector<int> hc((size_t)m);
    bool ok = true;
    if (m > 0) {
        ok = ok && hipMemcpy(h.data(), stats, h.size() * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess;
        ok = ok && hipMemcpy(hc.data(), cnt, hc.size() * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
    }
    if (cam_nobs) memcpy(cam_nobs, hc.data(), hc.size() * sizeof(int));
    if (cam_mean) memcpy(cam_mean, h.data(), (size_t)m * sizeof(double));
    if (cam_kth80) memcpy(cam_kth80, h.data() + m, (size_t)m * sizeof(double));
    if (cam_kth50) memcpy(cam_kth50, h.data() + 2 * (size_t)m, (size_t)m * sizeof(double));
    if (cam_thresh) memcpy(cam_thresh, h.data() + 3 * (size_t)m, (size_t)m * sizeof(double));
    if (global_mean) {            // Bundle.cpp:846-849: sum of the per-camera sums over all observations
        double tot = 0.0; long long cntall = 0;
        for (int j = 0; j < m; ++j) { tot += h[j] * hc[j]; cntall += hc[j]; }
        *global_mean = cntall ? tot / (double)cntall : 0.0;
    }
    if (point_outlier && n > 0) ok = ok && hipMemcpy(point_outlier, pflag, (size_t)n, hipMemcpyDeviceToHost) == hipSuccess;
    if (point_err && n > 0) ok = ok && hipMemcpy(point_err, perr, (size_t)n * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess;
    cleanup();
    return ok ? 0 : BSFM_ERROR;
}

int bsfm_problem_ray_angles(bsfm_problem_t* pb, double ray_angle_threshold, double* max_angle_deg, unsigned char* prune,
                            int* num_pruned)
{
    const int m = pb->P.m, n = pb->P.n, nvis = pb->P.nvis;
    if (pb->mot) { fprintf(stderr, "[bsfm] ray angles: the camera-only problem does not hold the points on the device\n"); return BSFM_ERROR; }
    double *rays = nullptr, *deg = nullptr; unsigned char* flag = nullptr;
    auto cleanup = [&] { for (void* q : { (void*)rays, (void*)deg, (void*)flag }) bsfm::dev_free(q); };
    if (dmalloc(&rays, (size_t)3 * nvis + 1) != hipSuccess || dmalloc(&deg, (size_t)n + 1) != hipSuccess ||
        dmalloc(&flag, (size_t)n + 1) != hipSuccess) { cleanup(); return BSFM_ERROR; }
    if (nvis > 0)
        hipLaunchKernelGGL(k_obs_rays, dim3(grid_for(nvis, 256)), dim3(256), 0, pb->stream, nvis, pb->cnp, pb->d_obs_cam, pb->d_obs_pt,
                           pb->d_p, pb->d_p + (size_t)m * pb->cnp, rays);
    if (n > 0)
        hipLaunchKernelGGL(k_point_ray_angle, dim3(grid_for(n, 256)), dim3(256), 0, pb->stream, n, pb->d_rowptr, rays,
                           0.5 * ray_angle_threshold, deg, flag);
    if (hipStreamSynchronize(pb->stream) != hipSuccess) { cleanup(); return BSFM_ERROR; }
    std::vector<unsigned char> hf((size_t)n);
    bool ok = n == 0 || hipMemcpy(hf.data(), flag, (size_t)n, hipMemcpyDeviceToHost) == hipSuccess;
    if (max_angle_deg && n > 0) ok = ok && hipMemcpy(max_angle_deg, deg, (size_t)n * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess;
    if (prune && n > 0) memcpy(prune, hf.data(), (size_t)n);
    if (num_pruned) { int c = 0; for (int i = 0; i < n; ++i) c += hf[i]; *num_pruned = c; }
    cleanup();
    return ok ? 0 : BSFM_ERROR;
}

int bsfm_eval_residuals(bsfm_problem_t* pb, double* e_out, double* cost)
{
    launch_cam_table(pb, pb->d_p, pb->d_camtab);
    launch_residual(pb, pb->d_camtab, pb->d_p, pb->d_e, nullptr, SC_COST);
    hipLaunchKernelGGL(k_constraint_cost, dim3(1), dim3(256), 0, pb->stream, pb->P, pb->d_p,
                       pb->d_p + (size_t)pb->P.m * pb->cnp, 1, pb->d_scal + SC_CCOST);
    if (read_scalars(pb)) return BSFM_ERROR;
    if (cost) *cost = pb->h_scal[SC_COST] + pb->h_scal[SC_CCOST];
    if (e_out && pb->P.nvis) {   // the device copy is camera-major: back to observation (CRS) order through the trial buffer
        hipLaunchKernelGGL(k_permute16, dim3(grid_for(pb->P.nvis, 256)), dim3(256), 0, pb->stream, pb->P.nvis, pb->d_campos, pb->d_e, pb->d_hx);
        HIP_OK(hipStreamSynchronize(pb->stream));
        HIP_OK(hipMemcpy(e_out, pb->d_hx, 2 * (size_t)pb->P.nvis * sizeof(double), hipMemcpyDeviceToHost));
    }
    return 0;
}

int bsfm_eval_normal_equations(bsfm_problem_t* pb, double mu, double* U, double* ea, double* V, double* eb,
                               double* J, double* S, double* E)
{
    const int cnp = pb->cnp, m = pb->P.m, n = pb->P.n;
    launch_cam_table(pb, pb->d_p, pb->d_camtab);
    launch_residual(pb, pb->d_camtab, pb->d_p, pb->d_e, nullptr, SC_COST);
    if (compute_normal_blocks(pb)) return BSFM_ERROR;
    (void)hipMemsetAsync(pb->d_flags, 0, 4 * sizeof(int), pb->stream);
    if (n > 0) hipLaunchKernelGGL(k_point_invert, dim3(grid_for(n, 256)), dim3(256), 0, pb->stream, n, mu, pb->d_V, pb->d_Vinv, pb->d_flags);
    pb->export_full_s = 1;            // S is copied out below as a full symmetric matrix
    const int rc_schur = compute_schur(pb, mu);
    pb->export_full_s = 0;
    if (rc_schur) return BSFM_ERROR;
    HIP_OK(hipStreamSynchronize(pb->stream));
    if (U) {
        std::vector<double> h((size_t)m * cnp * cnp);
        HIP_OK(hipMemcpy(h.data(), pb->d_U, h.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int j = pb->P.mcon; j < m; ++j) for (int q = 0; q < cnp; ++q) h[(size_t)j * cnp * cnp + q * cnp + q] += mu;
        memcpy(U, h.data(), h.size() * sizeof(double));
    }
    if (ea) HIP_OK(hipMemcpy(ea, pb->d_ea, (size_t)m * cnp * sizeof(double), hipMemcpyDeviceToHost));
    if (V && n) {
        double* tmp = nullptr;
        HIP_OK(dmalloc(&tmp, 9 * (size_t)n));
        hipLaunchKernelGGL(k_expand_v, dim3(grid_for(n, 256)), dim3(256), 0, pb->stream, n, mu, pb->d_V, tmp);
        HIP_OK(hipStreamSynchronize(pb->stream));
        HIP_OK(hipMemcpy(V, tmp, 9 * (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
        bsfm::dev_free(tmp);
    }
    if (eb && n) HIP_OK(hipMemcpy(eb, pb->d_eb, 3 * (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    if (J && pb->P.nvis) {   // the device copies are camera-major and chunked: back to the reference's record A (2 x cnp) || B (2 x 3) in
                             // observation order on the host (export path only)
        const size_t nv = (size_t)pb->P.nvis, js = (size_t)(2 * cnp + 6);
        std::vector<double> ac(nv * 2 * cnp), bc(nv * 8); std::vector<int> pos(nv);
        HIP_OK(hipMemcpy(ac.data(), pb->d_Ac, ac.size() * sizeof(double), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(bc.data(), pb->d_Bc, bc.size() * sizeof(double), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(pos.data(), pb->d_campos, nv * sizeof(int), hipMemcpyDeviceToHost));
        for (size_t k = 0; k < nv; ++k) {
            const double* a = ac.data() + (size_t)pos[k] * 2 * cnp; const double* b = bc.data() + (size_t)pos[k] * 8;
            double* o = J + k * js;
            for (int c = 0; c < cnp; ++c) { o[c] = a[2 * c]; o[cnp + c] = a[2 * c + 1]; }
            for (int q = 0; q < 6; ++q) o[2 * cnp + q] = b[q];
        }
    }
    if (S && pb->Sdim) HIP_OK(hipMemcpy2D(S, (size_t)pb->Sdim * sizeof(double), pb->d_S, (size_t)pb->ld * sizeof(double),
                                          (size_t)pb->Sdim * sizeof(double), pb->Sdim, hipMemcpyDeviceToHost));
    if (E && pb->Sdim) HIP_OK(hipMemcpy(E, pb->d_E, (size_t)pb->Sdim * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int bsfm_chol_flow_schedule(int nblk, const int* last, int np_max, int slots, void* tasks_out, int capacity, double* sim_us)
{
    if (nblk <= 0) return -1;
    FlowParams prm = flow_params_from_env();
    if (np_max > 0) prm.np_max = std::min(8, np_max);
    if (slots > 0) prm.slots = std::max(32, slots);      // fewer than 16 can never hold the 16 parts of TRSM32
    std::vector<int> lv;
    if (last) lv.assign(last, last + nblk);
    FlowSchedule sc;
    if (flow_cached_schedule(nblk, lv, prm, sc) != 0) return -1;        // the product's path: built and checked once per process and shape
    if (sim_us) *sim_us = sc.sim_us;
    const int nt = (int)sc.tasks.size();
    if (tasks_out) memcpy(tasks_out, sc.tasks.data(), (size_t)std::min(nt, std::max(0, capacity)) * sizeof(FlowTask));
    return nt;
}

int bsfm_chol_dyn_plan(int nblk, const int* last, void* chain_out, int chain_cap, void* potrf_out, int potrf_cap, unsigned* init_out, int init_cap, int* meta)
{
    std::vector<int> lv;
    if (last && nblk > 0) lv.assign(last, last + nblk);
    DynPlan p;
    if (dyn_build_plan(nblk, lv, p) != 0) return -1;
    if (chain_out) memcpy(chain_out, p.chain.data(), (size_t)std::min<long long>((long long)p.chain.size(), std::max(0, chain_cap)) * sizeof(FlowTask));
    if (potrf_out) memcpy(potrf_out, p.potrf.data(), (size_t)std::min<long long>((long long)p.potrf.size(), std::max(0, potrf_cap)) * sizeof(FlowTask));
    if (init_out) memcpy(init_out, p.init.data(), (size_t)std::min<long long>((long long)p.init.size(), std::max(0, init_cap)) * sizeof(unsigned));
    if (meta) {
        meta[0] = (int)p.chain.size(); meta[1] = (int)p.potrf.size(); meta[2] = (int)p.nwords; meta[3] = (int)p.ofs_c32; meta[4] = (int)p.ofs_c10;
        meta[5] = (int)p.ofs_wd; meta[6] = (int)p.ofs_rd; meta[7] = (int)p.ofs_tw;
    }
    return 0;
}

static int dense_chol_solve_impl(int n, const double* A, const double* b, double* x, int backend, int reps_in, double* ms_out, double* flow_ms_out, double* flow_gflop_out,
                                 bsfm_comm_t* comm = nullptr);

int bsfm_dense_chol_solve(int n, const double* A, const double* b, double* x, int backend)
{
    return dense_chol_solve_impl(n, A, b, x, backend, 0, nullptr, nullptr, nullptr);
}

// The same solve by the ranks of a communicator TOGETHER (test entry of the distributed factorisation, chol_flow.hip.h: FlowDist): every rank calls it
// with the same A and b and receives the same x -- bit-identical to bsfm_dense_chol_solve's -- and the same return value (0, dpotrf's k, BSFM_ERROR).
// COLLECTIVE.  Transports: "ipc", "loopback".  backend 0 = dense task list, 2 = tile envelope of A.
int bsfm_dense_chol_solve_dist(bsfm_comm_t* comm, int n, const double* A, const double* b, double* x, int backend)
{
    if (!comm || bsfm_comm_world(comm) < 2) return dense_chol_solve_impl(n, A, b, x, backend, 0, nullptr, nullptr, nullptr);
    return dense_chol_solve_impl(n, A, b, x, backend, 0, nullptr, nullptr, nullptr, comm);
}

// The same solve `reps` times with the device time of every repetition (ms_out[reps]: factorisation + both substitutions, HIP events
// on the solve's stream), the mean HIP-event time of the k_chol_flow launches and the flops the library scheduled for one launch --
// bench.py's `dense_valued_S` leg: the headline task list on a matrix whose tiles all hold numbers (missing #5).
int bsfm_dense_chol_solve_timed(int n, const double* A, const double* b, double* x, int backend, int reps, double* ms_out, double* flow_ms_out,
                                double* flow_gflop_out)
{
    return dense_chol_solve_impl(n, A, b, x, backend, std::max(1, reps), ms_out, flow_ms_out, flow_gflop_out);
}

// ranks' dpotrf words -> the system's: a time-out anywhere is a time-out; else the smallest positive k; collective
static int combine_potrf_info(bsfm_comm_t* comm, int info)
{
    double v[2] = { info > 0 ? 1e9 - (double)info : 0.0, info < 0 ? 1.0 : 0.0 };
    if (bsfm_comm_allreduce_host(comm, v, 2, 1) != 0) return POTRF_INFO_TIMEOUT;
    if (v[1] != 0.0) return POTRF_INFO_TIMEOUT;
    return v[0] > 0.0 ? (int)(1e9 - v[0] + 0.5) : 0;
}

static int dense_chol_solve_impl(int n, const double* A, const double* b, double* x, int backend, int reps_in, double* ms_out, double* flow_ms_out, double* flow_gflop_out,
                                 bsfm_comm_t* comm)
{
    if (bsfm_device_count() <= 0) { fprintf(stderr, "[bsfm] FATAL: no usable HIP device\n"); return BSFM_ERROR; }
    if (n <= 0) return BSFM_ERROR;
    const int ld = std::max(POTRF_NB, (n + POTRF_NB - 1) / POTRF_NB * POTRF_NB);
    const bool envelope = backend == 2;          // 2: the tiled factorisation restricted to the tile envelope of A (test entry)
    if (envelope) backend = 0;
    PotrfWorkspace ws;
    if (potrf_init(ws, ld, backend)) return BSFM_ERROR;
    if (comm && n > POTRF_NB && backend == 0) ws.dist_comm = comm;
    double *dS = nullptr, *dE = nullptr, *dx = nullptr; int* dinfo = nullptr;
    int rc = BSFM_ERROR, info = 0;
    hipStream_t st = nullptr;
    do {
        if (hipStreamCreate(&st) != hipSuccess) break;
        if (dmalloc(&dS, (size_t)ld * ld) != hipSuccess || dmalloc(&dE, ld) != hipSuccess || dmalloc(&dx, ld) != hipSuccess || dmalloc(&dinfo, 1) != hipSuccess) break;
        if (hipMemset(dS, 0, (size_t)ld * ld * sizeof(double)) != hipSuccess || hipMemset(dE, 0, ld * sizeof(double)) != hipSuccess || hipMemset(dinfo, 0, sizeof(int)) != hipSuccess) break;
        if (hipMemcpy2D(dS, (size_t)ld * sizeof(double), A, (size_t)n * sizeof(double), (size_t)n * sizeof(double), n, hipMemcpyHostToDevice) != hipSuccess) break;
        if (hipMemcpy(dE, b, n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) break;
        if (envelope) {      // tile envelope of A's own zero pattern (lower triangle), no reordering: the test entry of the envelope solver
            const int nt = ld / POTRF_NB;
            std::vector<int> first((size_t)nt), last((size_t)nt);
            for (int I = 0; I < nt; ++I) { first[I] = I; last[I] = I; }
            for (int r = 0; r < n; ++r)
                for (int c = 0; c <= r; ++c)
                    if (A[(size_t)r * n + c] != 0.0) { first[r / POTRF_NB] = std::min(first[r / POTRF_NB], c / POTRF_NB); break; }
            for (int I = 0; I < nt; ++I) for (int K = first[I]; K <= I; ++K) last[K] = std::max(last[K], I);
            ws.env_rows.assign((size_t)nt, 0);
            for (int K = 0; K < nt; ++K) ws.env_rows[K] = last[K] - K;
            if (bsfm::dev_alloc((void**)&ws.d_last, (size_t)nt * sizeof(int)) != hipSuccess) break;
            if (hipMemcpy(ws.d_last, last.data(), (size_t)nt * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) break;
        }
        // BSFM_CHOL_REPS=n (diagnostics): solve n times (S is destroyed by a solve and uploaded again) and print the device time of each
        int reps = 1;
        if (const char* e = getenv("BSFM_CHOL_REPS")) reps = std::max(1, atoi(e));
        if (reps_in > 0) reps = reps_in;
        const bool timed_api = reps_in > 0;
        bool failed = false;
        for (int rep = 0; rep < reps && !failed; ++rep) {
            if (rep > 0) {
                if (hipMemset(dinfo, 0, sizeof(int)) != hipSuccess) { failed = true; break; }
                if (hipMemcpy2D(dS, (size_t)ld * sizeof(double), A, (size_t)n * sizeof(double), (size_t)n * sizeof(double), n, hipMemcpyHostToDevice) != hipSuccess) { failed = true; break; }
            }
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (reps > 1 || timed_api) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0, st); }
            if (potrf_solve(ws, dS, ld, n, dE, dx, dinfo, st)) { failed = true; break; }
            if (e1) (void)hipEventRecord(e1, st);
            if (hipStreamSynchronize(st) != hipSuccess) { failed = true; break; }
            if (e0 && e1) {
                float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
                if (timed_api) { if (ms_out) ms_out[rep] = ms; }
                else fprintf(stderr, "[bsfm] dense_chol_solve n = %d rep %d: %.3f ms\n", n, rep, ms);
                (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
            }
        }
        if (failed) break;
        if (hipMemcpy(&info, dinfo, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) break;
        if (ws.dist_comm) {      // every rank holds the verdict of its own tile columns: combine (and fall back TOGETHER on a time-out)
            info = combine_potrf_info(ws.dist_comm, info);
            if (hipMemcpy(dinfo, &info, sizeof(int), hipMemcpyHostToDevice) != hipSuccess) break;
            if (info == POTRF_INFO_TIMEOUT) ws.dist_comm = nullptr;      // the repetition below is every rank's own (replicated) solve
        }
        if (info == POTRF_INFO_TIMEOUT && ws.use_flow && backend == 0 && ld / POTRF_NB <= POTRF_MAX_TILES) {
            // starved dataflow launch: the same system once more on the stream-ordered schedule (see bsfm_lm_iterate)
            fprintf(stderr, "[bsfm] WARNING: the tile-dataflow Cholesky launch timed out waiting for its own workgroups; repeating the solve on the "
                            "stream-ordered schedule\n");
            ws.use_flow = 0;
            if (hipMemset(dinfo, 0, sizeof(int)) != hipSuccess) break;
            if (hipMemcpy2D(dS, (size_t)ld * sizeof(double), A, (size_t)n * sizeof(double), (size_t)n * sizeof(double), n, hipMemcpyHostToDevice) != hipSuccess) break;
            if (potrf_solve(ws, dS, ld, n, dE, dx, dinfo, st)) break;
            if (hipStreamSynchronize(st) != hipSuccess) break;
        }
        if (timed_api) {
            potrf_collect_time(ws);
            if (ws.flow) flow_collect_time(*ws.flow);
            if (flow_ms_out) *flow_ms_out = ws.flow && ws.flow->kern_cnt ? ws.flow->kern_ms / (double)ws.flow->kern_cnt : -1.0;
            if (flow_gflop_out) *flow_gflop_out = ws.flow && ws.flow->kern_cnt ? (ws.flow->flops + (double)ws.flow->nblk * POTRF_NB * POTRF_NB * POTRF_NB) * 1e-9 : -1.0;
        }
        if (hipMemcpy(&info, dinfo, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) break;
        if (hipMemcpy(x, dx, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) break;
        rc = info;
    } while (0);
    (void)hipDeviceSynchronize();
    bsfm::dev_free(dS, true); bsfm::dev_free(dE, true); bsfm::dev_free(dx, true); bsfm::dev_free(dinfo, true);
    if (st) (void)hipStreamDestroy(st);
    potrf_free(ws);
    return rc;
}

}  // extern "C"
