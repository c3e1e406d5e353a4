// index_build.hip -- index bookkeeping of the bundle-adjustment core, built ON THE DEVICE (integer work, bit-exact).
//
// What it replaces in the reference (paths relative to the reference tree):
//   * struct sba_crsm + the fill loop                        lib/sba-1.5/sba_levmar.c:653-663, lib/sba-1.5/sba.h:70-78
//   * sba_crsm_col_elmidxs (camera-major traversal, re-derived by binary search for every camera in every U_j / Q / Jacobian
//     loop)                                                  lib/sba-1.5/sba_crsm.c:183-212
//   * the co-visibility search inside the Schur loop (range test + sba_crsm_elmidxp per camera pair and point)
//                                                            lib/sba-1.5/sba_levmar.c:1218-1268
// The observation order is the reference's contract (k-th set bit of vmask in row-major order = k-th measurement), so every
// array here is a pure function of (rowptr, colidx): camera-major order = stable sort of the observations by camera (points
// ascending inside a camera, exactly the order sba_crsm_col_elmidxs returns them); Schur triples = all pairs (a <= b) of the
// free cameras of a point, stable-sorted by block (j, k) so that inside a block they stay in point order -- the order the
// reference accumulates Y_ij W_ik^T in.  Round 1 built all of this with single-threaded host loops (0.4 - 7.4 s at
// 1 000 cameras / 5 M observations, paid by every run_sfm call); here it is a handful of kernels plus rocPRIM's stable radix
// sort / scan / run-length primitives (through the hipcub headers), a few milliseconds.
#include <hip/hip_runtime.h>
#include "prim.hip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <climits>
#include <algorithm>
#include <mutex>
#include <chrono>
#include "index_build.h"
#include "devcache.h"

namespace bsfm {
namespace {

#define IX_OK(call)                                                                                     \
    do { hipError_t _e = (call); if (_e != hipSuccess) {                                               \
        fprintf(stderr, "[bsfm] index build: HIP error %s at %s:%d\n", hipGetErrorName(_e), __FILE__, __LINE__); \
        return -1; } } while (0)

inline int grid_for(size_t count, int block) { return (int)std::max<size_t>(1, (count + block - 1) / block); }
inline int bits_for(unsigned long long maxval) { int b = 1; while (b < 64 && (maxval >> b)) ++b; return b; }

// flag bit 0: rowptr not monotone / out of range; bit 1: colidx out of range; bit 2: colidx not strictly ascending in a row;
// bit 3 (not an error): a row is empty
__global__ void k_validate_rows(int n, int m, int nvis, const int* __restrict__ rowptr, const int* __restrict__ colidx, int* __restrict__ flag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r0 = rowptr[i], r1 = rowptr[i + 1];
    if (r0 < 0 || r1 < r0 || r1 > nvis) { atomicOr(flag, 1); return; }
    if (r1 == r0) atomicOr(flag, 8);
    int prev = -1;
    for (int k = r0; k < r1; ++k) {
        const int c = colidx[k];
        if (c < 0 || c >= m) { atomicOr(flag, 2); return; }
        if (c <= prev) { atomicOr(flag, 4); return; }
        prev = c;
    }
}

// obs_pt, and the number of co-visibility triples of every point: c (c + 1) / 2 with c = cameras >= mcon of the point
// (colidx ascends inside a row, so the free cameras are a suffix)
__global__ void k_rows(int n, int mcon, const int* __restrict__ rowptr, const int* __restrict__ colidx, int* __restrict__ obs_pt,
                       long long* __restrict__ tcount)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r0 = rowptr[i], r1 = rowptr[i + 1];
    int c = 0;
    for (int k = r0; k < r1; ++k) { obs_pt[k] = i; c += colidx[k] >= mcon; }
    if (tcount) tcount[i] = (long long)c * (c + 1) / 2;
}

__global__ void k_iota(int cnt, int* __restrict__ v)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < cnt) v[k] = k;
}

// camptr[j] = first camera-major position whose camera is >= j (lower bound in the sorted camera array)
__global__ void k_camptr(int m, int nvis, const int* __restrict__ cam_sorted, int* __restrict__ camptr)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > m) return;
    int lo = 0, hi = nvis;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (cam_sorted[mid] < j) lo = mid + 1; else hi = mid; }
    camptr[j] = lo;
}

__global__ void k_cam_maps(int nvis, const int* __restrict__ camobs, const int* __restrict__ obs_pt, int* __restrict__ campos,
                           int* __restrict__ cam_pt)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nvis) return;
    const int k = camobs[t];
    campos[k] = t;
    cam_pt[t] = obs_pt[k];
}

// one thread per point: its triples in (a, b >= a) order at offset toff[i]; key = block (j - mcon) * mm + (k - mcon),
// value = (camera-major position of (i, j), of (i, k)) packed low / high
template <typename KeyT>
__global__ void k_gen_triples(int n, int mcon, int mm, const int* __restrict__ rowptr, const int* __restrict__ colidx,
                              const int* __restrict__ campos, const long long* __restrict__ toff, KeyT* __restrict__ keys,
                              unsigned long long* __restrict__ vals)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r1 = rowptr[i + 1];
    int r0 = rowptr[i];
    while (r0 < r1 && colidx[r0] < mcon) ++r0;
    long long o = toff[i];
    for (int a = r0; a < r1; ++a) {
        const unsigned long long ja = (unsigned long long)(colidx[a] - mcon) * (unsigned long long)mm;
        const unsigned pa = (unsigned)campos[a];
        for (int b = a; b < r1; ++b, ++o) {
            keys[o] = (KeyT)(ja + (unsigned long long)(colidx[b] - mcon));
            vals[o] = ((unsigned long long)(unsigned)campos[b] << 32) | pa;
        }
    }
}

template <typename KeyT>
__global__ void k_blocks(int nblk, int mcon, int mm, const KeyT* __restrict__ ukeys, const int* __restrict__ counts,
                         int* __restrict__ blk_j, int* __restrict__ blk_k, int* __restrict__ ntask, int chunk)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblk) return;
    const unsigned long long key = (unsigned long long)ukeys[b];
    blk_j[b] = mcon + (int)(key / (unsigned long long)mm);
    blk_k[b] = mcon + (int)(key % (unsigned long long)mm);
    ntask[b] = (counts[b] + chunk - 1) / chunk;
}

// tasks in block order; blk_start / blk_task0 are the exclusive scans of the triple and task counts (entry nblk = totals)
__global__ void k_tasks(int nblk, const int* __restrict__ blk_start, const int* __restrict__ blk_task0, const int* __restrict__ blk_j,
                        const int* __restrict__ blk_k, SchurTask* __restrict__ tasks, int chunk)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblk) return;
    const int s0 = blk_start[b], s1 = blk_start[b + 1], diag = blk_j[b] == blk_k[b] ? 1 : 0;
    int t = blk_task0[b];
    for (int s = s0; s < s1; s += chunk, ++t) {
        SchurTask tk; tk.start = s; tk.count = min(chunk, s1 - s); tk.diag = diag; tk.out = t;
        tasks[t] = tk;
    }
}

__global__ void k_tri_pt(int ntri, const unsigned long long* __restrict__ vals, const int* __restrict__ cam_pt, int* __restrict__ tri_pt)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < ntri) tri_pt[s] = cam_pt[(unsigned)(vals[s] & 0xffffffffULL)];
}

__global__ void k_task_keys(int ntasks, const SchurTask* __restrict__ tasks, const int* __restrict__ tri_pt, int* __restrict__ keys)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < ntasks) keys[t] = tri_pt[tasks[t].start];
}

// Key of the clustered launch order: (slice of the point range the task starts in, the two cameras in breadth-first numbering).
__global__ void k_task_keys_clustered(int nblk, int mcon, const int* __restrict__ blk_task0, const int* __restrict__ blk_j, const int* __restrict__ blk_k,
                                      const int* __restrict__ rank, const SchurTask* __restrict__ tasks, const int* __restrict__ tri_pt,
                                      int n, int slices, unsigned long long* __restrict__ keys)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblk) return;
    const unsigned ra = (unsigned)rank[blk_j[b] - mcon], rb = (unsigned)rank[blk_k[b] - mcon];
    const unsigned long long cams = ((unsigned long long)min(ra, rb) << 24) | (unsigned long long)max(ra, rb);
    for (int t = blk_task0[b]; t < blk_task0[b + 1]; ++t) {
        const unsigned long long slice = (unsigned long long)((long long)tri_pt[tasks[t].start] * slices / max(n, 1));
        keys[t] = (slice << 48) | cams;
    }
}

// Launch order (schur.hip.h): tasks sorted by the first point they touch (`ord`), then every XCD (workgroup index % 8, four
// tasks per workgroup) gets ONE contiguous stretch of that order: the workgroups wg = x, x + 8, x + 16 ... take consecutive
// four-task pieces.  Slots past the end are padding (out = -1).
__global__ void k_launch_order(int ntasks, int nwg, const SchurTask* __restrict__ tasks, const int* __restrict__ ord,
                               SchurTask* __restrict__ launch)
{
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= nwg * 4) return;
    const int wg = slot >> 2, w = slot & 3, x = wg & 7;
    int next = wg >> 3;
    for (int r = 0; r < x; ++r) next += nwg > r ? (nwg - r + 7) / 8 : 0;
    const long long src = (long long)next * 4 + w;
    SchurTask tk; tk.start = 0; tk.count = 0; tk.diag = 0; tk.out = -1;
    if (src < ntasks) tk = tasks[ord[src]];
    launch[slot] = tk;
}

// slots k_schur_assemble / k_schur_pack add for block b: its tasks', in task order
__global__ void k_blk_ranges(int nblk, const int* __restrict__ blk_task0, int2* __restrict__ range)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblk) return;
    range[b] = make_int2(blk_task0[b], blk_task0[b + 1]);
}

// Temporaries of one build come from the device's stream-ordered memory pool (hipMallocAsync): the pool keeps what it is given
// back (release threshold = unlimited, set once), so the ~0.9 GB of sort buffers cost an allocation only on the FIRST run_sfm
// call of a process -- an incremental reconstruction calls run_sfm hundreds of times.  Freed on every exit path.
// Temporaries come from a PRIVATE stream-ordered pool per device (release threshold unlimited, so the ~0.9 GB of sort buffers of one
// problem_create are still there for the next run_sfm call).  Round 2 raised the threshold of the process-wide DEFAULT pool instead,
// which changed the host application's own allocation behaviour; the default pool is no longer touched.
static hipMemPool_t g_scratch_pools[64] = {};
inline hipMemPool_t scratch_pool()
{
    static std::once_flag once[64];
    int dev = 0; (void)hipGetDevice(&dev);
    const int slot = dev & 63;
    std::call_once(once[slot], [dev, slot] {
        hipMemPoolProps props;
        memset(&props, 0, sizeof(props));
        props.allocType = hipMemAllocationTypePinned;
        props.handleTypes = hipMemHandleTypeNone;
        props.location.type = hipMemLocationTypeDevice;
        props.location.id = dev;
        hipMemPool_t pool = nullptr;
        if (hipMemPoolCreate(&pool, &props) == hipSuccess && pool) {
            unsigned long long thr = ~0ULL;
            (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
            g_scratch_pools[slot] = pool;
        } else (void)hipGetLastError();          // no private pool: plain hipMallocAsync on the default pool, untouched
    });
    return g_scratch_pools[slot];
}

struct Scratch {
    hipStream_t st = nullptr;
    hipMemPool_t pool = nullptr;
    std::vector<void*> ptrs;
    explicit Scratch(hipStream_t s) : st(s), pool(scratch_pool()) {}
    ~Scratch() { for (void* p : ptrs) if (p) (void)hipFreeAsync(p, st); }
    template <typename T> hipError_t alloc(T** p, size_t count)
    {
        const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
        hipError_t e = pool ? hipMallocFromPoolAsync(reinterpret_cast<void**>(p), bytes, pool, st) : hipMallocAsync(reinterpret_cast<void**>(p), bytes, st);
        if (e == hipSuccess) ptrs.push_back(*p);
        return e;
    }
};

// destroys its events on every exit path (the early returns of build_index_device leaked them)
struct EventPair {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    EventPair() { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); }
    ~EventPair() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
};

// what a build hands to its caller comes from the problems' block cache (devcache.h): the caller releases it with dev_free
template <typename T> hipError_t keep(T** p, size_t count) { return bsfm::dev_alloc(reinterpret_cast<void**>(p), std::max<size_t>(count, 1) * sizeof(T)); }

// breadth-first numbering of the free cameras in the co-visibility graph (components contiguous, neighbours close)
void bfs_rank(int mm2, int mcon, const std::vector<int>& bj, const std::vector<int>& bk, std::vector<int>& rank)
{
    const int nblk = (int)bj.size();
    rank.assign((size_t)mm2, 0);
    std::vector<int> adj_ptr((size_t)mm2 + 1, 0);
    for (int b = 0; b < nblk; ++b) { const int a = bj[b] - mcon, c = bk[b] - mcon; if (a != c) { ++adj_ptr[a + 1]; ++adj_ptr[c + 1]; } }
    for (int j = 0; j < mm2; ++j) adj_ptr[j + 1] += adj_ptr[j];
    std::vector<int> adj((size_t)adj_ptr[mm2]), fill(adj_ptr.begin(), adj_ptr.end() - 1);
    for (int b = 0; b < nblk; ++b) { const int a = bj[b] - mcon, c = bk[b] - mcon; if (a != c) { adj[fill[a]++] = c; adj[fill[c]++] = a; } }
    std::vector<char> seen((size_t)mm2, 0);
    std::vector<int> q; q.reserve((size_t)mm2);
    for (int s0 = 0; s0 < mm2; ++s0) {
        if (seen[s0]) continue;
        seen[s0] = 1; q.push_back(s0);
        for (size_t h = q.size() - 1; h < q.size(); ++h)
            for (int e = adj_ptr[q[h]]; e < adj_ptr[q[h] + 1]; ++e) if (!seen[adj[e]]) { seen[adj[e]] = 1; q.push_back(adj[e]); }
    }
    for (int p = 0; p < mm2; ++p) rank[q[p]] = p;
}

template <typename KeyT>
int build_schur(int n, int m, int mcon, int nvis, const int* d_rowptr, const int* d_colidx, const long long* d_toff, long long total,
                int order_mode, DeviceIndex& ix, hipStream_t st)
{
    const int mm = m - mcon;
    std::vector<int> rank;                  // breadth-first numbering of the free cameras (clustered launch order, row plan)
    // a problem whose Jacobian records (272 bytes per observation) fit one XCD's L2 needs no launch order at all: block order, no task sort
    if (order_mode == SCHUR_ORDER_CLUSTERED && (size_t)nvis * 272u <= (size_t)(4u << 20)) order_mode = SCHUR_ORDER_BLOCK;
    Scratch tmp(st);
    const size_t nt = (size_t)total;
    ix.ntriples = (int)total;
    IX_OK(keep(&ix.triples, nt)); IX_OK(keep(&ix.tri_pt, nt));
    if (total == 0) {
        ix.ntasks = ix.nblk = ix.nslots = 0;
        IX_OK(keep(&ix.tasks, 1)); IX_OK(keep(&ix.blk_j, 1)); IX_OK(keep(&ix.blk_k, 1)); IX_OK(keep(&ix.blk_task0, 1));
        IX_OK(keep(&ix.blk_range, 1));
        IX_OK(hipMemsetAsync(ix.blk_task0, 0, sizeof(int), st));
        ix.tasks_launch = ix.tasks;
        return 0;
    }
    KeyT *keys_in = nullptr, *keys_out = nullptr; unsigned long long* vals_in = nullptr;
    IX_OK(tmp.alloc(&keys_in, nt)); IX_OK(tmp.alloc(&keys_out, nt)); IX_OK(tmp.alloc(&vals_in, nt));
    hipLaunchKernelGGL((k_gen_triples<KeyT>), dim3(grid_for(n, 128)), dim3(128), 0, st, n, mcon, mm, d_rowptr, d_colidx, ix.campos,
                       d_toff, keys_in, vals_in);
    // stable sort by block key; the values land directly in the triple array (int2 = low / high word)
    unsigned long long* vals_out = reinterpret_cast<unsigned long long*>(ix.triples);
    const int kbits = bits_for((unsigned long long)mm * (unsigned long long)mm - 1ULL);
    size_t tb = 0;
    IX_OK(prim::sort_pairs(nullptr, tb, keys_in, keys_out, vals_in, vals_out, (int)nt, 0, kbits, st));
    void* d_tmp = nullptr;
    IX_OK(tmp.alloc(reinterpret_cast<char**>(&d_tmp), tb));
    IX_OK(prim::sort_pairs(d_tmp, tb, keys_in, keys_out, vals_in, vals_out, (int)nt, 0, kbits, st));
    // blocks = runs of equal keys
    KeyT* ukeys = keys_in;                     // reuse: the unsorted keys are no longer needed
    int *counts = nullptr, *nruns = nullptr;
    IX_OK(tmp.alloc(&counts, nt + 1)); IX_OK(tmp.alloc(&nruns, 1));
    size_t rb = 0;
    IX_OK(prim::run_length_encode(nullptr, rb, keys_out, ukeys, counts, nruns, (int)nt, st));
    void* d_rle = nullptr;
    IX_OK(tmp.alloc(reinterpret_cast<char**>(&d_rle), rb));
    IX_OK(prim::run_length_encode(d_rle, rb, keys_out, ukeys, counts, nruns, (int)nt, st));
    int nblk = 0;
    IX_OK(hipMemcpyAsync(&nblk, nruns, sizeof(int), hipMemcpyDeviceToHost, st));
    IX_OK(hipStreamSynchronize(st));
    ix.nblk = nblk;
    IX_OK(keep(&ix.blk_j, (size_t)nblk)); IX_OK(keep(&ix.blk_k, (size_t)nblk)); IX_OK(keep(&ix.blk_task0, (size_t)nblk + 1));
    int *ntask = nullptr, *blk_start = nullptr;
    IX_OK(tmp.alloc(&ntask, (size_t)nblk + 1)); IX_OK(tmp.alloc(&blk_start, (size_t)nblk + 1));
    IX_OK(hipMemsetAsync(ntask + nblk, 0, sizeof(int), st));
    IX_OK(hipMemsetAsync(counts + nblk, 0, sizeof(int), st));          // counts has nt + 1 >= nblk + 1 entries
    hipLaunchKernelGGL((k_blocks<KeyT>), dim3(grid_for(nblk, 256)), dim3(256), 0, st, nblk, mcon, mm, ukeys, counts, ix.blk_j, ix.blk_k, ntask, schur_chunk());
    size_t sb = 0;
    IX_OK(prim::exclusive_sum(nullptr, sb, ntask, ix.blk_task0, nblk + 1, st));
    void* d_scan = nullptr;
    IX_OK(tmp.alloc(reinterpret_cast<char**>(&d_scan), sb));
    IX_OK(prim::exclusive_sum(d_scan, sb, ntask, ix.blk_task0, nblk + 1, st));
    IX_OK(prim::exclusive_sum(d_scan, sb, counts, blk_start, nblk + 1, st));
    int ntasks = 0;
    IX_OK(hipMemcpyAsync(&ntasks, ix.blk_task0 + nblk, sizeof(int), hipMemcpyDeviceToHost, st));
    IX_OK(hipStreamSynchronize(st));
    ix.ntasks = ntasks;
    SchurTask* tasks = nullptr;
    if (order_mode == SCHUR_ORDER_BLOCK) {
        IX_OK(bsfm::dev_alloc((void**)&tasks, std::max<size_t>(1, (size_t)ntasks) * sizeof(SchurTask)));
        ix.tasks = tasks; ix.nslots = ntasks;          // owned by ix from here on: free_index_device releases it on every error path
    } else IX_OK(tmp.alloc(&tasks, (size_t)ntasks));
    hipLaunchKernelGGL(k_tasks, dim3(grid_for(nblk, 256)), dim3(256), 0, st, nblk, blk_start, ix.blk_task0, ix.blk_j, ix.blk_k, tasks, schur_chunk());
    hipLaunchKernelGGL(k_tri_pt, dim3(grid_for(nt, 256)), dim3(256), 0, st, (int)nt, vals_out, ix.cam_pt, ix.tri_pt);
    ix.h_blk_j.resize((size_t)nblk); ix.h_blk_k.resize((size_t)nblk);
    std::vector<int> h_ntask((size_t)nblk);
    if (nblk) {
        IX_OK(hipMemcpyAsync(ix.h_blk_j.data(), ix.blk_j, (size_t)nblk * sizeof(int), hipMemcpyDeviceToHost, st));
        IX_OK(hipMemcpyAsync(ix.h_blk_k.data(), ix.blk_k, (size_t)nblk * sizeof(int), hipMemcpyDeviceToHost, st));
        if (order_mode == SCHUR_ORDER_CLUSTERED) IX_OK(hipMemcpyAsync(h_ntask.data(), ntask, (size_t)nblk * sizeof(int), hipMemcpyDeviceToHost, st));
    }
    if (order_mode != SCHUR_ORDER_BLOCK) {
        int *id_in = nullptr, *ord = nullptr;
        IX_OK(tmp.alloc(&id_in, (size_t)ntasks)); IX_OK(tmp.alloc(&ord, (size_t)ntasks));
        hipLaunchKernelGGL(k_iota, dim3(grid_for(ntasks, 256)), dim3(256), 0, st, ntasks, id_in);
        if (order_mode == SCHUR_ORDER_POINT) {
            int *tk_in = nullptr, *tk_out = nullptr;
            IX_OK(tmp.alloc(&tk_in, (size_t)ntasks)); IX_OK(tmp.alloc(&tk_out, (size_t)ntasks));
            hipLaunchKernelGGL(k_task_keys, dim3(grid_for(ntasks, 256)), dim3(256), 0, st, ntasks, tasks, ix.tri_pt, tk_in);
            size_t ob = 0;
            const int pbits = bits_for((unsigned long long)std::max(n, 1) - 1ULL);
            IX_OK(prim::sort_pairs(nullptr, ob, tk_in, tk_out, id_in, ord, ntasks, 0, pbits, st));
            void* d_ob = nullptr;
            IX_OK(tmp.alloc(reinterpret_cast<char**>(&d_ob), ob));
            IX_OK(prim::sort_pairs(d_ob, ob, tk_in, tk_out, id_in, ord, ntasks, 0, pbits, st));
        } else {
            // Clustered order.  Two tasks read the same Jacobian records iff they share a camera AND their point ranges overlap, and what
            // is in flight on one XCD (~500 tasks, ~100 KB of records each) has to overlap enough to fit its 4 MB of L2.  Sorting by the
            // first point alone does that only when all blocks of a camera group cut their point lists at the same places (the generator's
            // cliques); on a connected scene the first point of a chunk is noise, neighbours in the order are unrelated blocks, the L2
            // hit rate is 9 % and the kernel fetches 17 GB per launch (profiles/r04_schur_connected_counters.txt).  So: the point range
            // in as many SLICES as the largest block has tasks; within a slice the cameras in breadth-first numbering of the
            // co-visibility graph (components contiguous, neighbours close) -- (slice, lower camera, higher camera).
            IX_OK(hipStreamSynchronize(st));
            const int mm2 = m - mcon;
            bfs_rank(mm2, mcon, ix.h_blk_j, ix.h_blk_k, rank);
            int slices = 1;
            for (int b = 0; b < nblk; ++b) slices = std::max(slices, h_ntask[b]);
            slices = std::min(slices, 4096);
            int* d_rank = nullptr;
            IX_OK(tmp.alloc(&d_rank, (size_t)mm2));
            IX_OK(hipMemcpyAsync(d_rank, rank.data(), (size_t)mm2 * sizeof(int), hipMemcpyHostToDevice, st));
            unsigned long long *ck_in = nullptr, *ck_out = nullptr;
            IX_OK(tmp.alloc(&ck_in, (size_t)ntasks)); IX_OK(tmp.alloc(&ck_out, (size_t)ntasks));
            hipLaunchKernelGGL(k_task_keys_clustered, dim3(grid_for(nblk, 256)), dim3(256), 0, st, nblk, mcon, ix.blk_task0, ix.blk_j, ix.blk_k, d_rank,
                               tasks, ix.tri_pt, n, slices, ck_in);
            size_t ob = 0;
            const int kb2 = 48 + bits_for((unsigned long long)slices);
            IX_OK(prim::sort_pairs(nullptr, ob, ck_in, ck_out, id_in, ord, ntasks, 0, kb2, st));
            void* d_ob = nullptr;
            IX_OK(tmp.alloc(reinterpret_cast<char**>(&d_ob), ob));
            IX_OK(prim::sort_pairs(d_ob, ob, ck_in, ck_out, id_in, ord, ntasks, 0, kb2, st));
            IX_OK(hipStreamSynchronize(st));          // `rank` (host) was the source of an asynchronous copy
        }
        const int nwg = (ntasks + 3) / 4;
        ix.nslots = nwg * 4;
        IX_OK(keep(&ix.tasks, (size_t)ix.nslots));
        hipLaunchKernelGGL(k_launch_order, dim3(grid_for((size_t)ix.nslots, 256)), dim3(256), 0, st, ntasks, nwg, tasks, ord, ix.tasks);
    }
    IX_OK(hipStreamSynchronize(st));
    // slot ranges per block; the task kernel is given the whole launch list (round 5's row kernel for dense blocks -- measured slower on both
    // scenes, profiles/r05_schur_kernels.txt -- was removed in round 6)
    ix.tasks_launch = ix.tasks;
    IX_OK(keep(&ix.blk_range, (size_t)nblk));
    hipLaunchKernelGGL(k_blk_ranges, dim3(grid_for((size_t)nblk, 256)), dim3(256), 0, st, nblk, ix.blk_task0, ix.blk_range);
    IX_OK(hipStreamSynchronize(st));          // temporaries are freed when `tmp` goes out of scope
    return 0;
}

}  // namespace

// bsfm_device_cache_trim: the private scratch pools give their pages back too (every device that built an index)
void index_pool_trim()
{
    for (hipMemPool_t pool : g_scratch_pools) if (pool) (void)hipMemPoolTrimTo(pool, 0);
}

namespace {

__global__ void k_merge_keys(int nvis, int nadd, int n_new, int m_new, const int* __restrict__ obs_pt, const int* __restrict__ colidx,
                             const int* __restrict__ add_pt, const int* __restrict__ add_cam, unsigned long long* __restrict__ keys,
                             int* __restrict__ vals, int* __restrict__ flag)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nvis + nadd) return;
    int pt, cam;
    if (t < nvis) { pt = obs_pt[t]; cam = colidx[t]; }
    else { pt = add_pt[t - nvis]; cam = add_cam[t - nvis]; if (pt < 0 || pt >= n_new || cam < 0 || cam >= m_new) { atomicOr(flag, 1); pt = 0; cam = 0; } }
    keys[t] = (unsigned long long)pt * (unsigned long long)m_new + (unsigned long long)cam;
    vals[t] = t;
}

__global__ void k_merge_scatter(int total, int nvis, int m_new, const unsigned long long* __restrict__ keys, const int* __restrict__ src,
                                const double* __restrict__ x, const double* __restrict__ add_xy, int* __restrict__ colidx_out,
                                double* __restrict__ x_out, int* __restrict__ flag)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const unsigned long long key = keys[t];
    if (t > 0 && keys[t - 1] == key) atomicOr(flag, 2);                     // the same (point, camera) twice
    colidx_out[t] = (int)(key % (unsigned long long)m_new);
    const int s = src[t];
    const double* xs = s < nvis ? x + 2 * (size_t)s : add_xy + 2 * (size_t)(s - nvis);
    x_out[2 * (size_t)t] = xs[0]; x_out[2 * (size_t)t + 1] = xs[1];
}

__global__ void k_merge_rowptr(int n_new, int m_new, int total, const unsigned long long* __restrict__ keys, int* __restrict__ rowptr)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_new) return;
    const unsigned long long target = (unsigned long long)i * (unsigned long long)m_new;
    int lo = 0, hi = total;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] < target) lo = mid + 1; else hi = mid; }
    rowptr[i] = lo;
}

}  // namespace

int merge_observations_device(int n_new, int m_new, int nvis, const int* d_obs_pt, const int* d_colidx, const double* d_x,
                              int nadd, const int* d_add_pt, const int* d_add_cam, const double* d_add_xy,
                              int** rowptr_out, int** colidx_out, double** x_out, hipStream_t st)
{
    (void)hipGetLastError();
    const long long total64 = (long long)nvis + nadd;
    if (total64 > INT_MAX) { fprintf(stderr, "[bsfm] append: too many observations\n"); return -1; }
    const int total = (int)total64;
    Scratch tmp(st);
    unsigned long long *keys = nullptr, *keys_s = nullptr; int *vals = nullptr, *vals_s = nullptr, *flag = nullptr;
    IX_OK(tmp.alloc(&keys, (size_t)total)); IX_OK(tmp.alloc(&keys_s, (size_t)total));
    IX_OK(tmp.alloc(&vals, (size_t)total)); IX_OK(tmp.alloc(&vals_s, (size_t)total)); IX_OK(tmp.alloc(&flag, 1));
    IX_OK(hipMemsetAsync(flag, 0, sizeof(int), st));
    IX_OK(keep(rowptr_out, (size_t)n_new + 1)); IX_OK(keep(colidx_out, (size_t)total)); IX_OK(keep(x_out, 2 * (size_t)total));
    if (total > 0) {
        hipLaunchKernelGGL(k_merge_keys, dim3(grid_for((size_t)total, 256)), dim3(256), 0, st, nvis, nadd, n_new, m_new, d_obs_pt, d_colidx,
                           d_add_pt, d_add_cam, keys, vals, flag);
        size_t tb = 0;
        const int kbits = bits_for((unsigned long long)n_new * (unsigned long long)m_new);
        IX_OK(prim::sort_pairs(nullptr, tb, keys, keys_s, vals, vals_s, total, 0, kbits, st));
        void* d_tmp = nullptr;
        IX_OK(tmp.alloc(reinterpret_cast<char**>(&d_tmp), tb));
        IX_OK(prim::sort_pairs(d_tmp, tb, keys, keys_s, vals, vals_s, total, 0, kbits, st));
        hipLaunchKernelGGL(k_merge_scatter, dim3(grid_for((size_t)total, 256)), dim3(256), 0, st, total, nvis, m_new, keys_s, vals_s, d_x, d_add_xy,
                           *colidx_out, *x_out, flag);
    }
    hipLaunchKernelGGL(k_merge_rowptr, dim3(grid_for((size_t)n_new + 1, 256)), dim3(256), 0, st, n_new, m_new, total, keys_s, *rowptr_out);
    int hflag = 0;
    IX_OK(hipMemcpyAsync(&hflag, flag, sizeof(int), hipMemcpyDeviceToHost, st));
    IX_OK(hipStreamSynchronize(st));
    if (hflag) {
        fprintf(stderr, "[bsfm] append:%s%s\n", (hflag & 1) ? " point / camera index out of range" : "", (hflag & 2) ? " an observation (point, camera) is given twice" : "");
        bsfm::dev_free(*rowptr_out); bsfm::dev_free(*colidx_out); bsfm::dev_free(*x_out);
        *rowptr_out = nullptr; *colidx_out = nullptr; *x_out = nullptr;
        return -1;
    }
    return 0;
}

// ---- shrinking (the outlier loop of RunSFM_SBA drops whole points, src/Bundle.cpp:784-913) ------------------------
namespace {
__global__ void k_keep_counts(int n, const int* __restrict__ rowptr, const unsigned char* __restrict__ remove, int* __restrict__ pk, int* __restrict__ oc)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    const bool keep_it = i < n && remove[i] == 0;
    pk[i] = keep_it ? 1 : 0;
    oc[i] = keep_it ? rowptr[i + 1] - rowptr[i] : 0;
}
__global__ void k_compact_rows(int n, const unsigned char* __restrict__ remove, const int* __restrict__ pnew, const int* __restrict__ onew,
                               int* __restrict__ rowptr_out, int* __restrict__ remap)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) { rowptr_out[pnew[n]] = onew[n]; return; }
    const bool keep_it = remove[i] == 0;
    remap[i] = keep_it ? pnew[i] : -1;
    if (keep_it) rowptr_out[pnew[i]] = onew[i];
}
__global__ void k_compact_obs(int nvis, const int* __restrict__ obs_pt, const int* __restrict__ rowptr, const int* __restrict__ colidx,
                              const double* __restrict__ x, const unsigned char* __restrict__ remove, const int* __restrict__ onew,
                              int* __restrict__ colidx_out, double* __restrict__ x_out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nvis) return;
    const int i = obs_pt[k];
    if (remove[i]) return;
    const int dst = onew[i] + (k - rowptr[i]);
    colidx_out[dst] = colidx[k];
    x_out[2 * (size_t)dst] = x[2 * (size_t)k]; x_out[2 * (size_t)dst + 1] = x[2 * (size_t)k + 1];
}
__global__ void k_gather_kept(int n, const int* __restrict__ remap, int width, const unsigned char* __restrict__ src, unsigned char* __restrict__ dst)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n * width) return;
    const int i = (int)(t / width), b = (int)(t % width);
    const int q = remap[i];
    if (q >= 0) dst[(size_t)q * width + b] = src[(size_t)i * width + b];
}
}  // namespace

int compact_points_device(int n, int nvis, const int* d_rowptr, const int* d_obs_pt, const int* d_colidx, const double* d_x,
                          const unsigned char* d_remove, int** rowptr_out, int** colidx_out, double** x_out, int** remap_out,
                          int* n_keep, int* nvis_keep, hipStream_t st)
{
    (void)hipGetLastError();
    Scratch tmp(st);
    int *pk = nullptr, *oc = nullptr, *pnew = nullptr, *onew = nullptr;
    IX_OK(tmp.alloc(&pk, (size_t)n + 1)); IX_OK(tmp.alloc(&oc, (size_t)n + 1)); IX_OK(tmp.alloc(&pnew, (size_t)n + 1)); IX_OK(tmp.alloc(&onew, (size_t)n + 1));
    hipLaunchKernelGGL(k_keep_counts, dim3(grid_for((size_t)n + 1, 256)), dim3(256), 0, st, n, d_rowptr, d_remove, pk, oc);
    size_t sb = 0;
    IX_OK(prim::exclusive_sum(nullptr, sb, pk, pnew, n + 1, st));
    void* d_scan = nullptr;
    IX_OK(tmp.alloc(reinterpret_cast<char**>(&d_scan), sb));
    IX_OK(prim::exclusive_sum(d_scan, sb, pk, pnew, n + 1, st));
    IX_OK(prim::exclusive_sum(d_scan, sb, oc, onew, n + 1, st));
    int h[2] = { 0, 0 };
    IX_OK(hipMemcpyAsync(&h[0], pnew + n, sizeof(int), hipMemcpyDeviceToHost, st));
    IX_OK(hipMemcpyAsync(&h[1], onew + n, sizeof(int), hipMemcpyDeviceToHost, st));
    IX_OK(hipStreamSynchronize(st));
    *n_keep = h[0]; *nvis_keep = h[1];
    IX_OK(keep(rowptr_out, (size_t)h[0] + 1)); IX_OK(keep(colidx_out, (size_t)h[1])); IX_OK(keep(x_out, 2 * (size_t)h[1])); IX_OK(keep(remap_out, (size_t)n));
    hipLaunchKernelGGL(k_compact_rows, dim3(grid_for((size_t)n + 1, 256)), dim3(256), 0, st, n, d_remove, pnew, onew, *rowptr_out, *remap_out);
    if (nvis > 0)
        hipLaunchKernelGGL(k_compact_obs, dim3(grid_for((size_t)nvis, 256)), dim3(256), 0, st, nvis, d_obs_pt, d_rowptr, d_colidx, d_x, d_remove, onew,
                           *colidx_out, *x_out);
    IX_OK(hipStreamSynchronize(st));
    return 0;
}

int gather_kept_device(int n, const int* d_remap, int width_bytes, const void* src, void* dst, hipStream_t st)
{
    if (n <= 0 || width_bytes <= 0) return 0;
    hipLaunchKernelGGL(k_gather_kept, dim3(grid_for((size_t)n * (size_t)width_bytes, 256)), dim3(256), 0, st, n, d_remap, width_bytes,
                       static_cast<const unsigned char*>(src), static_cast<unsigned char*>(dst));
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

void free_index_device(DeviceIndex& ix)
{
    void* ptrs[] = { ix.obs_pt, ix.camptr, ix.camobs, ix.campos, ix.cam_pt, ix.cam_cam, ix.triples, ix.tri_pt, ix.tasks,
                     ix.blk_j, ix.blk_k, ix.blk_task0, ix.blk_range };
    (void)hipDeviceSynchronize();
    for (void* p : ptrs) bsfm::dev_free(p, true);
    ix = DeviceIndex();
}

int build_index_device(int n, int m, int mcon, int nvis, const int* d_rowptr, const int* d_colidx, bool want_schur,
                       int order_mode, DeviceIndex& ix, hipStream_t st)
{
    // rocPRIM checks hipGetLastError() after its launches: a stale error of an EARLIER, unrelated call in this thread (e.g.
    // hipEventElapsedTime on a never-recorded phase event -> hipErrorInvalidHandle) would be reported as a sort failure
    (void)hipGetLastError();
    EventPair evp;
    hipEvent_t e0 = evp.e0, e1 = evp.e1;
    if (e0) (void)hipEventRecord(e0, st);
    Scratch tmp(st);
    int* flag = nullptr;
    IX_OK(tmp.alloc(&flag, 1));
    IX_OK(hipMemsetAsync(flag, 0, sizeof(int), st));
    if (n > 0) hipLaunchKernelGGL(k_validate_rows, dim3(grid_for(n, 256)), dim3(256), 0, st, n, m, nvis, d_rowptr, d_colidx, flag);
    int hflag = 0;
    IX_OK(hipMemcpyAsync(&hflag, flag, sizeof(int), hipMemcpyDeviceToHost, st));
    IX_OK(hipStreamSynchronize(st));
    ix.empty_rows = (hflag & 8) != 0 || n == 0;
    hflag &= 7;
    if (hflag) {
        fprintf(stderr, "[bsfm] bad visibility index:%s%s%s\n", (hflag & 1) ? " rowptr not monotone" : "",
                (hflag & 2) ? " colidx out of range" : "", (hflag & 4) ? " colidx not strictly ascending in a row" : "");
        return -1;
    }
    IX_OK(keep(&ix.obs_pt, (size_t)nvis)); IX_OK(keep(&ix.camptr, (size_t)m + 1)); IX_OK(keep(&ix.camobs, (size_t)nvis));
    IX_OK(keep(&ix.campos, (size_t)nvis)); IX_OK(keep(&ix.cam_pt, (size_t)nvis)); IX_OK(keep(&ix.cam_cam, (size_t)nvis));
    long long* tcount = nullptr; long long* toff = nullptr;
    if (want_schur) { IX_OK(tmp.alloc(&tcount, (size_t)n + 1)); IX_OK(tmp.alloc(&toff, (size_t)n + 1)); IX_OK(hipMemsetAsync(tcount, 0, ((size_t)n + 1) * sizeof(long long), st)); }
    if (n > 0) hipLaunchKernelGGL(k_rows, dim3(grid_for(n, 256)), dim3(256), 0, st, n, mcon, d_rowptr, d_colidx, ix.obs_pt, tcount);
    // camera-major order: stable sort of the observations by camera
    if (nvis > 0) {
        int* iota = nullptr;
        IX_OK(tmp.alloc(&iota, (size_t)nvis));
        hipLaunchKernelGGL(k_iota, dim3(grid_for(nvis, 256)), dim3(256), 0, st, nvis, iota);
        size_t tb = 0;
        const int cbits = bits_for((unsigned long long)m - 1ULL);
        IX_OK(prim::sort_pairs(nullptr, tb, d_colidx, ix.cam_cam, iota, ix.camobs, nvis, 0, cbits, st));
        void* d_tmp = nullptr;
        IX_OK(tmp.alloc(reinterpret_cast<char**>(&d_tmp), tb));
        IX_OK(prim::sort_pairs(d_tmp, tb, d_colidx, ix.cam_cam, iota, ix.camobs, nvis, 0, cbits, st));
        hipLaunchKernelGGL(k_cam_maps, dim3(grid_for(nvis, 256)), dim3(256), 0, st, nvis, ix.camobs, ix.obs_pt, ix.campos, ix.cam_pt);
    }
    hipLaunchKernelGGL(k_camptr, dim3(grid_for((size_t)m + 1, 256)), dim3(256), 0, st, m, nvis, ix.cam_cam, ix.camptr);
    if (want_schur) {
        size_t sb = 0;
        IX_OK(prim::exclusive_sum(nullptr, sb, tcount, toff, n + 1, st));
        void* d_scan = nullptr;
        IX_OK(tmp.alloc(reinterpret_cast<char**>(&d_scan), sb));
        IX_OK(prim::exclusive_sum(d_scan, sb, tcount, toff, n + 1, st));
        long long total = 0;
        IX_OK(hipMemcpyAsync(&total, toff + n, sizeof(long long), hipMemcpyDeviceToHost, st));
        IX_OK(hipStreamSynchronize(st));
        if (total > (long long)INT_MAX) { fprintf(stderr, "[bsfm] too many co-visibility triples (%lld)\n", total); return -1; }
        const unsigned long long maxkey = (unsigned long long)(m - mcon) * (unsigned long long)(m - mcon);
        const int rc = maxkey <= 0xffffffffULL
            ? build_schur<unsigned int>(n, m, mcon, nvis, d_rowptr, d_colidx, toff, total, order_mode, ix, st)
            : build_schur<unsigned long long>(n, m, mcon, nvis, d_rowptr, d_colidx, toff, total, order_mode, ix, st);
        if (rc) return rc;
    }
    if (getenv("BSFM_DEBUG_INDEX")) fprintf(stderr, "[bsfm] index build: n %d m %d nvis %d triples %d blocks %d tasks %d\n", n, m, nvis, ix.ntriples, ix.nblk, ix.ntasks);
    if (e1) {
        (void)hipEventRecord(e1, st);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        if (e0 && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) ix.build_ms = ms;
    } else IX_OK(hipStreamSynchronize(st));
    return 0;
}

int schur_chunk()
{
    static const int v = [] {
        int c = 192;
        if (const char* e = getenv("BSFM_SCHUR_CHUNK")) c = atoi(e);
        c = (std::max(16, std::min(SCHUR_CHUNK_MAX, c)) + 15) / 16 * 16;
        return std::min(c, SCHUR_CHUNK_MAX);
    }();
    return v;
}

// ---- dense visibility mask -> CRS on the device ------------------------------------------
// The reference's contract (lib/sba-1.5/sba_levmar.c:642-663): the k-th non-zero byte of vmask in row-major order is measurement k;
// rowptr[i] = number of non-zero bytes before row i, colidx[k] = column of the k-th one.  At 1 000 cameras / 500 000 points the mask
// is 500 MB and round 2 scanned it twice, byte by byte, on one host thread.  Here the mask is treated as ONE flat byte string of
// n*m bytes (independent of m): pass 1 counts the non-zero bytes of every 4 KB piece (one 256-thread workgroup) (16 bytes per lane, SWAR test on the four
// words), rocPRIM scans the piece counts, pass 2 repeats the count inside the piece (wave prefix by DPP-free shuffles + a 4-entry
// LDS scan), writes the column of every set byte to its slot and the row pointer of every row that starts inside the lane's 16
// bytes.  Integer work, bit-identical to the host loop (tests/test_index.py).
namespace {

__device__ __forceinline__ unsigned nz_mask(unsigned w)          // bit 7 of every non-zero byte
{
    return (((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) & 0x80808080u;
}


__global__ __launch_bounds__(256) void k_vmask_count(const uint4* __restrict__ vm, size_t nwords16, int* __restrict__ piece_count)
{
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    int c = 0;
    if (q < nwords16) {
        const uint4 w = vm[q];
        c = __popc(nz_mask(w.x)) + __popc(nz_mask(w.y)) + __popc(nz_mask(w.z)) + __popc(nz_mask(w.w));
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    __shared__ int ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) piece_count[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ __launch_bounds__(256) void k_vmask_fill(const uint4* __restrict__ vm, size_t nwords16, size_t total_bytes, int n, int m,
                                                    const int* __restrict__ piece_off, int* __restrict__ rowptr, int* __restrict__ colidx)
{
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned wd[4] = { 0u, 0u, 0u, 0u };
    if (q < nwords16) { const uint4 w = vm[q]; wd[0] = w.x; wd[1] = w.y; wd[2] = w.z; wd[3] = w.w; }
    unsigned mk[4];
    int c = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) { mk[t] = nz_mask(wd[t]); c += __popc(mk[t]); }
    // exclusive prefix of c over the workgroup
    int incl = c;
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
    __shared__ int ws[4];
    if (lane == 63) ws[wave] = incl;
    __syncthreads();
    int base = piece_off[blockIdx.x];
    for (int w2 = 0; w2 < wave; ++w2) base += ws[w2];
    int k = base + incl - c;                          // slot of this lane's first set byte
    if (q >= nwords16) return;
    const size_t b0 = q * 16;                         // first byte of this lane
    // row pointers of the rows that start inside [b0, b0 + 16): rowptr[i] = set bytes before byte i*m
    {
        size_t i = (b0 + (size_t)m - 1) / (size_t)m;
        for (; i <= (size_t)n && i * (size_t)m < b0 + 16; ++i) {
            const int off = (int)(i * (size_t)m - b0);          // 0 .. 15
            int before = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int lo = 4 * t;
                if (off >= lo + 4) before += __popc(mk[t]);
                else if (off > lo) before += __popc(mk[t] & ((1u << (8 * (off - lo))) - 1u));
            }
            rowptr[i] = k + before;
        }
    }
    if (c == 0) return;
    unsigned col = (unsigned)(b0 % (size_t)m);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (b0 + 4 * t + b < total_bytes && (mk[t] >> (8 * b + 7)) & 1u) colidx[k++] = (int)col;
            if (++col == (unsigned)m) col = 0;
        }
}

}  // namespace

namespace {
struct PlainAlloc {          // hipMalloc'ed temporaries, freed on scope exit
    std::vector<void*> p;
    ~PlainAlloc() { for (void* q : p) if (q) (void)hipFree(q); }
    template <typename T> hipError_t alloc(T** out, size_t count)
    {
        const hipError_t e = hipMalloc(reinterpret_cast<void**>(out), std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) p.push_back(*out);
        return e;
    }
};
}  // namespace

int crs_from_vmask_device(int n, int m, const char* h_vmask, int** d_rowptr_out, int** d_colidx_out, int* nvis_out, double ms_out[3],
                          hipStream_t st)
{
    (void)hipGetLastError();
    *d_rowptr_out = nullptr; *d_colidx_out = nullptr; *nvis_out = 0;
    if (ms_out) ms_out[0] = ms_out[1] = ms_out[2] = 0.0;
    const size_t total = (size_t)n * (size_t)m;
    const size_t nwords16 = (total + 15) / 16, padded = nwords16 * 16;
    const size_t npieces = std::max<size_t>(1, (nwords16 + 255) / 256);
    if (npieces > (size_t)INT_MAX) { fprintf(stderr, "[bsfm] visibility mask too large\n"); return -1; }
    const auto t0 = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    // Plain hipMalloc for everything hipMemcpy touches here.  With stream-ordered POOL memory (Scratch) this function returned zeros for
    // particular mask sizes when it was the first device work of a process: the uploaded mask was there for kernels but read back as
    // zeros by a device-to-host copy and vice versa (debug dumps, round 3) -- pool blocks and the copy engines do not mix reliably on
    // this runtime.  The three allocations cost ~0.1 ms against a 9 ms upload at the headline size.
    PlainAlloc tmp;
    char* d_vm = nullptr; int* d_cnt = nullptr; int* d_off = nullptr;
    IX_OK(tmp.alloc(&d_vm, padded)); IX_OK(tmp.alloc(&d_cnt, npieces + 1)); IX_OK(tmp.alloc(&d_off, npieces + 1));
    if (padded > total) { IX_OK(hipMemsetAsync(d_vm + (padded - 16), 0, 16, st)); IX_OK(hipStreamSynchronize(st)); }
    if (total) IX_OK(hipMemcpy(d_vm, h_vmask, total, hipMemcpyHostToDevice));
    IX_OK(hipStreamSynchronize(st));
    if (ms_out) ms_out[0] = ms_since(t0);                               // upload (pageable host memory: staged by the runtime)
    const auto t1 = std::chrono::steady_clock::now();
    IX_OK(hipMemsetAsync(d_cnt + npieces, 0, sizeof(int), st));
    hipLaunchKernelGGL(k_vmask_count, dim3((unsigned)npieces), dim3(256), 0, st, reinterpret_cast<const uint4*>(d_vm), nwords16, d_cnt);
    size_t sb = 0;
    IX_OK(prim::exclusive_sum(nullptr, sb, d_cnt, d_off, npieces + 1, st));
    void* d_scan = nullptr;
    IX_OK(tmp.alloc(reinterpret_cast<char**>(&d_scan), sb));
    IX_OK(prim::exclusive_sum(d_scan, sb, d_cnt, d_off, npieces + 1, st));
    int nvis = 0;
    IX_OK(hipStreamSynchronize(st));
    IX_OK(hipMemcpy(&nvis, d_off + npieces, sizeof(int), hipMemcpyDeviceToHost));
    if (nvis < 0) { fprintf(stderr, "[bsfm] visibility mask: more than 2^31-1 observations\n"); return -1; }
    int *rp = nullptr, *ci = nullptr;
    IX_OK(keep(&rp, (size_t)n + 1));
    if (keep(&ci, (size_t)nvis) != hipSuccess) { bsfm::dev_free(rp); return -1; }
    hipLaunchKernelGGL(k_vmask_fill, dim3((unsigned)npieces), dim3(256), 0, st, reinterpret_cast<const uint4*>(d_vm), nwords16, total, n, m,
                       d_off, rp, ci);
    // rowptr[n] = nvis: no lane covers byte n*m when the mask length is a multiple of 16 (and nothing at all is covered by an empty
    // mask); copied device-to-device from the scan's total -- not from a host temporary, whose lifetime an asynchronous copy would
    // outlive (round 3: a flaky rowptr[n] in the n*m = 65 536 test case)
    if (total == 0) { if (hipMemsetAsync(rp, 0, ((size_t)n + 1) * sizeof(int), st) != hipSuccess) { bsfm::dev_free(rp); bsfm::dev_free(ci); return -1; } }
    else if (hipMemcpyAsync(rp + n, d_off + npieces, sizeof(int), hipMemcpyDeviceToDevice, st) != hipSuccess) { bsfm::dev_free(rp); bsfm::dev_free(ci); return -1; }
    if (hipStreamSynchronize(st) != hipSuccess) { bsfm::dev_free(rp); bsfm::dev_free(ci); return -1; }
    if (ms_out) { ms_out[1] = ms_since(t1); ms_out[2] = ms_since(t0); }
    *d_rowptr_out = rp; *d_colidx_out = ci; *nvis_out = nvis;
    return 0;
}

}  // namespace bsfm
