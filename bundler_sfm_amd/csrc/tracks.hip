// tracks.hip -- linking pairwise matches into tracks on the device (second half).
//
// Reference: BundlerApp::ComputeTracks (src/ComputeTracks.cpp:36-313) on the symmetric match lists (MakeMatchListsSymmetric,
// src/MatchTracks.cpp:337-392).  It walks the images in order, the keys of an image in order, and from every key not yet claimed runs a
// breadth-first search over the match graph: from a feature (image, key) it looks, for each matching image k in ASCENDING image order that
// is not yet represented in the growing track, for the feature's match in k (binary search in the list sorted by first index), and claims
// it if it is unclaimed.  A track keeps at most one key per image; tracks with >= 2 projections are numbered in the order their first
// feature was met.  The result depends on that order, so it cannot be a plain connected-components labelling -- but two searches interact
// only inside one connected component of the feature graph, and components are tiny (a track's worth of features).  Hence:
//   1. feature ids g = key_offset[image] + key; both directions of every match sorted by (source, destination) with rocPRIM's radix sort
//      -> adjacency rows whose neighbours ascend by image exactly as the reference visits them;
//   2. connected components by min-label propagation with pointer jumping (label = smallest feature id of the component);
//   3. ONE THREAD PER COMPONENT replays the reference's search over its features in ascending id = the order the reference meets them;
//      the growing track doubles as the FIFO queue (the reference pushes to both at once);
//   4. tracks are ordered by their first feature's id (radix sort) = the reference's track numbering, and gathered.
// Integer bookkeeping only: the output equals the reference's bit for bit (tests/test_tracks.py, against src/ComputeTracks.cpp compiled
// verbatim, on the kermit example's matches.init.txt and on synthetic match graphs).
// Precondition (what PruneDoubleMatches, src/MatchTracks.cpp:394-440, and the matcher guarantee): inside a pair every key index occurs at most
// once on either side -- otherwise the reference's own result depends on the order std::sort leaves equal elements in; refused loudly.
#include <hip/hip_runtime.h>
#include "prim.hip.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../../include/bsfm.h"

namespace {

#define TK_OK(call)                                                                                              \
    do { hipError_t _e = (call); if (_e != hipSuccess) {                                                        \
        fprintf(stderr, "[bsfm] compute_tracks: HIP error %s at %s:%d\n", hipGetErrorName(_e), __FILE__, __LINE__); \
        return BSFM_ERROR; } } while (0)

inline int grid_for(size_t count, int block) { return (int)std::max<size_t>(1, (count + block - 1) / block); }
inline int bits_for(unsigned long long maxval) { int b = 1; while (b < 64 && (maxval >> b)) ++b; return b; }

struct Bufs {
    std::vector<void*> p;
    ~Bufs() { for (void* q : p) if (q) (void)hipFree(q); }
    template <typename T> hipError_t alloc(T** out, size_t count)
    {
        hipError_t e = hipMalloc(reinterpret_cast<void**>(out), std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) p.push_back(*out);
        return e;
    }
};

// directed edges of both directions: key = (source feature << 32) | destination feature
__global__ void k_edges(int npairs, const int* __restrict__ pair_i, const int* __restrict__ pair_j, const int* __restrict__ match_ptr,
                        const int* __restrict__ matches, const int* __restrict__ key_off, const int* __restrict__ num_keys,
                        unsigned long long* __restrict__ ekey, int* __restrict__ flag)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = match_ptr[npairs];
    if (q >= total) return;
    int lo = 0, hi = npairs - 1;                      // pair of match q
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (match_ptr[mid] <= q) lo = mid; else hi = mid - 1; }
    const int i1 = pair_i[lo], i2 = pair_j[lo], k1 = matches[2 * q], k2 = matches[2 * q + 1];
    if (k1 < 0 || k1 >= num_keys[i1] || k2 < 0 || k2 >= num_keys[i2]) { atomicOr(flag, 1); ekey[2 * (size_t)q] = ~0ULL; ekey[2 * (size_t)q + 1] = ~0ULL; return; }
    const unsigned long long g1 = (unsigned long long)(key_off[i1] + k1), g2 = (unsigned long long)(key_off[i2] + k2);
    ekey[2 * (size_t)q] = (g1 << 32) | g2;
    ekey[2 * (size_t)q + 1] = (g2 << 32) | g1;
}

__global__ void k_img_of(int num_images, const int* __restrict__ key_off, int* __restrict__ img_of)
{
    const int i = blockIdx.x;
    for (int k = key_off[i] + threadIdx.x; k < key_off[i + 1]; k += blockDim.x) img_of[k] = i;
}

// adjacency rows: adj_ptr[g] = first edge whose source is >= g; also: two edges of one feature into the same image = refused
__global__ void k_adj_ptr(int F, int E2, const unsigned long long* __restrict__ ekey, int* __restrict__ adj_ptr)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g > F) return;
    const unsigned long long target = (unsigned long long)g << 32;
    int lo = 0, hi = E2;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (ekey[mid] < target) lo = mid + 1; else hi = mid; }
    adj_ptr[g] = lo;
}
__global__ void k_check_dups(int E2, const unsigned long long* __restrict__ ekey, const int* __restrict__ img_of, int* __restrict__ flag)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e <= 0 || e >= E2) return;
    const unsigned long long a = ekey[e - 1], b = ekey[e];
    if ((a >> 32) == (b >> 32) && img_of[(unsigned)(a & 0xffffffffULL)] == img_of[(unsigned)(b & 0xffffffffULL)]) atomicOr(flag, 2);
}

__global__ void k_label_init(int F, int* __restrict__ label)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < F) label[g] = g;
}
// one sweep: every edge pulls the larger label down to the smaller one; then one pointer jump per feature
__global__ void k_label_edges(int E2, const unsigned long long* __restrict__ ekey, int* __restrict__ label, int* __restrict__ changed)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E2) return;
    const int u = (int)(ekey[e] >> 32), v = (int)(ekey[e] & 0xffffffffULL);
    const int lu = label[u], lv = label[v];
    if (lu < lv) { atomicMin(&label[v], lu); *changed = 1; }
}
__global__ void k_label_jump(int F, int* __restrict__ label, int* __restrict__ changed)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= F) return;
    const int l = label[g], ll = label[l];
    if (ll < l) { label[g] = ll; *changed = 1; }
}

// nodes = features with at least one match; key for grouping = (label << 32) | feature
__global__ void k_node_keys(int F, const int* __restrict__ adj_ptr, const int* __restrict__ label, unsigned long long* __restrict__ nkey)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= F) return;
    nkey[g] = adj_ptr[g + 1] > adj_ptr[g] ? ((unsigned long long)(unsigned)label[g] << 32) | (unsigned)g : ~0ULL;
}
__global__ void k_comp_starts(int nnodes, const unsigned long long* __restrict__ nkey, int* __restrict__ is_start)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nnodes) return;
    is_start[t] = (t == 0 || (nkey[t] >> 32) != (nkey[t - 1] >> 32)) ? 1 : 0;
}

// ONE THREAD PER COMPONENT: the reference's search (src/ComputeTracks.cpp:106-268) over the component's features in ascending id.
// out_views[c0 .. c1) receives the tracks back to back; seg_len[first slot of a track] = its length (>= 2), 0 elsewhere.
__global__ void k_replay(int ncomp, int nnodes, const int* __restrict__ comp_start, const unsigned long long* __restrict__ nkey,
                         const int* __restrict__ adj_ptr, const unsigned long long* __restrict__ ekey, const int* __restrict__ img_of,
                         unsigned char* __restrict__ visited, int* __restrict__ out_views, int* __restrict__ seg_len)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncomp) return;
    const int c0 = comp_start[c], c1 = c + 1 < ncomp ? comp_start[c + 1] : nnodes;
    int w = c0;                                        // write position inside the component's output region
    for (int t = c0; t < c1; ++t) {
        const int seed = (int)(nkey[t] & 0xffffffffULL);
        if (visited[seed]) continue;
        visited[seed] = 1;
        int len = 1, head = 0;
        out_views[w] = seed;
        while (head < len) {                           // the track is its own FIFO queue
            const int cur = out_views[w + head++];
            for (int e = adj_ptr[cur]; e < adj_ptr[cur + 1]; ++e) {       // matching images in ascending order
                const int v = (int)(ekey[e] & 0xffffffffULL);
                const int img = img_of[v];
                bool marked = false;                    // img_marked[k]: the image already has a key in this track
                for (int s = 0; s < len && !marked; ++s) marked = img_of[out_views[w + s]] == img;
                if (marked || visited[v]) continue;
                visited[v] = 1;
                out_views[w + len++] = v;
            }
        }
        if (len >= 2) { seg_len[w] = len; w += len; }
    }
}

__global__ void k_track_keys(int nnodes, const int* __restrict__ seg_len, const int* __restrict__ out_views, unsigned long long* __restrict__ tkey)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nnodes) return;
    // tracks first, ordered by the id of their first feature; everything else behind them
    tkey[t] = seg_len[t] > 0 ? (unsigned long long)(unsigned)out_views[t] : ~0ULL;
}
__global__ void k_track_len(int ntracks, const int* __restrict__ pos_sorted, const int* __restrict__ seg_len, int* __restrict__ len_out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < ntracks) len_out[t] = seg_len[pos_sorted[t]];
    if (t == ntracks) len_out[t] = 0;
}
__global__ void k_track_gather(int ntracks, const int* __restrict__ pos_sorted, const int* __restrict__ track_ptr, const int* __restrict__ out_views,
                               const int* __restrict__ img_of, const int* __restrict__ key_off, int* __restrict__ views)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntracks) return;
    const int src = pos_sorted[t], dst = track_ptr[t], len = track_ptr[t + 1] - dst;
    for (int s = 0; s < len; ++s) {
        const int g = out_views[src + s], img = img_of[g];
        views[2 * (size_t)(dst + s)] = img; views[2 * (size_t)(dst + s) + 1] = g - key_off[img];
    }
}

}  // namespace

extern "C" int bsfm_compute_tracks(int num_images, const int* num_keys, int num_pairs, const int* pair_i, const int* pair_j,
                                   const int* match_ptr, const int* matches, int new_image_start,
                                   int* track_ptr, int* views, int max_tracks, int max_views, int* num_views)
{
    (void)new_image_start;      // the reference computes a start index from it and never uses it (src/ComputeTracks.cpp:147-154)
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { fprintf(stderr, "[bsfm] FATAL: no usable HIP device; compute_tracks has no CPU fallback\n"); return BSFM_ERROR; }
    if (num_images < 0 || num_pairs < 0 || (num_images && !num_keys) || (num_pairs && (!pair_i || !pair_j || !match_ptr))) return BSFM_ERROR;
    (void)hipGetLastError();
    std::vector<int> off((size_t)num_images + 1, 0);
    for (int i = 0; i < num_images; ++i) {
        if (num_keys[i] < 0) return BSFM_ERROR;
        const long long nx = (long long)off[i] + num_keys[i];
        if (nx > 0x7fffffffLL) { fprintf(stderr, "[bsfm] compute_tracks: too many keys\n"); return BSFM_ERROR; }
        off[i + 1] = (int)nx;
    }
    const int F = off[num_images];
    const int E = num_pairs ? match_ptr[num_pairs] : 0;
    for (int p = 0; p < num_pairs; ++p)
        if (pair_i[p] < 0 || pair_j[p] < 0 || pair_i[p] >= num_images || pair_j[p] >= num_images || pair_i[p] == pair_j[p] || match_ptr[p + 1] < match_ptr[p]) {
            fprintf(stderr, "[bsfm] compute_tracks: bad pair %d\n", p); return BSFM_ERROR; }
    if (num_views) *num_views = 0;
    if (E == 0 || F == 0) { if (track_ptr && max_tracks >= 0) track_ptr[0] = 0; return 0; }
    if ((long long)E * 2 > 0x7fffffffLL) { fprintf(stderr, "[bsfm] compute_tracks: too many matches\n"); return BSFM_ERROR; }
    const int E2 = 2 * E;
    hipStream_t st = nullptr;
    Bufs B;
    int *d_off = nullptr, *d_nk = nullptr, *d_pi = nullptr, *d_pj = nullptr, *d_mp = nullptr, *d_mt = nullptr, *d_flag = nullptr, *d_img = nullptr, *d_adj = nullptr, *d_label = nullptr;
    unsigned long long *d_ek = nullptr, *d_ek2 = nullptr;
    TK_OK(B.alloc(&d_off, (size_t)num_images + 1)); TK_OK(B.alloc(&d_nk, (size_t)num_images)); TK_OK(B.alloc(&d_pi, (size_t)num_pairs)); TK_OK(B.alloc(&d_pj, (size_t)num_pairs));
    TK_OK(B.alloc(&d_mp, (size_t)num_pairs + 1)); TK_OK(B.alloc(&d_mt, 2 * (size_t)E)); TK_OK(B.alloc(&d_flag, 4)); TK_OK(B.alloc(&d_img, (size_t)F));
    TK_OK(B.alloc(&d_adj, (size_t)F + 1)); TK_OK(B.alloc(&d_label, (size_t)F)); TK_OK(B.alloc(&d_ek, (size_t)E2)); TK_OK(B.alloc(&d_ek2, (size_t)E2));
    TK_OK(hipMemcpy(d_off, off.data(), off.size() * sizeof(int), hipMemcpyHostToDevice));
    TK_OK(hipMemcpy(d_nk, num_keys, (size_t)num_images * sizeof(int), hipMemcpyHostToDevice));
    TK_OK(hipMemcpy(d_pi, pair_i, (size_t)num_pairs * sizeof(int), hipMemcpyHostToDevice));
    TK_OK(hipMemcpy(d_pj, pair_j, (size_t)num_pairs * sizeof(int), hipMemcpyHostToDevice));
    TK_OK(hipMemcpy(d_mp, match_ptr, ((size_t)num_pairs + 1) * sizeof(int), hipMemcpyHostToDevice));
    TK_OK(hipMemcpy(d_mt, matches, 2 * (size_t)E * sizeof(int), hipMemcpyHostToDevice));
    TK_OK(hipMemset(d_flag, 0, 4 * sizeof(int)));
    hipLaunchKernelGGL(k_img_of, dim3(num_images), dim3(256), 0, st, num_images, d_off, d_img);
    hipLaunchKernelGGL(k_edges, dim3(grid_for((size_t)E, 256)), dim3(256), 0, st, num_pairs, d_pi, d_pj, d_mp, d_mt, d_off, d_nk, d_ek, d_flag);
    {   // adjacency: edges sorted by (source, destination)
        size_t tb = 0;
        const int kbits = 32 + bits_for((unsigned long long)F);
        TK_OK(bsfm::prim::sort_keys(nullptr, tb, d_ek, d_ek2, E2, 0, kbits, st));
        char* d_tmp = nullptr;
        TK_OK(B.alloc(&d_tmp, tb));
        TK_OK(bsfm::prim::sort_keys(d_tmp, tb, d_ek, d_ek2, E2, 0, kbits, st));
    }
    int hflag[4] = { 0, 0, 0, 0 };
    TK_OK(hipMemcpy(hflag, d_flag, sizeof(hflag), hipMemcpyDeviceToHost));
    if (hflag[0] & 1) { fprintf(stderr, "[bsfm] compute_tracks: a match refers to a key index outside its image\n"); return BSFM_ERROR; }
    hipLaunchKernelGGL(k_adj_ptr, dim3(grid_for((size_t)F + 1, 256)), dim3(256), 0, st, F, E2, d_ek2, d_adj);
    hipLaunchKernelGGL(k_check_dups, dim3(grid_for((size_t)E2, 256)), dim3(256), 0, st, E2, d_ek2, d_img, d_flag);
    // connected components
    hipLaunchKernelGGL(k_label_init, dim3(grid_for((size_t)F, 256)), dim3(256), 0, st, F, d_label);
    for (int it = 0; it < 100000; ++it) {
        TK_OK(hipMemsetAsync(d_flag + 1, 0, sizeof(int), st));
        hipLaunchKernelGGL(k_label_edges, dim3(grid_for((size_t)E2, 256)), dim3(256), 0, st, E2, d_ek2, d_label, d_flag + 1);
        hipLaunchKernelGGL(k_label_jump, dim3(grid_for((size_t)F, 256)), dim3(256), 0, st, F, d_label, d_flag + 1);
        hipLaunchKernelGGL(k_label_jump, dim3(grid_for((size_t)F, 256)), dim3(256), 0, st, F, d_label, d_flag + 1);
        TK_OK(hipMemcpy(hflag, d_flag, sizeof(hflag), hipMemcpyDeviceToHost));
        if (!hflag[1]) break;
    }
    if (hflag[0] & 2) {
        fprintf(stderr, "[bsfm] compute_tracks: a key is matched twice into the same image (run PruneDoubleMatches first, src/MatchTracks.cpp:394-440); "
                        "the reference's result would depend on std::sort's order of equal elements\n");
        return BSFM_ERROR;
    }
    // nodes grouped by component
    unsigned long long *d_nk1 = nullptr, *d_nk2 = nullptr; int *d_start = nullptr, *d_cstart = nullptr, *d_cnt = nullptr;
    TK_OK(B.alloc(&d_nk1, (size_t)F)); TK_OK(B.alloc(&d_nk2, (size_t)F)); TK_OK(B.alloc(&d_start, (size_t)F)); TK_OK(B.alloc(&d_cstart, (size_t)F + 1)); TK_OK(B.alloc(&d_cnt, 2));
    hipLaunchKernelGGL(k_node_keys, dim3(grid_for((size_t)F, 256)), dim3(256), 0, st, F, d_adj, d_label, d_nk1);
    {
        size_t tb = 0;
        TK_OK(bsfm::prim::sort_keys(nullptr, tb, d_nk1, d_nk2, F, 0, 64, st));
        char* d_tmp = nullptr;
        TK_OK(B.alloc(&d_tmp, tb));
        TK_OK(bsfm::prim::sort_keys(d_tmp, tb, d_nk1, d_nk2, F, 0, 64, st));
    }
    // number of nodes = features with matches (their keys sort ahead of the ~0 fillers): count on the host from adj_ptr is not available
    // without a download, so count the edges' distinct sources with a tiny reduction: nnodes = #(adj rows non-empty)
    std::vector<int> h_adj((size_t)F + 1);
    TK_OK(hipMemcpy(h_adj.data(), d_adj, h_adj.size() * sizeof(int), hipMemcpyDeviceToHost));
    int nnodes = 0;
    for (int g = 0; g < F; ++g) nnodes += h_adj[g + 1] > h_adj[g];
    hipLaunchKernelGGL(k_comp_starts, dim3(grid_for((size_t)nnodes, 256)), dim3(256), 0, st, nnodes, d_nk2, d_start);
    int* d_iota = nullptr;
    TK_OK(B.alloc(&d_iota, (size_t)nnodes));
    {   // component start positions = positions t with is_start[t]
        std::vector<int> iota((size_t)nnodes);
        for (int t = 0; t < nnodes; ++t) iota[t] = t;
        TK_OK(hipMemcpy(d_iota, iota.data(), iota.size() * sizeof(int), hipMemcpyHostToDevice));
        size_t tb = 0;
        TK_OK(bsfm::prim::select_flagged(nullptr, tb, d_iota, d_start, d_cstart, d_cnt, nnodes, st));
        char* d_tmp = nullptr;
        TK_OK(B.alloc(&d_tmp, tb));
        TK_OK(bsfm::prim::select_flagged(d_tmp, tb, d_iota, d_start, d_cstart, d_cnt, nnodes, st));
    }
    int ncomp = 0;
    TK_OK(hipMemcpy(&ncomp, d_cnt, sizeof(int), hipMemcpyDeviceToHost));
    // replay
    unsigned char* d_vis = nullptr; int *d_out = nullptr, *d_seg = nullptr;
    TK_OK(B.alloc(&d_vis, (size_t)F)); TK_OK(B.alloc(&d_out, (size_t)nnodes)); TK_OK(B.alloc(&d_seg, (size_t)nnodes));
    TK_OK(hipMemsetAsync(d_vis, 0, (size_t)F, st)); TK_OK(hipMemsetAsync(d_seg, 0, (size_t)nnodes * sizeof(int), st));
    hipLaunchKernelGGL(k_replay, dim3(grid_for((size_t)ncomp, 64)), dim3(64), 0, st, ncomp, nnodes, d_cstart, d_nk2, d_adj, d_ek2, d_img, d_vis, d_out, d_seg);
    // tracks in the reference's numbering: by the id of the first feature
    unsigned long long *d_tk1 = nullptr, *d_tk2 = nullptr; int *d_pos = nullptr, *d_len = nullptr, *d_tptr = nullptr;
    TK_OK(B.alloc(&d_tk1, (size_t)nnodes)); TK_OK(B.alloc(&d_tk2, (size_t)nnodes)); TK_OK(B.alloc(&d_pos, (size_t)nnodes));
    hipLaunchKernelGGL(k_track_keys, dim3(grid_for((size_t)nnodes, 256)), dim3(256), 0, st, nnodes, d_seg, d_out, d_tk1);
    {
        size_t tb = 0;
        TK_OK(bsfm::prim::sort_pairs(nullptr, tb, d_tk1, d_tk2, d_iota, d_pos, nnodes, 0, 64, st));
        char* d_tmp = nullptr;
        TK_OK(B.alloc(&d_tmp, tb));
        TK_OK(bsfm::prim::sort_pairs(d_tmp, tb, d_tk1, d_tk2, d_iota, d_pos, nnodes, 0, 64, st));
    }
    std::vector<int> h_seg((size_t)nnodes);
    TK_OK(hipMemcpy(h_seg.data(), d_seg, h_seg.size() * sizeof(int), hipMemcpyDeviceToHost));
    int ntracks = 0; long long nviews = 0;
    for (int t = 0; t < nnodes; ++t) if (h_seg[t] > 0) { ++ntracks; nviews += h_seg[t]; }
    if (num_views) *num_views = (int)nviews;
    if (!track_ptr || !views) return ntracks;                       // size query: the counts only
    if (ntracks > max_tracks || nviews > max_views) { fprintf(stderr, "[bsfm] compute_tracks: output buffers too small (%d tracks, %lld views)\n", ntracks, nviews); return BSFM_ERROR; }
    TK_OK(B.alloc(&d_len, (size_t)ntracks + 1)); TK_OK(B.alloc(&d_tptr, (size_t)ntracks + 1));
    hipLaunchKernelGGL(k_track_len, dim3(grid_for((size_t)ntracks + 1, 256)), dim3(256), 0, st, ntracks, d_pos, d_seg, d_len);
    {
        size_t tb = 0;
        TK_OK(bsfm::prim::exclusive_sum(nullptr, tb, d_len, d_tptr, ntracks + 1, st));
        char* d_tmp = nullptr;
        TK_OK(B.alloc(&d_tmp, tb));
        TK_OK(bsfm::prim::exclusive_sum(d_tmp, tb, d_len, d_tptr, ntracks + 1, st));
    }
    int* d_views = nullptr;
    TK_OK(B.alloc(&d_views, 2 * (size_t)std::max<long long>(nviews, 1)));
    hipLaunchKernelGGL(k_track_gather, dim3(grid_for((size_t)ntracks, 256)), dim3(256), 0, st, ntracks, d_pos, d_tptr, d_out, d_img, d_off, d_views);
    TK_OK(hipMemcpy(track_ptr, d_tptr, ((size_t)ntracks + 1) * sizeof(int), hipMemcpyDeviceToHost));
    if (nviews) TK_OK(hipMemcpy(views, d_views, 2 * (size_t)nviews * sizeof(int), hipMemcpyDeviceToHost));
    return ntracks;
}
