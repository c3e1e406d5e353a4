// index_build.h -- interface of the device-side index construction (index_build.hip) used by solver.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <vector>

namespace bsfm {

// diag: the task's block is S_jj (it also produces its part of e_j); out: slot of its partial sums (tasks of a block hold
// consecutive slots, which fixes the summation order; the array itself is in LAUNCH order, see index_build.hip) or -1 = padding
struct SchurTask { int start; int count; int diag; int out; };

constexpr int SCHUR_CHUNK_MAX = 192;  // upper bound of the co-visibility triples per task (LDS of the task kernel, schur.hip.h)
// Triples per task: a multiple of the kernel's pass of 16, BSFM_SCHUR_CHUNK overrides the default.  It is also the knob for the L2
// working set of the task kernel: the ~55 tasks of a camera clique over one point range re-read each other's records, and what
// is in flight per XCD (tasks x triples x 2 x 272 bytes) has to stay below its 4 MB.
int schur_chunk();

// Everything the LM kernels index with, built on the device from the CRS (rowptr, colidx) the caller hands over.
// All pointers are device memory owned by the receiver; they come from bsfm::dev_alloc (devcache.h): release each non-null one with
// bsfm::dev_free, NOT hipFree (the block cache keeps a record of every block it hands out).
struct DeviceIndex {
    int* obs_pt = nullptr;        // nvis: point of observation k
    int* camptr = nullptr;        // m + 1
    int* camobs = nullptr;        // nvis: camera-major position -> observation (the traversal order of sba_crsm_col_elmidxs)
    int* campos = nullptr;        // nvis: observation -> camera-major position
    int* cam_pt = nullptr;        // nvis: camera-major position -> point
    int* cam_cam = nullptr;       // nvis: camera-major position -> camera
    // Schur structure (absent for the camera-only problem)
    int2* triples = nullptr;      // ntriples: (camera-major position of (i,j), of (i,k)), grouped by block (j <= k)
    int* tri_pt = nullptr;        // ntriples: point i of the triple
    SchurTask* tasks = nullptr;   // nslots, launch order
    int* blk_j = nullptr; int* blk_k = nullptr; int* blk_task0 = nullptr;     // nblk, nblk, nblk + 1
    int ntriples = 0, ntasks = 0, nblk = 0, nslots = 0;
    int2* blk_range = nullptr;          // nblk: the slots (= tasks, in task order) k_schur_assemble adds for block b
    SchurTask* tasks_launch = nullptr;  // what the task kernel is given (== tasks; a separate name since round 5's row kernel, removed in round 6)
    std::vector<int> h_blk_j, h_blk_k;      // host copies of the block list (component analysis / multi-GPU union)
    bool empty_rows = false;      // some point has no observation (k_schur_prep's fused point inversion needs every point to have one)
    double build_ms = 0.0;        // device time of the whole construction (HIP events)
};

// rowptr (n+1) / colidx (nvis) are DEVICE arrays.  Returns 0, or -1 with a message on stderr (bad CRS, allocation failure,
// more than 2^31-1 co-visibility triples).  order_mode: launch order of the Schur tasks (results do not depend on it: the partial sums
// keep their block order) -- clustered (default: point slice, then the cameras in breadth-first numbering), block order
// (BSFM_SCHUR_ORDER=block) or by first point (BSFM_SCHUR_ORDER=point, the order of rounds 1-3).
enum { SCHUR_ORDER_CLUSTERED = 0, SCHUR_ORDER_BLOCK = 1, SCHUR_ORDER_POINT = 2 };
int build_index_device(int n, int m, int mcon, int nvis, const int* d_rowptr, const int* d_colidx, bool want_schur,
                       int order_mode, DeviceIndex& out, hipStream_t st);
void free_index_device(DeviceIndex& ix);

// Growing a resident problem: merges `nadd` new observations (point, camera, x, y -- device arrays, any order) into
// an existing CRS (rowptr / obs_pt / colidx / x of nvis observations) for n_new points and m_new cameras.  Outputs (device, owned by
// the caller, bsfm::dev_free): rowptr_out (n_new + 1), colidx_out and x_out (nvis + nadd; 2 doubles per observation), ordered by
// (point, camera) = the reference's measurement order.  Returns 0, or -1 (index out of range, an observation given twice).
int merge_observations_device(int n_new, int m_new, int nvis, const int* d_obs_pt, const int* d_colidx, const double* d_x,
                              int nadd, const int* d_add_pt, const int* d_add_cam, const double* d_add_xy,
                              int** rowptr_out, int** colidx_out, double** x_out, hipStream_t st);

// Shrinking a resident problem (the outlier loop of RunSFM_SBA, src/Bundle.cpp:784-913): drops the points with
// d_remove[i] != 0 (device, n bytes) and all their observations; the others keep their order.  Outputs (device, owned by the caller,
// bsfm::dev_free): the CRS of the kept points, remap_out (n: new index or -1); *n_keep / *nvis_keep.  Returns 0 or -1.
int compact_points_device(int n, int nvis, const int* d_rowptr, const int* d_obs_pt, const int* d_colidx, const double* d_x,
                          const unsigned char* d_remove, int** rowptr_out, int** colidx_out, double** x_out, int** remap_out,
                          int* n_keep, int* nvis_keep, hipStream_t st);
// dst[remap[i]] = src[i] for the kept rows of a per-point array with rows of width_bytes
int gather_kept_device(int n, const int* d_remap, int width_bytes, const void* src, void* dst, hipStream_t st);

// Dense visibility mask (host, n*m bytes, row-major, the reference's vmask) -> CRS ON THE DEVICE: uploads the mask and builds
// rowptr (n + 1) / colidx (nvis) there (device arrays owned by the caller, bsfm::dev_free).  Bit-identical to the reference's fill loop
// (lib/sba-1.5/sba_levmar.c:642-663).  ms_out (optional): upload / kernels / total wall milliseconds.  Returns 0 or -1.
int crs_from_vmask_device(int n, int m, const char* h_vmask, int** d_rowptr_out, int** d_colidx_out, int* nvis_out, double ms_out[3],
                          hipStream_t st);

// Gives the pages of the private stream-ordered scratch pools back to the driver (bsfm_device_cache_trim).
void index_pool_trim();
}  // namespace bsfm
