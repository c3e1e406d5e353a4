// comm.hip -- the collective behind the multi-GPU path, on the C/C++ side of the boundary (north star: "host code stays C/C++ ...
// observations shard across the 8 GPUs of one node with RCCL reduce over xGMI into the camera-block J^T J").
//
// One communicator object per rank; the LM driver (solver.hip) enqueues its exchanges on the problem's own compute stream --
// no host hop, no stream synchronisation around the collective.  Three ways to get one:
//   * bsfm_comm_create_from_env      one rank per PROCESS, launched the way `python -m torch.distributed.run` / mpirun do it:
//                                    RANK, WORLD_SIZE, LOCAL_RANK, MASTER_PORT from the environment; rank 0 draws the
//                                    ncclUniqueId and hands it over through a file under /dev/shm (single node);
//   * bsfm_comm_create_all           one rank per THREAD of one process (ncclCommInitAll over the visible devices): what
//                                    run_sfm uses when it is asked for more than one GPU (boundary.hip).  Every thread owns ONE
//                                    communicator and ONE device and issues its all-reduces on its own stream in the same order
//                                    as its peers: that is RCCL's "one thread per device" usage, which needs no
//                                    ncclGroupStart / ncclGroupEnd (grouping is for ONE thread that drives SEVERAL communicators,
//                                    where un-grouped blocking calls would deadlock); the communicators are created together
//                                    by ncclCommInitAll before the threads start;
//   * the loopback transport         ranks of one process that share ONE device (RCCL refuses duplicate devices in a
//                                    communicator): a barrier + a peer-summing kernel, summed in rank order on every rank.  Only
//                                    there so that the multi-rank control flow can be tested on a 1-GPU box.
// RCCL is dlopen'ed (librccl.so) on first use: single-GPU runs never load it.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <unistd.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <sys/mman.h>
#include <atomic>
#include <cctype>
#include <memory>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <mutex>
#include <condition_variable>
#include <chrono>
#include <thread>
#include "../../include/bsfm.h"
#include "idfile.h"

namespace {

double comm_timeout_s();

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        r.lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!r.lib) r.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!r.lib) { fprintf(stderr, "[bsfm] multi-GPU: cannot load librccl.so: %s\n", dlerror()); return; }
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
        r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.lib, "ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
        r.AllReduce = (decltype(r.AllReduce))dlsym(r.lib, "ncclAllReduce");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommInitAll && r.CommDestroy && r.AllReduce && r.GetErrorString;
        if (!r.ok) fprintf(stderr, "[bsfm] multi-GPU: librccl.so lacks an expected symbol\n");
    });
    return r;
}

// ---- loopback transport: in-process ranks on one device -------------------------------------------------------------------
struct Loopback {
    int world = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0; long generation = 0;
    std::vector<double*> bufs;          // this round's buffers, by rank
    std::vector<void*> shared;          // bsfm_comm_share: this round's allocations, by rank
    std::vector<double*> scratch;       // per-rank result buffers
    std::vector<size_t> scratch_cap;
    int refs = 0;
    bool broken = false;
    // false when a rank never arrived (it died, or left the LM loop on an error of its own): nobody waits forever
    bool barrier(double timeout_s = 300.0)
    {
        std::unique_lock<std::mutex> lk(mu);
        if (broken) return false;
        const long gen = generation;
        if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); return true; }
        if (!cv.wait_for(lk, std::chrono::duration<double>(timeout_s), [&] { return generation != gen || broken; }) || broken) {
            broken = true; cv.notify_all();
            return false;
        }
        return true;
    }
};

__global__ void k_peer_reduce(int world, size_t count, int op, double* const* __restrict__ srcs, double* __restrict__ dst)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    double v = srcs[0][i];
    for (int r = 1; r < world; ++r) { const double t = srcs[r][i]; v = op == 0 ? v + t : (t > v ? t : v); }      // rank order: same bits on every rank
    dst[i] = v;
}

// ---- ipc transport: ranks in DIFFERENT processes that share a device (or a node): tests of the multi-process control flow on a
// 1-GPU box, where RCCL refuses two ranks on one device.  Rank 0 creates a control block in POSIX shared memory
// whose name derives from a random nonce that travels through the same id file as the ncclUniqueId (idfile.h); every rank
// publishes the hipIpcMemHandle of a fixed-capacity exchange buffer there; an all-reduce copies the contribution into the own
// exchange buffer, meets the others at a barrier in the control block (generation counter, bounded wait), sums all ranks' buffers
// in RANK ORDER (the same bits on every rank) straight into the caller's buffer, and meets them again before the exchange buffers
// are reused.  Larger messages go in pieces of the buffer's capacity.
constexpr int IPC_MAX_WORLD = 16;
struct IpcShared {
    unsigned long long magic; int world; int reserved;
    std::atomic<int> arrived; std::atomic<long> generation; std::atomic<int> broken; std::atomic<int> published;
    unsigned long long capacity;                   // doubles per exchange buffer
    hipIpcMemHandle_t handles[IPC_MAX_WORLD];
    hipIpcMemHandle_t board[IPC_MAX_WORLD];        // bsfm_comm_share: the handle of the allocation every rank is currently publishing
};
struct IpcPeer {
    IpcShared* sh = nullptr; size_t map_bytes = 0; std::string shm_path;
    double* mine = nullptr; double* peers[IPC_MAX_WORLD] = {}; size_t cap = 0;
    bool barrier(int world, double timeout_s)
    {
        if (sh->broken.load()) return false;
        const long gen = sh->generation.load();
        if (sh->arrived.fetch_add(1) + 1 == world) { sh->arrived.store(0); sh->generation.fetch_add(1); return true; }
        const auto t0 = std::chrono::steady_clock::now();
        while (sh->generation.load() == gen) {
            if (sh->broken.load() || std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) { sh->broken.store(1); return false; }
            std::this_thread::yield();
        }
        return !sh->broken.load();
    }
};

}  // namespace

struct bsfm_comm {
    int rank = 0, world = 1, device = 0;
    ncclComm_t nccl = nullptr;          // RCCL transport
    IpcPeer* ipc = nullptr;             // ipc transport (processes sharing a device / node)
    Loopback* loop = nullptr;           // loopback transport (shared by the ranks of one process)
    double** d_srcs = nullptr;          // loopback: device array of the ranks' buffer pointers
    double* d_small = nullptr;          // staging for host-scalar reductions
    std::string id_file;                // rank 0 of a process group removes it on destroy
};

namespace {

int comm_alloc_common(bsfm_comm* c)
{
    if (hipMalloc((void**)&c->d_small, 256 * sizeof(double)) != hipSuccess) return -1;
    if ((c->loop || c->ipc) && hipMalloc((void**)&c->d_srcs, (size_t)c->world * sizeof(double*)) != hipSuccess) return -1;
    return 0;
}

}  // namespace

extern "C" {

int bsfm_comm_rank(const bsfm_comm_t* c) { return c ? c->rank : 0; }
int bsfm_comm_world(const bsfm_comm_t* c) { return c ? c->world : 1; }
const char* bsfm_comm_transport(const bsfm_comm_t* c) { return !c ? "none" : (c->nccl ? "rccl" : (c->ipc ? "ipc" : (c->loop ? "loopback" : "none"))); }

void bsfm_comm_destroy(bsfm_comm_t* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->nccl && rccl().ok) (void)rccl().CommDestroy(c->nccl);
    if (c->d_small) (void)hipFree(c->d_small);
    if (c->d_srcs) (void)hipFree(c->d_srcs);
    if (c->ipc) {
        (void)hipDeviceSynchronize();
        for (int r = 0; r < c->world; ++r) if (r != c->rank && c->ipc->peers[r]) (void)hipIpcCloseMemHandle(c->ipc->peers[r]);
        if (c->ipc->mine) (void)hipFree(c->ipc->mine);
        if (c->ipc->sh) (void)munmap(c->ipc->sh, c->ipc->map_bytes);
        if (c->rank == 0 && !c->ipc->shm_path.empty()) (void)unlink(c->ipc->shm_path.c_str());
        delete c->ipc;
    }
    if (c->loop) {
        bool last;
        { std::lock_guard<std::mutex> lk(c->loop->mu); last = --c->loop->refs == 0; }
        if (last) { for (double* p : c->loop->scratch) if (p) (void)hipFree(p); delete c->loop; }
    }
    if (!c->id_file.empty() && c->rank == 0) (void)unlink(c->id_file.c_str());
    delete c;
}

// In-place all-reduce of `count` doubles at device pointer `buf`, enqueued on `stream` (op 0 = sum, 1 = max).
int bsfm_comm_allreduce(bsfm_comm_t* c, void* buf, size_t count, int op, void* stream)
{
    if (!c || c->world <= 1 || count == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (c->nccl) {
        const ncclResult_t r = rccl().AllReduce(buf, buf, count, ncclDouble, op == 0 ? ncclSum : ncclMax, c->nccl, st);
        if (r != ncclSuccess) { fprintf(stderr, "[bsfm] ncclAllReduce failed: %s\n", rccl().GetErrorString(r)); return BSFM_ERROR; }
        return 0;
    }
    if (c->ipc) {
        IpcPeer* I = c->ipc;
        const double tmo = comm_timeout_s();
        for (size_t off = 0; off < count; off += I->cap) {
            const size_t n = std::min(I->cap, count - off);
            double* part = (double*)buf + off;
            if (hipMemcpyAsync(I->mine, part, n * sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess) return BSFM_ERROR;
            if (hipStreamSynchronize(st) != hipSuccess) return BSFM_ERROR;              // this rank's contribution is in its exchange buffer
            if (!I->barrier(c->world, tmo)) { fprintf(stderr, "[bsfm] comm: a rank of the ipc group never reached the exchange (rank %d gives up)\n", c->rank); return BSFM_ERROR; }
            hipLaunchKernelGGL(k_peer_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, c->world, n, op, (double* const*)c->d_srcs, part);
            if (hipStreamSynchronize(st) != hipSuccess) return BSFM_ERROR;
            if (!I->barrier(c->world, tmo)) return BSFM_ERROR;                           // every rank has read every exchange buffer
        }
        return 0;
    }
    if (c->loop) {
        Loopback* L = c->loop;
        if (hipStreamSynchronize(st) != hipSuccess) return BSFM_ERROR;      // this rank's contribution is complete
        {
            std::lock_guard<std::mutex> lk(L->mu);
            L->bufs[c->rank] = (double*)buf;
            if (L->scratch_cap[c->rank] < count) {
                if (L->scratch[c->rank]) (void)hipFree(L->scratch[c->rank]);
                if (hipMalloc((void**)&L->scratch[c->rank], count * sizeof(double)) != hipSuccess) return BSFM_ERROR;
                L->scratch_cap[c->rank] = count;
            }
        }
        if (!L->barrier()) { fprintf(stderr, "[bsfm] comm: a rank of the in-process group never reached the exchange (rank %d gives up)\n", c->rank); return BSFM_ERROR; }   // every rank's pointer is published
        std::vector<double*> h(L->bufs);
        if (hipMemcpyAsync(c->d_srcs, h.data(), (size_t)c->world * sizeof(double*), hipMemcpyHostToDevice, st) != hipSuccess) return BSFM_ERROR;
        hipLaunchKernelGGL(k_peer_reduce, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, c->world, count, op,
                           (double* const*)c->d_srcs, L->scratch[c->rank]);
        if (hipStreamSynchronize(st) != hipSuccess) return BSFM_ERROR;
        if (!L->barrier()) return BSFM_ERROR;                                // every rank has read every buffer
        if (hipMemcpyAsync(buf, L->scratch[c->rank], count * sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess) return BSFM_ERROR;
        return 0;
    }
    return BSFM_ERROR;
}

// Host scalars (count <= 256): staged through a small device buffer, synchronous.  op 0 = sum, 1 = max.
int bsfm_comm_allreduce_host(bsfm_comm_t* c, double* vals, int count, int op)
{
    if (!c || c->world <= 1 || count <= 0) return 0;
    if (count > 256) return BSFM_ERROR;
    (void)hipSetDevice(c->device);
    if (hipMemcpy(c->d_small, vals, (size_t)count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return BSFM_ERROR;
    if (bsfm_comm_allreduce(c, c->d_small, (size_t)count, op, nullptr) != 0) return BSFM_ERROR;
    if (hipStreamSynchronize(nullptr) != hipSuccess) return BSFM_ERROR;
    if (hipMemcpy(vals, c->d_small, (size_t)count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return BSFM_ERROR;
    return 0;
}

// Collective: every rank passes one device allocation of its own (the BASE pointer of a hipMalloc) and receives, in peers[0 .. world), pointers
// through which it can address every rank's allocation -- its own pointer at peers[rank].  ipc transport: hipIpc handles through the control block
// (same device: shared memory; other devices of the node: peer access over xGMI); loopback transport (ranks = threads of one process): the pointers
// themselves.  RCCL transport: not available (BSFM_ERROR) -- the distributed factorisation (chol_flow.hip.h) is offered on the other two.
// bsfm_comm_unshare closes what share opened (every rank, before it frees its own allocation).
int bsfm_comm_share(bsfm_comm_t* c, void* mine, void** peers)
{
    if (!c || !peers || !mine) return BSFM_ERROR;
    if (c->world <= 1) { peers[0] = mine; return 0; }
    const double tmo = comm_timeout_s();
    if (c->ipc) {
        IpcPeer* I = c->ipc;
        if (hipIpcGetMemHandle(&I->sh->board[c->rank], mine) != hipSuccess) { fprintf(stderr, "[bsfm] comm: rank %d cannot export an allocation\n", c->rank); I->sh->broken.store(1); return BSFM_ERROR; }
        if (!I->barrier(c->world, tmo)) return BSFM_ERROR;                     // every handle is on the board
        int bad = 0;
        for (int r = 0; r < c->world; ++r) {
            if (r == c->rank) { peers[r] = mine; continue; }
            if (hipIpcOpenMemHandle(&peers[r], I->sh->board[r], hipIpcMemLazyEnablePeerAccess) != hipSuccess) { peers[r] = nullptr; bad = 1; }
        }
        if (bad) I->sh->broken.store(1);
        if (!I->barrier(c->world, tmo) || bad) return BSFM_ERROR;              // every rank has opened every handle: the board may be reused
        return 0;
    }
    if (c->loop) {
        Loopback* L = c->loop;
        { std::lock_guard<std::mutex> lk(L->mu); if ((int)L->shared.size() < c->world) L->shared.assign((size_t)c->world, nullptr); L->shared[c->rank] = mine; }
        if (!L->barrier(tmo)) return BSFM_ERROR;
        { std::lock_guard<std::mutex> lk(L->mu); for (int r = 0; r < c->world; ++r) peers[r] = L->shared[r]; }
        if (!L->barrier(tmo)) return BSFM_ERROR;
        return 0;
    }
    return BSFM_ERROR;
}
int bsfm_comm_unshare(bsfm_comm_t* c, void** peers)
{
    if (!c || !peers) return BSFM_ERROR;
    if (c->ipc)
        for (int r = 0; r < c->world; ++r) if (r != c->rank && peers[r]) { (void)hipIpcCloseMemHandle(peers[r]); peers[r] = nullptr; }
    return 0;
}

int bsfm_comm_barrier(bsfm_comm_t* c)
{
    double one = 1.0;
    return bsfm_comm_allreduce_host(c, &one, 1, 0);
}

// One rank per process.  Environment (what torch.distributed.run / mpirun wrappers export): RANK, WORLD_SIZE, LOCAL_RANK,
// MASTER_ADDR, MASTER_PORT (+ TORCHELASTIC_RUN_ID when present).  The 128-byte ncclUniqueId travels through a file under /dev/shm
// (single node) whose name is derived from what every rank of ONE job shares and two jobs do not: MASTER_ADDR, MASTER_PORT,
// WORLD_SIZE and the launcher's run id -- not from getppid() (round 2), which differs between ranks under launchers that fork per
// rank from different parents.  BSFM_COMM_ID_FILE overrides the path.  Rank 0 removes a stale file of an earlier, crashed job
// before it creates its own (O_CREAT | O_EXCL | O_NOFOLLOW, mode 0600, written under a temporary name and renamed); the payload
// carries a magic word, the world size and rank 0's creation time, and the other ranks only accept a file created after they
// themselves started (minus a grace period for launcher skew), so an id left behind by an earlier job is never picked up.
// ncclCommInitRank runs under a watchdog: if the ranks do not meet within BSFM_COMM_TIMEOUT_S (default 180 s) the process says so
// and exits -- RCCL itself would wait forever.
}  // extern "C"
namespace {

static_assert(sizeof(ncclUniqueId) == bsfm::IDFILE_PAYLOAD, "the id file carries exactly one ncclUniqueId");
const long long g_process_start_ns = bsfm::idfile_now_ns();          // static initialisation = library load

std::string sanitize(const char* s)
{
    std::string o;
    for (const char* q = s ? s : ""; *q; ++q) o += (isalnum((unsigned char)*q) || *q == '-' || *q == '.') ? *q : '_';
    return o.empty() ? std::string("none") : o;
}

// runs fn() on a helper thread; false when it did not finish within `seconds`
template <typename F> bool with_timeout(double seconds, F fn)
{
    struct St { std::mutex mu; std::condition_variable cv; bool done = false; };
    auto st = std::make_shared<St>();
    std::thread([st, fn]() mutable { fn(); { std::lock_guard<std::mutex> lk(st->mu); st->done = true; } st->cv.notify_all(); }).detach();
    std::unique_lock<std::mutex> lk(st->mu);
    return st->cv.wait_for(lk, std::chrono::duration<double>(seconds), [&] { return st->done; });
}

double comm_timeout_s()
{
    const char* e = getenv("BSFM_COMM_TIMEOUT_S");
    const double v = e ? atof(e) : 180.0;
    return v > 1.0 ? v : 1.0;
}

}  // namespace
extern "C" {

bsfm_comm_t* bsfm_comm_create_from_env(void)
{
    const char* er = getenv("RANK"); const char* ew = getenv("WORLD_SIZE"); const char* el = getenv("LOCAL_RANK");
    const int world = ew ? atoi(ew) : 1, rank = er ? atoi(er) : 0, local = el ? atoi(el) : rank;
    if (world < 1 || rank < 0 || rank >= world) { fprintf(stderr, "[bsfm] comm: bad RANK / WORLD_SIZE\n"); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { fprintf(stderr, "[bsfm] comm: no HIP device\n"); return nullptr; }
    bsfm_comm* c = new bsfm_comm();
    c->rank = rank; c->world = world; c->device = local % ndev;
    if (hipSetDevice(c->device) != hipSuccess) { delete c; return nullptr; }
    if (world == 1 && !getenv("BSFM_COMM_FORCE_RCCL")) { if (comm_alloc_common(c)) { bsfm_comm_destroy(c); return nullptr; } return c; }
    const char* tr = getenv("BSFM_COMM_TRANSPORT");
    const bool want_ipc = tr && !strcmp(tr, "ipc");
    if (!want_ipc && !rccl().ok) { delete c; return nullptr; }
    if (want_ipc && world > IPC_MAX_WORLD) { fprintf(stderr, "[bsfm] comm: the ipc transport takes at most %d ranks\n", IPC_MAX_WORLD); delete c; return nullptr; }
    std::string path;
    if (const char* e = getenv("BSFM_COMM_ID_FILE")) path = e;
    else
        path = "/dev/shm/bsfm_nccl_" + sanitize(getenv("MASTER_ADDR")) + "_" + sanitize(getenv("MASTER_PORT")) + "_w" + std::to_string(world) + "_" +
               sanitize(getenv("TORCHELASTIC_RUN_ID")) + ".id";
    c->id_file = path;
    const double tmo = comm_timeout_s();
    if (want_ipc) {
        // ---- ipc transport: nonce through the id file -> control block in shared memory -> exchange-buffer handles
        unsigned char nonce[bsfm::IDFILE_PAYLOAD];
        memset(nonce, 0, sizeof(nonce));
        if (rank == 0) {
            const int rfd = open("/dev/urandom", O_RDONLY);
            const bool ok = rfd >= 0 && read(rfd, nonce, 16) == 16;
            if (rfd >= 0) (void)close(rfd);
            if (!ok) { fprintf(stderr, "[bsfm] comm: no random nonce\n"); delete c; return nullptr; }
        }
        char hex[33];
        auto shm_name = [&] { for (int q = 0; q < 16; ++q) snprintf(hex + 2 * q, 3, "%02x", nonce[q]); return std::string("/dev/shm/bsfm_ipc_") + hex; };
        IpcPeer* I = new IpcPeer();
        c->ipc = I;
        I->map_bytes = sizeof(IpcShared);
        size_t cap_mb = 64;
        if (const char* e = getenv("BSFM_COMM_IPC_MB")) cap_mb = (size_t)std::max(1, atoi(e));
        I->cap = cap_mb * (1u << 20) / sizeof(double);
        int fd = -1;
        if (rank == 0) {
            I->shm_path = shm_name();
            fd = open(I->shm_path.c_str(), O_RDWR | O_CREAT | O_EXCL | O_NOFOLLOW, 0600);
            if (fd < 0 || ftruncate(fd, (off_t)I->map_bytes) != 0) { fprintf(stderr, "[bsfm] comm: cannot create %s\n", I->shm_path.c_str()); if (fd >= 0) (void)close(fd); bsfm_comm_destroy(c); return nullptr; }
        }
        void* mp = MAP_FAILED;
        if (rank == 0) {
            mp = mmap(nullptr, I->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            (void)close(fd);
        } else {
            // The id file may still be the record of a CRASHED job on the same address and port (inside the 120 s grace window, before this
            // job's rank 0 has replaced it): its nonce names a control block that no longer exists, or one of another world size.  That is
            // not an error but "not yet": keep polling the id file until the time-out.
            const auto t_first = std::chrono::steady_clock::now();
            for (;;) {
                const double left = tmo - std::chrono::duration<double>(std::chrono::steady_clock::now() - t_first).count();
                if (left <= 0.0 || bsfm::idfile_wait(path, world, rank, left, g_process_start_ns, 120.0, nonce) != 0) {
                    fprintf(stderr, "[bsfm] comm: rank %d found no live ipc control block within %.0f s (id file %s)\n", rank, tmo, path.c_str());
                    bsfm_comm_destroy(c); return nullptr;
                }
                I->shm_path = shm_name();
                fd = open(I->shm_path.c_str(), O_RDWR | O_NOFOLLOW);
                if (fd >= 0) {
                    struct stat stb;
                    if (fstat(fd, &stb) == 0 && (size_t)stb.st_size >= I->map_bytes) {
                        mp = mmap(nullptr, I->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
                        if (mp != MAP_FAILED) {
                            IpcShared* sh = (IpcShared*)mp;
                            if (sh->magic == bsfm::IDFILE_MAGIC && sh->world == world) { (void)close(fd); break; }
                            (void)munmap(mp, I->map_bytes); mp = MAP_FAILED;        // another job's block
                        }
                    }
                    (void)close(fd);
                }
                std::this_thread::sleep_for(std::chrono::milliseconds(20));
            }
        }
        if (mp == MAP_FAILED) { fprintf(stderr, "[bsfm] comm: mmap of %s failed\n", I->shm_path.c_str()); bsfm_comm_destroy(c); return nullptr; }
        I->sh = (IpcShared*)mp;
        if (rank == 0) {
            // (a fresh file is zero-filled: counters start at 0); the block exists BEFORE the nonce is published
            I->sh->world = world; I->sh->capacity = I->cap; I->sh->magic = bsfm::IDFILE_MAGIC;
            if (bsfm::idfile_publish(path, world, nonce) != 0) { bsfm_comm_destroy(c); return nullptr; }
        }
        if (hipMalloc((void**)&I->mine, I->cap * sizeof(double)) != hipSuccess ||
            hipIpcGetMemHandle(&I->sh->handles[rank], I->mine) != hipSuccess) { fprintf(stderr, "[bsfm] comm: cannot export the exchange buffer of rank %d\n", rank); bsfm_comm_destroy(c); return nullptr; }
        I->sh->published.fetch_add(1);
        const auto t0 = std::chrono::steady_clock::now();
        while (I->sh->published.load() < world) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > tmo) {
                fprintf(stderr, "[bsfm] comm: only %d of %d ranks reached the ipc group within %.0f s\n", I->sh->published.load(), world, tmo);
                bsfm_comm_destroy(c); return nullptr;
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
        for (int r = 0; r < world; ++r) {
            if (r == rank) { I->peers[r] = I->mine; continue; }
            if (hipIpcOpenMemHandle((void**)&I->peers[r], I->sh->handles[r], hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
                fprintf(stderr, "[bsfm] comm: rank %d cannot open the exchange buffer of rank %d\n", rank, r); I->peers[r] = nullptr; bsfm_comm_destroy(c); return nullptr;
            }
        }
        if (comm_alloc_common(c)) { bsfm_comm_destroy(c); return nullptr; }
        if (hipMemcpy(c->d_srcs, I->peers, (size_t)world * sizeof(double*), hipMemcpyHostToDevice) != hipSuccess) { bsfm_comm_destroy(c); return nullptr; }
        return c;
    }
    ncclUniqueId uid;
    memset(&uid, 0, sizeof(uid));
    if (rank == 0) {
        if (rccl().GetUniqueId(&uid) != ncclSuccess) { fprintf(stderr, "[bsfm] comm: ncclGetUniqueId failed\n"); delete c; return nullptr; }
        if (bsfm::idfile_publish(path, world, reinterpret_cast<const unsigned char*>(&uid)) != 0) { delete c; return nullptr; }
    } else {
        // launchers start their ranks within seconds of each other: 120 s of grace
        if (bsfm::idfile_wait(path, world, rank, tmo, g_process_start_ns, 120.0, reinterpret_cast<unsigned char*>(&uid)) != 0) { delete c; return nullptr; }
    }
    ncclResult_t r = ncclSuccess;
    ncclComm_t* slot = &c->nccl;
    const int dev = c->device;
    const ncclUniqueId id = uid;
    if (!with_timeout(tmo, [&r, slot, world, id, rank, dev] { (void)hipSetDevice(dev); r = rccl().CommInitRank(slot, world, id, rank); })) {
        fprintf(stderr, "[bsfm] FATAL: ncclCommInitRank did not complete within %.0f s on rank %d of %d (a rank is missing or holds another job's id); RCCL cannot be cancelled -- exiting\n", tmo, rank, world);
        fflush(stderr);
        _exit(3);
    }
    if (r != ncclSuccess) { fprintf(stderr, "[bsfm] ncclCommInitRank failed: %s\n", rccl().GetErrorString(r)); c->nccl = nullptr; bsfm_comm_destroy(c); return nullptr; }
    if (comm_alloc_common(c)) { bsfm_comm_destroy(c); return nullptr; }
    return c;
}

// Test hook (no device, no RCCL): the id hand-over of bsfm_comm_create_from_env on its own.  rank 0 publishes id[128] at `path`, any
// other rank waits up to timeout_s for a record that passes every check of idfile.h and receives it in id[128].  grace_s = how much
// older than this process a record may be (120 in production).  Returns 0, or BSFM_ERROR (message on stderr).
int bsfm_comm_idfile_exchange(const char* path, int rank, int world, double timeout_s, double grace_s, unsigned char* id)
{
    if (!path || !id || world < 1 || rank < 0 || rank >= world) return BSFM_ERROR;
    if (rank == 0) return bsfm::idfile_publish(path, world, id) == 0 ? 0 : BSFM_ERROR;
    return bsfm::idfile_wait(path, world, rank, timeout_s, g_process_start_ns, grace_s, id) == 0 ? 0 : BSFM_ERROR;
}

// One rank per thread of THIS process: fills comms[0 .. ndev-1] for the devices devs[.] (the caller's threads then each take one
// and call hipSetDevice(devs[g]) themselves).  Distinct devices -> ncclCommInitAll; a repeated device -> loopback transport.
int bsfm_comm_create_all(int ndev, const int* devs, bsfm_comm_t** comms)
{
    if (ndev < 1 || !devs || !comms) return BSFM_ERROR;
    bool distinct = true;
    for (int a = 0; a < ndev; ++a) for (int b = a + 1; b < ndev; ++b) if (devs[a] == devs[b]) distinct = false;
    int saved = 0; (void)hipGetDevice(&saved);
    for (int g = 0; g < ndev; ++g) { comms[g] = new bsfm_comm(); comms[g]->rank = g; comms[g]->world = ndev; comms[g]->device = devs[g]; }
    auto fail = [&] { for (int g = 0; g < ndev; ++g) { bsfm_comm_destroy(comms[g]); comms[g] = nullptr; } (void)hipSetDevice(saved); return BSFM_ERROR; };
    if (ndev > 1) {
        if (distinct) {
            if (!rccl().ok) return fail();
            std::vector<ncclComm_t> nc((size_t)ndev);
            ncclResult_t r = ncclSuccess;
            ncclComm_t* ncp = nc.data();
            if (!with_timeout(comm_timeout_s(), [&r, ncp, ndev, devs] { r = rccl().CommInitAll(ncp, ndev, devs); })) {
                fprintf(stderr, "[bsfm] FATAL: ncclCommInitAll over %d devices did not complete within %.0f s -- exiting\n", ndev, comm_timeout_s());
                fflush(stderr);
                _exit(3);
            }
            if (r != ncclSuccess) { fprintf(stderr, "[bsfm] ncclCommInitAll failed: %s\n", rccl().GetErrorString(r)); return fail(); }
            for (int g = 0; g < ndev; ++g) comms[g]->nccl = nc[g];
        } else {
            Loopback* L = new Loopback();
            L->world = ndev; L->bufs.assign((size_t)ndev, nullptr); L->scratch.assign((size_t)ndev, nullptr); L->scratch_cap.assign((size_t)ndev, 0);
            L->refs = ndev;
            for (int g = 0; g < ndev; ++g) comms[g]->loop = L;
        }
    }
    for (int g = 0; g < ndev; ++g) {
        if (hipSetDevice(devs[g]) != hipSuccess || comm_alloc_common(comms[g])) return fail();
    }
    (void)hipSetDevice(saved);
    return 0;
}

}  // extern "C"
