// idfile.h -- hand-over of a small secret (the 128-byte ncclUniqueId) from rank 0 to the other ranks of ONE job through a file.
// Host only; comm.hip's bsfm_comm_create_from_env uses it, and bsfm_comm_idfile_exchange exposes it to the two-process CPU tests
// (tests/test_comm_idfile.py), so the protocol is exercised without RCCL or a second GPU.
//
// Rank 0: removes whatever sits at `path`, writes magic | world | creation time | payload under a temporary name
// (O_CREAT | O_EXCL | O_NOFOLLOW, mode 0600) and renames it into place.
// Other ranks: poll `path` until a file appears that (a) opens without following a symlink, (b) is a regular file OWNED BY THIS
// USER with no group / other permission bits (another local user must not be able to plant an id), (c) carries the
// magic word and this job's world size, (d) was created after this process started, minus a grace period for launcher skew -- an id
// left behind by an earlier job on the same address and port is never accepted.  A file that fails a check is ignored and polling
// goes on (rank 0 may still be about to replace it); after `timeout_s` the caller gets an error and a message that names the path.
#pragma once
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>

namespace bsfm {

constexpr unsigned long long IDFILE_MAGIC = 0x6273666d5f6e6363ULL;      // "bsfm_ncc"
constexpr int IDFILE_PAYLOAD = 128;
struct IdFileRecord { unsigned long long magic; int world; int reserved; long long created_ns; unsigned char payload[IDFILE_PAYLOAD]; };

inline long long idfile_now_ns()
{
    return (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
}

// 0 on success
inline int idfile_publish(const std::string& path, int world, const unsigned char* payload)
{
    IdFileRecord rec;
    memset(&rec, 0, sizeof(rec));
    rec.magic = IDFILE_MAGIC; rec.world = world; rec.created_ns = idfile_now_ns();
    memcpy(rec.payload, payload, IDFILE_PAYLOAD);
    const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
    (void)unlink(path.c_str());                               // a stale id of an earlier job with the same address / port
    (void)unlink(tmp.c_str());
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW, 0600);
    bool ok = fd >= 0 && write(fd, &rec, sizeof(rec)) == (ssize_t)sizeof(rec);
    if (fd >= 0) ok = (close(fd) == 0) && ok;
    if (!ok || rename(tmp.c_str(), path.c_str()) != 0) {
        fprintf(stderr, "[bsfm] comm: cannot publish %s\n", path.c_str());
        (void)unlink(tmp.c_str());
        return -1;
    }
    return 0;
}

// 0 on success (payload filled); -1 after timeout_s without an acceptable file
inline int idfile_wait(const std::string& path, int world, int rank, double timeout_s, long long process_start_ns, double grace_s, unsigned char* payload)
{
    const long long grace_ns = (long long)(grace_s * 1e9);
    const auto t0 = std::chrono::steady_clock::now();
    const char* why = "no file";
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < timeout_s) {
        const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW);
        if (fd >= 0) {
            struct stat sb;
            IdFileRecord q;
            if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) why = "not a regular file";
            else if (sb.st_uid != geteuid()) why = "owned by another user";
            else if ((sb.st_mode & 077) != 0) why = "readable or writable by others";
            else if (read(fd, &q, sizeof(q)) != (ssize_t)sizeof(q)) why = "short file";
            else if (q.magic != IDFILE_MAGIC || q.world != world) why = "another job's record";
            else if (q.created_ns < process_start_ns - grace_ns) why = "a stale id of an earlier job";
            else { memcpy(payload, q.payload, IDFILE_PAYLOAD); (void)close(fd); return 0; }
            (void)close(fd);
        } else why = "no file (or a symbolic link)";
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
    fprintf(stderr, "[bsfm] comm: rank %d found no fresh id at %s within %.0f s (last look: %s; is rank 0 running with the same "
                    "MASTER_ADDR / MASTER_PORT / WORLD_SIZE?)\n", rank, path.c_str(), timeout_s, why);
    return -1;
}

}  // namespace bsfm
