// Comm.hpp -- what a C++ host needs to run one admm::System per GPU (one process per rank).
//
// The reference is single-process (System.cpp:57-58 is an OpenMP loop); here the element loop of
// System::step shards across ranks and the partial right-hand sides meet in one all-reduce per ADMM
// iteration.  The all-reduce itself is RCCL inside libadmm_hip.so
// (admm_hip_rccl_init); a communicator needs ONE 128-byte id to travel from rank 0 to the other ranks
// before it exists, and a C++ host has no torch process group for that.  Two helpers, plain POSIX:
//
//   rccl_id_via_file   rank 0 publishes the ncclUniqueId in a file (written under a temporary name,
//                      renamed into place), the other ranks wait for it.  Works across nodes on a
//                      shared filesystem.  The path must be unique per job (e.g. contain the job id).
//   ShmAllReduce       a host-memory all-reduce between the ranks of ONE node through a POSIX shared
//                      memory segment (sums in rank order: every rank gets the same bits).  Not the
//                      production transport -- RCCL over xGMI is -- but it needs no RCCL, so several
//                      ranks can share one GPU (RCCL refuses that): bring-up and the tests use it via
//                      admm_hip_set_host_allreduce.
#pragma once
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace admm {
namespace comm {

inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// A 64-bit tag every rank of ONE launch derives from the same environment (so that ranks never pick up the id of another job):
// FNV-1a over whatever the launcher exports that is unique to the launch.  0 = nothing job-unique found.
inline uint64_t launch_nonce() {
    // (the restart counters: a launcher that restarts a crashed group under the same port / job id gives the new attempt another nonce,
    //  so the dead attempt's file -- should it have survived, see below -- is never taken for the new one's)
    static const char *const keys[] = {"ADMM_HIP_JOB_TAG", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "MASTER_ADDR", "MASTER_PORT", "SLURM_JOB_ID", "SLURM_STEP_ID",
                                       "SLURM_RESTART_COUNT", "OMPI_MCA_ess_base_jobid", "OMPI_MCA_orte_hnp_uri", "PMIX_NAMESPACE", "PMI_JOBID"};
    uint64_t h = 1469598103934665603ull; bool any = false;
    for (const char *k : keys) {
        const char *v = std::getenv(k);
        if (!v) continue;
        if (std::strcmp(k, "MASTER_ADDR") != 0 && std::strcmp(k, "TORCHELASTIC_RUN_ID") != 0 && std::strstr(k, "RESTART_COUNT") == nullptr) any = true;      // (an address / the default run id "none" alone tell no two jobs apart)
        for (const char *c = k; *c; ++c) { h ^= (unsigned char)*c; h *= 1099511628211ull; }
        for (const char *c = v; *c; ++c) { h ^= (unsigned char)*c; h *= 1099511628211ull; }
    }
    return any ? (h ? h : 1) : 0;
}

// rank 0: id holds the 128 bytes to publish; other ranks: id receives them.  false + *why on failure / timeout.
// File = 8 bytes magic, 8 bytes nonce (launch_nonce() unless the caller passes one), 128 bytes id.  Rank 0 removes whatever sits
// under the name first and creates its file exclusively (O_EXCL | O_NOFOLLOW, mode 0600: no symlink tricks in a shared /tmp),
// under a temporary name renamed into place.  The others take a file only if it is a regular file of this user with the right
// size, magic and nonce that was written no earlier than max_age_s before they began to wait (<= 0: any age) -- a leftover of a
// crashed earlier launch (same port, same tag, same restart count) is older than that and is ignored; the wait goes on until rank 0's
// fresh file appears.  The caller removes the file on rank 0 once the communicator exists (System.hpp setup_shard), so a leftover
// needs a crash between publishing and joining; keep max_age_s generous (default 600 s: a rank may reach its rendezvous minutes
// after rank 0 published -- staggered starts, a slow scene load -- and must still take the file).
inline bool rccl_id_via_file(const std::string &path, int rank, unsigned char id[128], double timeout_s, double max_age_s, std::string *why, uint64_t nonce = 0) {
    static const unsigned char kMagic[8] = {'A', 'D', 'M', 'M', 'r', 'c', 'c', '1'};
    if (path.empty()) { if (why) *why = "no rendezvous file given"; return false; }
    if (!nonce) nonce = launch_nonce();
    if (rank == 0) {
        (void)::unlink(path.c_str());
        const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
        (void)::unlink(tmp.c_str());
        const int fd = ::open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW, 0600);
        unsigned char buf[144];
        std::memcpy(buf, kMagic, 8); std::memcpy(buf + 8, &nonce, 8); std::memcpy(buf + 16, id, 128);
        if (fd < 0 || ::write(fd, buf, sizeof(buf)) != (ssize_t)sizeof(buf)) { if (fd >= 0) ::close(fd); if (why) *why = "cannot write " + tmp; return false; }
        ::close(fd);
        if (std::rename(tmp.c_str(), path.c_str()) != 0) { if (why) *why = "cannot rename " + tmp + " to " + path; return false; }
        return true;
    }
    const double t0 = now_s();
    const time_t wait_begin = ::time(nullptr);
    for (;;) {
        const int fd = ::open(path.c_str(), O_RDONLY | O_NOFOLLOW);
        if (fd >= 0) {
            struct stat st; unsigned char buf[144]; uint64_t got_nonce = 0;
            const bool shape = ::fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_uid == ::getuid() && st.st_size == (off_t)sizeof(buf) &&
                               (max_age_s <= 0.0 || std::difftime(wait_begin, st.st_mtime) <= max_age_s);
            const bool ok = shape && ::read(fd, buf, sizeof(buf)) == (ssize_t)sizeof(buf) && std::memcmp(buf, kMagic, 8) == 0 && (std::memcpy(&got_nonce, buf + 8, 8), got_nonce == nonce);
            ::close(fd);
            if (ok) { std::memcpy(id, buf + 16, 128); return true; }
        }
        if (now_s() - t0 > timeout_s) { if (why) *why = "timed out waiting for rank 0's id in " + path; return false; }
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
}

class ShmAllReduce {
public:
    ShmAllReduce() : rank_(0), world_(1), cap_(0), hdr_(nullptr), data_(nullptr), bytes_(0), sense_(0), timeout_s_(120.0) {}
    ~ShmAllReduce() { close(); }
    ShmAllReduce(const ShmAllReduce &) = delete;
    ShmAllReduce &operator=(const ShmAllReduce &) = delete;

    // name: "/something-unique-per-job"; capacity: doubles per rank and round (longer buffers go in rounds)
    bool open(const std::string &name, int rank, int world, size_t capacity = (size_t)1 << 20, double timeout_s = 120.0) {
        close();
        rank_ = rank; world_ = world; cap_ = capacity; timeout_s_ = timeout_s; name_ = name;
        bytes_ = sizeof(Header) + sizeof(double) * cap_ * (size_t)world_;
        sense_ = 0;
        const double t0 = now_s();
        for (;;) {      // (ranks > 0 come round again when the segment they attached to turns out to be a crashed run's leftover)
            int fd = -1; ino_t ino = 0;
            if (rank == 0) {
                // a leftover of a crashed run under the same name: mark it dead for whoever attached to it already, then remove the name
                const int old = ::shm_open(name.c_str(), O_RDWR, 0600);
                if (old >= 0) {
                    struct stat st;
                    if (::fstat(old, &st) == 0 && (size_t)st.st_size >= sizeof(Header)) {
                        void *q = ::mmap(nullptr, sizeof(Header), PROT_READ | PROT_WRITE, MAP_SHARED, old, 0);
                        if (q != MAP_FAILED) { Header *h = static_cast<Header *>(q); h->failed.store(1); h->magic.store(0); ::munmap(q, sizeof(Header)); }
                    }
                    ::close(old);
                }
                ::shm_unlink(name.c_str());
                fd = ::shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
                if (fd < 0 || ::ftruncate(fd, (off_t)bytes_) != 0) { if (fd >= 0) ::close(fd); return false; }
            } else {
                for (;;) {
                    fd = ::shm_open(name.c_str(), O_RDWR, 0600);
                    struct stat st;
                    if (fd >= 0 && ::fstat(fd, &st) == 0 && (size_t)st.st_size == bytes_) { ino = st.st_ino; break; }
                    if (fd >= 0) { ::close(fd); fd = -1; }
                    if (now_s() - t0 > timeout_s) return false;
                    std::this_thread::sleep_for(std::chrono::milliseconds(2));
                }
            }
            void *p = ::mmap(nullptr, bytes_, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            ::close(fd);
            if (p == MAP_FAILED) { if (rank == 0) ::shm_unlink(name.c_str()); return false; }
            hdr_ = static_cast<Header *>(p);
            data_ = reinterpret_cast<double *>(static_cast<char *>(p) + sizeof(Header));
            bool stale = false;
            if (rank == 0) {
                hdr_->count.store(0); hdr_->sense.store(0); hdr_->failed.store(0); hdr_->world = world;
                hdr_->magic.store(kMagic, std::memory_order_release);
            } else {
                double t_check = now_s();
                while (hdr_->magic.load(std::memory_order_acquire) != kMagic || hdr_->failed.load()) {
                    if (now_s() - t0 > timeout_s) { close(); return false; }
                    if (hdr_->failed.load() || now_s() - t_check > 0.02) {      // dead, or slow: is the name still this segment?
                        t_check = now_s();
                        const int again = ::shm_open(name.c_str(), O_RDWR, 0600);
                        struct stat st;
                        const bool same = again >= 0 && ::fstat(again, &st) == 0 && st.st_ino == ino;
                        if (again >= 0) ::close(again);
                        if (!same) { stale = true; break; }
                    }
                    std::this_thread::yield();
                }
                if (!stale && hdr_->world != world) { close(); return false; }
            }
            if (!stale) {
                const bool ok = barrier();       // everybody has mapped the segment: the name can go (nothing is left behind by a crash)
                if (rank == 0) ::shm_unlink(name.c_str());
                if (ok || rank == 0 || now_s() - t0 > timeout_s) { if (!ok) close(); return ok; }
                // a rank > 0 whose barrier failed: rank 0 of a NEW run has marked this (old) segment dead -- attach again
            }
            close(); sense_ = 0;
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
    }
    void close() {
        if (hdr_) { ::munmap(hdr_, bytes_); hdr_ = nullptr; data_ = nullptr; }
    }
    // buf[0..count) <- sum over the ranks, added in rank order on every rank (bitwise the same everywhere)
    bool allreduce(double *buf, int64_t count) {
        if (!hdr_) return false;
        for (int64_t off = 0; off < count; off += (int64_t)cap_) {
            const size_t n = (size_t)((count - off < (int64_t)cap_) ? count - off : (int64_t)cap_);
            std::memcpy(data_ + cap_ * (size_t)rank_, buf + off, sizeof(double) * n);
            if (!barrier()) return false;
            for (size_t i = 0; i < n; ++i) {
                double s = data_[i];
                for (int r = 1; r < world_; ++r) s += data_[cap_ * (size_t)r + i];
                buf[off + i] = s;
            }
            if (!barrier()) return false;       // nobody overwrites its slot before everybody has read it
        }
        return true;
    }
    // admm_hip_host_allreduce_fn
    static int hook(void *self, double *host_buf, int64_t count) { return static_cast<ShmAllReduce *>(self)->allreduce(host_buf, count) ? 0 : 1; }

private:
    static constexpr uint32_t kMagic = 0xADB17E55u;
    struct Header { std::atomic<uint32_t> magic; std::atomic<int> count, sense, failed; int world; char pad[44]; };
    int rank_, world_; size_t cap_; Header *hdr_; double *data_; size_t bytes_; int sense_; double timeout_s_; std::string name_;

    // sense-reversing barrier; a rank that gives up (a peer died) marks the segment failed so that the others give up too
    bool barrier() {
        sense_ ^= 1;
        if (hdr_->count.fetch_add(1, std::memory_order_acq_rel) == world_ - 1) {
            hdr_->count.store(0, std::memory_order_relaxed);
            hdr_->sense.store(sense_, std::memory_order_release);
            return hdr_->failed.load() == 0;
        }
        const double t0 = now_s();
        unsigned spins = 0;
        while (hdr_->sense.load(std::memory_order_acquire) != sense_) {
            if (hdr_->failed.load(std::memory_order_relaxed)) return false;
            if ((++spins & 1023u) == 0) {
                if (now_s() - t0 > timeout_s_) { hdr_->failed.store(1); return false; }
                std::this_thread::yield();
            }
        }
        return hdr_->failed.load() == 0;
    }
};

} // namespace comm
} // namespace admm
