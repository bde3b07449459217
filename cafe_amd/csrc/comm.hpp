// comm.hpp -- what one rank (one process per GPU, one node) needs to talk to the others, below the C ABI:
//
//   * a POSIX shared-memory control segment named by the caller's 128-byte id: rendezvous, a sense-reversing
//     barrier, a 64-byte mailbox per rank (IPC memory handles, the RCCL id) and an all-gather of HOST blocks of any
//     size through a second, per-call segment (the report phase: Monte-Carlo null, p-values, Viterbi sizes);
//   * the exchange buffer of the objective evaluation: a small UNCACHED device allocation per rank, mapped into every
//     other rank with hipIpc handles, into which the peers' score kernels store their packed (chunk sums,
//     first-zero index) rows directly over xGMI -- no collective launch, no host in the loop (k3_score<2>,
//     k_misc.hip); every link of the point-to-point fabric carries one rank's few hundred bytes at the same time;
//   * RCCL (resolved with dlopen when first needed) as the other exchange mode: ONE ncclAllGather of the same packed
//     rows on the context's stream -- the map + sum of cafe/lambda.cpp:698-722 either way.
//
//   * mode agreement: "mapped" is not "reachable".  After the mapping every rank runs a FUNCTIONAL probe (a one-
//     workgroup kernel stores a nonce into every peer's probe words and waits <= 1 s for theirs, cafehip_comm_init),
//     and the ranks agree through the mailboxes: direct only if EVERY rank saw EVERY peer, else RCCL if every rank
//     could join one communicator, else the communicator fails on every rank -- together, at set-up, never inside
//     the first evaluation (decide_mode).
//
// Nothing here touches family data: the exchange moves (chunks + 1) doubles per rank and evaluation.
#pragma once
#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

// (see "RCCL, resolved on demand" below: the real header, where the toolchain has one, checks the hand-declared ABI)
#if defined(__has_include)
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#define CAFEHIP_HAVE_RCCL_HEADER 1
#endif
#endif

namespace cafehip {

constexpr int kCommMaxWorld = 16;          // ranks of one node (xGMI: 8 GPUs)
constexpr int kCommSlotCap = 8192;         // chunk slots per rank the exchange buffer is sized for (2 M families per rank)
constexpr int kCommIdBytes = 128;
// a rank that never shows up fails the call instead of hanging it (CAFEHIP_COMM_TIMEOUT_S, read once, for tests)
inline double comm_timeout_s()
{
    static const double t = [] {
        const char* e = getenv("CAFEHIP_COMM_TIMEOUT_S");
        const double v = e ? atof(e) : 0.0;
        return v > 0 ? v : 120.0;
    }();
    return t;
}

// ---- RCCL, resolved on demand -----------------------------------------------------------------------------------
// The library is dlopen'ed (a build box or a one-GPU box need not have it), so the few types and constants of its ABI that
// cross the function pointers below are declared by hand -- and, wherever the toolchain ships <rccl/rccl.h> (this image does),
// the header itself is compiled in instead and the hand-written values are checked against it.
#ifdef CAFEHIP_HAVE_RCCL_HEADER
static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes in the hand-declared ABI (kCommIdBytes)");
static_assert((int)ncclSuccess == 0 && (int)ncclChar == 0 && (int)ncclDouble == 8, "hand-declared RCCL constants");
static_assert(sizeof(ncclDataType_t) == sizeof(int) && sizeof(ncclResult_t) == sizeof(int), "enums travel as int through the function pointers");
#else
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclChar = 0, ncclDouble = 8 };
#endif
static_assert(sizeof(ncclUniqueId) == 128, "the id travels in kCommIdBytes = 128 bytes");

struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*CommCount)(ncclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;

    bool load()
    {
        if (lib) return true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) {
            error = std::string("cannot load librccl: ") + dlerror();
            return false;
        }
        GetUniqueId = (decltype(GetUniqueId))dlsym(lib, "ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))dlsym(lib, "ncclCommInitRank");
        AllGather = (decltype(AllGather))dlsym(lib, "ncclAllGather");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        CommCount = (decltype(CommCount))dlsym(lib, "ncclCommCount");   // optional: what RCCL itself says its world is
        if (!GetUniqueId || !CommInitRank || !AllGather || !CommDestroy || !GetErrorString) {
            error = "librccl lacks an expected entry point";
            dlclose(lib);
            lib = nullptr;
            return false;
        }
        return true;
    }
};
inline RcclApi& rccl_api()
{
    static RcclApi api;
    return api;
}

// ---- exchange buffers: uncached device memory, pooled for the life of the process ----------------------------------
// Measured on MI355X / ROCm 7.2 (round 3): after hipFree of a hipDeviceMallocUncached allocation, a later ordinary
// hipMalloc that reuses the range returned slightly wrong data to kernels of an unrelated context (matrices of a later
// session off by 1e-8 in a few entries; tests/test_gpu_native_comm.py caught it).  A communicator's buffer is
// therefore never freed: close() hands it back to this pool and the next communicator on the device takes it.
#include <map>
#include <mutex>
inline std::mutex& xbuf_pool_mutex()
{
    static std::mutex m;
    return m;
}
inline std::map<int, std::vector<void*>>& xbuf_pool()
{
    static std::map<int, std::vector<void*>> pool;
    return pool;
}

// ---- control segment --------------------------------------------------------------------------------------------
struct CommControl {
    std::atomic<uint32_t> joined;             // ranks that have mapped the segment
    std::atomic<uint32_t> bar_count;
    std::atomic<uint32_t> bar_gen;
    std::atomic<uint32_t> gather_serial;      // per-call data segments: name suffix
    std::atomic<uint32_t> failed;             // a rank hit an error inside a collective step: everybody gives up
    std::atomic<uint32_t> fail_owner;         // 1 + the rank whose message stands in fail_msg (first failure wins)
    char fail_msg[160];
    uint32_t world;
    unsigned char mail[kCommMaxWorld][128];   // one mailbox per rank (IPC handle / RCCL id / flags)
};

class CommLink {
public:
    int rank = 0, world = 1, device = 0;
    std::string error;

    // exchange buffer of the objective evaluation: [2 parities][flags: kCommMaxWorld u64][rows: world x (slots + 1) f64]
    void* xbuf = nullptr;                      // mine (uncached device memory)
    void* peer_xbuf[kCommMaxWorld] = {};       // everyone's, mapped here (peer_xbuf[rank] == xbuf)
    bool p2p_ok = false;                       // every rank MAPPED every buffer (API success only)
    bool direct_ok = false;                    // ... and every rank's probe kernel SAW every peer's store: the agreed verdict
    bool probe_ran = false;
    int peers_mapped = 0, peers_seen = 0;      // this rank's own view (itself included)
    double probe_ms = 0.0;
    uint64_t nonce = 0;                        // same on every rank of the job (from the id)
    size_t xbuf_bytes = 0;
    // RCCL
    ncclComm_t rccl = nullptr;
    bool rccl_tried = false;
    int rccl_count = 0;                        // ncclCommCount of the live communicator (0: none)

    // [parity 0][parity 1][probe words: kCommMaxWorld u64]
    static uint64_t* probe_of(void* base) { return reinterpret_cast<uint64_t*>(static_cast<char*>(base) + 2 * parity_stride_bytes()); }
    static size_t total_bytes() { return 2 * parity_stride_bytes() + sizeof(uint64_t) * kCommMaxWorld; }
    static size_t parity_stride_bytes() { return sizeof(uint64_t) * kCommMaxWorld + sizeof(double) * (size_t)kCommMaxWorld * (kCommSlotCap + 1); }
    static uint64_t* flags_of(void* base, int parity) { return reinterpret_cast<uint64_t*>(static_cast<char*>(base) + parity * parity_stride_bytes()); }
    static double* rows_of(void* base, int parity) { return reinterpret_cast<double*>(flags_of(base, parity) + kCommMaxWorld); }

    ~CommLink() { close(); }

    bool fail(const std::string& m)
    {
        error = m;
        if (ctl_) {
            // the first failure names itself for the others: they report THIS message, not a barrier time-out
            uint32_t nobody = 0;
            if (ctl_->fail_owner.compare_exchange_strong(nobody, (uint32_t)rank + 1)) {
                snprintf(ctl_->fail_msg, sizeof ctl_->fail_msg, "%s", m.c_str());
                std::atomic_thread_fence(std::memory_order_release);
            }
            ctl_->failed.store(1);
        }
        return false;
    }

    // what another rank reported when it gave up (empty: nobody did)
    std::string peer_failure() const
    {
        if (!ctl_ || !ctl_->failed.load()) return "";
        const uint32_t who = ctl_->fail_owner.load();
        std::atomic_thread_fence(std::memory_order_acquire);
        if (!who) return "a rank failed";
        char buf[sizeof ctl_->fail_msg + 1];
        memcpy(buf, ctl_->fail_msg, sizeof ctl_->fail_msg);
        buf[sizeof ctl_->fail_msg] = 0;
        return "rank " + std::to_string(who - 1) + " failed: " + buf;
    }

    // names this communicator's id maps to under /dev/shm (control segment + per-call gather segments): a launcher
    // that had to kill its ranks removes what they could not (cafehip_comm_cleanup)
    static int unlink_names(const void* id)
    {
        uint64_t h = 1469598103934665603ull;
        for (int i = 0; i < kCommIdBytes; ++i) h = (h ^ static_cast<const unsigned char*>(id)[i]) * 1099511628211ull;
        char nm[96];
        snprintf(nm, sizeof nm, "/cafehip_%016llx", (unsigned long long)h);
        int n = shm_unlink(nm) == 0 ? 1 : 0;
        int misses = 0;
        for (uint32_t serial = 1; misses < 64; ++serial) {
            char g[128];
            snprintf(g, sizeof g, "%s_g%u", nm, serial);
            if (shm_unlink(g) == 0) {
                ++n;
                misses = 0;
            } else {
                ++misses;
            }
        }
        return n;
    }

    // Every rank of the job calls this with the same id (any 128 bytes unique to the job).
    bool init(int device_id, int rank_, int world_, const void* id)
    {
        rank = rank_;
        world = world_;
        device = device_id;
        if (world < 1 || world > kCommMaxWorld || rank < 0 || rank >= world) return fail("rank/world out of range (at most " + std::to_string(kCommMaxWorld) + " ranks of one node)");
        // name from the id
        uint64_t h = 1469598103934665603ull;
        for (int i = 0; i < kCommIdBytes; ++i) h = (h ^ static_cast<const unsigned char*>(id)[i]) * 1099511628211ull;
        char nm[64];
        snprintf(nm, sizeof nm, "/cafehip_%016llx", (unsigned long long)h);
        name_ = nm;
        nonce = (h << 8) | 0x80ull;   // low byte left for the rank; never 0
        int fd = shm_open(nm, O_CREAT | O_RDWR, 0600);
        if (fd < 0) return fail(std::string("shm_open ") + nm + ": " + strerror(errno));
        if (ftruncate(fd, sizeof(CommControl)) != 0) {
            ::close(fd);
            return fail(std::string("ftruncate ") + nm + ": " + strerror(errno));
        }
        void* p = mmap(nullptr, sizeof(CommControl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        ::close(fd);
        if (p == MAP_FAILED) return fail("mmap of the control segment failed");
        ctl_ = static_cast<CommControl*>(p);   // a fresh segment is zero-filled: valid initial state of every field
        if (rank == 0) ctl_->world = (uint32_t)world;
        ctl_->joined.fetch_add(1);
        if (!wait_until([&] { return ctl_->joined.load() >= (uint32_t)world; })) {
            shm_unlink(nm);   // nobody else will: do not leave the name behind
            return fail("rendezvous timed out: " + std::to_string(ctl_->joined.load()) + " of " + std::to_string(world) + " ranks arrived");
        }
        if (!barrier()) {
            shm_unlink(nm);
            return false;
        }
        if (rank == 0) shm_unlink(nm);   // everybody holds a mapping: the name can go
        return setup_p2p();
    }

    bool barrier()
    {
        if (world == 1) return true;
        const uint32_t gen = ctl_->bar_gen.load();
        if (ctl_->bar_count.fetch_add(1) + 1 == (uint32_t)world) {
            ctl_->bar_count.store(0);
            ctl_->bar_gen.fetch_add(1);
            return true;
        }
        if (!wait_until([&] { return ctl_->bar_gen.load() != gen; })) {
            const std::string other = peer_failure();
            return fail(other.empty() ? "barrier timed out (a rank died?)" : other);
        }
        return true;
    }

    // 128-byte mailboxes: everyone posts, barrier, everyone reads all
    bool exchange_mail(const void* mine, size_t n, unsigned char (*all)[128])
    {
        memset(ctl_->mail[rank], 0, 128);
        memcpy(ctl_->mail[rank], mine, n);
        if (!barrier()) return false;
        for (int r = 0; r < world; ++r) memcpy(all[r], ctl_->mail[r], 128);
        return barrier();   // nobody overwrites a mailbox somebody still reads
    }

    // All-gather of host blocks: rank r contributes nbytes_mine bytes, `all` receives world slots of nbytes_slot
    // bytes in rank order.  One shared segment per call (created by rank 0, unlinked once everyone has mapped it).
    bool host_allgather(const void* mine, size_t nbytes_mine, void* all, size_t nbytes_slot)
    {
        if (nbytes_mine > nbytes_slot) return fail("all-gather: a block of " + std::to_string(nbytes_mine) + " bytes does not fit its " + std::to_string(nbytes_slot) + "-byte slot");
        if (world == 1) {
            if (nbytes_mine) memcpy(all, mine, nbytes_mine);
            return true;
        }
        const size_t total = std::max<size_t>(nbytes_slot, 1) * world;
        uint32_t serial = 0;
        if (rank == 0) serial = ctl_->gather_serial.fetch_add(1) + 1;
        if (!barrier()) return false;
        serial = ctl_->gather_serial.load();
        char nm[96];
        snprintf(nm, sizeof nm, "%s_g%u", name_.c_str(), serial);
        int fd = -1;
        if (rank == 0) {
            fd = shm_open(nm, O_CREAT | O_RDWR, 0600);
            if (fd < 0 || ftruncate(fd, (off_t)total) != 0) {
                if (fd >= 0) ::close(fd);
                fail(std::string("gather segment: ") + strerror(errno));
            }
        }
        if (!barrier()) return false;
        if (ctl_->failed.load()) return fail(peer_failure());
        if (rank != 0) {
            fd = shm_open(nm, O_RDWR, 0600);
            if (fd < 0) fail(std::string("gather segment open: ") + strerror(errno));
        }
        char* p = nullptr;
        if (fd >= 0) {
            void* m = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            ::close(fd);
            if (m == MAP_FAILED) fail("gather segment mmap failed");
            else p = static_cast<char*>(m);
        }
        if (p && nbytes_mine) memcpy(p + (size_t)rank * nbytes_slot, mine, nbytes_mine);
        bool ok = barrier() && !ctl_->failed.load();
        if (ok && p) memcpy(all, p, nbytes_slot * world);
        ok = barrier() && ok;
        if (p) munmap(p, total);
        if (rank == 0) shm_unlink(nm);
        return ok && p != nullptr;
    }

    // RCCL communicator (collective: every rank calls it); the id travels through the mailboxes
    bool ensure_rccl()
    {
        if (rccl) return true;
        if (rccl_tried) return false;
        rccl_tried = true;
        RcclApi& api = rccl_api();
        unsigned char all[kCommMaxWorld][128];
        ncclUniqueId id;
        memset(&id, 0, sizeof id);
        int have = api.load() ? 1 : 0;
        if (have && rank == 0 && api.GetUniqueId(&id) != ncclSuccess) have = 0;
        unsigned char post[128];
        memset(post, 0, sizeof post);
        if (rank == 0) memcpy(post, &id, sizeof id);
        if (!exchange_mail(post, sizeof post, all)) return false;
        memcpy(&id, all[0], sizeof id);
        // everybody must be able to join, or nobody tries (ncclCommInitRank would wait for the missing rank)
        unsigned char ok1[128] = {(unsigned char)have};
        if (!exchange_mail(ok1, 1, all)) return false;
        for (int r = 0; r < world; ++r)
            if (!all[r][0]) {
                error = "RCCL not available on every rank: " + api.error;
                return false;
            }
        if (hipSetDevice(device) != hipSuccess) return fail("hipSetDevice failed");
        const int rc = api.CommInitRank(&rccl, world, id, rank);
        std::string why;
        if (rc != ncclSuccess) {
            rccl = nullptr;
            why = std::string("ncclCommInitRank: ") + api.GetErrorString(rc);
        }
        rccl_count = 0;
        if (rccl && api.CommCount && api.CommCount(rccl, &rccl_count) != ncclSuccess) rccl_count = 0;
        if (rccl && !api.CommCount) rccl_count = world;   // (an RCCL without the query: the init succeeded with `world`)
        // a communicator is usable only if EVERY rank holds one of the full size
        bool everyone = false;
        if (!all_ranks(rccl != nullptr && rccl_count == world, &everyone)) return false;
        if (!everyone) {
            if (rccl) api.CommDestroy(rccl);
            rccl = nullptr;
            rccl_count = 0;
            error = why.empty() ? "RCCL communicator could not be formed on every rank" : why;
            return false;
        }
        return true;
    }

    // collective AND: *result = every rank posted true
    bool all_ranks(bool mine, bool* result)
    {
        *result = mine;
        if (world == 1) return true;
        unsigned char post[128] = {(unsigned char)(mine ? 1 : 0)}, all[kCommMaxWorld][128];
        if (!exchange_mail(post, 1, all)) return false;
        for (int r = 0; r < world; ++r) *result = *result && all[r][0] != 0;
        return true;
    }

    // The exchange mode of the job, agreed by every rank at set-up (collective): 2 direct when every rank's probe saw
    // every peer; otherwise 1 rccl when `join_rccl` (collective itself: ensure_rccl) succeeds everywhere; otherwise 0 --
    // the same value on every rank, so either all carry on in one mode or all fail together.  -1: the control segment
    // itself failed (a rank died).
    template <class JoinRccl>
    int decide_mode(bool my_probe_ok, bool my_rccl_allowed, JoinRccl join_rccl)
    {
        bool all_direct = false;
        if (!all_ranks(my_probe_ok, &all_direct)) return -1;
        direct_ok = all_direct;
        if (all_direct) return 2;
        // RCCL is tried only if NO rank rules it out: joining it is collective (two mailbox rounds), so a rank-local switch
        // (option comm=direct, CAFEHIP_COMM in one process's environment) must not let the ranks run different numbers of rounds
        bool all_allow = false;
        if (!all_ranks(my_rccl_allowed, &all_allow)) return -1;
        if (!all_allow) return 0;
        const bool mine = join_rccl();
        bool all_rccl = false;
        if (!all_ranks(mine, &all_rccl)) return -1;
        return all_rccl ? 1 : 0;
    }

    void close()
    {
        for (int r = 0; r < kCommMaxWorld; ++r)
            if (peer_xbuf[r] && peer_xbuf[r] != xbuf) (void)hipIpcCloseMemHandle(peer_xbuf[r]);
        memset(peer_xbuf, 0, sizeof peer_xbuf);
        if (xbuf) {
            std::lock_guard<std::mutex> g(xbuf_pool_mutex());
            xbuf_pool()[device].push_back(xbuf);
        }
        xbuf = nullptr;
        if (rccl) rccl_api().CommDestroy(rccl);
        rccl = nullptr;
        rccl_count = 0;
        if (ctl_) munmap(ctl_, sizeof(CommControl));
        ctl_ = nullptr;
    }

private:
    CommControl* ctl_ = nullptr;
    std::string name_;

    template <class Pred>
    bool wait_until(Pred done)
    {
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (!done()) {
            if (ctl_->failed.load()) return false;
            if ((++spins & 1023) == 0) {
                sched_yield();
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > comm_timeout_s()) return false;
            }
        }
        return true;
    }

    // Allocate my exchange buffer (uncached: stores of the peers and my kernel's polling loads bypass the caches),
    // publish its IPC handle, map everyone else's.  p2p_ok only if EVERY rank mapped every buffer.
    bool setup_p2p()
    {
        xbuf_bytes = total_bytes();
        int good = device >= 0 ? 1 : 0;   // (device < 0: host-only link, cafehip_comm_host_selftest)
        if (good && hipSetDevice(device) != hipSuccess) {
            (void)hipGetLastError();
            good = 0;
        }
        if (good) {
            std::lock_guard<std::mutex> g(xbuf_pool_mutex());
            auto& pool = xbuf_pool()[device];
            if (!pool.empty()) {
                xbuf = pool.back();
                pool.pop_back();
            }
        }
        if (good && !xbuf && hipExtMallocWithFlags(&xbuf, xbuf_bytes, hipDeviceMallocUncached) != hipSuccess) {
            (void)hipGetLastError();
            xbuf = nullptr;
            good = 0;
        }
        if (good && hipMemset(xbuf, 0, xbuf_bytes) != hipSuccess) good = 0;
        if (good && hipDeviceSynchronize() != hipSuccess) good = 0;
        unsigned char post[128], all[kCommMaxWorld][128];
        memset(post, 0, sizeof post);
        hipIpcMemHandle_t h;
        static_assert(sizeof(hipIpcMemHandle_t) <= 120, "IPC handle must fit a mailbox");
        if (good && world > 1) {
            if (hipIpcGetMemHandle(&h, xbuf) != hipSuccess) {
                (void)hipGetLastError();
                good = 0;
            } else {
                memcpy(post + 8, &h, sizeof h);
            }
        }
        post[0] = (unsigned char)good;
        if (world == 1) {
            peer_xbuf[0] = xbuf;
            p2p_ok = good != 0;
            peers_mapped = good;
            return true;
        }
        if (!exchange_mail(post, sizeof post, all)) return false;
        for (int r = 0; r < world && good; ++r) {
            if (!all[r][0]) good = 0;
            else if (r == rank) peer_xbuf[r] = xbuf;
            else {
                hipIpcMemHandle_t hr;
                memcpy(&hr, all[r] + 8, sizeof hr);
                if (hipIpcOpenMemHandle(&peer_xbuf[r], hr, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
                    (void)hipGetLastError();
                    peer_xbuf[r] = nullptr;
                    good = 0;
                }
            }
        }
        for (int r = 0; r < world; ++r) peers_mapped += peer_xbuf[r] != nullptr;
        unsigned char ok1[128] = {(unsigned char)good};
        if (!exchange_mail(ok1, 1, all)) return false;
        p2p_ok = true;
        for (int r = 0; r < world; ++r) p2p_ok = p2p_ok && all[r][0];
        return true;
    }
};

}  // namespace cafehip
