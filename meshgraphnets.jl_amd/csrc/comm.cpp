// Engine-owned communicator: RCCL (run-time bound) and a shared-memory host transport.  See comm.h.
#include "comm.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>   // types and prototypes only: the library itself is bound with dlopen (see rccl_api)
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

namespace mgn {
namespace {

// ---------------------------------------------------------------------------------------------------------------------
// RCCL, bound at run time.  A host that already carries an RCCL (PyTorch-ROCm bundles librccl.so.1 next to its own HIP
// runtime) must not get a second copy with a second HIP runtime behind it; dlopen by soname returns the loaded one.
// ---------------------------------------------------------------------------------------------------------------------
struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

RcclApi* rccl_api(std::string& why) {
    static std::mutex mu;
    static RcclApi api;
    static std::string failed;
    std::lock_guard<std::mutex> lock(mu);
    if (api.lib) return &api;
    if (!failed.empty()) { why = failed; return nullptr; }
    const char* names[] = {getenv("MGN_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    for (const char* n : names) {
        if (!n || !*n) continue;
        lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (lib) break;
    }
    if (!lib) {
        failed = std::string("RCCL is not available: dlopen(librccl.so.1) failed: ") + (dlerror() ? dlerror() : "?");
        why = failed;
        return nullptr;
    }
    bool ok = true;
    auto bind = [&](auto& fn, const char* name) {
        fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(lib, name));
        if (!fn) { ok = false; failed = std::string("RCCL symbol missing: ") + name; }
    };
    bind(api.GetUniqueId, "ncclGetUniqueId");
    bind(api.CommInitRank, "ncclCommInitRank");
    bind(api.CommDestroy, "ncclCommDestroy");
    bind(api.GroupStart, "ncclGroupStart");
    bind(api.GroupEnd, "ncclGroupEnd");
    bind(api.Send, "ncclSend");
    bind(api.Recv, "ncclRecv");
    bind(api.AllGather, "ncclAllGather");
    bind(api.AllReduce, "ncclAllReduce");
    bind(api.GetErrorString, "ncclGetErrorString");
    if (!ok) { why = failed; dlclose(lib); return nullptr; }
    api.lib = lib;
    return &api;
}

#define HIPC(expr)                                                                                       \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess) {                                                                          \
            err = std::string(#expr) + " failed: " + hipGetErrorString(_e);                              \
            return -1;                                                                                   \
        }                                                                                                \
    } while (0)
// host transport: a failing rank also raises the shared abort flag, so that its peers leave their barrier at once
#define HIPC_ABORT(expr)                                                                                 \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess) {                                                                          \
            err = std::string(#expr) + " failed: " + hipGetErrorString(_e);                              \
            return fail_abort();                                                                         \
        }                                                                                                \
    } while (0)
#define NCCLC(expr)                                                                                      \
    do {                                                                                                 \
        ncclResult_t _r = (expr);                                                                        \
        if (_r != ncclSuccess) {                                                                         \
            err = std::string(#expr) + " failed: " + api->GetErrorString(_r);                            \
            return -1;                                                                                   \
        }                                                                                                \
    } while (0)

struct RcclComm final : Comm {
    RcclApi* api = nullptr;
    ncclComm_t comm = nullptr;
    hipStream_t cs = nullptr;            // communication stream
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;
    double* d_red = nullptr;             // 64 doubles of device scratch for allreduce_f64

    ~RcclComm() override {
        if (cs) (void)hipStreamSynchronize(cs);
        if (comm && api) (void)api->CommDestroy(comm);
        if (ev_ready) (void)hipEventDestroy(ev_ready);
        if (ev_done) (void)hipEventDestroy(ev_done);
        if (cs) (void)hipStreamDestroy(cs);
        if (d_red) (void)hipFree(d_red);
    }
    int init(const void* id, std::string& why) {
        api = rccl_api(why);
        if (!api) return -1;
        ncclUniqueId uid;
        static_assert(sizeof(uid) == COMM_ID_BYTES, "ncclUniqueId size");
        memcpy(&uid, id, sizeof uid);
        auto bad = [&](const std::string& s) { why = s; return -1; };
        if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) return bad("hipStreamCreate (communication stream) failed");
        if (hipEventCreateWithFlags(&ev_ready, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ev_done, hipEventDisableTiming) != hipSuccess)
            return bad("hipEventCreate failed");
        if (hipMalloc(reinterpret_cast<void**>(&d_red), 64 * sizeof(double)) != hipSuccess) return bad("hipMalloc failed");
        const ncclResult_t r = api->CommInitRank(&comm, nranks, uid, rank);
        if (r != ncclSuccess) return bad(std::string("ncclCommInitRank failed: ") + api->GetErrorString(r));
        return 0;
    }
    int a2a_start(const void* send, const size_t* sbytes, const size_t* soff, void* recv, const size_t* rbytes, const size_t* roff,
                  hipStream_t compute) override {
        HIPC(hipEventRecord(ev_ready, compute));
        HIPC(hipStreamWaitEvent(cs, ev_ready, 0));
        bool any = false;
        for (int q = 0; q < nranks; ++q) any = any || sbytes[q] || rbytes[q];
        if (any) {
            NCCLC(api->GroupStart());
            for (int q = 0; q < nranks; ++q) {
                if (q == rank) continue;
                if (sbytes[q]) NCCLC(api->Send(static_cast<const char*>(send) + soff[q], sbytes[q], ncclInt8, q, comm, cs));
                if (rbytes[q]) NCCLC(api->Recv(static_cast<char*>(recv) + roff[q], rbytes[q], ncclInt8, q, comm, cs));
            }
            NCCLC(api->GroupEnd());
            if (sbytes[rank] && sbytes[rank] == rbytes[rank])
                HIPC(hipMemcpyAsync(static_cast<char*>(recv) + roff[rank], static_cast<const char*>(send) + soff[rank], sbytes[rank],
                                    hipMemcpyDeviceToDevice, cs));
        }
        HIPC(hipEventRecord(ev_done, cs));
        return 0;
    }
    int a2a_finish(hipStream_t compute) override {
        HIPC(hipStreamWaitEvent(compute, ev_done, 0));
        return 0;
    }
    int a2a_host(const void*, const size_t*, const size_t*, void*, const size_t*, const size_t*) override {
        err = "host-memory exchange needs the MGN_COMM_HOST transport";
        return -1;
    }
    int allgather(const void* send, size_t bytes, void* recv, hipStream_t compute) override {
        NCCLC(api->AllGather(send, recv, bytes, ncclInt8, comm, compute));
        return 0;
    }
    int allreduce_f64(double* x, int n, int op, hipStream_t compute) override {
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int m = n - i0 < 64 ? n - i0 : 64;
            HIPC(hipMemcpyAsync(d_red, x + i0, m * sizeof(double), hipMemcpyHostToDevice, compute));
            NCCLC(api->AllReduce(d_red, d_red, m, ncclDouble, op == 1 ? ncclMax : ncclSum, comm, compute));
            HIPC(hipMemcpyAsync(x + i0, d_red, m * sizeof(double), hipMemcpyDeviceToHost, compute));
            HIPC(hipStreamSynchronize(compute));
        }
        return 0;
    }
    int barrier(hipStream_t compute) override {
        HIPC(hipStreamSynchronize(compute));
        double one = 1.0;
        return allreduce_f64(&one, 1, 0, compute);
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// Host transport: ranks of ONE node meet in POSIX shared memory.  Every collective is "publish my outgoing bytes in my own
// outbox -> barrier -> copy what is addressed to me out of every peer's outbox -> barrier".
// ---------------------------------------------------------------------------------------------------------------------
constexpr int HOST_MAX_RANKS = 64;
constexpr uint32_t HOST_MAGIC = 0x4D474E48u;   // "MGNH"

struct ShmCtl {
    std::atomic<int32_t> bar_count, bar_gen, abort_flag, attached;
    struct Slot {
        std::atomic<uint64_t> cap;
        std::atomic<uint32_t> gen;
        char pad[52];
    } slot[HOST_MAX_RANKS];
};

double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

struct HostComm final : Comm {
    std::string name;       // "/mgn_<hex>"
    ShmCtl* ctl = nullptr;
    bool device_ok = false;
    double timeout_s = 120.0;
    struct Box { char* p = nullptr; size_t cap = 0; uint32_t gen = 0; };
    std::vector<Box> box;   // box[rank] is my outbox (writable); the others are read-only maps of the peers' outboxes
    // staging (pinned when a device is present)
    char *stage_s = nullptr, *stage_r = nullptr;
    size_t cap_s = 0, cap_r = 0;
    // pending device exchange (between a2a_start and a2a_finish)
    void* pend_recv = nullptr;
    std::vector<size_t> pend_rbytes, pend_roff;

    ~HostComm() override {
        for (int q = 0; q < (int)box.size(); ++q)
            if (box[q].p) munmap(box[q].p, box[q].cap);
        if (!box.empty() && box[rank].gen) shm_unlink(box_name(rank, box[rank].gen).c_str());
        if (ctl) {
            if (ctl->attached.fetch_sub(1) == 1) shm_unlink(name.c_str());   // last one out (no-op after the early unlink)
            munmap(ctl, sizeof(ShmCtl));
        }
        free_stage(stage_s);
        free_stage(stage_r);
    }
    void free_stage(char* p) {
        if (!p) return;
        if (device_ok) (void)hipHostFree(p);
        else free(p);
    }
    int ensure_stage(char*& p, size_t& cap, size_t need) {
        if (need <= cap) return 0;
        free_stage(p);
        p = nullptr;
        cap = 0;
        const size_t n = need + need / 2 + 4096;
        if (device_ok) {
            if (hipHostMalloc(reinterpret_cast<void**>(&p), n, hipHostMallocDefault) != hipSuccess) { err = "hipHostMalloc failed"; p = nullptr; return -1; }
        } else {
            p = static_cast<char*>(malloc(n));
            if (!p) { err = "host allocation failed"; return -1; }
        }
        cap = n;
        return 0;
    }
    std::string box_name(int r, uint32_t gen) const {
        char b[96];
        snprintf(b, sizeof b, "%s_o%d_%u", name.c_str(), r, gen);
        return b;
    }
    int init(const void* id, std::string& why) {
        const unsigned char* u = static_cast<const unsigned char*>(id);
        uint32_t magic;
        memcpy(&magic, u, 4);
        if (magic != HOST_MAGIC) { why = "the communicator id was not made for the MGN_COMM_HOST transport"; return -1; }
        if (nranks > HOST_MAX_RANKS) { why = "MGN_COMM_HOST supports up to 64 ranks"; return -1; }
        char hex[40];
        for (int i = 0; i < 12; ++i) snprintf(hex + 2 * i, 3, "%02x", u[4 + i]);
        name = std::string("/mgn_") + hex;
        if (const char* e = getenv("MGN_COMM_TIMEOUT_S")) timeout_s = atof(e);
        const int fd = shm_open(name.c_str(), O_CREAT | O_RDWR, 0600);
        if (fd < 0) { why = "shm_open failed for the communicator's control segment"; return -1; }
        if (ftruncate(fd, sizeof(ShmCtl)) != 0) { close(fd); why = "ftruncate failed on the control segment"; return -1; }
        void* p = mmap(nullptr, sizeof(ShmCtl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) { why = "mmap failed on the control segment"; return -1; }
        ctl = static_cast<ShmCtl*>(p);     // a fresh segment is zero-filled: a valid initial state of every field
        ctl->attached.fetch_add(1);
        box.assign(nranks, Box());
        if (wait_barrier() != 0) { why = err; return -1; }
        if (rank == 0) shm_unlink(name.c_str());   // everyone is mapped: the name can go (nothing is left behind on a crash)
        return 0;
    }
    int wait_barrier() {
        const int32_t g = ctl->bar_gen.load(std::memory_order_acquire);
        if (ctl->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == nranks) {
            ctl->bar_count.store(0, std::memory_order_relaxed);
            ctl->bar_gen.store(g + 1, std::memory_order_release);
            return 0;
        }
        const double t0 = now_s();
        int spins = 0;
        while (ctl->bar_gen.load(std::memory_order_acquire) == g) {
            if (ctl->abort_flag.load(std::memory_order_relaxed)) { err = "a peer rank aborted the exchange"; return -1; }
            if (++spins > 200) {
                sched_yield();
                if ((spins & 1023) == 0 && now_s() - t0 > timeout_s) {
                    ctl->abort_flag.store(1);
                    err = "timed out waiting for the peer ranks (MGN_COMM_TIMEOUT_S)";
                    return -1;
                }
            }
        }
        return 0;
    }
    // outbox layout: uint64 bcast, uint64 off[nranks + 1], payload
    size_t header_bytes() const { return 8 * (size_t)(nranks + 2); }
    int publish(const void* send, const size_t* sbytes, const size_t* soff, bool bcast, size_t bcast_bytes) {
        size_t total = 0;
        if (bcast) total = bcast_bytes;
        else for (int q = 0; q < nranks; ++q) total += sbytes[q];
        const size_t need = header_bytes() + total;
        Box& b = box[rank];
        if (need > b.cap) {
            if (b.p) { munmap(b.p, b.cap); shm_unlink(box_name(rank, b.gen).c_str()); }
            b.p = nullptr;
            const uint32_t gen = b.gen + 1;
            size_t cap = need + need / 2;
            if (cap < (1u << 16)) cap = 1u << 16;
            const std::string nm = box_name(rank, gen);
            const int fd = shm_open(nm.c_str(), O_CREAT | O_RDWR | O_TRUNC, 0600);
            if (fd < 0) { err = "shm_open failed for an outbox"; return fail_abort(); }
            if (ftruncate(fd, (off_t)cap) != 0) { close(fd); err = "ftruncate failed on an outbox (is /dev/shm full?)"; return fail_abort(); }
            void* p = mmap(nullptr, cap, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            close(fd);
            if (p == MAP_FAILED) { err = "mmap failed on an outbox"; return fail_abort(); }
            b.p = static_cast<char*>(p);
            b.cap = cap;
            b.gen = gen;
            ctl->slot[rank].cap.store(cap, std::memory_order_relaxed);
            ctl->slot[rank].gen.store(gen, std::memory_order_release);
        }
        uint64_t* hd = reinterpret_cast<uint64_t*>(b.p);
        hd[0] = bcast ? 1 : 0;
        char* pay = b.p + header_bytes();
        if (bcast) {
            hd[1] = 0;
            hd[2] = bcast_bytes;
            memcpy(pay, send, bcast_bytes);
        } else {
            size_t o = 0;
            for (int q = 0; q < nranks; ++q) {
                hd[1 + q] = o;
                if (sbytes[q]) memcpy(pay + o, static_cast<const char*>(send) + soff[q], sbytes[q]);
                o += sbytes[q];
            }
            hd[1 + nranks] = o;
        }
        std::atomic_thread_fence(std::memory_order_release);
        return 0;
    }
    int fail_abort() {
        if (ctl) ctl->abort_flag.store(1);
        return -1;
    }
    // map peer `src`'s current outbox (read-only)
    int peer_box(int src, const char** p) {
        if (src == rank) { *p = box[rank].p; return 0; }
        const uint32_t gen = ctl->slot[src].gen.load(std::memory_order_acquire);
        Box& b = box[src];
        if (b.gen != gen || !b.p) {
            if (b.p) munmap(b.p, b.cap);
            b.p = nullptr;
            const size_t cap = ctl->slot[src].cap.load(std::memory_order_relaxed);
            const int fd = shm_open(box_name(src, gen).c_str(), O_RDONLY, 0600);
            if (fd < 0) { err = "shm_open failed for a peer's outbox"; return fail_abort(); }
            void* m = mmap(nullptr, cap, PROT_READ, MAP_SHARED, fd, 0);
            close(fd);
            if (m == MAP_FAILED) { err = "mmap failed on a peer's outbox"; return fail_abort(); }
            b.p = static_cast<char*>(m);
            b.cap = cap;
            b.gen = gen;
        }
        *p = b.p;
        return 0;
    }
    int collect(void* recv, const size_t* rbytes, const size_t* roff) {
        std::atomic_thread_fence(std::memory_order_acquire);
        for (int q = 0; q < nranks; ++q) {
            if (!rbytes[q]) continue;
            const char* p = nullptr;
            if (peer_box(q, &p) != 0) return -1;
            const uint64_t* hd = reinterpret_cast<const uint64_t*>(p);
            const size_t o = hd[0] ? 0 : hd[1 + rank], len = hd[0] ? hd[2] : hd[2 + rank] - hd[1 + rank];
            if (len != rbytes[q]) {
                err = "halo exchange size mismatch between ranks (did every rank call mgn_set_graph with the same mesh?)";
                return fail_abort();
            }
            memcpy(static_cast<char*>(recv) + roff[q], p + header_bytes() + o, len);
        }
        return 0;
    }
    int a2a_host(const void* send, const size_t* sbytes, const size_t* soff, void* recv, const size_t* rbytes, const size_t* roff) override {
        if (publish(send, sbytes, soff, false, 0) != 0) return fail_abort();
        if (wait_barrier() != 0) return fail_abort();
        if (collect(recv, rbytes, roff) != 0) return fail_abort();
        return wait_barrier() != 0 ? fail_abort() : 0;
    }
    static size_t span(const size_t* bytes, const size_t* off, int n) {
        size_t m = 0;
        for (int q = 0; q < n; ++q)
            if (bytes[q] && off[q] + bytes[q] > m) m = off[q] + bytes[q];
        return m;
    }
    int a2a_start(const void* send, const size_t* sbytes, const size_t* soff, void* recv, const size_t* rbytes, const size_t* roff,
                  hipStream_t compute) override {
        pend_recv = nullptr;                         // a failed start must not leave the previous exchange's target behind
        pend_rbytes.clear();
        pend_roff.clear();
        const size_t ns = span(sbytes, soff, nranks);
        if (ensure_stage(stage_s, cap_s, ns) != 0) return fail_abort();
        if (ns) HIPC_ABORT(hipMemcpyAsync(stage_s, send, ns, hipMemcpyDeviceToHost, compute));
        HIPC_ABORT(hipStreamSynchronize(compute));     // also: the H2D of the previous exchange has left stage_r
        if (publish(stage_s, sbytes, soff, false, 0) != 0) return fail_abort();
        if (wait_barrier() != 0) return fail_abort();
        pend_recv = recv;
        pend_rbytes.assign(rbytes, rbytes + nranks);
        pend_roff.assign(roff, roff + nranks);
        return 0;
    }
    int a2a_finish(hipStream_t compute) override {
        if (!pend_recv && pend_rbytes.empty()) { err = "a2a_finish without a pending a2a_start"; return fail_abort(); }
        const size_t nr = span(pend_rbytes.data(), pend_roff.data(), nranks);
        if (ensure_stage(stage_r, cap_r, nr) != 0) return fail_abort();
        if (collect(stage_r, pend_rbytes.data(), pend_roff.data()) != 0) return fail_abort();
        if (wait_barrier() != 0) return fail_abort();
        for (int q = 0; q < nranks; ++q)
            if (pend_rbytes[q])
                HIPC_ABORT(hipMemcpyAsync(static_cast<char*>(pend_recv) + pend_roff[q], stage_r + pend_roff[q], pend_rbytes[q], hipMemcpyHostToDevice, compute));
        pend_recv = nullptr;
        pend_rbytes.clear();
        pend_roff.clear();
        return 0;
    }
    int allgather(const void* send, size_t bytes, void* recv, hipStream_t compute) override {
        if (ensure_stage(stage_s, cap_s, bytes) != 0 || ensure_stage(stage_r, cap_r, bytes * nranks) != 0) return fail_abort();
        if (bytes) HIPC_ABORT(hipMemcpyAsync(stage_s, send, bytes, hipMemcpyDeviceToHost, compute));
        HIPC_ABORT(hipStreamSynchronize(compute));
        if (publish(stage_s, nullptr, nullptr, true, bytes) != 0) return fail_abort();
        if (wait_barrier() != 0) return fail_abort();
        std::vector<size_t> rb(nranks, bytes), ro(nranks);
        for (int q = 0; q < nranks; ++q) ro[q] = (size_t)q * bytes;
        if (collect(stage_r, rb.data(), ro.data()) != 0) return fail_abort();
        if (wait_barrier() != 0) return fail_abort();
        if (bytes) HIPC_ABORT(hipMemcpyAsync(recv, stage_r, bytes * nranks, hipMemcpyHostToDevice, compute));
        HIPC_ABORT(hipStreamSynchronize(compute));
        return 0;
    }
    int allreduce_f64(double* x, int n, int op, hipStream_t) override {
        const size_t bytes = (size_t)n * sizeof(double);
        if (publish(x, nullptr, nullptr, true, bytes) != 0) return -1;
        if (wait_barrier() != 0) return -1;
        std::vector<double> all((size_t)n * nranks);
        std::vector<size_t> rb(nranks, bytes), ro(nranks);
        for (int q = 0; q < nranks; ++q) ro[q] = (size_t)q * bytes;
        if (collect(all.data(), rb.data(), ro.data()) != 0) return -1;
        if (wait_barrier() != 0) return -1;
        for (int i = 0; i < n; ++i) {                       // rank order: the same bits on every rank
            double a = all[i];
            for (int q = 1; q < nranks; ++q) {
                const double v = all[(size_t)q * n + i];
                a = op == 1 ? (v > a ? v : a) : a + v;
            }
            x[i] = a;
        }
        return 0;
    }
    int barrier(hipStream_t compute) override {
        if (device_ok) HIPC(hipStreamSynchronize(compute));
        return wait_barrier();
    }
};

}  // namespace

int comm_unique_id(void* id, int transport, std::string& why) {
    memset(id, 0, COMM_ID_BYTES);
    if (transport == 0) {
        RcclApi* api = rccl_api(why);
        if (!api) return -1;
        ncclUniqueId uid;
        const ncclResult_t r = api->GetUniqueId(&uid);
        if (r != ncclSuccess) { why = std::string("ncclGetUniqueId failed: ") + api->GetErrorString(r); return -1; }
        memcpy(id, &uid, sizeof uid);
        return 0;
    }
    if (transport == 1) {
        unsigned char* u = static_cast<unsigned char*>(id);
        memcpy(u, &HOST_MAGIC, 4);
        bool ok = false;
        if (FILE* f = fopen("/dev/urandom", "rb")) {
            ok = fread(u + 4, 1, 12, f) == 12;
            fclose(f);
        }
        if (!ok) {
            uint64_t x = (uint64_t)getpid() * 0x9E3779B97F4A7C15ull ^ (uint64_t)(now_s() * 1e9);
            memcpy(u + 4, &x, 8);
            x = x * 0xBF58476D1CE4E5B9ull + 12345;
            memcpy(u + 12, &x, 4);
        }
        return 0;
    }
    why = "unknown transport";
    return -1;
}

Comm* comm_create(const void* id, int transport, int rank, int nranks, bool device_ok, std::string& why) {
    if (transport == 0) {
        if (!device_ok) { why = "the RCCL transport needs a device handle"; return nullptr; }
        RcclComm* c = new (std::nothrow) RcclComm();
        if (!c) { why = "host allocation failed"; return nullptr; }
        c->rank = rank; c->nranks = nranks; c->transport = 0;
        if (c->init(id, why) != 0) { delete c; return nullptr; }
        return c;
    }
    if (transport == 1) {
        HostComm* c = new (std::nothrow) HostComm();
        if (!c) { why = "host allocation failed"; return nullptr; }
        c->rank = rank; c->nranks = nranks; c->transport = 1; c->device_ok = device_ok;
        if (c->init(id, why) != 0) { delete c; return nullptr; }
        return c;
    }
    why = "unknown transport";
    return nullptr;
}

}  // namespace mgn
