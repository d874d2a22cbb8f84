// Transport of the time-parallel hand-over behind the C-ABI: the forward transfer uend -> u[0] of the next time rank
// (controller_MPI.py:218-305 send_full / recv_full, mesh.py:85-125 isend / irecv / bcast) and the end-of-block broadcast
// (controller_MPI.py:125-130), modelled on the reference's NCCL wrapper (helpers/NCCL_communicator.py:12-20: unique id
// from rank 0, one communicator per process; :128-135 Bcast).
//
// Two wires carry the same protocol:
//   * RCCL over xGMI (one process per GPU).  librccl is bound at run time (dlopen) the first time a communicator is asked
//     for: single-GPU runs never load it, and a process in which torch.distributed has already loaded its librccl shares
//     that copy.
//   * host mailboxes in POSIX shared memory ("shm:" ids): ranks that cannot have an RCCL communicator - several ranks on
//     ONE GPU (RCCL refuses duplicate devices), ranks that are threads of one process (tests), or no GPU at all (the
//     protocol's piece arithmetic is exercised on plain host buffers by sdc_comm_selftest in the CPU suite).  Same calls,
//     same ordering against the engine's stream; the data makes a round trip through host memory.
// Messages travel on a stream of their own: a send waits (on the device) for the event that marks UEND complete, a
// receive lands in an inbox and is handed to the level through sdc_replace_u0 on the engine's stream once it has
// arrived, so residual passes and messages overlap and the host never blocks (RCCL wire).
#pragma once
#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <memory>
#include <thread>

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi* rccl_api(std::string* why) {
    static RcclApi api;
    static std::mutex guard;
    std::lock_guard<std::mutex> lock(guard);
    if (api.handle) return &api;
    const char* names[] = {"librccl.so", "librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names)  // a copy that is already in the process (torch.distributed's) wins
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    for (const char* n : {"librccl.so.1", "librccl.so"})
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        if (why) *why = std::string("cannot load librccl: ") + dlerror();
        return nullptr;
    }
#define BIND(field, sym)                                                      \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(h, sym));          \
    if (!api.field) {                                                          \
        if (why) *why = std::string("librccl does not export ") + sym;         \
        return nullptr;                                                        \
    }
    BIND(GetUniqueId, "ncclGetUniqueId")
    BIND(CommInitRank, "ncclCommInitRank")
    BIND(CommDestroy, "ncclCommDestroy")
    BIND(Send, "ncclSend")
    BIND(Recv, "ncclRecv")
    BIND(Broadcast, "ncclBroadcast")
    BIND(GroupStart, "ncclGroupStart")
    BIND(GroupEnd, "ncclGroupEnd")
    BIND(GetErrorString, "ncclGetErrorString")
#undef BIND
    api.handle = h;
    return &api;
}

// ------------------------------------------------------------------------------------------------------------------
// wires
// ------------------------------------------------------------------------------------------------------------------
// One rank's end of a communicator.  Point-to-point operations are issued between group_begin / group_end (one group =
// operations that must progress together, ncclGroupStart / End); everything is ordered on `stream`.  `c` only receives
// error texts (may be null).
struct Wire {
    int rank = 0, size = 1;
    hipStream_t stream = nullptr;  // device wires: where the messages are ordered
    size_t chunk = 0;              // > 0: messages are cut into pieces of this many doubles (one group)
    virtual ~Wire() {}
    virtual const char* kind() const = 0;
    virtual int group_begin(sdc_ctx* c) = 0;
    virtual int send(sdc_ctx* c, const double* buf, size_t n, int peer) = 0;
    virtual int recv(sdc_ctx* c, double* buf, size_t n, int peer) = 0;
    virtual int group_end(sdc_ctx* c) = 0;
    virtual int bcast(sdc_ctx* c, double* buf, size_t n, int root) = 0;
    virtual int sync(sdc_ctx* c) = 0;  // host waits for everything posted so far
};

struct RcclWire : Wire {
    RcclApi* api = nullptr;
    ncclComm_t comm = nullptr;
    ncclResult_t first_err = ncclSuccess;
    const char* kind() const override { return "rccl"; }
    ~RcclWire() override {
        if (stream) (void)hipStreamSynchronize(stream);
        if (comm && api) (void)api->CommDestroy(comm);
        if (stream) (void)hipStreamDestroy(stream);
    }
    int group_begin(sdc_ctx* c) override {
        first_err = ncclSuccess;
        ncclResult_t r = api->GroupStart();
        if (r != ncclSuccess) return fail(c, SDC_ERR_COMM, "ncclGroupStart: %s", api->GetErrorString(r));
        return SDC_OK;
    }
    int send(sdc_ctx*, const double* buf, size_t n, int peer) override {
        const size_t step = chunk ? chunk : n;
        for (size_t o = 0; o < n && first_err == ncclSuccess; o += step)
            first_err = api->Send(buf + o, std::min(step, n - o), ncclDouble, peer, comm, stream);
        return SDC_OK;  // (reported by group_end: the group has to be closed whatever happened inside)
    }
    int recv(sdc_ctx*, double* buf, size_t n, int peer) override {
        const size_t step = chunk ? chunk : n;
        for (size_t o = 0; o < n && first_err == ncclSuccess; o += step)
            first_err = api->Recv(buf + o, std::min(step, n - o), ncclDouble, peer, comm, stream);
        return SDC_OK;
    }
    int group_end(sdc_ctx* c) override {
        ncclResult_t re = api->GroupEnd();
        if (first_err != ncclSuccess || re != ncclSuccess)
            return fail(c, SDC_ERR_COMM, "send/recv group: %s", api->GetErrorString(first_err != ncclSuccess ? first_err : re));
        return SDC_OK;
    }
    int bcast(sdc_ctx* c, double* buf, size_t n, int root) override {
        ncclResult_t r = api->Broadcast(buf, buf, n, ncclDouble, root, comm, stream);
        if (r != ncclSuccess) return fail(c, SDC_ERR_COMM, "ncclBroadcast: %s", api->GetErrorString(r));
        return SDC_OK;
    }
    int sync(sdc_ctx* c) override {
        HIPCHK(c, hipStreamSynchronize(stream));
        return SDC_OK;
    }
};

// Mailboxes in POSIX shared memory: one single-slot box per ordered pair (src, dst), file /dev/shm/<job>.<src>.<dst>,
// created by whoever gets there first (a fresh file is zero-filled: both counters start at 0).  The sender waits until the
// box is free (consumed == posted), copies its data in and bumps `posted`; the receiver waits for posted > consumed,
// copies out, bumps `consumed`.  A group collects its operations and carries them out at group_end - all sends, then
// all receives - so that the ranks of a group never wait for each other in a cycle (a send only waits for the
// receiver to have taken the PREVIOUS message of that pair, which it does in a group it entered earlier).  Blocks the
// host (that is what makes it a rehearsal wire, not a fast one); waits end with SDC_ERR_COMM after `timeout_s`.
struct ShmWire : Wire {
    struct Head {
        std::atomic<unsigned long long> posted, consumed;
        unsigned long long bytes;
        char pad[40];
    };
    struct Box {
        int fd = -1;
        unsigned char* base = nullptr;
        size_t map_bytes = 0;
        Head* head() const { return reinterpret_cast<Head*>(base); }
        unsigned char* data() const { return base + sizeof(Head); }
    };
    struct Op {
        bool is_send;
        double* buf;
        size_t n;
        int peer;
    };
    std::string job;
    size_t cap = 0;        // largest message, doubles
    bool device = true;    // buffers are device memory (hipMemcpy through `stream`); false: plain host memory
    double timeout_s = 120.0;
    std::map<std::pair<int, int>, Box> boxes;
    std::vector<Op> ops;
    const char* kind() const override { return "shm"; }
    static_assert(sizeof(Head) == 64, "mailbox header");

    std::string path(int src, int dst) const { return "/" + job + "." + std::to_string(src) + "." + std::to_string(dst); }
    ~ShmWire() override {
        if (device && stream) (void)hipStreamSynchronize(stream);
        for (auto& kv : boxes) {
            // a box this rank filled is only removed once its message has been taken: the receiver may not even have
            // opened it yet, and would otherwise create an empty one of the same name and wait for ever
            if (kv.first.first == rank && kv.second.base) {
                Head* h = kv.second.head();
                for (int spin = 0; spin < 100000 && h->consumed.load(std::memory_order_acquire) != h->posted.load(); ++spin)
                    usleep(50);
            }
            if (kv.second.base) munmap(kv.second.base, kv.second.map_bytes);
            if (kv.second.fd >= 0) close(kv.second.fd);
            shm_unlink(path(kv.first.first, kv.first.second).c_str());  // (the peer's mapping stays valid until it goes too)
        }
        if (device && stream) (void)hipStreamDestroy(stream);
    }
    int box(sdc_ctx* c, int src, int dst, Box** out) {
        auto key = std::make_pair(src, dst);
        auto it = boxes.find(key);
        if (it == boxes.end()) {
            Box b;
            const std::string p = path(src, dst);
            b.fd = shm_open(p.c_str(), O_CREAT | O_RDWR, 0600);
            if (b.fd < 0) return fail(c, SDC_ERR_COMM, "shm_open(%s): %s", p.c_str(), strerror(errno));
            b.map_bytes = sizeof(Head) + cap * sizeof(double);
            if (ftruncate(b.fd, (off_t)b.map_bytes) != 0) {
                close(b.fd);
                return fail(c, SDC_ERR_COMM, "ftruncate(%s, %zu): %s", p.c_str(), b.map_bytes, strerror(errno));
            }
            void* m = mmap(nullptr, b.map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, b.fd, 0);
            if (m == MAP_FAILED) {
                close(b.fd);
                return fail(c, SDC_ERR_COMM, "mmap(%s): %s", p.c_str(), strerror(errno));
            }
            b.base = static_cast<unsigned char*>(m);
            it = boxes.emplace(key, b).first;
        }
        *out = &it->second;
        return SDC_OK;
    }
    template <class Pred>
    int wait_for(sdc_ctx* c, Pred ready, const char* what, int peer) {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spin = 0; !ready(); ++spin) {
            if (spin < 2000) sched_yield();
            else usleep(50);
            if ((spin & 1023) == 1023 &&
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
                return fail(c, SDC_ERR_COMM, "rank %d: %s rank %d timed out after %.0f s (shm wire %s)", rank, what, peer,
                            timeout_s, job.c_str());
        }
        return SDC_OK;
    }
    int copy(sdc_ctx* c, void* dst, const void* src, size_t bytes, bool to_host) {
        if (!device) {
            memcpy(dst, src, bytes);
            return SDC_OK;
        }
        HIPCHK(c, hipMemcpyAsync(dst, src, bytes, to_host ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice, stream));
        return SDC_OK;
    }
    int drain(sdc_ctx* c) {
        if (device) HIPCHK(c, hipStreamSynchronize(stream));
        return SDC_OK;
    }
    int put(sdc_ctx* c, const double* buf, size_t n, int dst) {
        if (n > cap) return fail(c, SDC_ERR_COMM, "message of %zu doubles exceeds the mailbox (%zu)", n, cap);
        Box* b;
        int rc = box(c, rank, dst, &b);
        if (rc != SDC_OK) return rc;
        Head* h = b->head();
        rc = wait_for(c, [h] { return h->consumed.load(std::memory_order_acquire) == h->posted.load(std::memory_order_relaxed); },
                      "waiting for the previous message to be taken by", dst);
        if (rc != SDC_OK) return rc;
        h->bytes = n * sizeof(double);
        return copy(c, b->data(), buf, n * sizeof(double), true);
    }
    int take(sdc_ctx* c, double* buf, size_t n, int src) {
        Box* b;
        int rc = box(c, src, rank, &b);
        if (rc != SDC_OK) return rc;
        Head* h = b->head();
        rc = wait_for(c, [h] { return h->posted.load(std::memory_order_acquire) > h->consumed.load(std::memory_order_relaxed); },
                      "waiting for a message from", src);
        if (rc != SDC_OK) return rc;
        if (h->bytes != n * sizeof(double))
            return fail(c, SDC_ERR_COMM, "rank %d expected %zu bytes from rank %d, the message has %llu", rank,
                        n * sizeof(double), src, h->bytes);
        return copy(c, buf, b->data(), n * sizeof(double), false);
    }
    int group_begin(sdc_ctx*) override {
        ops.clear();
        return SDC_OK;
    }
    int send(sdc_ctx*, const double* buf, size_t n, int peer) override {
        ops.push_back(Op{true, const_cast<double*>(buf), n, peer});
        return SDC_OK;
    }
    int recv(sdc_ctx*, double* buf, size_t n, int peer) override {
        ops.push_back(Op{false, buf, n, peer});
        return SDC_OK;
    }
    int group_end(sdc_ctx* c) override {
        int rc = SDC_OK;
        // single-slot boxes: one message per ordered pair and group (what every pattern above it sends); a second one would
        // wait for a receiver that only takes messages after its own sends
        for (size_t i = 0; i < ops.size(); ++i)
            for (size_t k = i + 1; k < ops.size(); ++k)
                if (ops[i].is_send == ops[k].is_send && ops[i].peer == ops[k].peer) {
                    ops.clear();
                    return fail(c, SDC_ERR_COMM, "shm wire: two messages for one peer in one group");
                }
        for (const Op& o : ops)
            if (o.is_send && (rc = put(c, o.buf, o.n, o.peer)) != SDC_OK) return rc;
        if ((rc = drain(c)) != SDC_OK) return rc;  // the data is in the boxes: tell the receivers
        for (const Op& o : ops)
            if (o.is_send) boxes[std::make_pair(rank, o.peer)].head()->posted.fetch_add(1, std::memory_order_release);
        for (const Op& o : ops)
            if (!o.is_send && (rc = take(c, o.buf, o.n, o.peer)) != SDC_OK) return rc;
        if ((rc = drain(c)) != SDC_OK) return rc;
        for (const Op& o : ops)
            if (!o.is_send) boxes[std::make_pair(o.peer, rank)].head()->consumed.fetch_add(1, std::memory_order_release);
        ops.clear();
        return SDC_OK;
    }
    int bcast(sdc_ctx* c, double* buf, size_t n, int root) override {
        int rc = group_begin(c);
        if (rank == root) {
            for (int k = 0; k < size && rc == SDC_OK; ++k)
                if (k != root) rc = send(c, buf, n, k);
        } else {
            rc = recv(c, buf, n, root);
        }
        return rc == SDC_OK ? group_end(c) : rc;
    }
    int sync(sdc_ctx* c) override { return drain(c); }
};

// A second path for TWO ranks.  The direct message of a two-rank run travels over ONE of a GPU's seven xGMI links (8.6 GB at
// 1024^3: ~134 ms at the ~64 GB/s a link sustains) while the host link idles; a share of it - the tail of the message - goes
// through pinned host memory instead: the sender copies chunks into a ring of `depth` slots in POSIX shared memory (registered
// with the HIP runtime: DMA at the host link's rate), the receiver copies them out as they arrive; both sides run their loop on
// a helper thread and a stream of their own, beside the RCCL transfer of the head of the message.  With 45 % of the message on
// this path both finish in ~80 ms (measured on one box: 48 GB/s end to end, scripts/probe_host_pipe.py).  One-directional: file /dev/shm/<job>.pipe.<src>.<dst>.
// The ring is made ONCE per communicator and direction, its slots sized for the largest message the context can send (a field
// or a half spectrum) - the piece a hand-over sends per slot (`chunk`) is a logical length inside them, so a change of the
// message kind or of the share never re-opens the file (two ranks re-opening on their own could end up on different files).
// The file's name is dropped as soon as BOTH ends have mapped it (a counter in its header; the second to attach unlinks - the
// mapping outlives the name), or by the only end that ever opened it when that end goes away.
struct HostPipe {
    struct Head {
        std::atomic<unsigned long long> posted, consumed;
        std::atomic<unsigned> attached;   // ends that have mapped the file
        char pad[44];
    };
    static_assert(sizeof(Head) == 64, "pipe header");
    std::string path;
    int fd = -1;
    unsigned char* base = nullptr;
    size_t map_bytes = 0, cap = 0;     // cap: doubles per slot
    int depth = 2;
    bool device = true, registered = false, named = false;
    hipStream_t stream = nullptr;
    double timeout_s = 0.0;            // <= 0: wait for ever, like a receive on the wire (SDC_COMM_TIMEOUT > 0 bounds both)
    Head* head() const { return reinterpret_cast<Head*>(base); }
    double* slot(unsigned long long k) const { return reinterpret_cast<double*>(base + sizeof(Head)) + (size_t)(k % (unsigned)depth) * cap; }
    ~HostPipe() {
        if (stream) {
            (void)hipStreamSynchronize(stream);
            (void)hipStreamDestroy(stream);
        }
        if (base) {
            if (named && head()->attached.load(std::memory_order_acquire) < 2) shm_unlink(path.c_str());   // (nobody else ever came)
            if (registered) (void)hipHostUnregister(base);
            munmap(base, map_bytes);
        }
        if (fd >= 0) close(fd);
    }
    int open(sdc_ctx* c, const std::string& job, int src, int dst, size_t slot_doubles, bool on_device) {
        path = "/" + job + ".pipe." + std::to_string(src) + "." + std::to_string(dst);
        cap = slot_doubles;
        device = on_device;
        if (const char* t = getenv("SDC_COMM_TIMEOUT")) timeout_s = atof(t);
        fd = shm_open(path.c_str(), O_CREAT | O_RDWR, 0600);
        if (fd < 0) return fail(c, SDC_ERR_COMM, "shm_open(%s): %s", path.c_str(), strerror(errno));
        named = true;
        map_bytes = sizeof(Head) + (size_t)depth * cap * sizeof(double);
        if (ftruncate(fd, (off_t)map_bytes) != 0) return fail(c, SDC_ERR_COMM, "ftruncate(%s): %s", path.c_str(), strerror(errno));
        void* m = mmap(nullptr, map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (m == MAP_FAILED) return fail(c, SDC_ERR_COMM, "mmap(%s): %s", path.c_str(), strerror(errno));
        base = static_cast<unsigned char*>(m);
        if (head()->attached.fetch_add(1, std::memory_order_acq_rel) + 1 >= 2) {   // both ends hold the mapping: the name can go
            shm_unlink(path.c_str());
            named = false;
        }
        if (device) {
            registered = hipHostRegister(base, map_bytes, hipHostRegisterDefault) == hipSuccess;   // (pageable copies still work)
            if (!registered) (void)hipGetLastError();
            if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) return fail(c, SDC_ERR_HIP, "cannot create the pipe's stream");
        }
        return SDC_OK;
    }
    template <class Pred>
    int wait_for(Pred ready, std::string* err, const char* what) {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spin = 0; !ready(); ++spin) {
            if (spin < 4000) sched_yield();
            else usleep(20);
            if (timeout_s > 0.0 && (spin & 1023) == 1023 &&
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
                *err = std::string("host pipe ") + path + ": " + what + " timed out";
                return SDC_ERR_COMM;
            }
        }
        return SDC_OK;
    }
    // host loops (helper threads): n doubles from / to a buffer of the pipe's kind, `chunk` (<= cap) of them per slot
    int send(const double* buf, size_t n, size_t chunk, std::string* err) {
        Head* h = head();
        for (size_t o = 0; o < n; o += chunk) {
            const size_t len = std::min(chunk, n - o);
            int rc = wait_for([&] { return h->posted.load(std::memory_order_relaxed) - h->consumed.load(std::memory_order_acquire) < (unsigned)depth; },
                              err, "waiting for a free slot");
            if (rc != SDC_OK) return rc;
            double* at = slot(h->posted.load(std::memory_order_relaxed));
            if (device) {
                if (hipMemcpyAsync(at, buf + o, len * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess ||
                    hipStreamSynchronize(stream) != hipSuccess) {
                    *err = "host pipe " + path + ": device-to-host copy failed";
                    return SDC_ERR_HIP;
                }
            } else {
                memcpy(at, buf + o, len * sizeof(double));
            }
            h->posted.fetch_add(1, std::memory_order_release);
        }
        return SDC_OK;
    }
    int recv(double* buf, size_t n, size_t chunk, std::string* err) {
        Head* h = head();
        for (size_t o = 0; o < n; o += chunk) {
            const size_t len = std::min(chunk, n - o);
            int rc = wait_for([&] { return h->posted.load(std::memory_order_acquire) > h->consumed.load(std::memory_order_relaxed); }, err,
                              "waiting for a chunk");
            if (rc != SDC_OK) return rc;
            const double* at = slot(h->consumed.load(std::memory_order_relaxed));
            if (device) {
                if (hipMemcpyAsync(buf + o, at, len * sizeof(double), hipMemcpyHostToDevice, stream) != hipSuccess ||
                    hipStreamSynchronize(stream) != hipSuccess) {
                    *err = "host pipe " + path + ": host-to-device copy failed";
                    return SDC_ERR_HIP;
                }
            } else {
                memcpy(buf + o, at, len * sizeof(double));
            }
            h->consumed.fetch_add(1, std::memory_order_release);
        }
        return SDC_OK;
    }
};

// ------------------------------------------------------------------------------------------------------------------
// the exchange patterns (any wire, any buffers of that wire's kind)
// ------------------------------------------------------------------------------------------------------------------
// piece j of a message of n values cut into `parts` pieces: offset and length (the last ones may be short or empty)
static inline void msg_piece(size_t n, int parts, int j, size_t* off, size_t* len) {
    const size_t csz = (n + (size_t)parts - 1) / (size_t)parts;
    const size_t lo = std::min(n, (size_t)j * csz), hi = std::min(n, (size_t)(j + 1) * csz);
    *off = lo;
    *len = hi - lo;
}
// values this rank relays per message in a two-hop hand-over over P ranks (size of one staging slot)
static inline size_t two_hop_slot(size_t n, int P, int r) {
    size_t off, len;
    msg_piece(n, P, r, &off, &len);
    return len;
}

// The forward hand-over src(rank) -> dst(rank + 1) of ALL P active ranks at once, over two hops.
//
// xGMI is a full mesh of point-to-point links: a direct message uses one of a GPU's seven links while six idle.  Every
// message is cut into P pieces; piece j travels via rank j (phase 1: owner -> relay, phase 2: relay -> destination; the
// pieces whose relay is the owner or the destination go directly).  Each link then carries 1/P of a message per phase:
// 2/P of the direct transfer time.  Both phases are one group in which every active rank takes part, which is why the
// caller must have established lock step.  stage: (P - 1) slots of two_hop_slot() values.  Bit-identical to a copy.
static int two_hop_handover(Wire* w, sdc_ctx* c, int P, const double* src, double* dst, double* stage, size_t n) {
    const int r = w->rank;
    size_t off, len, moff, mine;
    msg_piece(n, P, r, &moff, &mine);
    int rc = w->group_begin(c);  // phase 1: owners hand piece j to rank j (the destination's own piece lands in place)
    if (rc != SDC_OK) return rc;
    if (r <= P - 2)
        for (int j = 0; j < P; ++j) {
            msg_piece(n, P, j, &off, &len);
            if (j != r && len > 0) w->send(c, src + off, len, j);
        }
    if (mine > 0)
        for (int origin = 0; origin < P - 1; ++origin)
            if (origin != r) w->recv(c, origin == r - 1 ? dst + moff : stage + (size_t)origin * mine, mine, origin);
    if ((rc = w->group_end(c)) != SDC_OK) return rc;
    if ((rc = w->group_begin(c)) != SDC_OK) return rc;  // phase 2: relays forward to the destinations
    if (mine > 0)
        for (int d = 1; d < P; ++d)
            if (d != r) {
                const int origin = d - 1;
                w->send(c, origin == r ? src + moff : stage + (size_t)origin * mine, mine, d);
            }
    if (r >= 1)
        for (int j = 0; j < P; ++j) {
            msg_piece(n, P, j, &off, &len);
            if (j != r && len > 0) w->recv(c, dst + off, len, j);
        }
    return w->group_end(c);
}

// buf of rank `root` to every rank, as scatter + all-gather over the mesh instead of the library broadcast: the message
// is cut into size-1 pieces, the root hands piece j to the j-th other rank (its links carry one piece each, all at
// once), then those ranks exchange their pieces among themselves - two phases in which every link carries 1/(size-1)
// of the message.  Bit-identical to a copy.
static int mesh_bcast(Wire* w, sdc_ctx* c, double* buf, size_t n, int root) {
    const int P = w->size, r = w->rank;
    if (P <= 2) return w->bcast(c, buf, n, root);
    auto slot_of = [root](int k) { return k < root ? k : k - 1; };  // position of rank k among the non-root ranks
    size_t off, len, moff = 0, mine = 0;
    if (r != root) msg_piece(n, P - 1, slot_of(r), &moff, &mine);
    int rc = w->group_begin(c);
    if (rc != SDC_OK) return rc;
    if (r == root) {
        for (int k = 0; k < P; ++k) {
            if (k == root) continue;
            msg_piece(n, P - 1, slot_of(k), &off, &len);
            if (len > 0) w->send(c, buf + off, len, k);
        }
    } else if (mine > 0) {
        w->recv(c, buf + moff, mine, root);
    }
    if ((rc = w->group_end(c)) != SDC_OK) return rc;
    if (r == root) return SDC_OK;
    if ((rc = w->group_begin(c)) != SDC_OK) return rc;
    for (int k = 0; k < P; ++k) {
        if (k == root || k == r) continue;
        if (mine > 0) w->send(c, buf + moff, mine, k);
        msg_piece(n, P - 1, slot_of(k), &off, &len);
        if (len > 0) w->recv(c, buf + off, len, k);
    }
    return w->group_end(c);
}

// ------------------------------------------------------------------------------------------------------------------
// per-context state
// ------------------------------------------------------------------------------------------------------------------
struct CommState {
    std::shared_ptr<Wire> wire;     // shared by the levels of one time rank (sdc_comm_attach)
    hipEvent_t inbox_free = nullptr;  // engine stream -> message stream: the inbox has been read (sdc_replace_u0)
    hipEvent_t ready = nullptr;     // engine stream -> message stream: the buffers a message touches are settled
    hipEvent_t done = nullptr;      // message stream -> engine stream: the last message has completed
    bool inbox_busy = false;        // ... an event is recorded that the next receive has to wait for
    double* inbox = nullptr;        // where a received u[0] lands before sdc_replace_u0 hands it to the level
    double* stage = nullptr;        // staging slots of the two-hop hand-over
    size_t stage_len = 0;
    bool send_pending = false;      // a send may still be reading UEND
    bool relay = true;              // more than two ranks: two-hop hand-over / mesh broadcast instead of direct messages
    int posted_recv = -1;           // sdc_comm_handover_post: 1 / 2 = a field / a spectrum is being received, 0 = only a send, -1 = nothing posted
    unsigned long long two_hop_calls = 0, mesh_bcast_calls = 0;
    // two ranks: a share of the direct message through pinned host memory (HostPipe), on helper threads
    double host_share = 0.0;        // 0: off
    std::string job;                // names the pipes' files (the same on both ranks)
    std::unique_ptr<HostPipe> pipe_out, pipe_in;
    std::thread helper_out, helper_in;
    int helper_rc_out = SDC_OK, helper_rc_in = SDC_OK;
    std::string helper_err_out, helper_err_in;
    hipEvent_t pipe_ready = nullptr;   // engine / message side -> pipe streams: source complete, destination free
    unsigned long long host_path_calls = 0;
};
// the helper threads of a posted hand-over have finished (their data is in place: they synchronise their own streams)
static int join_helpers(sdc_ctx* c, CommState* cs) {
    if (cs->helper_out.joinable()) cs->helper_out.join();
    if (cs->helper_in.joinable()) cs->helper_in.join();
    int rc = cs->helper_rc_out != SDC_OK ? cs->helper_rc_out : cs->helper_rc_in;
    if (rc != SDC_OK) {
        const std::string& e = cs->helper_rc_out != SDC_OK ? cs->helper_err_out : cs->helper_err_in;
        cs->helper_rc_out = cs->helper_rc_in = SDC_OK;
        return fail(c, rc, "%s", e.c_str());
    }
    return SDC_OK;
}

// UEND is about to be overwritten: a send that reads it has to be through first (device-side wait, no host block)
static int uend_write_fence(sdc_ctx* c) {
    CommState* cs = c->comm;
    if (cs && cs->send_pending) {
        HIPCHK(c, hipStreamWaitEvent(c->stream, cs->done, 0));
        cs->send_pending = false;
    }
    return SDC_OK;
}

static void comm_free(sdc_ctx* c) {
    CommState* cs = c->comm;
    if (!cs) return;
    (void)join_helpers(nullptr, cs);
    cs->pipe_out.reset();
    cs->pipe_in.reset();
    if (cs->pipe_ready) (void)hipEventDestroy(cs->pipe_ready);
    if (cs->wire) (void)cs->wire->sync(nullptr);
    cs->wire.reset();
    if (cs->ready) (void)hipEventDestroy(cs->ready);
    if (cs->done) (void)hipEventDestroy(cs->done);
    if (cs->inbox_free) (void)hipEventDestroy(cs->inbox_free);
    (void)hipFree(cs->inbox);
    (void)hipFree(cs->stage);
    delete cs;
    c->comm = nullptr;
}

static int comm_events(sdc_ctx* c, CommState* cs) {
    if (hipEventCreateWithFlags(&cs->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&cs->done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&cs->inbox_free, hipEventDisableTiming) != hipSuccess)
        return fail(c, SDC_ERR_HIP, "cannot create the message events");
    return SDC_OK;
}

static std::shared_ptr<Wire> make_shm_wire(const char* uid128, int nranks, int rank, size_t cap, bool device) {
    auto w = std::make_shared<ShmWire>();
    char name[129];
    memcpy(name, uid128, 128);
    name[128] = 0;
    w->job = std::string("sdcmi.") + (name + 4);
    for (char& ch : w->job)
        if (!(isalnum((unsigned char)ch) || ch == '.' || ch == '_' || ch == '-')) ch = '_';
    w->rank = rank;
    w->size = nranks;
    w->cap = cap;
    w->device = device;
    if (const char* t = getenv("SDC_COMM_TIMEOUT")) w->timeout_s = atof(t) > 0 ? atof(t) : w->timeout_s;
    return w;
}

extern "C" int sdc_comm_unique_id(char* out128) {
    if (!out128) return fail(nullptr, SDC_ERR_PARAM, "null pointer");
    std::string why;
    RcclApi* api = rccl_api(&why);
    if (!api) return fail(nullptr, SDC_ERR_COMM, "%s", why.c_str());
    ncclUniqueId id;
    ncclResult_t r = api->GetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, SDC_ERR_COMM, "ncclGetUniqueId: %s", api->GetErrorString(r));
    static_assert(sizeof(id) == 128, "ncclUniqueId");
    memcpy(out128, &id, sizeof id);
    return SDC_OK;
}

extern "C" int sdc_comm_init(sdc_ctx* c, const char* uid128, int nranks, int rank) {
    if (!c || !uid128) return fail(c, SDC_ERR_PARAM, "null pointer");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(c, SDC_ERR_PARAM, "rank %d of %d", rank, nranks);
    if (c->comm) return fail(c, SDC_ERR_STATE, "communicator exists already (sdc_comm_destroy first)");
    HIPCHK(c, hipSetDevice(c->device));
    std::unique_ptr<CommState> cs(new CommState);
    if (strncmp(uid128, "shm:", 4) == 0) {
        // the largest message of this rank's levels: a field, or (later) its half spectrum
        auto w = make_shm_wire(uid128, nranks, rank, std::max(c->N, 2 * c->Nc) + 8, true);
        if (hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking) != hipSuccess)
            return fail(c, SDC_ERR_HIP, "cannot create the message stream");
        cs->job = static_cast<ShmWire*>(w.get())->job;
        cs->wire = w;
    } else {
        std::string why;
        RcclApi* api = rccl_api(&why);
        if (!api) return fail(c, SDC_ERR_COMM, "%s", why.c_str());
        auto w = std::make_shared<RcclWire>();
        w->api = api;
        w->rank = rank;
        w->size = nranks;
        ncclUniqueId id;
        memcpy(&id, uid128, sizeof id);
        ncclResult_t r = api->CommInitRank(&w->comm, nranks, id, rank);
        if (r != ncclSuccess) {
            w->comm = nullptr;
            return fail(c, SDC_ERR_COMM, "ncclCommInitRank(rank %d of %d): %s", rank, nranks, api->GetErrorString(r));
        }
        if (hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking) != hipSuccess)
            return fail(c, SDC_ERR_HIP, "cannot create the message stream");
        {   // (a name both ranks derive from the unique id: files of the host pipes)
            unsigned long long hsh = 1469598103934665603ull;
            for (int i = 0; i < 128; ++i) hsh = (hsh ^ (unsigned char)uid128[i]) * 1099511628211ull;
            char nm[64];
            snprintf(nm, sizeof nm, "sdcmi.r%016llx", hsh);
            cs->job = nm;
        }
        cs->wire = w;
    }
    int rc = comm_events(c, cs.get());
    if (rc != SDC_OK) {
        c->comm = cs.release();
        comm_free(c);
        return rc;
    }
    c->comm = cs.release();
    return SDC_OK;
}

// a further level of the same time rank (MLSDC / PFASST: controller_MPI.py sends on every level) shares the owner's
// communicator and message stream; messages of all levels are matched by their order, like the tags level*100 + iter do
extern "C" int sdc_comm_attach(sdc_ctx* c, sdc_ctx* owner) {
    if (!c || !owner) return fail(c, SDC_ERR_PARAM, "null pointer");
    if (!owner->comm) return fail(c, SDC_ERR_STATE, "the owner has no communicator (sdc_comm_init)");
    if (c->comm) return fail(c, SDC_ERR_STATE, "communicator exists already (sdc_comm_destroy first)");
    if (c->device != owner->device) return fail(c, SDC_ERR_PARAM, "levels of one time rank live on one device");
    if (ShmWire* sw = dynamic_cast<ShmWire*>(owner->comm->wire.get()))
        if (std::max(c->N, 2 * c->Nc) + 8 > sw->cap)
            return fail(c, SDC_ERR_PARAM, "attach the smaller level to the larger one (mailbox capacity)");
    std::unique_ptr<CommState> cs(new CommState);
    cs->wire = owner->comm->wire;
    cs->relay = owner->comm->relay;
    int rc = comm_events(c, cs.get());
    c->comm = cs.release();
    if (rc != SDC_OK) comm_free(c);
    return rc;
}

extern "C" int sdc_comm_destroy(sdc_ctx* c) {
    if (!c) return SDC_ERR_PARAM;
    comm_free(c);
    return SDC_OK;
}

#define NEED_COMM(c)                                                                   \
    if (!(c) || !(c)->comm) return fail(c, SDC_ERR_STATE, "no communicator (sdc_comm_init)"); \
    CommState* cs = (c)->comm;                                                         \
    Wire* w = cs->wire.get()

extern "C" int sdc_comm_set_chunk(sdc_ctx* c, size_t doubles_per_piece) {
    NEED_COMM(c);
    w->chunk = doubles_per_piece;
    return SDC_OK;
}

extern "C" int sdc_comm_set_relay(sdc_ctx* c, int on) {
    NEED_COMM(c);
    (void)w;
    cs->relay = on != 0;
    return SDC_OK;
}

// two ranks: `share` of every lock-step hand-over (0 <= share < 1; 0 = off) travels through pinned host memory beside the
// direct message (HostPipe above).  Both ranks must make the same choice.
extern "C" int sdc_comm_set_host_share(sdc_ctx* c, double share) {
    NEED_COMM(c);
    (void)w;
    if (!(share >= 0.0 && share < 1.0)) return fail(c, SDC_ERR_PARAM, "share of the message on the host path: 0 <= share < 1");
    int rc = join_helpers(c, cs);
    if (rc != SDC_OK) return rc;
    cs->host_share = share;
    return SDC_OK;
}

// what travels in a lock-step hand-over: 0 = the end value as a field (N doubles), 1 = its half spectrum (2 Nc doubles) -
// between levels that sweep in Fourier space (sdc_spectral_handover_ok) neither the sender's inverse transform nor the
// receiver's forward transform is needed then.  Every rank of the communicator has to make the same choice.
extern "C" int sdc_comm_set_format(sdc_ctx* c, int spectra) {
    NEED_COMM(c);
    (void)w;
    (void)cs;
    return sdc_set_wire_spectral(c, spectra);
}

extern "C" int sdc_comm_info(sdc_ctx* c, int* rank, int* size, unsigned long long* two_hop_calls,
                             unsigned long long* mesh_bcast_calls, char* kind16) {
    NEED_COMM(c);
    if (rank) *rank = w->rank;
    if (size) *size = w->size;
    if (two_hop_calls) *two_hop_calls = cs->two_hop_calls;
    if (mesh_bcast_calls) *mesh_bcast_calls = cs->mesh_bcast_calls;
    if (kind16) {
        strncpy(kind16, w->kind(), 15);
        kind16[15] = 0;
    }
    return SDC_OK;
}

static int ensure_inbox(sdc_ctx* c, CommState* cs) {
    if (!cs->inbox) {
        HIPCHK(c, hipMalloc((void**)&cs->inbox, c->N * sizeof(double)));
        c->bytes += c->N * sizeof(double);
    }
    return SDC_OK;
}
// the message stream may write the inbox once the sdc_replace_u0 of the previous receive has read it - and waits for
// nothing that was queued on the engine's stream after that
static int inbox_writable(sdc_ctx* c, CommState* cs, Wire* w) {
    if (cs->inbox_busy) {
        HIPCHK(c, hipStreamWaitEvent(w->stream, cs->inbox_free, 0));
        cs->inbox_busy = false;
    }
    return SDC_OK;
}
static int deliver_inbox(sdc_ctx* c, CommState* cs) {
    int rc = sdc_replace_u0(c, cs->inbox);
    if (rc != SDC_OK) return rc;
    HIPCHK(c, hipEventRecord(cs->inbox_free, c->stream));
    cs->inbox_busy = true;
    return SDC_OK;
}

// send UEND to send_peer and / or receive the new u[0] from recv_peer as ONE group (ncclGroupStart / End): the two
// directions progress concurrently instead of unwinding rank by rank.  A peer < 0 skips that direction (first /
// last rank, or a predecessor that is done: controller_MPI.py:235-305).
extern "C" int sdc_comm_exchange(sdc_ctx* c, int send_peer, int recv_peer) {
    NEED_COMM(c);
    if (send_peer >= w->size || recv_peer >= w->size) return fail(c, SDC_ERR_PARAM, "peer out of range");
    if (cs->posted_recv >= 0) return fail(c, SDC_ERR_STATE, "a posted hand-over is still open (sdc_comm_handover_complete)");
    if (send_peer < 0 && recv_peer < 0) return SDC_OK;
    int rc;
    if (send_peer >= 0) {  // behind the point where UEND is complete, and behind nothing queued later
        if ((rc = sdc_stream_wait_uend(c, w->stream)) != SDC_OK) return rc;
    }
    if (recv_peer >= 0) {
        if ((rc = ensure_inbox(c, cs)) != SDC_OK) return rc;
        if ((rc = inbox_writable(c, cs, w)) != SDC_OK) return rc;
    }
    if ((rc = w->group_begin(c)) != SDC_OK) return rc;
    if (send_peer >= 0) w->send(c, c->UEND, c->N, send_peer);
    if (recv_peer >= 0) w->recv(c, cs->inbox, c->N, recv_peer);
    if ((rc = w->group_end(c)) != SDC_OK) return rc;
    HIPCHK(c, hipEventRecord(cs->done, w->stream));
    if (send_peer >= 0) cs->send_pending = true;
    if (recv_peer >= 0) {
        HIPCHK(c, hipStreamWaitEvent(c->stream, cs->done, 0));
        cs->send_pending = false;  // the engine's stream now runs behind the whole group
        return deliver_inbox(c, cs);
    }
    return SDC_OK;
}

extern "C" int sdc_send_uend(sdc_ctx* c, int peer) { return sdc_comm_exchange(c, peer, -1); }
extern "C" int sdc_recv_u0(sdc_ctx* c, int peer) { return sdc_comm_exchange(c, -1, peer); }

// Lock-step runs (every active rank iterates in step: single level, Jacobi-type multi-step SDC, fixed number of sweeps
// or all_to_done): the hand-over uend(r) -> u[0](r + 1) of ALL `nactive` ranks, posted now and completed later.  The
// message stream waits for UEND only (sdc_set_early_end_point: the sweep produced it before its residual passes), so
// the message travels while the engine's stream still reduces the residual; sdc_comm_handover_complete then makes the
// engine's stream wait for the message and hands the received value to the level (sdc_replace_u0).  More than two
// ranks (and sdc_comm_set_relay on): two hops over the whole mesh instead of one link per message.
extern "C" int sdc_comm_handover_post(sdc_ctx* c, int nactive) {
    NEED_COMM(c);
    if (nactive < 1 || nactive > w->size) return fail(c, SDC_ERR_PARAM, "%d active ranks of %d", nactive, w->size);
    if (cs->posted_recv >= 0) return fail(c, SDC_ERR_STATE, "a posted hand-over is still open (sdc_comm_handover_complete)");
    const int r = w->rank;
    if (r >= nactive) return SDC_OK;
    const bool sending = r < nactive - 1, receiving = r >= 1;
    const bool two_hop = cs->relay && nactive > 2;
    const bool spectra = c->wire_spectral;
    const size_t n = spectra ? 2 * c->Nc : c->N;
    const double* src = nullptr;
    double* dst = nullptr;
    int rc;
    if (sending) {  // (a relay that does not send never reads its own end value)
        if (spectra) {
            src = (const double*)sdc_end_spectrum(c, w->stream);
            if (!src) return c->err.empty() ? fail(c, SDC_ERR_STATE, "no end value to send") : SDC_ERR_STATE;
        } else {
            if ((rc = sdc_stream_wait_uend(c, w->stream)) != SDC_OK) return rc;
            src = c->UEND;
        }
    }
    if (receiving) {
        if (spectra) {
            dst = (double*)sdc_spectrum_inbox(c);
            if (!dst) return SDC_ERR_NOMEM;
        } else {
            if ((rc = ensure_inbox(c, cs)) != SDC_OK) return rc;
            dst = cs->inbox;
        }
        if ((rc = inbox_writable(c, cs, w)) != SDC_OK) return rc;
    }
    if (two_hop) {
        const size_t slot = two_hop_slot(n, nactive, r), need = slot * (size_t)(nactive - 1);
        if (need > cs->stage_len) {
            if (cs->stage) {
                HIPCHK(c, hipStreamSynchronize(w->stream));
                (void)hipFree(cs->stage);
                c->bytes -= cs->stage_len * sizeof(double);
                cs->stage = nullptr;
            }
            HIPCHK(c, hipMalloc((void**)&cs->stage, std::max<size_t>(need, 1) * sizeof(double)));
            cs->stage_len = std::max<size_t>(need, 1);
            c->bytes += cs->stage_len * sizeof(double);
        }
        cs->two_hop_calls++;
        if ((rc = two_hop_handover(w, c, nactive, src, dst, cs->stage, n)) != SDC_OK) return rc;
    } else {
        // two ranks: the tail of the message through pinned host memory, on helper threads, beside the direct transfer
        size_t n_host = 0;
        // (only messages of tens of megabytes: below that the helper threads and the per-slot synchronisation cost more than the
        // share of the link they free - SDC_PIPE_MIN_BYTES, default 32 MB; tests lower it)
        size_t min_bytes = (size_t)32 << 20;
        if (const char* pm = getenv("SDC_PIPE_MIN_BYTES")) min_bytes = strtoull(pm, nullptr, 10);
        if (nactive == 2 && cs->host_share > 0.0 && n * sizeof(double) >= min_bytes && n >= 1024) {
            n_host = ((size_t)((double)n * cs->host_share) / 512) * 512;
            size_t slot_doubles = (size_t)16 << 20;   // 128 MB slots (SDC_PIPE_CHUNK: doubles per slot, for tests)
            if (const char* pc = getenv("SDC_PIPE_CHUNK")) slot_doubles = std::max<size_t>(512, strtoull(pc, nullptr, 10));
            // slots sized ONCE for the largest message of this context (a field or a half spectrum), whatever travels now
            const size_t cap = std::min<size_t>(slot_doubles, ((std::max<size_t>(c->N, 2 * c->Nc) + 511) / 512) * 512);
            if ((rc = join_helpers(c, cs)) != SDC_OK) return rc;
            if (!cs->pipe_ready) HIPCHK(c, hipEventCreateWithFlags(&cs->pipe_ready, hipEventDisableTiming));
            // (whatever the message stream waits for - the end value complete, the inbox free - the pipes wait for too)
            HIPCHK(c, hipEventRecord(cs->pipe_ready, w->stream));
            if (sending) {
                if (!cs->pipe_out) {
                    cs->pipe_out.reset(new HostPipe);
                    if ((rc = cs->pipe_out->open(c, cs->job, r, r + 1, cap, true)) != SDC_OK) return rc;
                }
                HIPCHK(c, hipStreamWaitEvent(cs->pipe_out->stream, cs->pipe_ready, 0));
                HostPipe* pp = cs->pipe_out.get();
                const size_t chunk = std::min<size_t>(std::max<size_t>(n_host, 512), pp->cap);   // (the ring's own slot size)
                const double* from = src + (n - n_host);
                const int dev = c->device;
                cs->helper_out = std::thread([cs, pp, from, n_host, chunk, dev] {
                    (void)hipSetDevice(dev);
                    cs->helper_rc_out = pp->send(from, n_host, chunk, &cs->helper_err_out);
                });
            }
            if (receiving) {
                if (!cs->pipe_in) {
                    cs->pipe_in.reset(new HostPipe);
                    if ((rc = cs->pipe_in->open(c, cs->job, r - 1, r, cap, true)) != SDC_OK) return rc;
                }
                HIPCHK(c, hipStreamWaitEvent(cs->pipe_in->stream, cs->pipe_ready, 0));
                HostPipe* pp = cs->pipe_in.get();
                const size_t chunk = std::min<size_t>(std::max<size_t>(n_host, 512), pp->cap);
                double* to = dst + (n - n_host);
                const int dev = c->device;
                cs->helper_in = std::thread([cs, pp, to, n_host, chunk, dev] {
                    (void)hipSetDevice(dev);
                    cs->helper_rc_in = pp->recv(to, n_host, chunk, &cs->helper_err_in);
                });
            }
            cs->host_path_calls++;
        }
        if ((rc = w->group_begin(c)) != SDC_OK) return rc;
        if (sending) w->send(c, src, n - n_host, r + 1);
        if (receiving) w->recv(c, dst, n - n_host, r - 1);
        if ((rc = w->group_end(c)) != SDC_OK) return rc;
    }
    HIPCHK(c, hipEventRecord(cs->done, w->stream));
    cs->send_pending = sending && !spectra;
    cs->posted_recv = receiving ? (spectra ? 2 : 1) : 0;
    if (spectra) {
        // the end value of this iteration existed for the wire, as a spectrum; its field form was put off (sdc_end_point)
        // and would otherwise be produced on account of the next sweep, which replaces the iterate it belongs to
        c->uend_pending = false;
        c->uend_gen = -1;
    }
    return SDC_OK;
}

extern "C" int sdc_comm_handover_complete(sdc_ctx* c) {
    NEED_COMM(c);
    (void)w;
    if (cs->posted_recv < 0) return SDC_OK;
    const int kind = cs->posted_recv;
    cs->posted_recv = -1;
    {   // (the share that went through host memory: its helper threads end with the data in place)
        int rch = join_helpers(c, cs);
        if (rch != SDC_OK) return rch;
    }
    // (also on a rank that only sent: its next sweep rewrites the spectrum / end value the message reads)
    HIPCHK(c, hipStreamWaitEvent(c->stream, cs->done, 0));
    cs->send_pending = false;
    if (kind == 1) return deliver_inbox(c, cs);
    if (kind == 2) {
        int rc = sdc_replace_u0_spectrum(c);
        if (rc != SDC_OK) return rc;
        HIPCHK(c, hipEventRecord(cs->inbox_free, c->stream));  // (the buffer that is the inbox NOW: the old start spectrum)
        cs->inbox_busy = true;
    }
    return SDC_OK;
}

// a device buffer of rank `root` to every rank, in place, ordered on the engine's stream on both sides
static int bcast_buffer(sdc_ctx* c, CommState* cs, Wire* w, double* buf, size_t n, int root) {
    if (root < 0 || root >= w->size) return fail(c, SDC_ERR_PARAM, "root out of range");
    HIPCHK(c, hipEventRecord(cs->ready, c->stream));
    HIPCHK(c, hipStreamWaitEvent(w->stream, cs->ready, 0));
    int rc;
    if (cs->relay && w->size > 2) {
        cs->mesh_bcast_calls++;
        rc = mesh_bcast(w, c, buf, n, root);
    } else {
        rc = w->bcast(c, buf, n, root);
    }
    if (rc != SDC_OK) return rc;
    HIPCHK(c, hipEventRecord(cs->done, w->stream));
    HIPCHK(c, hipStreamWaitEvent(c->stream, cs->done, 0));
    return SDC_OK;
}

// one slab field of rank `root` to every rank, in place (the end value of a block: controller_MPI.py:125-130 bcast
// of uend; mesh.py:113-125)
extern "C" int sdc_bcast(sdc_ctx* c, int slot, int m, int root) {
    NEED_COMM(c);
    double* buf = (double*)sdc_slot_ptr(c, slot, m, 0);  // (stores deferred node fields; marks UEND as rewritten)
    if (!buf) return fail(c, SDC_ERR_PARAM, "bad slot (%d, %d)", slot, m);
    int rc = uend_write_fence(c);
    if (rc != SDC_OK) return rc;
    if ((rc = bcast_buffer(c, cs, w, buf, c->N, root)) != SDC_OK) return rc;
    if (w->rank != root && slot == SDC_SLOT_U) return sdc_invalidate_spectra(c, m == 0 ? 1 : 2);
    return SDC_OK;
}

// any device buffer of n doubles (the controller's own copy of the end value)
extern "C" int sdc_comm_bcast_buffer(sdc_ctx* c, double* buf, size_t n, int root) {
    NEED_COMM(c);
    if (!buf) return fail(c, SDC_ERR_PARAM, "null pointer");
    if (ShmWire* sw = dynamic_cast<ShmWire*>(w))
        if (n > sw->cap) return fail(c, SDC_ERR_PARAM, "buffer of %zu doubles exceeds the mailbox (%zu)", n, sw->cap);
    return bcast_buffer(c, cs, w, buf, n, root);
}

// The end value of a block from rank `root` to every rank AS ITS HALF SPECTRUM (levels that sweep in Fourier space): the
// root sends the last node's spectrum as it lies in its cache, the others receive into their spectrum inbox
// (sdc_start_from_spectrum makes it the start value of the next block; the root itself continues with sdc_advance) - no
// inverse transform on the root, no forward transform anywhere (controller_MPI.py:125-130 without leaving Fourier space).
extern "C" int sdc_comm_bcast_end_spectrum(sdc_ctx* c, int root) {
    NEED_COMM(c);
    if (root < 0 || root >= w->size) return fail(c, SDC_ERR_PARAM, "root out of range");
    if (!sdc_spectral_handover_ok(c)) return fail(c, SDC_ERR_STATE, "this level does not sweep in Fourier space");
    if (cs->posted_recv >= 0) return fail(c, SDC_ERR_STATE, "a posted hand-over is still open (sdc_comm_handover_complete)");
    const size_t n = 2 * c->Nc;
    double* buf;
    int rc;
    if (w->rank == root) {
        buf = (double*)sdc_end_spectrum(c, w->stream);
        if (!buf) return c->err.empty() ? fail(c, SDC_ERR_STATE, "no end value to broadcast") : SDC_ERR_STATE;
    } else {
        buf = (double*)sdc_spectrum_inbox(c);
        if (!buf) return SDC_ERR_NOMEM;
        if ((rc = inbox_writable(c, cs, w)) != SDC_OK) return rc;
        HIPCHK(c, hipEventRecord(cs->ready, c->stream));  // (whatever still reads the buffer that is the inbox now)
        HIPCHK(c, hipStreamWaitEvent(w->stream, cs->ready, 0));
    }
    if (cs->relay && w->size > 2) {
        cs->mesh_bcast_calls++;
        rc = mesh_bcast(w, c, buf, n, root);
    } else {
        rc = w->bcast(c, buf, n, root);
    }
    if (rc != SDC_OK) return rc;
    HIPCHK(c, hipEventRecord(cs->done, w->stream));
    HIPCHK(c, hipStreamWaitEvent(c->stream, cs->done, 0));
    return SDC_OK;
}

// host waits until every message posted so far has completed
extern "C" int sdc_comm_sync(sdc_ctx* c) {
    NEED_COMM(c);
    (void)cs;
    return w->sync(c);
}

// The exchange patterns on plain host buffers over a host-mode shm wire: rank `rank` of `nranks` (threads or processes
// that call this with the same job name) fills a message with a pattern only it can have made, runs
//   what = 0: direct hand-over r -> r + 1 among the first `arg` ranks (0: all), 1: the same over two hops,
//   what = 2: mesh broadcast from rank `arg`
// `rounds` times over the same mailboxes, and checks every value it received.  No GPU involved: this is how the CPU suite covers the piece arithmetic and the
// mailbox protocol.  Returns SDC_OK, or SDC_ERR_COMM with the first mismatch in sdc_last_error(NULL).
extern "C" int sdc_comm_selftest(const char* job, int nranks, int rank, size_t n, int what, int arg, int rounds) {
    if (!job || nranks < 1 || rank < 0 || rank >= nranks || rounds < 1) return fail(nullptr, SDC_ERR_PARAM, "bad self-test arguments");
    char uid[128];
    memset(uid, 0, sizeof uid);
    snprintf(uid, sizeof uid, "shm:%s", job);
    auto w = make_shm_wire(uid, nranks, rank, n + 8, false);
    auto value = [](int origin, int round, size_t i) { return (double)origin * 1e6 + (double)round * 1e3 + (double)(i % 997) + 0.25; };
    std::vector<double> src(std::max<size_t>(n, 1)), dst(std::max<size_t>(n, 1)), stage;
    for (int round = 0; round < rounds; ++round) {
        for (size_t i = 0; i < n; ++i) {
            src[i] = value(rank, round, i);
            dst[i] = -1.0;
        }
        int rc = SDC_OK;
        int expect_from = -1;
        if (what == 0 || what == 1) {
            const int P = arg > 0 ? arg : nranks;  // active ranks
            if (rank < P) {
                if (what == 1 && P > 2) {
                    stage.assign(std::max<size_t>(two_hop_slot(n, P, rank) * (size_t)(P - 1), 1), -2.0);
                    rc = two_hop_handover(w.get(), nullptr, P, src.data(), dst.data(), stage.data(), n);
                } else {
                    rc = w->group_begin(nullptr);
                    if (rank < P - 1) w->send(nullptr, src.data(), n, rank + 1);
                    if (rank >= 1) w->recv(nullptr, dst.data(), n, rank - 1);
                    if (rc == SDC_OK) rc = w->group_end(nullptr);
                }
                if (rank >= 1) expect_from = rank - 1;
            }
        } else if (what == 2) {
            const int root = arg;
            std::vector<double>& buf = rank == root ? src : dst;
            rc = mesh_bcast(w.get(), nullptr, buf.data(), n, root);
            if (rank != root) expect_from = root;
        } else {
            return fail(nullptr, SDC_ERR_PARAM, "self-test %d", what);
        }
        if (rc != SDC_OK) return rc;
        if (expect_from >= 0)
            for (size_t i = 0; i < n; ++i)
                if (dst[i] != value(expect_from, round, i))
                    return fail(nullptr, SDC_ERR_COMM, "rank %d, round %d: value %zu is %.17g, expected %.17g (from rank %d)", rank,
                                round, i, dst[i], value(expect_from, round, i), expect_from);
        for (size_t i = 0; i < n; ++i)
            if (src[i] != value(rank, round, i) && !(what == 2 && rank != arg))
                return fail(nullptr, SDC_ERR_COMM, "rank %d, round %d: the message itself was modified at %zu", rank, round, i);
    }
    return SDC_OK;
}
