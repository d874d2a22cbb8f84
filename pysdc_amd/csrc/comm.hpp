// RCCL point-to-point transport of the time-parallel hand-over behind the C-ABI: the forward transfer
// uend -> u[0] of the next time rank (controller_MPI.py:218-305 send_full / recv_full, mesh.py:85-125 isend / irecv /
// bcast) and the end-of-block broadcast (controller_MPI.py:125-130), modelled on the reference's NCCL wrapper
// (helpers/NCCL_communicator.py:12-20: unique id from rank 0, one communicator per process; :128-135 Bcast).
//
// librccl is bound at run time (dlopen) the first time a communicator is asked for: single-GPU runs never load it,
// and a process in which torch.distributed has already loaded its librccl shares that copy.  Messages travel on a
// stream of their own: a send waits (on the device) for the event that marks UEND complete, a receive lands in an
// inbox and is handed to the level through sdc_replace_u0 on the engine's stream once it has arrived, so residual
// passes and messages overlap and the host never blocks.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

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

#define RCCLCHK(c, api, call)                                                                       \
    do {                                                                                            \
        ncclResult_t r_ = (call);                                                                   \
        if (r_ != ncclSuccess) return fail(c, SDC_ERR_COMM, "%s: %s", #call, (api)->GetErrorString(r_)); \
    } while (0)

struct CommState {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;   // messages travel here, not on the engine's stream
    hipEvent_t ready = nullptr;     // engine stream -> message stream: the buffers a message touches are settled
    hipEvent_t done = nullptr;      // message stream -> engine stream: the last message has completed
    double* inbox = nullptr;        // where a received u[0] lands before sdc_replace_u0 hands it to the level
    bool send_pending = false;      // a send may still be reading UEND
    int rank = 0, size = 1;
    size_t chunk = 0;               // > 0: messages are cut into pieces of this many doubles (one group)
};

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
    if (cs->stream) (void)hipStreamSynchronize(cs->stream);
    if (cs->comm) {
        RcclApi* api = rccl_api(nullptr);
        if (api) (void)api->CommDestroy(cs->comm);
    }
    if (cs->ready) (void)hipEventDestroy(cs->ready);
    if (cs->done) (void)hipEventDestroy(cs->done);
    if (cs->stream) (void)hipStreamDestroy(cs->stream);
    (void)hipFree(cs->inbox);
    delete cs;
    c->comm = nullptr;
}

extern "C" int sdc_comm_unique_id(char* out128) {
    if (!out128) return fail(nullptr, SDC_ERR_PARAM, "null pointer");
    std::string why;
    RcclApi* api = rccl_api(&why);
    if (!api) return fail(nullptr, SDC_ERR_COMM, "%s", why.c_str());
    ncclUniqueId id;
    RCCLCHK(nullptr, api, api->GetUniqueId(&id));
    static_assert(sizeof(id) == 128, "ncclUniqueId");
    memcpy(out128, &id, sizeof id);
    return SDC_OK;
}

extern "C" int sdc_comm_init(sdc_ctx* c, const char* uid128, int nranks, int rank) {
    if (!c || !uid128) return fail(c, SDC_ERR_PARAM, "null pointer");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(c, SDC_ERR_PARAM, "rank %d of %d", rank, nranks);
    if (c->comm) return fail(c, SDC_ERR_STATE, "communicator exists already (sdc_comm_destroy first)");
    std::string why;
    RcclApi* api = rccl_api(&why);
    if (!api) return fail(c, SDC_ERR_COMM, "%s", why.c_str());
    HIPCHK(c, hipSetDevice(c->device));
    CommState* cs = new CommState;
    c->comm = cs;
    cs->rank = rank;
    cs->size = nranks;
    ncclUniqueId id;
    memcpy(&id, uid128, sizeof id);
    ncclResult_t r = api->CommInitRank(&cs->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        cs->comm = nullptr;
        comm_free(c);
        return fail(c, SDC_ERR_COMM, "ncclCommInitRank(rank %d of %d): %s", rank, nranks, api->GetErrorString(r));
    }
    if (hipStreamCreateWithFlags(&cs->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&cs->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&cs->done, hipEventDisableTiming) != hipSuccess) {
        comm_free(c);  // (the communicator and whatever was created are released again)
        return fail(c, SDC_ERR_HIP, "cannot create the message stream / events");
    }
    return SDC_OK;
}

extern "C" int sdc_comm_destroy(sdc_ctx* c) {
    if (!c) return SDC_ERR_PARAM;
    comm_free(c);
    return SDC_OK;
}

extern "C" int sdc_comm_set_chunk(sdc_ctx* c, size_t doubles_per_piece) {
    if (!c || !c->comm) return fail(c, SDC_ERR_STATE, "no communicator (sdc_comm_init)");
    c->comm->chunk = doubles_per_piece;
    return SDC_OK;
}

// pieces of one message inside the open group (all pieces of a message go to / come from the same peer, in order)
static ncclResult_t put(RcclApi* api, CommState* cs, const double* buf, size_t n, int peer) {
    const size_t step = cs->chunk ? cs->chunk : n;
    for (size_t o = 0; o < n; o += step) {
        ncclResult_t r = api->Send(buf + o, std::min(step, n - o), ncclDouble, peer, cs->comm, cs->stream);
        if (r != ncclSuccess) return r;
    }
    return ncclSuccess;
}
static ncclResult_t get(RcclApi* api, CommState* cs, double* buf, size_t n, int peer) {
    const size_t step = cs->chunk ? cs->chunk : n;
    for (size_t o = 0; o < n; o += step) {
        ncclResult_t r = api->Recv(buf + o, std::min(step, n - o), ncclDouble, peer, cs->comm, cs->stream);
        if (r != ncclSuccess) return r;
    }
    return ncclSuccess;
}

// send UEND to send_peer and / or receive the new u[0] from recv_peer as ONE group (ncclGroupStart / End): the two
// directions progress concurrently instead of unwinding rank by rank.  A peer < 0 skips that direction (first /
// last rank, or a predecessor that is done: controller_MPI.py:235-305).
extern "C" int sdc_comm_exchange(sdc_ctx* c, int send_peer, int recv_peer) {
    if (!c || !c->comm) return fail(c, SDC_ERR_STATE, "no communicator (sdc_comm_init)");
    CommState* cs = c->comm;
    RcclApi* api = rccl_api(nullptr);
    if (send_peer >= cs->size || recv_peer >= cs->size) return fail(c, SDC_ERR_PARAM, "peer out of range");
    if (send_peer < 0 && recv_peer < 0) return SDC_OK;
    if (send_peer >= 0) {  // behind the point where UEND is complete, and behind nothing queued later
        int rc = sdc_stream_wait_uend(c, cs->stream);
        if (rc != SDC_OK) return rc;
    }
    if (recv_peer >= 0 && !cs->inbox) {
        HIPCHK(c, hipMalloc((void**)&cs->inbox, c->N * sizeof(double)));
        c->bytes += c->N * sizeof(double);
    }
    if (recv_peer >= 0) {
        // the inbox may still be read by the sdc_replace_u0 of the previous receive
        HIPCHK(c, hipEventRecord(cs->ready, c->stream));
        HIPCHK(c, hipStreamWaitEvent(cs->stream, cs->ready, 0));
    }
    RCCLCHK(c, api, api->GroupStart());
    ncclResult_t r = ncclSuccess;
    if (send_peer >= 0) r = put(api, cs, c->UEND, c->N, send_peer);
    if (r == ncclSuccess && recv_peer >= 0) r = get(api, cs, cs->inbox, c->N, recv_peer);
    ncclResult_t re = api->GroupEnd();
    if (r != ncclSuccess || re != ncclSuccess)
        return fail(c, SDC_ERR_COMM, "send/recv group: %s", api->GetErrorString(r != ncclSuccess ? r : re));
    HIPCHK(c, hipEventRecord(cs->done, cs->stream));
    if (send_peer >= 0) cs->send_pending = true;
    if (recv_peer >= 0) {
        HIPCHK(c, hipStreamWaitEvent(c->stream, cs->done, 0));
        cs->send_pending = false;  // the engine's stream now runs behind the whole group
        return sdc_replace_u0(c, cs->inbox);
    }
    return SDC_OK;
}

extern "C" int sdc_send_uend(sdc_ctx* c, int peer) { return sdc_comm_exchange(c, peer, -1); }
extern "C" int sdc_recv_u0(sdc_ctx* c, int peer) { return sdc_comm_exchange(c, -1, peer); }

// one slab field of rank `root` to every rank, in place (the end value of a block: controller_MPI.py:125-130 bcast
// of uend; mesh.py:113-125)
extern "C" int sdc_bcast(sdc_ctx* c, int slot, int m, int root) {
    if (!c || !c->comm) return fail(c, SDC_ERR_STATE, "no communicator (sdc_comm_init)");
    CommState* cs = c->comm;
    RcclApi* api = rccl_api(nullptr);
    if (root < 0 || root >= cs->size) return fail(c, SDC_ERR_PARAM, "root out of range");
    double* buf = (double*)sdc_slot_ptr(c, slot, m, 0);  // (stores deferred node fields; marks UEND as rewritten)
    if (!buf) return fail(c, SDC_ERR_PARAM, "bad slot (%d, %d)", slot, m);
    int rc = uend_write_fence(c);
    if (rc != SDC_OK) return rc;
    HIPCHK(c, hipEventRecord(cs->ready, c->stream));
    HIPCHK(c, hipStreamWaitEvent(cs->stream, cs->ready, 0));
    RCCLCHK(c, api, api->Broadcast(buf, buf, c->N, ncclDouble, root, cs->comm, cs->stream));
    HIPCHK(c, hipEventRecord(cs->done, cs->stream));
    HIPCHK(c, hipStreamWaitEvent(c->stream, cs->done, 0));
    if (cs->rank != root && slot == SDC_SLOT_U) return sdc_invalidate_spectra(c, m == 0 ? 1 : 2);
    return SDC_OK;
}

// host waits until every message posted so far has completed
extern "C" int sdc_comm_sync(sdc_ctx* c) {
    if (!c || !c->comm) return fail(c, SDC_ERR_STATE, "no communicator (sdc_comm_init)");
    HIPCHK(c, hipStreamSynchronize(c->comm->stream));
    return SDC_OK;
}
