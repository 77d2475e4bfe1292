// libmicrodit_comm.so: the gradient exchange of the data-parallel MicroDiT step on RCCL, behind the C ABI of
// include/microdit_comm.h.  Host-only code.  RCCL is bound at run time (dlopen): the process usually has PyTorch's librccl.so
// mapped already, and a second copy of the library (a different build of it) in one address space is exactly what must not
// happen -- so the loader first asks for the image that is already there.
#include "../../../include/microdit_comm.h"

#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

namespace {

constexpr int TICKETS = 1024;
thread_local std::string g_err;

int fail(const std::string& what) {
    g_err = what;
    return MD_COMM_FAILED;
}

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names)                       // the image PyTorch (or the caller) has mapped already, if any
            if ((r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        const char* env = getenv("MD_COMM_RCCL");
        if (!r.handle && env) r.handle = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
        for (const char* n : names)
            if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!r.handle) return;
#define BIND(field, sym) *reinterpret_cast<void**>(&r.field) = dlsym(r.handle, sym)
        BIND(GetUniqueId, "ncclGetUniqueId");
        BIND(CommInitRank, "ncclCommInitRank");
        BIND(CommDestroy, "ncclCommDestroy");
        BIND(AllReduce, "ncclAllReduce");
        BIND(ReduceScatter, "ncclReduceScatter");
        BIND(AllGather, "ncclAllGather");
        BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.ReduceScatter && r.AllGather && r.GetErrorString;
    });
    return r;
}

bool dtype_of(int32_t d, ncclDataType_t* out) {
    if (d == MD_COMM_BF16) *out = ncclBfloat16;
    else if (d == MD_COMM_F32) *out = ncclFloat32;
    else return false;
    return true;
}

}  // namespace

struct md_comm {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;       // the communicator's own stream: collectives never queue behind compute kernels
    hipEvent_t order = nullptr;         // "the bucket is ready" on the caller's stream (re-recorded per collective)
    hipEvent_t done[TICKETS] = {};      // ring of completion events
    int64_t issued = 0;
    int rank = 0, world = 1, device = 0;
};

#define HIP_TRY(call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) return fail(std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define NCCL_TRY(call)                                                                          \
    do {                                                                                        \
        ncclResult_t r_ = (call);                                                               \
        if (r_ != ncclSuccess) return fail(std::string(#call) + ": " + rccl().GetErrorString(r_)); \
    } while (0)

extern "C" int md_comm_abi_version(void) { return MD_COMM_ABI_VERSION; }
extern "C" const char* md_comm_last_error(void) { return g_err.c_str(); }

extern "C" int md_comm_unique_id(void* out) {
    static_assert(sizeof(ncclUniqueId) == MD_COMM_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");
    if (!out) return MD_COMM_BAD_ARG;
    if (!rccl().ok) return MD_COMM_NO_RCCL;
    ncclUniqueId id;
    NCCL_TRY(rccl().GetUniqueId(&id));
    memcpy(out, &id, sizeof(id));
    return 0;
}

extern "C" int md_comm_destroy(md_comm* c);

namespace {
// Everything md_comm_init can fail on after the struct exists; a failure leaves *c for the caller to tear down.
int comm_setup(md_comm* c, const void* uid) {
    int lo = 0, hi = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));        // hi = the numerically lowest = highest priority
    HIP_TRY(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi));
    HIP_TRY(hipEventCreateWithFlags(&c->order, hipEventDisableTiming));
    for (int i = 0; i < TICKETS; ++i) HIP_TRY(hipEventCreateWithFlags(&c->done[i], hipEventDisableTiming));
    ncclUniqueId id;
    memcpy(&id, uid, sizeof(id));
    NCCL_TRY(rccl().CommInitRank(&c->comm, c->world, id, c->rank));
    return 0;
}
}  // namespace

extern "C" int md_comm_init(md_comm** out, const void* uid, int32_t rank, int32_t world, int32_t device) {
    if (!out || !uid || world < 1 || rank < 0 || rank >= world || device < 0) return MD_COMM_BAD_ARG;
    *out = nullptr;
    if (!rccl().ok) return MD_COMM_NO_RCCL;
    HIP_TRY(hipSetDevice(device));
    md_comm* c = new md_comm();        // value-initialised: every handle null until created
    c->rank = rank;
    c->world = world;
    c->device = device;
    const int rc = comm_setup(c, uid);
    if (rc) {                          // nothing leaks: the stream, the events created so far and the struct go with destroy
        const std::string why = g_err;   // the message of the call that failed, not of the teardown
        (void)md_comm_destroy(c);
        g_err = why;
        return rc;
    }
    *out = c;
    return 0;
}

extern "C" int md_comm_destroy(md_comm* c) {
    if (!c) return MD_COMM_BAD_ARG;
    int rc = 0;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) {
        const ncclResult_t r = rccl().CommDestroy(c->comm);
        if (r != ncclSuccess) rc = fail(std::string("ncclCommDestroy: ") + rccl().GetErrorString(r));
    }
    for (int i = 0; i < TICKETS; ++i)
        if (c->done[i]) (void)hipEventDestroy(c->done[i]);
    if (c->order) (void)hipEventDestroy(c->order);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return rc;
}

extern "C" int md_comm_rank(const md_comm* c) { return c ? c->rank : MD_COMM_BAD_ARG; }
extern "C" int md_comm_world(const md_comm* c) { return c ? c->world : MD_COMM_BAD_ARG; }

namespace {

// Order the communicator's stream behind `after`, run `issue` on it, publish the completion event as the next ticket.
template <typename F>
int collective(md_comm* c, md_comm_stream after, int64_t* ticket, F&& issue) {
    if (!c) return MD_COMM_BAD_ARG;
    HIP_TRY(hipEventRecord(c->order, static_cast<hipStream_t>(after)));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->order, 0));
    const int rc = issue();
    if (rc) return rc;
    const int64_t t = ++c->issued;
    HIP_TRY(hipEventRecord(c->done[t % TICKETS], c->stream));
    if (ticket) *ticket = t;
    return 0;
}

}  // namespace

extern "C" int md_comm_allreduce_bucket(md_comm* c, void* buf, int64_t count, int32_t dtype, md_comm_stream after, int64_t* ticket) {
    ncclDataType_t dt;
    if (!c || !buf || count <= 0 || !dtype_of(dtype, &dt)) return MD_COMM_BAD_ARG;
    return collective(c, after, ticket, [&]() -> int {
        NCCL_TRY(rccl().AllReduce(buf, buf, (size_t)count, dt, ncclSum, c->comm, c->stream));
        return 0;
    });
}

extern "C" int md_comm_reduce_scatter_bucket(md_comm* c, const void* send, void* recv, int64_t recv_count, int32_t dtype,
                                             md_comm_stream after, int64_t* ticket) {
    ncclDataType_t dt;
    if (!c || !send || !recv || recv_count <= 0 || !dtype_of(dtype, &dt)) return MD_COMM_BAD_ARG;
    return collective(c, after, ticket, [&]() -> int {
        NCCL_TRY(rccl().ReduceScatter(send, recv, (size_t)recv_count, dt, ncclSum, c->comm, c->stream));
        return 0;
    });
}

extern "C" int md_comm_all_gather_bucket(md_comm* c, const void* send, void* recv, int64_t send_count, int32_t dtype,
                                         md_comm_stream after, int64_t* ticket) {
    ncclDataType_t dt;
    if (!c || !send || !recv || send_count <= 0 || !dtype_of(dtype, &dt)) return MD_COMM_BAD_ARG;
    return collective(c, after, ticket, [&]() -> int {
        NCCL_TRY(rccl().AllGather(send, recv, (size_t)send_count, dt, c->comm, c->stream));
        return 0;
    });
}

extern "C" int md_comm_wait(md_comm* c, int64_t ticket, md_comm_stream stream) {
    if (!c) return MD_COMM_BAD_ARG;
    if (ticket <= 0) return 0;
    if (ticket > c->issued || ticket + TICKETS <= c->issued) return MD_COMM_BAD_ARG;     // never issued / its event was recycled
    HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(stream), c->done[ticket % TICKETS], 0));
    return 0;
}

extern "C" int md_comm_query(md_comm* c, int64_t ticket) {
    if (!c) return MD_COMM_BAD_ARG;
    if (ticket <= 0) return 1;
    if (ticket > c->issued) return MD_COMM_BAD_ARG;
    if (ticket + TICKETS <= c->issued) return 1;         // its event was recycled: 1024 younger collectives have been issued since
    const hipError_t e = hipEventQuery(c->done[ticket % TICKETS]);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) return 0;
    HIP_TRY(e);
    return 0;
}

extern "C" int md_comm_synchronize(md_comm* c) {
    if (!c) return MD_COMM_BAD_ARG;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}
