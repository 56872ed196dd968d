// Data-parallel collectives behind the C ABI (bd_comm_*): RCCL over xGMI, one communicator per process / GPU.
//
// Replaces dist.bcast_list_ (configs/detection_cfg.py:80-82), dist.make_allreduce_cb (solver/default_solver.py:58-63,121) and
// the two-scalar all_reduce_mean of FCOS (models/det/fcos.py:143-144).  RCCL is resolved at run time (dlopen): a copy that is
// already mapped into the process wins (torch ships one), else BD_RCCL_LIB, else /opt/rocm/lib/librccl.so.1 -- the library
// itself has no link-time dependency on it, so the one-GPU path never touches RCCL.
//
// Stream model: the communicator owns ONE communication stream.  bd_comm_allreduce_async records an event on every producer
// stream the caller names (main stream, weight-gradient side stream), makes the communication stream wait for them and
// enqueues the collective there; compute streams never wait for each other or for the collective.  bd_comm_wait makes a
// consumer stream (the SGD launch) wait for everything enqueued so far on the communication stream.  Nothing blocks the host.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <rccl/rccl.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string path;
};

RcclApi g_api;
std::mutex g_api_mu;

bool load_rccl() {
    std::lock_guard<std::mutex> lk(g_api_mu);
    if (g_api.handle) return true;
    std::vector<std::string> names;
    if (const char* e = getenv("BD_RCCL_LIB")) names.push_back(e);
    void* h = nullptr;
    // a copy already mapped into the process (torch's wheel carries one): share it instead of mapping a second RCCL
    for (const char* n : {"librccl.so", "librccl.so.1"}) {
        if (!names.empty()) break;
        h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        if (h) {
            g_api.path = std::string(n) + " (already mapped)";
            break;
        }
    }
    if (!h) {
        names.push_back("/opt/rocm/lib/librccl.so.1");
        names.push_back("librccl.so.1");
        names.push_back("librccl.so");
        for (auto& n : names) {
            h = dlopen(n.c_str(), RTLD_NOW | RTLD_GLOBAL);
            if (h) {
                g_api.path = n;
                break;
            }
        }
    }
    if (!h) {
        bd_set_error("bd_comm: cannot load RCCL (%s)", dlerror());
        return false;
    }
#define BD_SYM(field, sym)                                            \
    g_api.field = (decltype(g_api.field))dlsym(h, sym);               \
    if (!g_api.field) {                                               \
        bd_set_error("bd_comm: %s lacks %s", g_api.path.c_str(), sym); \
        dlclose(h);                                                   \
        return false;                                                 \
    }
    BD_SYM(GetUniqueId, "ncclGetUniqueId")
    BD_SYM(CommInitRank, "ncclCommInitRank")
    BD_SYM(CommDestroy, "ncclCommDestroy")
    BD_SYM(CommAbort, "ncclCommAbort")
    BD_SYM(AllReduce, "ncclAllReduce")
    BD_SYM(Broadcast, "ncclBroadcast")
    BD_SYM(GetErrorString, "ncclGetErrorString")
#undef BD_SYM
    g_api.handle = h;
    return true;
}

}  // namespace

struct bd_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;       // the communication stream
    std::vector<hipEvent_t> ev;         // ring of producer events (reused; an event may be re-recorded once its wait is enqueued)
    size_t ev_next = 0;
    hipEvent_t done = nullptr;
};

#define BD_NCCL(call, what)                                                                \
    do {                                                                                   \
        ncclResult_t r__ = (call);                                                         \
        if (r__ != ncclSuccess) {                                                          \
            bd_set_error("%s: RCCL error %d (%s)", what, (int)r__, g_api.GetErrorString(r__)); \
            return BD_ELAUNCH;                                                             \
        }                                                                                  \
    } while (0)
#define BD_HIP(call, what)                                                   \
    do {                                                                     \
        hipError_t e__ = (call);                                             \
        if (e__ != hipSuccess) {                                             \
            bd_set_error("%s: %s", what, hipGetErrorString(e__));            \
            return BD_ELAUNCH;                                               \
        }                                                                    \
    } while (0)

static int to_nccl_dtype(int dtype, ncclDataType_t* out, size_t* esize) {
    switch (dtype) {
        case BD_COMM_F32: *out = ncclFloat32; *esize = 4; return 0;
        case BD_COMM_BF16: *out = ncclBfloat16; *esize = 2; return 0;
        case BD_COMM_I32: *out = ncclInt32; *esize = 4; return 0;
        case BD_COMM_F64: *out = ncclFloat64; *esize = 8; return 0;
    }
    return -1;
}

extern "C" {

int bd_comm_unique_id(void* id128_host) {
    BD_REQUIRE(id128_host != nullptr, "bd_comm_unique_id: null buffer");
    if (!load_rccl()) return BD_ELAUNCH;
    static_assert(sizeof(ncclUniqueId) == BD_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    BD_NCCL(g_api.GetUniqueId(&id), "bd_comm_unique_id");
    memcpy(id128_host, &id, sizeof(id));
    return BD_OK;
}

int bd_comm_init(bd_comm_t* out, const void* id128_host, int rank, int world, int device) {
    BD_REQUIRE(out && id128_host, "bd_comm_init: null argument");
    BD_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bd_comm_init: rank %d of %d", rank, world);
    if (!load_rccl()) return BD_ELAUNCH;
    BD_HIP(hipSetDevice(device), "bd_comm_init: hipSetDevice");
    bd_comm* c = new bd_comm();
    c->rank = rank; c->world = world; c->device = device;
    ncclUniqueId id;
    memcpy(&id, id128_host, sizeof(id));
    ncclResult_t r = g_api.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        bd_set_error("bd_comm_init: ncclCommInitRank failed: %d (%s) [%s]", (int)r, g_api.GetErrorString(r), g_api.path.c_str());
        delete c;
        return BD_ELAUNCH;
    }
    // high priority: a bucket's collective should not queue behind the backward kernels still being issued
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    c->ev.assign(16, nullptr);
    hipError_t e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi);
    for (size_t i = 0; e == hipSuccess && i < c->ev.size(); ++i) e = hipEventCreateWithFlags(&c->ev[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done, hipEventDisableTiming);
    if (e != hipSuccess) {          // release what exists: the communicator (abort: the peers may not have arrived), events, stream
        bd_set_error("bd_comm_init: stream / event creation: %s", hipGetErrorString(e));
        if (c->comm) g_api.CommAbort(c->comm);
        for (auto& ev : c->ev) if (ev) (void)hipEventDestroy(ev);
        if (c->done) (void)hipEventDestroy(c->done);
        if (c->stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return BD_ELAUNCH;
    }
    *out = c;
    return BD_OK;
}

int bd_comm_rank(bd_comm_t c) { return c ? c->rank : 0; }
int bd_comm_world(bd_comm_t c) { return c ? c->world : 1; }
bd_stream_t bd_comm_stream(bd_comm_t c) { return c ? (bd_stream_t)c->stream : nullptr; }

int bd_comm_bcast(bd_comm_t c, void* buf, size_t count, int dtype, int root, bd_stream_t stream) {
    BD_REQUIRE(c && buf, "bd_comm_bcast: null argument");
    ncclDataType_t dt; size_t es;
    BD_REQUIRE(to_nccl_dtype(dtype, &dt, &es) == 0, "bd_comm_bcast: dtype %d", dtype);
    BD_NCCL(g_api.Broadcast(buf, buf, count, dt, root, c->comm, (hipStream_t)stream), "bd_comm_bcast");
    return BD_OK;
}

int bd_comm_allreduce(bd_comm_t c, void* buf, size_t count, int dtype, int op, bd_stream_t stream) {
    BD_REQUIRE(c && buf, "bd_comm_allreduce: null argument");
    ncclDataType_t dt; size_t es;
    BD_REQUIRE(to_nccl_dtype(dtype, &dt, &es) == 0, "bd_comm_allreduce: dtype %d", dtype);
    BD_REQUIRE(op == BD_COMM_SUM || op == BD_COMM_MAX || op == BD_COMM_AVG, "bd_comm_allreduce: op %d", op);
    const ncclRedOp_t rop = op == BD_COMM_SUM ? ncclSum : (op == BD_COMM_MAX ? ncclMax : ncclAvg);
    BD_NCCL(g_api.AllReduce(buf, buf, count, dt, rop, c->comm, (hipStream_t)stream), "bd_comm_allreduce");
    return BD_OK;
}

int bd_comm_allreduce_async(bd_comm_t c, void* buf, size_t count, int dtype, int op, const bd_stream_t* producers,
                            int n_producers) {
    BD_REQUIRE(c && buf, "bd_comm_allreduce_async: null argument");
    BD_REQUIRE(n_producers >= 0 && n_producers <= 8, "bd_comm_allreduce_async: %d producer streams", n_producers);
    for (int i = 0; i < n_producers; ++i) {
        hipEvent_t e = c->ev[c->ev_next++ % c->ev.size()];
        BD_HIP(hipEventRecord(e, (hipStream_t)producers[i]), "bd_comm_allreduce_async: record");
        BD_HIP(hipStreamWaitEvent(c->stream, e, 0), "bd_comm_allreduce_async: wait");
    }
    return bd_comm_allreduce(c, buf, count, dtype, op, (bd_stream_t)c->stream);
}

namespace {
__global__ void comm_f32_to_bf16_kernel(const float* __restrict__ src, bf16_raw* __restrict__ dst, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] = f2bf(src[i]);
}
__global__ void comm_bf16_to_f32_kernel(const bf16_raw* __restrict__ src, float* __restrict__ dst, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] = bf2f(src[i]);
}
}  // namespace

int bd_comm_allreduce_async_bf16(bd_comm_t c, float* buf, void* tmp_bf16, size_t count, int op, const bd_stream_t* producers,
                                 int n_producers) {
    BD_REQUIRE(c && buf && tmp_bf16, "bd_comm_allreduce_async_bf16: null argument");
    BD_REQUIRE(n_producers >= 0 && n_producers <= 8, "bd_comm_allreduce_async_bf16: %d producer streams", n_producers);
    BD_REQUIRE(op == BD_COMM_SUM || op == BD_COMM_AVG, "bd_comm_allreduce_async_bf16: op %d", op);
    for (int i = 0; i < n_producers; ++i) {
        hipEvent_t e = c->ev[c->ev_next++ % c->ev.size()];
        BD_HIP(hipEventRecord(e, (hipStream_t)producers[i]), "bd_comm_allreduce_async_bf16: record");
        BD_HIP(hipStreamWaitEvent(c->stream, e, 0), "bd_comm_allreduce_async_bf16: wait");
    }
    if (count == 0) return BD_OK;
    const unsigned grid = (unsigned)((count + 255) / 256 > 4096 ? 4096 : (count + 255) / 256);
    hipLaunchKernelGGL(comm_f32_to_bf16_kernel, dim3(grid), dim3(256), 0, c->stream, (const float*)buf, (bf16_raw*)tmp_bf16, (long long)count);
    const int rc = bd_comm_allreduce(c, tmp_bf16, count, BD_COMM_BF16, op, (bd_stream_t)c->stream);
    if (rc != BD_OK) return rc;
    hipLaunchKernelGGL(comm_bf16_to_f32_kernel, dim3(grid), dim3(256), 0, c->stream, (const bf16_raw*)tmp_bf16, buf, (long long)count);
    BD_CHECK_LAUNCH("bd_comm_allreduce_async_bf16");
    return BD_OK;
}

int bd_comm_wait(bd_comm_t c, bd_stream_t consumer) {
    BD_REQUIRE(c != nullptr, "bd_comm_wait: null communicator");
    BD_HIP(hipEventRecord(c->done, c->stream), "bd_comm_wait: record");
    BD_HIP(hipStreamWaitEvent((hipStream_t)consumer, c->done, 0), "bd_comm_wait: wait");
    return BD_OK;
}

int bd_comm_destroy(bd_comm_t c) {
    if (!c) return BD_OK;
    (void)hipStreamSynchronize(c->stream);
    if (c->comm) g_api.CommDestroy(c->comm);
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return BD_OK;
}

}  // extern "C"
