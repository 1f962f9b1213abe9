// RCCL collectives behind the C ABI (`jamie_allreduce(flat_g, n, comm, stream)`), for the data-parallel exchange
// of the flat gradient / the sharded optimiser (jamie_amd/distributed.py).  The reference has no distributed code
// (SURVEY.md 2.1); the exchange sits between `batch_loss.backward()` (jamie.py:734) and `clip_grad_norm_` (jamie.py:739).
//
// Why not torch.distributed: every collective enqueued through it costs the host ~28 us (Python -> c10d -> work object), and the
// sharded step makes ten of them: 480 us of host time per step against 465 us of GPU work per rank (tools/
// bench_sharded_host_time.py).  Here a collective is ONE foreign call, recordable in the launch plan like a kernel launch:
//     record an event on the caller's stream -> the communicator's own stream waits for it -> ncclXxx on that stream ->
//     record the slot's completion event;   jamie_comm_wait(slot) makes a stream wait for that event.
// The collective therefore overlaps the kernels the caller launches afterwards, exactly as ProcessGroupNCCL arranges it.
//
// librccl is bound at RUN TIME (dlopen; RTLD_NOLOAD first, so that the RCCL instance PyTorch has already loaded is shared):
// libjamie_hip.so itself has no link-time dependency on it and loads on a box without it.
#include "common.h"
#include <dlfcn.h>

typedef struct { char internal[128]; } jc_unique_id;
typedef void* jc_comm_t;
typedef int jc_result_t;
enum { JC_SUM = 0, JC_F32 = 7, JC_BF16 = 9 };          // ncclSum, ncclFloat32, ncclBfloat16 (rccl.h)

struct JcApi {
    void* lib;
    jc_result_t (*GetUniqueId)(jc_unique_id*);
    jc_result_t (*CommInitRank)(jc_comm_t*, int, jc_unique_id, int);
    jc_result_t (*CommDestroy)(jc_comm_t);
    jc_result_t (*AllReduce)(const void*, void*, size_t, int, int, jc_comm_t, hipStream_t);
    jc_result_t (*ReduceScatter)(const void*, void*, size_t, int, int, jc_comm_t, hipStream_t);
    jc_result_t (*AllGather)(const void*, void*, size_t, int, jc_comm_t, hipStream_t);
    const char* (*GetErrorString)(jc_result_t);
    int (*GetVersion)(int*);
};
static JcApi g_api;

static int jc_bind() {
    if (g_api.lib) return 0;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    for (const char* n : names) if ((lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;       // the instance already in the process
    if (!lib) for (const char* n : names) if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!lib) return jamie_fail(-2, "%s: librccl.so not found (%lld %lld)", "jamie_comm", 0, 0);
#define JC_SYM(field, name)                                                                     \
    *(void**)(&g_api.field) = dlsym(lib, name);                                                 \
    if (!g_api.field) return jamie_fail(-2, "%s: symbol missing in librccl [%lld %lld]", name, 0, 0);
    JC_SYM(GetUniqueId, "ncclGetUniqueId") JC_SYM(CommInitRank, "ncclCommInitRank") JC_SYM(CommDestroy, "ncclCommDestroy")
    JC_SYM(AllReduce, "ncclAllReduce") JC_SYM(ReduceScatter, "ncclReduceScatter") JC_SYM(AllGather, "ncclAllGather")
    JC_SYM(GetErrorString, "ncclGetErrorString") JC_SYM(GetVersion, "ncclGetVersion")
#undef JC_SYM
    g_api.lib = lib;
    return 0;
}

#define JC_SLOTS 32
struct JamieComm {
    jc_comm_t comm;
    int rank, world;
    hipStream_t stream;                 // the communicator's own stream: collectives run beside the caller's kernels
    hipEvent_t ready;                   // caller's stream -> communicator stream
    hipEvent_t done[JC_SLOTS];          // completion of the collective last issued with that slot
};

static int jc_fail(const char* who, jc_result_t r) {
    snprintf(g_jamie_err, sizeof(g_jamie_err), "%s: RCCL error %d: %s", who, (int)r, g_api.GetErrorString ? g_api.GetErrorString(r) : "?");
    return 1000 + (int)r;
}
#define JC_HIP(call)                                                                                       \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess) {                                                                            \
            snprintf(g_jamie_err, sizeof(g_jamie_err), "%s: %s", __func__, hipGetErrorString(e_));         \
            return (int)e_;                                                                                \
        }                                                                                                  \
    } while (0)

extern "C" int jamie_comm_version(int* version /*host*/) {
    JAMIE_ARG(version != nullptr, "null pointer");
    const int rc = jc_bind();
    if (rc) return rc;
    return g_api.GetVersion(version) == 0 ? 0 : jamie_fail(-2, "%s: ncclGetVersion failed [%lld %lld]", "jamie_comm_version", 0, 0);
}

extern "C" int jamie_comm_unique_id(void* id128 /*host, 128 bytes*/) {
    JAMIE_ARG(id128 != nullptr, "null pointer");
    const int rc = jc_bind();
    if (rc) return rc;
    const jc_result_t r = g_api.GetUniqueId((jc_unique_id*)id128);
    return r == 0 ? 0 : jc_fail("jamie_comm_unique_id", r);
}

extern "C" int jamie_comm_create(const void* id128 /*host*/, int rank, int world, void** comm_out /*host*/) {
    JAMIE_ARG(id128 != nullptr && comm_out != nullptr && world >= 1 && rank >= 0 && rank < world, "0 <= rank < world");
    const int rc = jc_bind();
    if (rc) return rc;
    JamieComm* c = new JamieComm();
    memset(c, 0, sizeof(*c));
    c->rank = rank; c->world = world;
    jc_unique_id id;
    memcpy(&id, id128, sizeof(id));
    const jc_result_t r = g_api.CommInitRank(&c->comm, world, id, rank);
    if (r != 0) { delete c; return jc_fail("jamie_comm_create", r); }
    // a failure from here on must not leak the communicator (RCCL holds device buffers and proxy threads for it), the stream or
    // the events made so far: everything is torn down again before the error goes back
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ready, hipEventDisableTiming);
    for (int i = 0; i < JC_SLOTS && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&c->done[i], hipEventDisableTiming);
    if (e != hipSuccess) {
        snprintf(g_jamie_err, sizeof(g_jamie_err), "jamie_comm_create: %s", hipGetErrorString(e));
        if (g_api.CommDestroy) g_api.CommDestroy(c->comm);
        for (int i = 0; i < JC_SLOTS; ++i)
            if (c->done[i]) (void)hipEventDestroy(c->done[i]);
        if (c->ready) (void)hipEventDestroy(c->ready);
        if (c->stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return (int)e;
    }
    *comm_out = c;
    return 0;
}

extern "C" int jamie_comm_destroy(void* comm) {
    JamieComm* c = (JamieComm*)comm;
    if (!c) return 0;
    (void)hipStreamSynchronize(c->stream);
    if (g_api.CommDestroy) g_api.CommDestroy(c->comm);
    for (int i = 0; i < JC_SLOTS; ++i) (void)hipEventDestroy(c->done[i]);
    (void)hipEventDestroy(c->ready);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

static int jc_dtype(int dtype, int* nccl, size_t* size) {
    if (dtype == 0) { *nccl = JC_F32; *size = 4; return 0; }
    if (dtype == 1) { *nccl = JC_BF16; *size = 2; return 0; }
    return jamie_fail(-1, "%s: dtype must be 0 (fp32) or 1 (bf16) [%lld %lld]", "jamie_comm", dtype, 0);
}

// every collective: the communicator's stream first waits for what the caller's stream has launched so far
static int jc_enter(JamieComm* c, int slot, hipStream_t st) {
    JAMIE_ARG(c != nullptr && slot >= 0 && slot < JC_SLOTS, "communicator, 0 <= slot < 32");
    JC_HIP(hipEventRecord(c->ready, st));
    JC_HIP(hipStreamWaitEvent(c->stream, c->ready, 0));
    return 0;
}

/* In-place SUM all-reduce of buf[0 .. count) over the ranks. */
extern "C" int jamie_allreduce(void* comm, void* buf, long long count, int dtype, int slot, void* stream) {
    JamieComm* c = (JamieComm*)comm;
    int nt; size_t sz;
    int rc = jc_dtype(dtype, &nt, &sz);
    if (rc) return rc;
    JAMIE_ARG(buf != nullptr && count > 0, "buffer");
    if ((rc = jc_enter(c, slot, (hipStream_t)stream))) return rc;
    const jc_result_t r = g_api.AllReduce(buf, buf, (size_t)count, nt, JC_SUM, c->comm, c->stream);
    if (r != 0) return jc_fail("jamie_allreduce", r);
    JC_HIP(hipEventRecord(c->done[slot], c->stream));
    return 0;
}

/* recv[0 .. recv_count) = SUM over ranks of send[rank * recv_count .. (rank + 1) * recv_count). */
extern "C" int jamie_reduce_scatter(void* comm, const void* send, void* recv, long long recv_count, int dtype, int slot, void* stream) {
    JamieComm* c = (JamieComm*)comm;
    int nt; size_t sz;
    int rc = jc_dtype(dtype, &nt, &sz);
    if (rc) return rc;
    JAMIE_ARG(send != nullptr && recv != nullptr && recv_count > 0, "buffers");
    if ((rc = jc_enter(c, slot, (hipStream_t)stream))) return rc;
    const jc_result_t r = g_api.ReduceScatter(send, recv, (size_t)recv_count, nt, JC_SUM, c->comm, c->stream);
    if (r != 0) return jc_fail("jamie_reduce_scatter", r);
    JC_HIP(hipEventRecord(c->done[slot], c->stream));
    return 0;
}

/* recv[r * send_count .. (r + 1) * send_count) = rank r's send[0 .. send_count). */
extern "C" int jamie_all_gather(void* comm, const void* send, void* recv, long long send_count, int dtype, int slot, void* stream) {
    JamieComm* c = (JamieComm*)comm;
    int nt; size_t sz;
    int rc = jc_dtype(dtype, &nt, &sz);
    if (rc) return rc;
    JAMIE_ARG(send != nullptr && recv != nullptr && send_count > 0, "buffers");
    if ((rc = jc_enter(c, slot, (hipStream_t)stream))) return rc;
    const jc_result_t r = g_api.AllGather(send, recv, (size_t)send_count, nt, c->comm, c->stream);
    if (r != 0) return jc_fail("jamie_all_gather", r);
    JC_HIP(hipEventRecord(c->done[slot], c->stream));
    return 0;
}

/* `stream` waits for the collective last issued with `slot` (a no-op on the host: the wait is queued on the device). */
extern "C" int jamie_comm_wait(void* comm, int slot, void* stream) {
    JamieComm* c = (JamieComm*)comm;
    JAMIE_ARG(c != nullptr && slot >= 0 && slot < JC_SLOTS, "communicator, 0 <= slot < 32");
    JC_HIP(hipStreamWaitEvent((hipStream_t)stream, c->done[slot], 0));
    return 0;
}
