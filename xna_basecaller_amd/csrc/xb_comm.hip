// xb_comm.hip -- the path's one collective behind the C ABI: the gather of called sequences over RCCL (xGMI).
//
// SURVEY.md 8(b),(e): reads shard over the GPUs with no data-path collective; what is exchanged is the packed sequences
// and their lengths of one batch per rank (n * T + 4 n bytes, ~1 MB at n = 512).  The reference has no distributed code at
// all; this replaces what xna_basecaller_amd/dist.py did through torch.distributed ("nccl" backend = the same RCCL).
// librccl is opened at run time (dlopen, RTLD_LOCAL) so that libxnacall.so neither links RCCL nor clashes with the copy a
// torch process has already loaded; XB_RCCL_LIB overrides the library name.
// Ordering: xb_gather_called makes the communicator's own stream wait for an event recorded on the context's result stream
// (the stream that produces d_seq / d_seq_len), then enqueues the two all-gathers there: the next batch computes meanwhile.
// xb_comm_fence is the opposite edge (the context's streams wait for the gathers issued so far) for callers that rotate
// a small set of output buffers; xb_comm_synchronize is the host-side completion point.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/xna_basecaller.h"

namespace {

typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;      // NCCL_UNIQUE_ID_BYTES (rccl.h:40-43)
enum { NCCL_SUCCESS = 0, NCCL_INT8 = 0, NCCL_INT32 = 2 };  // ncclDataType_t values (rccl.h:459-461)

struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string err;
} g_rccl;

thread_local std::string g_comm_error;

bool load_rccl()
{
    if (g_rccl.lib) return true;
    const char *names[] = {getenv("XB_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        if (!n || !*n) continue;
        if ((g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    }
    if (!g_rccl.lib) {
        g_rccl.err = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?");
        return false;
    }
#define XB_SYM(field, name)                                                                   \
    *reinterpret_cast<void **>(&g_rccl.field) = dlsym(g_rccl.lib, name);                      \
    if (!g_rccl.field) { g_rccl.err = std::string("librccl lacks ") + name; dlclose(g_rccl.lib); g_rccl.lib = nullptr; return false; }
    XB_SYM(GetUniqueId, "ncclGetUniqueId")
    XB_SYM(CommInitRank, "ncclCommInitRank")
    XB_SYM(CommDestroy, "ncclCommDestroy")
    XB_SYM(AllGather, "ncclAllGather")
    XB_SYM(GroupStart, "ncclGroupStart")
    XB_SYM(GroupEnd, "ncclGroupEnd")
    XB_SYM(GetErrorString, "ncclGetErrorString")
#undef XB_SYM
    return true;
}

}  // namespace

struct xb_comm {
    int device = 0, rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ready = nullptr, done[2] = {};       // producer -> gather; gather -> producer (the last two gathers)
    unsigned issued = 0;          // gathers enqueued on the stream so far
    unsigned queued = 0;          // gathers asked for so far (a gather can wait for a held-back basecall: xb_gather_called)
    xb_ctx *waiting_on = nullptr; // the context whose held-back basecall carries this communicator's deferred gather (queued >
                                  // issued): xb_comm_synchronize / xb_comm_destroy launch that call first, so that no deferred
                                  // gather ever outlives its communicator
    int failed_rc = 0;            // a collective of this communicator failed (or could not be enqueued): STICKY.  The other ranks
                                  // have entered, or will enter, that all-gather; skipping it here and carrying on would leave
                                  // them blocked in it.  Every later gather / fence / synchronize fails with this code and the
                                  // caller has to abort all ranks (no collective of a failed communicator can be trusted).
    std::string err;
};

namespace {
int cfail(xb_comm *c, int code, const std::string &msg)
{
    if (c) c->err = msg; else g_comm_error = msg;
    return code;
}
}  // namespace

extern "C" {

XB_API const char *xb_comm_last_error(const xb_comm *comm) { return comm ? comm->err.c_str() : g_comm_error.c_str(); }

XB_API int xb_comm_unique_id(char id[XB_COMM_ID_BYTES])
{
    if (!id) return cfail(nullptr, XB_ERR_INVALID, "null argument");
    if (!load_rccl()) return cfail(nullptr, XB_ERR_STATE, g_rccl.err);
    ncclUniqueId u;
    const int rc = g_rccl.GetUniqueId(&u);
    if (rc != NCCL_SUCCESS) return cfail(nullptr, XB_ERR_DEVICE, std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(rc));
    static_assert(sizeof u == XB_COMM_ID_BYTES, "unique id size");
    memcpy(id, &u, sizeof u);
    return XB_OK;
}

XB_API int xb_comm_create(xb_comm **out, int device, int rank, int world, const char id[XB_COMM_ID_BYTES])
{
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return cfail(nullptr, XB_ERR_INVALID, "bad argument");
    *out = nullptr;
    if (!load_rccl()) return cfail(nullptr, XB_ERR_STATE, g_rccl.err);
    if (hipSetDevice(device) != hipSuccess) return cfail(nullptr, XB_ERR_HIP, "hipSetDevice failed");
    xb_comm *c = new (std::nothrow) xb_comm();
    if (!c) return cfail(nullptr, XB_ERR_NOMEM, "out of host memory");
    c->device = device; c->rank = rank; c->world = world;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    const int rc = g_rccl.CommInitRank(&c->comm, world, u, rank);
    if (rc != NCCL_SUCCESS) {
        const int code = cfail(nullptr, XB_ERR_DEVICE, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(rc));
        delete c;
        return code;
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->done[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->done[1], hipEventDisableTiming) != hipSuccess) {
        xb_comm_destroy(c);
        return cfail(nullptr, XB_ERR_HIP, "stream / event creation failed");
    }
    *out = c;
    return XB_OK;
}

namespace {
// a gather deferred behind a held-back basecall (xb_gather_called): launch that call now -- its gather runs right behind it
void flush_deferred(xb_comm *c)
{
    if (c->queued != c->issued && c->waiting_on) (void)xb_result_stream(c->waiting_on);
    c->waiting_on = nullptr;
}
}  // namespace

XB_API void xb_comm_destroy(xb_comm *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    flush_deferred(c);            // the caller's buffers and the context of a deferred gather must still be alive here (header)
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    if (c->ready) (void)hipEventDestroy(c->ready);
    for (auto &e : c->done) if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

XB_API int xb_comm_rank(const xb_comm *c) { return c ? c->rank : -1; }
XB_API int xb_comm_world(const xb_comm *c) { return c ? c->world : 0; }

extern "C" int xb_internal_defer_after(xb_ctx *ctx, const void *d_seq, void (*fn)(void *), void *arg);

namespace {
struct GatherArgs {
    xb_comm *c; xb_ctx *ctx; const int8_t *d_seq; const int32_t *d_len; int n, T; int8_t *all_seq; int32_t *all_len;
};
int gather_now(xb_comm *c, xb_ctx *ctx, const int8_t *d_seq, const int32_t *d_seq_len, int n, int T, int8_t *d_all_seq,
               int32_t *d_all_len);
// the gather of a call that was held back when xb_gather_called came: runs right behind that call's launch
void gather_later(void *p)
{
    GatherArgs *g = static_cast<GatherArgs *>(p);
    g->c->waiting_on = nullptr;
    const int rc = gather_now(g->c, g->ctx, g->d_seq, g->d_len, g->n, g->T, g->all_seq, g->all_len);
    (void)rc;                                       // a failure is sticky in the communicator: the next call reports it
    delete g;
}
}  // namespace

XB_API int xb_gather_called(xb_comm *c, xb_ctx *ctx, const int8_t *d_seq, const int32_t *d_seq_len, int n, int T,
                            int8_t *d_all_seq, int32_t *d_all_len)
{
    if (!c) return XB_ERR_INVALID;
    if (!d_seq || !d_seq_len || !d_all_seq || !d_all_len || n < 1 || T < 1) return cfail(c, XB_ERR_INVALID, "bad argument");
    if (c->failed_rc != XB_OK) return c->failed_rc;           // xb_comm_last_error holds the first failure
    if (ctx) {
        // the basecall that writes d_seq may be held back to share a pass with the next one (xb_basecall_chunks_dev): the gather
        // is then enqueued right behind its launch, in the order of the xb_gather_called calls
        GatherArgs *g = new GatherArgs{c, ctx, d_seq, d_seq_len, n, T, d_all_seq, d_all_len};
        if (xb_internal_defer_after(ctx, d_seq, &gather_later, g)) { c->queued += 1; c->waiting_on = ctx; return XB_OK; }
        delete g;
        if (c->queued != c->issued) (void)xb_result_stream(ctx);     // an older gather is still waiting for its basecall: launch it first
    }
    c->queued += 1;
    return gather_now(c, ctx, d_seq, d_seq_len, n, T, d_all_seq, d_all_len);
}

namespace {
int gather_now(xb_comm *c, xb_ctx *ctx, const int8_t *d_seq, const int32_t *d_seq_len, int n, int T, int8_t *d_all_seq,
               int32_t *d_all_len)
{
    if (c->failed_rc != XB_OK) return c->failed_rc;
    // any failure from here on poisons the communicator (failed_rc): this rank's collective sequence no longer matches the others'
    auto poison = [c](int code, const std::string &msg) { c->failed_rc = code; return cfail(c, code, msg); };
    if (hipSetDevice(c->device) != hipSuccess) return poison(XB_ERR_HIP, "hipSetDevice failed");
    // order the gather behind the stream that produces (d_seq, d_seq_len); without a context the caller has synchronised
    if (ctx) {
        hipStream_t rs = static_cast<hipStream_t>(xb_result_stream(ctx));
        if (hipEventRecord(c->ready, rs) != hipSuccess || hipStreamWaitEvent(c->stream, c->ready, 0) != hipSuccess)
            return poison(XB_ERR_HIP, "event hand-off to the gather stream failed");
    }
    int rc = g_rccl.GroupStart();
    if (rc != NCCL_SUCCESS) return poison(XB_ERR_DEVICE, std::string("ncclGroupStart: ") + g_rccl.GetErrorString(rc));
    rc = g_rccl.AllGather(d_seq_len, d_all_len, (size_t)n, NCCL_INT32, c->comm, c->stream);
    if (rc == NCCL_SUCCESS) rc = g_rccl.AllGather(d_seq, d_all_seq, (size_t)n * T, NCCL_INT8, c->comm, c->stream);
    const int rc2 = g_rccl.GroupEnd();              // only after a successful GroupStart
    if (rc == NCCL_SUCCESS) rc = rc2;
    if (rc != NCCL_SUCCESS) return poison(XB_ERR_DEVICE, std::string("ncclAllGather: ") + g_rccl.GetErrorString(rc));
    if (hipEventRecord(c->done[c->issued & 1], c->stream) != hipSuccess) return poison(XB_ERR_HIP, "hipEventRecord failed");
    c->issued += 1;
    return XB_OK;
}
}  // namespace

XB_API int xb_comm_fence(xb_comm *c, xb_ctx *ctx, int lag)
{
    if (!c || !ctx || lag < 0 || lag > 1) return XB_ERR_INVALID;
    if (c->failed_rc != XB_OK) return c->failed_rc;
    if (c->queued <= (unsigned)lag) return XB_OK;           // nothing that old has been asked for
    if (hipSetDevice(c->device) != hipSuccess) return cfail(c, XB_ERR_HIP, "hipSetDevice failed");
    // the gathers run in order on one stream: waiting for gather (latest - lag) covers every earlier one
    const unsigned target = c->queued - 1 - (unsigned)lag;
    if (target >= c->issued) (void)xb_result_stream(ctx);  // it still waits for a held-back basecall: launch that now
    if (c->failed_rc != XB_OK) return c->failed_rc;         // ... whose gather may just have failed
    if (target >= c->issued || c->issued - target > 2) return cfail(c, XB_ERR_STATE, "fence: that gather's event is gone");
    hipEvent_t ev = c->done[target & 1];
    return xb_stream_wait_event(ctx, ev) == XB_OK ? XB_OK : cfail(c, XB_ERR_HIP, "stream wait failed");
}

XB_API int xb_comm_synchronize(xb_comm *c)
{
    if (!c) return XB_ERR_INVALID;
    if (hipSetDevice(c->device) != hipSuccess) return cfail(c, XB_ERR_HIP, "hipSetDevice failed");
    flush_deferred(c);            // "every gather asked for so far is complete" includes one still waiting for its basecall
    if (hipStreamSynchronize(c->stream) != hipSuccess) return cfail(c, XB_ERR_HIP, "gather stream failed");
    return c->failed_rc;            // XB_OK, or the sticky failure of an earlier (possibly deferred) gather
}

}  // extern "C"
