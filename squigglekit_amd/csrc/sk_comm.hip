// sk_comm.hip -- the one exchange the hot path has across GPUs: the final gather of fixed-size result
// records (sk_hit, 24 B per read; segments) over RCCL / xGMI.
//
// The reference has no counterpart: its per-read loops (segmenter.py:189-230, MotifSeq.py:261-298) carry no
// state between reads, so a multi-GPU job is a contiguous block split of the reads with no data-path
// collective, and one all-gather at the end.  Two launch shapes are supported:
//   * one process, one host thread per GPU      sk_comm_init_all()   -> ncclCommInitAll
//   * one process per GPU (torch.distributed.run style launchers)
//                                               sk_comm_unique_id() on rank 0, handed to the other ranks
//                                               by the launcher-side code, then sk_comm_init_rank()
// librccl.so is dlopen()ed on first use, so the library loads (and every single-GPU entry point works) on a
// host without RCCL; when it cannot be loaded or a communicator cannot be created the calls return
// SK_ERR_UNSUPPORTED and the caller falls back to concatenating the shards on the host.
#include "sk_common.h"
#include <rccl/rccl.h>      // types only: every entry point is resolved with dlsym
#include <dlfcn.h>
#include <mutex>
#include <string.h>

namespace {

struct Rccl {
    void *h = nullptr;
    bool tried = false;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

Rccl *rccl()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (!g_rccl.tried) {
        g_rccl.tried = true;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names)
            if ((g_rccl.h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (g_rccl.h) {
            bool ok = true;
#define SK_SYM(field, sym) ok = ok && (*(void **)(&g_rccl.field) = dlsym(g_rccl.h, sym)) != nullptr
            SK_SYM(GetUniqueId, "ncclGetUniqueId");
            SK_SYM(CommInitRank, "ncclCommInitRank");
            SK_SYM(CommInitAll, "ncclCommInitAll");
            SK_SYM(CommDestroy, "ncclCommDestroy");
            SK_SYM(CommCount, "ncclCommCount");
            SK_SYM(CommUserRank, "ncclCommUserRank");
            SK_SYM(AllGather, "ncclAllGather");
            SK_SYM(GetErrorString, "ncclGetErrorString");
#undef SK_SYM
            if (!ok) { dlclose(g_rccl.h); g_rccl.h = nullptr; }
        }
    }
    return g_rccl.h ? &g_rccl : nullptr;
}

int no_rccl()
{
    const char *why = dlerror();
    return sk_fail(SK_ERR_UNSUPPORTED, "librccl.so could not be loaded (%s): gather the shards on the host instead",
                   why ? why : "missing symbol");
}

#define SK_NCCL(R, call)                                                                       \
    do {                                                                                        \
        ncclResult_t e_ = (call);                                                               \
        if (e_ != ncclSuccess)                                                                  \
            return sk_fail(SK_ERR_UNSUPPORTED, "%s failed: %s", #call, (R)->GetErrorString(e_)); \
    } while (0)

} // namespace

extern "C" {

int sk_comm_unique_id(void *id128)
{
    if (!id128) return sk_fail(SK_ERR_INVALID, "NULL id");
    Rccl *R = rccl();
    if (!R) return no_rccl();
    ncclUniqueId id;
    SK_NCCL(R, R->GetUniqueId(&id));
    static_assert(sizeof id == 128, "ncclUniqueId is 128 bytes in the C ABI of this entry point");
    memcpy(id128, &id, sizeof id);
    return SK_OK;
}

int sk_comm_init_rank(const void *id128, int nranks, int rank)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    if (!id128 || nranks <= 0 || rank < 0 || rank >= nranks) return sk_fail(SK_ERR_INVALID, "bad id / rank / nranks");
    if (c->comm) return sk_fail(SK_ERR_INVALID, "device %d already has a communicator", c->device);
    Rccl *R = rccl();
    if (!R) return no_rccl();
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t comm = nullptr;
    SK_NCCL(R, R->CommInitRank(&comm, nranks, id, rank));
    c->comm = comm;
    return SK_OK;
}

int sk_comm_init_all(const int *devices, int ndev)
{
    if (!devices || ndev <= 0 || ndev > SK_MAX_DEVICES) return sk_fail(SK_ERR_INVALID, "bad device list");
    Rccl *R = rccl();
    if (!R) return no_rccl();
    const int before = sk_bound_device();
    for (int i = 0; i < ndev; i++) {                      // every device needs its context (stream)
        const int rc = sk_init(devices[i]);
        if (rc) return rc;
        if (sk_ctx_of(devices[i])->comm)
            return sk_fail(SK_ERR_INVALID, "device %d already has a communicator", devices[i]);
        for (int j = 0; j < i; j++)
            if (devices[j] == devices[i]) return sk_fail(SK_ERR_INVALID, "device %d listed twice", devices[i]);
    }
    ncclComm_t comms[SK_MAX_DEVICES];
    SK_NCCL(R, R->CommInitAll(comms, ndev, devices));
    for (int i = 0; i < ndev; i++) sk_ctx_of(devices[i])->comm = comms[i];
    if (before >= 0) return sk_init_slot(before, sk_ctx_of(before)->device);   // the calling thread keeps its binding
    return SK_OK;
}

int sk_comm_info(int *nranks, int *rank)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    if (!c->comm) return sk_fail(SK_ERR_INVALID, "no communicator on device %d", c->device);
    Rccl *R = rccl();
    if (!R) return no_rccl();
    int n = 0, r = 0;
    SK_NCCL(R, R->CommCount((ncclComm_t)c->comm, &n));
    SK_NCCL(R, R->CommUserRank((ncclComm_t)c->comm, &r));
    if (nranks) *nranks = n;
    if (rank) *rank = r;
    return SK_OK;
}

// every rank contributes `bytes` bytes; d_recv (nranks * bytes) gets them in rank order.  Enqueued on the
// bound device's stream behind the kernels that produced d_send; sk_sync() waits for it.
int sk_comm_allgather_dev(const void *d_send, void *d_recv, size_t bytes)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    if (!c->comm) return sk_fail(SK_ERR_INVALID, "no communicator on device %d", c->device);
    if (bytes && (!d_send || !d_recv)) return sk_fail(SK_ERR_INVALID, "NULL buffer");
    Rccl *R = rccl();
    if (!R) return no_rccl();
    SK_NCCL(R, R->AllGather(d_send, d_recv, bytes, ncclUint8, (ncclComm_t)c->comm, c->stream));
    return SK_OK;
}

// small host-side exchange (timings, counts, a barrier): staged through device scratch, synchronous
int sk_comm_allgather_host(const void *send, void *recv, size_t bytes)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    if (!c->comm) return sk_fail(SK_ERR_INVALID, "no communicator on device %d", c->device);
    if (!send || !recv || bytes == 0) return sk_fail(SK_ERR_INVALID, "bad buffer");
    int n = 0;
    int rc = sk_comm_info(&n, nullptr);
    if (rc) return rc;
    if ((rc = sk_reserve(c, &c->commbuf, bytes * (size_t)(n + 1)))) return rc;
    char *d_send = (char *)c->commbuf.p, *d_recv = d_send + bytes;
    SK_HIP(hipMemcpyAsync(d_send, send, bytes, hipMemcpyHostToDevice, c->stream));
    if ((rc = sk_comm_allgather_dev(d_send, d_recv, bytes))) return rc;
    SK_HIP(hipMemcpyAsync(recv, d_recv, bytes * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    SK_HIP(hipStreamSynchronize(c->stream));
    return SK_OK;
}

int sk_comm_destroy(void)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    if (!c->comm) return SK_OK;
    Rccl *R = rccl();
    if (!R) return no_rccl();
    SK_HIP(hipStreamSynchronize(c->stream));
    ncclComm_t comm = (ncclComm_t)c->comm;
    c->comm = nullptr;
    SK_NCCL(R, R->CommDestroy(comm));
    return SK_OK;
}

} // extern "C"
