// tbk_comm.hip -- the one exchange of the sharded path: an RCCL all-gather of per-rank eigenvalue
// slabs over xGMI.  One process per GPU; the launcher (bench.py / tbmodels_amd.sharding) hands the
// 128-byte unique id from rank 0 to the other ranks over its own rendezvous.
//
// The reference has no distributed code at all; k-points are independent
// (/root/reference/src/tbmodels/_tb_model.py:1111-1123 has no cross-k term), so the hoppings are
// replicated, the k list is split into contiguous slabs, and this gather is the only collective.

#include <rccl/rccl.h>

#include <cstring>
#include <new>

#include "tbk_internal.h"

struct tbk_comm {
    int device = 0;
    int world = 1;
    int rank = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;  // collectives that overlap the next batch's kernels run here
    hipEvent_t ready = nullptr;    // "the send buffer is complete" on the model's stream
    hipEvent_t done[2] = {nullptr, nullptr};  // gather of buffer slot 0 / 1 finished
};

#define TBK_NCCL(expr)                                                                           \
    do {                                                                                         \
        ncclResult_t r_ = (expr);                                                                \
        if (r_ != ncclSuccess) {                                                                 \
            tbk_set_error("%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__,      \
                          __LINE__);                                                             \
            return TBK_ERR_DEVICE;                                                               \
        }                                                                                        \
    } while (0)

static_assert(sizeof(ncclUniqueId) == 128, "tbk_comm_unique_id hands out 128 bytes");

extern "C" int tbk_comm_unique_id(void* id128) {
    TBK_ARG(id128 != nullptr, "id buffer is NULL");
    ncclUniqueId id;
    TBK_NCCL(ncclGetUniqueId(&id));
    std::memcpy(id128, &id, sizeof(id));
    return TBK_OK;
}

extern "C" int tbk_comm_create(int device, int world_size, int rank, const void* id128, tbk_comm** out) {
    TBK_ARG(out != nullptr && id128 != nullptr, "out / id is NULL");
    TBK_ARG(world_size >= 1 && rank >= 0 && rank < world_size, "bad rank / world size");
    *out = nullptr;
    TBK_HIP(hipSetDevice(device));
    tbk_comm* c = new (std::nothrow) tbk_comm();
    if (!c) {
        tbk_set_error("out of host memory");
        return TBK_ERR_MEMORY;
    }
    c->device = device;
    c->world = world_size;
    c->rank = rank;
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclResult_t r = ncclCommInitRank(&c->comm, world_size, id, rank);
    if (r != ncclSuccess) {
        tbk_set_error("ncclCommInitRank failed: %s", ncclGetErrorString(r));
        delete c;
        return TBK_ERR_DEVICE;
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->done[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->done[1], hipEventDisableTiming) != hipSuccess) {
        tbk_set_error("cannot create the communicator's stream / event");
        tbk_comm_destroy(c);
        return TBK_ERR_DEVICE;
    }
    *out = c;
    return TBK_OK;
}

extern "C" void tbk_comm_destroy(tbk_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->ready) (void)hipEventDestroy(c->ready);
    for (hipEvent_t e : c->done)
        if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int tbk_comm_allgather_f64(tbk_comm* c, tbk_model* m, const double* d_send, double* d_recv,
                                      int64_t count) {
    TBK_ARG(c != nullptr, "comm is NULL");
    TBK_ARG(count >= 0, "count < 0");
    if (count == 0) return TBK_OK;
    TBK_ARG(d_send && d_recv, "send / recv is NULL");
    TBK_HIP(hipSetDevice(c->device));
    // m == NULL (a rank whose staging failed still has to take part): the communicator's own stream
    const bool ranged = m != nullptr && m->timing;
    if (ranged) tbk_range_push("tbk:allgather_eigenvalues");
    const ncclResult_t r = ncclAllGather(d_send, d_recv, (size_t)count, ncclDouble, c->comm, m ? m->stream : c->stream);
    if (ranged) tbk_range_pop();
    TBK_NCCL(r);
    return TBK_OK;
}

// Same exchange, off the model's stream: the gather starts when everything enqueued on the model's stream so
// far has finished, and runs on the communicator's own stream -- the next batch's kernels overlap it.
// Callers alternate two (send, recv) buffer pairs, `slot` 0 / 1; before a pair is written again,
// tbk_comm_wait_slot makes the model's stream wait for that pair's previous gather (a stream dependency, the
// host does not block).
extern "C" int tbk_comm_allgather_f64_overlapped(tbk_comm* c, tbk_model* m, const double* d_send, double* d_recv,
                                                 int64_t count, int slot) {
    TBK_ARG(c != nullptr && m != nullptr, "comm / model is NULL");
    TBK_ARG(count >= 0 && (slot == 0 || slot == 1), "bad count / slot");
    if (count == 0) return TBK_OK;
    TBK_ARG(d_send && d_recv, "send / recv is NULL");
    TBK_HIP(hipSetDevice(c->device));
    TBK_HIP(hipEventRecord(c->ready, m->stream));
    TBK_HIP(hipStreamWaitEvent(c->stream, c->ready, 0));
    if (m->timing) tbk_range_push("tbk:allgather_eigenvalues(overlapped)");
    const ncclResult_t r = ncclAllGather(d_send, d_recv, (size_t)count, ncclDouble, c->comm, c->stream);
    if (m->timing) tbk_range_pop();
    TBK_NCCL(r);
    TBK_HIP(hipEventRecord(c->done[slot], c->stream));
    return TBK_OK;
}

extern "C" int tbk_comm_wait_slot(tbk_comm* c, tbk_model* m, int slot) {
    TBK_ARG(c != nullptr && m != nullptr, "comm / model is NULL");
    TBK_ARG(slot == 0 || slot == 1, "bad slot");
    TBK_HIP(hipSetDevice(c->device));
    TBK_HIP(hipStreamWaitEvent(m->stream, c->done[slot], 0));  // no-op if the slot was never used
    return TBK_OK;
}

// Ranks that joined the communicator, as RCCL itself reports it (ncclCommCount) -- evidence that a multi-GPU run
// really went through an N-rank communicator.
extern "C" int tbk_comm_ranks(tbk_comm* c, int* count, int* rank) {
    TBK_ARG(c != nullptr && count != nullptr, "comm / count is NULL");
    int n = 0, r = 0;
    TBK_NCCL(ncclCommCount(c->comm, &n));
    TBK_NCCL(ncclCommUserRank(c->comm, &r));
    *count = n;
    if (rank) *rank = r;
    return TBK_OK;
}

extern "C" int tbk_comm_synchronize(tbk_comm* c) {
    TBK_ARG(c != nullptr, "comm is NULL");
    TBK_HIP(hipSetDevice(c->device));
    TBK_HIP(hipStreamSynchronize(c->stream));
    return TBK_OK;
}
