// tbk_comm.hip -- the one exchange of the sharded path: an RCCL all-gather of per-rank eigenvalue
// slabs over xGMI.  One process per GPU; the launcher (bench.py / tbmodels_amd.sharding) hands the
// 128-byte unique id from rank 0 to the other ranks over its own rendezvous.
//
// The reference has no distributed code at all; k-points are independent
// (/root/reference/src/tbmodels/_tb_model.py:1111-1123 has no cross-k term), so the hoppings are
// replicated, the k list is split into contiguous slabs, and this gather is the only collective.

#include <rccl/rccl.h>

#include <algorithm>
#include <cstdlib>
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
    // chunk-pipelined gather of one call (tbk_eigenval_device_gather)
    double* d_stage = nullptr;      // [world][block rows * n_orb] landing area of one block's all-gather
    size_t stage_bytes = 0;
    double* d_status = nullptr;     // [1 + world] this rank's status word, then everybody's
    hipEvent_t tail = nullptr;      // "the main stream has everything of the call behind it"
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
        hipEventCreateWithFlags(&c->done[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->tail, hipEventDisableTiming) != hipSuccess ||
        hipMalloc((void**)&c->d_status, (size_t)(1 + world_size) * sizeof(double)) != hipSuccess) {
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
    if (c->tail) (void)hipEventDestroy(c->tail);
    if (c->d_stage) (void)hipFree(c->d_stage);
    if (c->d_status) (void)hipFree(c->d_status);
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


// ------------------------------------------------------------------------------------------------
// One sharded eigenvalue call with the gather pipelined behind the k chunks (BASELINE config 4: the 10^6-point mesh
// in 8 slabs is 13 ms of kernels per rank -- a 3 ms all-gather at the end of the call, with nothing to hide under,
// would be a fifth of the step).
//
// d_all is the RESULT in caller order: [world][per][n_orb], rank r's slab at row r * per.  This rank evaluates its
// nk <= per k-points straight into its own rows; the slab is cut into blocks of B rows (B from per and n_orb alone, so
// every rank cuts alike whatever its chunk pipeline does), and as soon as the chunk pipeline has enqueued the
// eigenvalues of a whole block (tbk_model::chunk_done) the communicator's stream waits for that chunk's event, gathers
// the block from every rank into a landing area ([rank][block], what ncclAllGather writes) and a copy kernel moves the
// ranks' pieces to their rows of d_all -- while the next chunk computes.  Only the last block's gather is exposed.
// Rows nk..per of a short slab are zero.  The status word (this rank's `host_status`, else what the solvers' flag words
// say: non-finite / no convergence) is gathered LAST, one double per rank, so that every rank sees every rank's status
// and all raise alike; d_status_all[world] (device) receives it.  Everything is enqueued; tbk_comm_synchronize waits.
// Reference invariants: k-points are independent (_tb_model.py:1111-1123), results in the order of k (:1148-1150).
// ------------------------------------------------------------------------------------------------
namespace {

__global__ void __launch_bounds__(256) gather_place_kernel(const double2* __restrict__ stage, double2* __restrict__ all, int64_t block_pairs,
                                                           int64_t slab_pairs, int64_t offset_pairs) {
    // piece r of the landing area -> rows [r * per + b0, ...) of the result; 16 bytes per thread and trip
    const int r = blockIdx.y;
    const double2* src = stage + (size_t)r * block_pairs;
    double2* dst = all + (size_t)r * slab_pairs + offset_pairs;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < block_pairs; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}

__global__ void status_word_kernel(int* __restrict__ flag, int host_status, double* __restrict__ word) {
    int st = host_status;
    if (st == 0 && flag != nullptr) st = flag[1] != 0 ? TBK_ERR_NOT_FINITE : (flag[0] != 0 ? TBK_ERR_NO_CONVERGENCE : 0);
    if (flag != nullptr) flag[0] = flag[1] = 0;
    *word = (double)st;
}

}  // namespace

int64_t tbk_gather_block_rows(int64_t per, int n_orb) {
    // ~4 MiB per rank and block (8192 rows at 64 orbitals): small enough that the exposed last gather is a few hundred
    // microseconds at 8 ranks, large enough that a gather is bandwidth, not latency; an even row count (16-byte copies)
    int64_t rows = std::max<int64_t>(64, (int64_t(4) << 20) / (8 * (int64_t)std::max(n_orb, 1)));
    // TBK_GATHER_BLOCK_ROWS: tests / measurements only (every rank of a run must see the same value)
    if (const char* env = getenv("TBK_GATHER_BLOCK_ROWS"))
        if (atoll(env) > 0) rows = atoll(env);
    rows = std::min(rows, std::max<int64_t>(per, 1));
    return rows;
}

// Agreement in front of a sharded call: every rank contributes its status (0 = ready), everybody receives all of them
// on the HOST -- one 8-byte all-gather through buffers that exist since tbk_comm_create, so nothing of it can fail on
// one rank alone.  Synchronous (the caller decides on the verdict before it enters the data collectives).
extern "C" int tbk_comm_agree(tbk_comm* c, int status, double* verdict) {
    TBK_ARG(c != nullptr && verdict != nullptr, "comm / verdict is NULL");
    TBK_HIP(hipSetDevice(c->device));
    hipLaunchKernelGGL(status_word_kernel, dim3(1), dim3(1), 0, c->stream, (int*)nullptr, status, c->d_status);
    TBK_HIP(hipGetLastError());
    TBK_NCCL(ncclAllGather(c->d_status, c->d_status + 1, 1, ncclDouble, c->comm, c->stream));
    TBK_HIP(hipMemcpyAsync(verdict, c->d_status + 1, (size_t)c->world * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TBK_HIP(hipStreamSynchronize(c->stream));
    return TBK_OK;
}

// The landing area of the pipelined gather ([world][block rows][n_orb] doubles) for slabs of `per` rows of `n_orb` values:
// grow-only, sized from (per, n_orb, world) alone.  Callers run this INSIDE the step whose failures are exchanged by
// tbk_comm_agree, so that an allocation failure on one rank reaches every rank's verdict before anybody enters the data
// collectives; tbk_eigenval_device_gather itself only allocates when this was skipped.
extern "C" int tbk_comm_prepare_gather(tbk_comm* c, int n_orb, int64_t per) {
    TBK_ARG(c != nullptr, "comm is NULL");
    TBK_ARG(n_orb >= 1 && per >= 0, "need n_orb >= 1, per >= 0");
    TBK_HIP(hipSetDevice(c->device));
    const int64_t B = tbk_gather_block_rows(per, n_orb);
    const size_t want = (size_t)c->world * (size_t)B * (size_t)n_orb * sizeof(double);
    if (want > c->stage_bytes) {
        // (a gather of an earlier call may still read the old area: it is released behind the communicator's stream)
        TBK_HIP(hipStreamSynchronize(c->stream));
        if (c->d_stage) TBK_HIP(hipFree(c->d_stage));
        c->d_stage = nullptr;
        c->stage_bytes = 0;
        TBK_HIP(hipMalloc((void**)&c->d_stage, want));
        c->stage_bytes = want;
    }
    return TBK_OK;
}

extern "C" int tbk_eigenval_device_gather(tbk_comm* c, tbk_model* m, const double* d_k, const double* h_k, int64_t nk, int64_t per,
                                          int host_status, double* d_all, double* d_status_all) {
    // Without these nothing can be enqueued at all (the peers' agreement step has vouched for them):
    TBK_ARG(c != nullptr && m != nullptr, "comm / model is NULL");
    TBK_ARG(per >= 0, "per < 0");
    TBK_ARG(d_all != nullptr || per == 0, "result is NULL");
    TBK_ARG(d_status_all != nullptr, "status array is NULL");
    TBK_ARG(m->device == c->device, "model and communicator live on different devices");
    TBK_LOCK(m);  // the hook below belongs to this call alone
    TBK_HIP(hipSetDevice(c->device));
    // From here on a failure of THIS rank must not end the call: the peers are entering n_blocks all-gathers and the status
    // gather, and a rank that returned early would leave them waiting.  It becomes this rank's status word instead -- the
    // call walks the same sequence of collectives (its rows are then meaningless) and every rank raises alike.
    int local = TBK_OK;
    const auto soft = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && local == TBK_OK) {
            tbk_set_error("%s failed: %s (tbk_eigenval_device_gather)", what, hipGetErrorString(e));
            local = (e == hipErrorOutOfMemory) ? TBK_ERR_MEMORY : TBK_ERR_DEVICE;
        }
    };
    if (nk < 0 || nk > per) {
        tbk_set_error("invalid argument: need 0 <= nk <= per");
        local = TBK_ERR_ARGUMENT;
        nk = 0;
    }
    if (host_status != 0) local = host_status;
    const auto soft_nccl = [&](ncclResult_t r, const char* what) {
        if (r != ncclSuccess && local == TBK_OK) {
            tbk_set_error("%s failed: %s (tbk_eigenval_device_gather)", what, ncclGetErrorString(r));
            local = TBK_ERR_DEVICE;
        }
    };
    const int64_t n = m->n_orb;
    const int64_t slab = per * n;                       // doubles per rank
    const int64_t B = tbk_gather_block_rows(per, (int)n);
    const int64_t n_blocks = per > 0 ? (per + B - 1) / B : 0;
    double* mine = d_all + (size_t)c->rank * slab;
    // the landing area FIRST: whether this rank computes at all is decided behind the last thing that can fail in front of
    // the solve (a rank whose fallback allocation failed used to run the whole pipeline into rows its own
    // all-gathers were landing on)
    double* landing = c->d_stage;
    if ((size_t)c->world * (size_t)B * n * sizeof(double) > c->stage_bytes) {
        // (tbk_comm_prepare_gather was skipped or asked for another shape)
        const int rc_prepare = tbk_comm_prepare_gather(c, (int)n, per);
        landing = c->d_stage;
        if (rc_prepare != TBK_OK) {
            if (local == TBK_OK) local = rc_prepare;
            // no landing area: the head of the result takes the pieces ([world][per][n] holds [world][B][n]) IN PLACE -- this rank
            // sends from its own piece of that area, the one aliasing NCCL defines; what it sends is meaningless and nothing
            // of its copy is placed: the status word says so to everybody
            landing = d_all;
        }
    }
    const bool compute = local == TBK_OK && nk > 0;
    // rows this rank does not compute are zero (short or empty slab, or a rank that arrives with a failure)
    const int64_t first_idle = compute ? nk : 0;
    if (per > first_idle && landing != d_all)
        soft(hipMemsetAsync(mine + (size_t)first_idle * n, 0, (size_t)(per - first_idle) * n * sizeof(double), m->stream), "hipMemsetAsync");
    int64_t next_block = 0;
    const bool ranged = m->timing;
    auto send_blocks = [&](int64_t rows_done) -> int {  // every block that ends at or before rows_done
        while (next_block < n_blocks && std::min(per, (next_block + 1) * B) <= rows_done) {
            const int64_t b0 = next_block * B, rows = std::min(per, b0 + B) - b0;
            const int64_t count = rows * n;
            const double* piece = landing == d_all ? d_all + (size_t)c->rank * count : mine + (size_t)b0 * n;
            if (ranged) tbk_range_push("tbk:allgather_eigenvalues(block)");
            const ncclResult_t r = ncclAllGather(piece, landing, (size_t)count, ncclDouble, c->comm, c->stream);
            if (ranged) tbk_range_pop();
            soft_nccl(r, "ncclAllGather");  // (never a return from here: the peers are in the same sequence of collectives)
            if (landing == d_all) {
                // (this rank has no landing area and reports a failure: its rows are not placed)
            } else if (((count | slab | (b0 * n)) & 1) == 0) {  // 16-byte copies when every piece starts on an even double
                const int64_t pairs = count / 2;
                const unsigned gx = (unsigned)std::min<int64_t>((pairs + 255) / 256, 512);
                hipLaunchKernelGGL(gather_place_kernel, dim3(gx, (unsigned)c->world), dim3(256), 0, c->stream,
                                   reinterpret_cast<const double2*>(landing), reinterpret_cast<double2*>(d_all), pairs, slab / 2,
                                   b0 * n / 2);
                soft(hipGetLastError(), "gather_place_kernel");
            } else {
                for (int r2 = 0; r2 < c->world; ++r2)
                    soft(hipMemcpyAsync(d_all + (size_t)r2 * slab + (size_t)b0 * n, landing + (size_t)r2 * count,
                                        (size_t)count * sizeof(double), hipMemcpyDeviceToDevice, c->stream), "hipMemcpyAsync");
            }
            ++next_block;
        }
        return TBK_OK;
    };
    int rc = TBK_OK;
    if (compute) {
        m->chunk_done = [&](int64_t c0, int64_t nkc, hipEvent_t done) -> int {
            // rows below c0 + nkc of this rank's slab are final once `done` has fired (chunks complete in order)
            TBK_HIP(hipStreamWaitEvent(c->stream, done, 0));
            return send_blocks(c0 + nkc);
        };
        rc = tbk_eigenval_device_hint(m, d_k, h_k, nk, mine);
        m->chunk_done = nullptr;
    }
    // whatever is left -- the zero rows of a short slab, paths without chunk events (rocSOLVER), a call that failed on
    // the way: the peers are waiting in the same sequence of collectives -- goes behind the main stream
    soft(hipEventRecord(c->tail, m->stream), "hipEventRecord");
    soft(hipStreamWaitEvent(c->stream, c->tail, 0), "hipStreamWaitEvent");
    (void)send_blocks(per);
    const int status = local != TBK_OK ? local : rc;
    // (soft, all of it: this rank must enter the status all-gather whatever happened to it -- the peers are waiting there)
    hipLaunchKernelGGL(status_word_kernel, dim3(1), dim3(1), 0, c->stream, m->ws_flag.as<int>(), status, c->d_status);
    soft(hipGetLastError(), "status_word_kernel");
    soft_nccl(ncclAllGather(c->d_status, c->d_status + 1, 1, ncclDouble, c->comm, c->stream), "ncclAllGather (status)");
    soft(hipMemcpyAsync(d_status_all, c->d_status + 1, (size_t)c->world * sizeof(double), hipMemcpyDeviceToDevice, c->stream), "hipMemcpyAsync");
    // the next call on the model's streams must not overwrite rows a gather is still reading
    soft(hipEventRecord(c->done[0], c->stream), "hipEventRecord");
    soft(hipStreamWaitEvent(m->stream, c->done[0], 0), "hipStreamWaitEvent");
    if (local != TBK_OK && local != host_status) return local;
    return rc;
}
