// tbk_api.hip -- the C ABI of include/tbk.h: staging, the chunked k pipeline, host/device entry
// points.  Kernels live in tbk_phase.hip / tbk_stage.hip / tbk_hk_dense.hip / tbk_hk_csr.hip /
// tbk_eig.hip; this file only owns memory, streams and ordering.

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <functional>
#include <new>
#include <vector>

#include "tbk_internal.h"

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";

void tbk_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* tbk_last_error(void) { return g_err; }
#ifdef TBK_EXPERIMENTS
extern "C" const char* tbk_version(void) { return "tbk 0.1 (gfx950) +experiments"; }  // (`make EXPERIMENTS=1`: tbk_exp_env reads the environment)
#else
extern "C" const char* tbk_version(void) { return "tbk 0.1 (gfx950)"; }
#endif

extern "C" int tbk_device_count(int* count) {
    TBK_ARG(count != nullptr, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return TBK_OK;
}

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
int DevBuf::reserve(size_t want) {
    if (want <= bytes) return TBK_OK;
    release();
    const size_t rounded = (want + (size_t(1) << 20) - 1) & ~((size_t(1) << 20) - 1);
    TBK_HIP(hipMalloc(&ptr, rounded));
    bytes = rounded;
    return TBK_OK;
}

void DevBuf::release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
}

// roctx ranges (rocprofv3 --marker-trace labels the timeline with them): the tools library is looked up at run time,
// so libtbk.so has no link-time dependency on it; ranges are only pushed while TBK_OPT_TIMING is on.
namespace {
struct RoctxApi {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
};
const RoctxApi& roctx_api() {
    static const RoctxApi api = [] {
        RoctxApi a;
        for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so"}) {
            void* h = dlopen(name, RTLD_LAZY | RTLD_GLOBAL);
            if (!h) continue;
            a.push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
            a.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
            if (a.push && a.pop) break;
            a.push = nullptr;
            a.pop = nullptr;
        }
        return a;
    }();
    return api;
}
const char* const kStageNames[TBK_T_COUNT] = {"tbk:phase_rows", "tbk:hk_contraction", "tbk:tridiag_reduction",
                                              "tbk:tridiag_eigenvalues"};
}  // namespace

void tbk_range_push(const char* name) {
    const RoctxApi& api = roctx_api();
    if (api.push) (void)api.push(name);
}

void tbk_range_pop() {
    const RoctxApi& api = roctx_api();
    if (api.pop) (void)api.pop();
}

StageTimer::StageTimer(tbk_model* m_, int stage, hipStream_t s)
    : m(m_), on(m_->timing), stream(s ? s : m_->stream) {
    ev.stage = stage;
    if (on) {
        if (hipEventCreate(&ev.start) != hipSuccess || hipEventCreate(&ev.stop) != hipSuccess) {
            on = false;
            return;
        }
        tbk_range_push(kStageNames[stage]);
        (void)hipEventRecord(ev.start, stream);
    }
}

StageTimer::~StageTimer() {
    if (on) {
        (void)hipEventRecord(ev.stop, stream);
        tbk_range_pop();
        m->events.push_back(ev);
    }
}

static inline int64_t round_up(int64_t x, int64_t q) { return (x + q - 1) / q * q; }

static int require_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        (void)hipGetLastError();
        tbk_set_error("no HIP device visible: libtbk has no CPU path");
        return TBK_ERR_DEVICE;
    }
    if (device < 0 || device >= n) {
        tbk_set_error("device %d out of range (have %d)", device, n);
        return TBK_ERR_ARGUMENT;
    }
    TBK_HIP(hipSetDevice(device));
    return TBK_OK;
}

// everything both model kinds share: stream, rocBLAS handle, lattice vectors, packed-element map
static int create_common(int device, int dim, int n_orb, int64_t n_r, const int32_t* R,
                         int64_t k_rows_per_r, tbk_model** out) {
    TBK_ARG(out != nullptr, "out is NULL");
    *out = nullptr;
    TBK_ARG(dim >= 1 && dim <= TBK_MAX_DIM, "dim must be in [1, 8]");
    TBK_ARG(n_orb >= 1 && n_orb <= 32768, "n_orb must be in [1, 32768]");
    TBK_ARG(n_r >= 0 && n_r < (int64_t(1) << 28), "n_r out of range");
    TBK_ARG(n_r == 0 || R != nullptr || k_rows_per_r == 1, "R is NULL");  // k.p rows carry no R
    TBK_CHECK(require_device(device));

    tbk_model* m = new (std::nothrow) tbk_model();
    if (!m) {
        tbk_set_error("out of host memory");
        return TBK_ERR_MEMORY;
    }
    m->device = device;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) m->n_cu = cus;
    }
    m->dim = dim;
    m->n_orb = n_orb;
    m->n_r = n_r;
    // K rows are padded to whole LDS stages (TBK_BK); padding rows carry zero hoppings
    m->k2 = round_up(n_r * k_rows_per_r, TBK_BK);
    m->n_r_pad = m->k2 / k_rows_per_r;
    m->ncol = (int)((int64_t)n_orb * (n_orb + 1) / 2);
    m->ncol_pad = (int)round_up(m->ncol, TBK_BNP);

    int rc = TBK_OK;
    auto fail = [&](int code) {
        tbk_model_destroy(m);
        return code;
    };
#define TBK_TRY(expr)                           \
    do {                                        \
        rc = [&]() -> int {                     \
            expr;                               \
            return TBK_OK;                      \
        }();                                    \
        if (rc != TBK_OK) return fail(rc);      \
    } while (0)

    {
        // TBK_MAIN_PRIORITY=1 (measurements): the main stream -- the H(k) kernels -- at the highest priority
        static const bool main_hi = tbk_exp_env("TBK_MAIN_PRIORITY") != nullptr && atoi(tbk_exp_env("TBK_MAIN_PRIORITY")) != 0;
        int lo = 0, hi = 0;
        if (main_hi && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess)
            TBK_TRY(TBK_HIP(hipStreamCreateWithPriority(&m->stream, hipStreamNonBlocking, hi)));
        else
            TBK_TRY(TBK_HIP(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking)));
    }
    TBK_TRY(TBK_HIP(hipStreamCreateWithFlags(&m->stream_eig, hipStreamNonBlocking)));
    TBK_TRY(TBK_HIP(hipStreamCreateWithFlags(&m->stream_ql, hipStreamNonBlocking)));
    // (stream_xl / ev_xl: created by launch_band_xl when a batch above 1024 orbitals first goes in groups)
    for (int b = 0; b < 2; ++b) {
        TBK_TRY(TBK_HIP(hipEventCreateWithFlags(&m->ev_hk[b], hipEventDisableTiming)));
        TBK_TRY(TBK_HIP(hipEventCreateWithFlags(&m->ev_tri[b], hipEventDisableTiming)));
        TBK_TRY(TBK_HIP(hipEventCreateWithFlags(&m->ev_ql[b], hipEventDisableTiming)));
        TBK_TRY(TBK_HIP(hipEventCreateWithFlags(&m->ev_out[b], hipEventDisableTiming)));
        // (release to system scope: results of small calls are read by the CPU from non-coherent pinned memory right
        // behind hipEventSynchronize on this event -- with a default event that visibility is the runtime's choice)
        if (b == 0) TBK_TRY(TBK_HIP(hipEventCreateWithFlags(&m->ev_sync, hipEventDisableTiming | hipEventReleaseToSystem)));
    }
    TBK_TRY(TBK_ROCBLAS(rocblas_create_handle(&m->blas)));
    TBK_TRY(TBK_ROCBLAS(rocblas_set_stream(m->blas, m->stream)));
    TBK_TRY(TBK_CHECK(m->ws_flag.reserve(2 * sizeof(int))));
    TBK_TRY(TBK_HIP(hipMemsetAsync(m->ws_flag.ptr, 0, 2 * sizeof(int), m->stream)));
    // one H(k) (up to 1024 orbitals: 16 MiB) with its k-point and positions, or a few hundred eigenvalue rows
    m->h_stage_bytes = std::max<size_t>(size_t(320) << 10,
                                        n_orb <= 1024 ? (size_t)n_orb * n_orb * 16 + (size_t)n_orb * dim * 8 + (size_t(64) << 10) : 0);
    static const int stage_mode = tbk_exp_env("TBK_STAGE_MODE") ? atoi(tbk_exp_env("TBK_STAGE_MODE")) : 1;  // 0 off, 1 non-coherent, 2 coherent
    if (stage_mode == 0 || hipHostMalloc(&m->h_stage, m->h_stage_bytes, stage_mode == 1 ? hipHostMallocNonCoherent : hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        m->h_stage = nullptr;  // no pinned memory: every call takes the pageable path
        m->h_stage_bytes = 0;
    }

    // packed upper-triangle map, row-major over (i <= j): consecutive e -> consecutive j
    {
        std::vector<int32_t> colmap((size_t)m->ncol_pad, -1);
        size_t e = 0;
        for (int i = 0; i < n_orb; ++i)
            for (int j = i; j < n_orb; ++j) colmap[e++] = (int32_t)((i << 16) | j);
        TBK_TRY(TBK_HIP(hipMalloc((void**)&m->d_colmap, colmap.size() * sizeof(int32_t))));
        TBK_TRY(TBK_HIP(hipMemcpy(m->d_colmap, colmap.data(), colmap.size() * sizeof(int32_t),
                                  hipMemcpyHostToDevice)));
        m->staged_bytes += (int64_t)(colmap.size() * sizeof(int32_t));
    }
    if (n_r > 0 && R != nullptr && k_rows_per_r == 2) m->h_R.assign(R, R + (size_t)n_r * dim);
    if (m->n_r_pad > 0 && R != nullptr) {
        std::vector<int32_t> r_pad((size_t)m->n_r_pad * dim, 0);
        std::memcpy(r_pad.data(), R, (size_t)n_r * dim * sizeof(int32_t));
        TBK_TRY(TBK_HIP(hipMalloc((void**)&m->d_R, r_pad.size() * sizeof(int32_t))));
        TBK_TRY(TBK_HIP(hipMemcpy(m->d_R, r_pad.data(), r_pad.size() * sizeof(int32_t),
                                  hipMemcpyHostToDevice)));
        m->staged_bytes += (int64_t)(r_pad.size() * sizeof(int32_t));
    }
#undef TBK_TRY
    *out = m;
    return TBK_OK;
}

// ------------------------------------------------------------------------------------------------
// model creation / destruction
// ------------------------------------------------------------------------------------------------
extern "C" int tbk_model_create_dense(int device, int dim, int n_orb, int64_t n_r, const int32_t* R,
                                      const double* hop, tbk_model** out) {
    TBK_ARG(out != nullptr, "out is NULL");
    *out = nullptr;
    TBK_ARG(n_r == 0 || hop != nullptr, "hop is NULL");
    tbk_model* m = nullptr;
    TBK_CHECK(create_common(device, dim, n_orb, n_r, R, 2, &m));
    m->sparse = false;
    int rc = TBK_OK;
    double* d_raw = nullptr;
    const size_t raw_bytes = (size_t)n_r * n_orb * n_orb * 2 * sizeof(double);
    rc = [&]() -> int {
        if (raw_bytes) {
            TBK_HIP(hipMalloc((void**)&d_raw, raw_bytes));
            TBK_HIP(hipMemcpyAsync(d_raw, hop, raw_bytes, hipMemcpyHostToDevice, m->stream));
        }
        TBK_CHECK(tbk_stage_dense(m, d_raw));
        TBK_HIP(hipStreamSynchronize(m->stream));
        return TBK_OK;
    }();
    if (d_raw) (void)hipFree(d_raw);
    if (rc != TBK_OK) {
        tbk_model_destroy(m);
        return rc;
    }
    *out = m;
    return TBK_OK;
}

extern "C" int tbk_model_create_csr(int device, int dim, int n_orb, int64_t n_r, const int32_t* R,
                                    const int64_t* r_ptr, const int32_t* row, const int32_t* col,
                                    const double* val, tbk_model** out) {
    TBK_ARG(out != nullptr, "out is NULL");
    *out = nullptr;
    TBK_ARG(n_r >= 0, "n_r out of range");
    TBK_ARG(n_r == 0 || r_ptr != nullptr, "r_ptr is NULL");
    const int64_t nnz = n_r > 0 ? r_ptr[n_r] : 0;
    TBK_ARG(n_r == 0 || r_ptr[0] == 0, "r_ptr[0] must be 0");
    TBK_ARG(nnz >= 0, "negative nnz");
    TBK_ARG(nnz == 0 || (row && col && val), "row/col/val is NULL");
    for (int64_t r = 0; r < n_r; ++r) TBK_ARG(r_ptr[r] <= r_ptr[r + 1], "r_ptr not monotone");
    for (int64_t t = 0; t < nnz; ++t)
        TBK_ARG(row[t] >= 0 && row[t] < n_orb && col[t] >= 0 && col[t] < n_orb,
                "row/col index out of range");

    tbk_model* m = nullptr;
    TBK_CHECK(create_common(device, dim, n_orb, n_r, R, 2, &m));
    m->sparse = true;

    // transpose "per lattice vector, which elements" into "per packed element, which lattice
    // vectors" with a counting sort; record order inside an element follows r (deterministic sums)
    auto packed_index = [n_orb](int i, int j) -> int64_t {  // i <= j
        return (int64_t)i * n_orb - (int64_t)i * (i - 1) / 2 + (j - i);
    };
    std::vector<int64_t> cptr((size_t)m->ncol + 1, 0);
    for (int64_t t = 0; t < nnz; ++t) {
        const int i = std::min(row[t], col[t]), j = std::max(row[t], col[t]);
        cptr[(size_t)packed_index(i, j) + 1]++;
    }
    for (int e = 0; e < m->ncol; ++e) cptr[(size_t)e + 1] += cptr[(size_t)e];
    std::vector<int32_t> rec_r((size_t)nnz);
    std::vector<double> rec_v((size_t)nnz * 2);
    {
        std::vector<int64_t> cursor(cptr.begin(), cptr.end() - 1);
        for (int64_t r = 0; r < n_r; ++r)
            for (int64_t t = r_ptr[r]; t < r_ptr[r + 1]; ++t) {
                const int i = row[t], j = col[t];
                const int kind = (i == j) ? 2 : (i < j ? 0 : 1);
                const int64_t e = packed_index(std::min(i, j), std::max(i, j));
                const int64_t slot = cursor[(size_t)e]++;
                rec_r[(size_t)slot] = (int32_t)((kind << 28) | (int32_t)r);
                rec_v[(size_t)slot * 2] = val[2 * t];
                rec_v[(size_t)slot * 2 + 1] = val[2 * t + 1];
            }
    }
    m->nnz_rec = nnz;
    int rc = [&]() -> int {
        TBK_HIP(hipMalloc((void**)&m->d_cptr, cptr.size() * sizeof(int64_t)));
        TBK_HIP(hipMemcpy(m->d_cptr, cptr.data(), cptr.size() * sizeof(int64_t), hipMemcpyHostToDevice));
        m->staged_bytes += (int64_t)(cptr.size() * sizeof(int64_t));
        if (nnz > 0) {
            TBK_HIP(hipMalloc((void**)&m->d_rec_r, (size_t)nnz * sizeof(int32_t)));
            TBK_HIP(hipMalloc((void**)&m->d_rec_v, (size_t)nnz * 2 * sizeof(double)));
            TBK_HIP(hipMemcpy(m->d_rec_r, rec_r.data(), (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice));
            TBK_HIP(hipMemcpy(m->d_rec_v, rec_v.data(), (size_t)nnz * 2 * sizeof(double), hipMemcpyHostToDevice));
            m->staged_bytes += nnz * (int64_t)(sizeof(int32_t) + 2 * sizeof(double));
            const int kt = tbk_csr_tile_kpoints(n_r);
            if (kt > 0) {
                std::vector<int64_t> sptr;
                std::vector<int32_t> srec_r;
                std::vector<double> srec_v;
                tbk_csr_schedule(m->ncol, kt, cptr, rec_r, rec_v, sptr, srec_r, srec_v);
                m->sched_steps = sptr.back();
                auto upload = [&](void** dst, const void* src, size_t bytes) -> int {
                    TBK_HIP(hipMalloc(dst, std::max<size_t>(bytes, 8)));
                    if (bytes) TBK_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
                    m->staged_bytes += (int64_t)bytes;
                    return TBK_OK;
                };
                TBK_CHECK(upload((void**)&m->d_sptr, sptr.data(), sptr.size() * sizeof(int64_t)));
                TBK_CHECK(upload((void**)&m->d_srec_r, srec_r.data(), srec_r.size() * sizeof(int32_t)));
                TBK_CHECK(upload((void**)&m->d_srec_v, srec_v.data(), srec_v.size() * sizeof(double)));
                m->sched_kt = kt;
            }
        }
        return TBK_OK;
    }();
    if (rc != TBK_OK) {
        tbk_model_destroy(m);
        return rc;
    }
    *out = m;
    return TBK_OK;
}

extern "C" void tbk_model_destroy(tbk_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    hipStream_t streams[] = {m->stream, m->stream_eig, m->stream_ql, m->stream_xl[0], m->stream_xl[1], m->stream_xl[2]};
    for (hipStream_t st : streams)
        if (st) (void)hipStreamSynchronize(st);
    for (hipEvent_t e : m->ev_xl)
        if (e) (void)hipEventDestroy(e);
    for (int b = 0; b < 2; ++b) {
        if (m->ev_hk[b]) (void)hipEventDestroy(m->ev_hk[b]);
        if (m->ev_tri[b]) (void)hipEventDestroy(m->ev_tri[b]);
        if (m->ev_ql[b]) (void)hipEventDestroy(m->ev_ql[b]);
        if (m->ev_out[b]) (void)hipEventDestroy(m->ev_out[b]);
        if (b == 0 && m->ev_sync) (void)hipEventDestroy(m->ev_sync);
    }
    for (auto& ev : m->events) {
        (void)hipEventDestroy(ev.start);
        (void)hipEventDestroy(ev.stop);
    }
    if (m->blas) (void)rocblas_destroy_handle(m->blas);
    if (m->h_stage) (void)hipHostFree(m->h_stage);
    for (hipStream_t st : streams)
        if (st) (void)hipStreamDestroy(st);
    void* ptrs[] = {m->d_R, m->d_colmap, m->d_B, m->d_cptr, m->d_rec_r, m->d_rec_v, m->d_powers, m->d_sptr, m->d_srec_r, m->d_srec_v};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    DevBuf* bufs[] = {&m->ws_phase, &m->ws_H, &m->ws_E,   &m->ws_E2,
                      &m->ws_info,  &m->ws_k, &m->ws_pos, &m->ws_out, &m->ws_out2, &m->ws_flag, &m->ws_orb, &m->ws_part, &m->ws_kfold, &m->ws_kline, &m->ws_band, &m->ws_bandmat[0], &m->ws_bandmat[1], &m->ws_H2, &m->ws_split, &m->ws_xl, &m->ws_posraw};
    for (DevBuf* b : bufs) b->release();
    tbk_fold_release(m);
    delete m;
}

extern "C" int tbk_model_set_option(tbk_model* m, int option, int64_t value) {
    TBK_ARG(m != nullptr, "model is NULL");
    TBK_LOCK(m);
    switch (option) {
        case TBK_OPT_EIGENSOLVER:
            TBK_ARG(value >= TBK_EIG_AUTO && value <= TBK_EIG_ROCSOLVER, "unknown eigensolver");
            m->eigensolver = (int)value;
            return TBK_OK;
        case TBK_OPT_K_CHUNK:
            TBK_ARG(value >= 0, "k chunk must be >= 0");
            m->k_chunk = value;
            return TBK_OK;
        case TBK_OPT_TIMING:
            m->timing = value != 0;
            return TBK_OK;
        case TBK_OPT_FOLD:
            m->fold_enabled = value != 0;
            return TBK_OK;
        default:
            tbk_set_error("unknown option %d", option);
            return TBK_ERR_ARGUMENT;
    }
}

extern "C" int tbk_model_info(const tbk_model* m, int* device, int* dim, int* n_orb, int64_t* n_r,
                              int* is_sparse, int64_t* staged_bytes) {
    TBK_ARG(m != nullptr, "model is NULL");
    if (device) *device = m->device;
    if (dim) *dim = m->dim;
    if (n_orb) *n_orb = m->n_orb;
    if (n_r) *n_r = m->n_r;
    if (is_sparse) *is_sparse = m->sparse ? 1 : 0;
    if (staged_bytes) *staged_bytes = m->staged_bytes;
    return TBK_OK;
}

// ------------------------------------------------------------------------------------------------
// the chunked pipeline
// ------------------------------------------------------------------------------------------------
static int64_t choose_chunk(tbk_model* m, int64_t nk, bool with_eig) {
    // (a call of up to one k tile is one chunk whatever the memory: no hipMemGetInfo -- a driver query -- on the one-k path)
    if (nk <= TBK_BM) return TBK_BM;
    const int64_t n = m->n_orb;
    int64_t per_k = m->k2 * 8 + (with_eig ? n * n * 16 + (int64_t)tbk_eig_scratch_per_k(m) : 0);
    if (per_k < 64) per_k = 64;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = size_t(8) << 30;
    // (what this handle's grow-only chunk workspaces hold already is as good as free: counted out, the second call of a model
    // chose a smaller chunk than the first -- cfg3: one chunk in the warm-up, two from then on)
    free_b += m->ws_H.bytes + m->ws_H2.bytes + m->ws_phase.bytes + m->ws_band.bytes + m->ws_bandmat[0].bytes + m->ws_bandmat[1].bytes +
              m->ws_E.bytes + m->ws_xl.bytes;
    // above 64 orbitals a chunk is a few thousand matrices: every kernel of the eigensolver ends on a partly filled
    // round of workgroups, and 2 - 3 times longer chunks were worth 3 - 4 % (cfg3 3846 -> 12500 matrices per chunk,
    // cfg5 1250 -> 5000)
    // Round 6: 65 - 256 orbitals take chunks as long as 64 GiB allow (up to 131072 k-points).  At these sizes the reduction is
    // ~90 % of a chunk and its neighbours in the pipeline (the next chunk's H(k), the previous chunk's bisection) run on the
    // same FP64 pipes: overlapping them buys nothing, every chunk boundary costs the partly filled last rounds of its kernels
    // -- whole eigenval, us per k-point, chunks of 16384 / 32768 / 65536 / 131072: 0.4095 / 0.4060 / 0.4041 / 0.4024 at 96
    // orbitals, 2.291 / 2.264 / 2.247 / 2.228 at 160, 5.668 / 5.586 / 5.531 / 5.518 at 256; cfg3 170.8 -> 176.4 k k-points/s in ONE
    // chunk.  Above 256 orbitals the second stage is a launch of its own that does fill gaps: cfg5 16.63 k in chunks of 4096,
    // 16.51 k in one.
    const bool mid = n > 64 && n <= 256;
    int64_t budget = std::min<int64_t>((int64_t)(free_b / 4), int64_t(mid ? 64 : n > 64 ? 24 : 6) << 30);
    int64_t chunk = budget / per_k / TBK_BM * TBK_BM;
    // 32768 k-points per chunk at 64 orbitals and above; small matrices take proportionally more (up to 1 M at 8
    // orbitals): a chunk is ~6 launches and one QL latency chain whatever its size, and 20 M k-points of an
    // 8-orbital model spent 42 of 92 GPU-ms in 612 of those chains
    int64_t cap = mid ? 131072 : 32768;
    if (n < 64) cap *= std::min<int64_t>(32, (64 / n) * (64 / n));
    chunk = std::max<int64_t>(TBK_BM, std::min<int64_t>(chunk, cap));
    if (m->k_chunk > 0) chunk = round_up(m->k_chunk, TBK_BM);
    else if (chunk >= 4096) chunk = chunk / 4096 * 4096;  // 32 k tiles: equal shares for the 8 XCDs
    return std::min(chunk, round_up(nk, TBK_BM));
}

// leading dimension (in k-points) of the phase-row matrix A[K][ld]: whole k tiles
static inline int64_t phase_ld(int64_t nk) { return round_up(nk, TBK_BM); }

static int fill_rows(tbk_model* m, const double* d_k, int64_t nk, int64_t nk_pad, double* d_A) {
    if (m->kdotp)
        return tbk_launch_monomials(m->stream, m->d_powers, m->dim, m->n_r, m->k2, d_k, nk, nk_pad, d_A);
    return tbk_launch_phase(m, d_k, nk, nk_pad, d_A);
}

static int build_h(tbk_model* m, const double* d_A, int64_t nk, int64_t nk_pad, int mode,
                   int convention, const double* d_k, const double* d_pos, double* d_H) {
    if (m->sparse)
        return tbk_launch_hk_csr(m, d_A, nk, nk_pad, mode, convention, d_k, d_pos, d_H);
    return tbk_launch_hk_dense(m, d_A, nk, nk_pad, mode, convention, d_k, d_pos, d_H);
}

extern "C" int tbk_hamilton_device(tbk_model* m, const double* d_k, int64_t nk, int convention,
                                   const double* d_pos, double* d_H) {
    TBK_ARG(m != nullptr, "model is NULL");
    TBK_LOCK(m);
    TBK_ARG(convention == 1 || convention == 2, "convention must be 1 or 2");
    TBK_ARG(nk >= 0, "nk < 0");
    if (nk == 0) return TBK_OK;
    TBK_ARG(d_k && d_H, "k / H is NULL");
    TBK_ARG(convention == 2 || d_pos != nullptr, "convention 1 needs pos");
    TBK_HIP(hipSetDevice(m->device));
    const int64_t chunk = choose_chunk(m, nk, false);
    const size_t nn2 = (size_t)m->n_orb * m->n_orb * 2;
    for (int64_t c0 = 0; c0 < nk; c0 += chunk) {
        const int64_t nkc = std::min(chunk, nk - c0);
        const int64_t nk_pad = phase_ld(nkc);
        TBK_CHECK(m->ws_phase.reserve((size_t)std::max<int64_t>(m->k2, 1) * nk_pad * sizeof(double)));
        double* d_A = m->ws_phase.as<double>();
        const double* kc = d_k + c0 * m->dim;
        const bool own_rows = tbk_hk_inline_phases(m, nkc);  // a few k-points: the H(k) kernel makes its phase rows
        if (!own_rows) TBK_CHECK(fill_rows(m, kc, nkc, nk_pad, d_A));
        const double* d_orb = nullptr;
        // (one-k host call on the matrix-vector path: the H(k) kernel forms the phases of its one k-point itself from the
        // raw positions -- no orbital_phase_kernel launch)
        const bool inline_orb = m->h_k_inline != nullptr && m->d_pos_inline != nullptr && nk == 1 && own_rows;
        if (convention == 1 && !inline_orb) {
            TBK_CHECK(m->ws_orb.reserve((size_t)nkc * m->n_orb * 2 * sizeof(double)));
            TBK_CHECK(tbk_launch_orbital_phases(m, kc, d_pos, nkc, m->ws_orb.as<double>()));
            d_orb = m->ws_orb.as<double>();
        }
        TBK_CHECK(build_h(m, own_rows ? nullptr : d_A, nkc, nk_pad, HK_FULL, convention, kc, d_orb, d_H + (size_t)c0 * nn2));
    }
    return TBK_OK;
}

// Eigenvalues with the hand-written solvers (register-resident reduction for n_orb <= 64, blocked streaming
// reduction up to 512), software-pipelined over k chunks on
// three streams:   main: phase(c) -> H(c)      eig: tridiag(c)      ql: QL(c - 1)
//
//     | H(c) | tridiag(c) || QL(c-1) | H(c+1) | tridiag(c+1) || QL(c) | ...
//
// The MFMA contraction runs alone: it fills the LDS (2 x 72 KiB per CU), so anything co-scheduled with
// it only displaces its workgroups.  The two eigensolver kernels are complementary -- the reduction is
// VALU-bound with 14 KiB of LDS per workgroup, the QL is a latency-bound serial chain with 64 KiB per
// workgroup and almost no issue pressure -- so QL(c-1) runs in the shadow of tridiag(c).
// tridiagonal stage on the QL stream: lane-per-matrix QL up to 64 orbitals, bisection above
// Calls of at most max(4096, 768 n) k-points take the bisection kernel for every chunk: the lane-per-matrix QL
// is a serial chain of ~n^2 rotations (1.6 ms at n = 64 however few matrices there are) that only pays when tens of
// thousands of matrices share it and it can hide under the next chunk's reduction; bisection spends a wave per
// matrix (VALU work ~ n per matrix) and got ~1.7x faster with the secant steps of tbk_eig_stream.hip.  Measured
// crossover (tools/bench_crossover.py, ms per call, QL vs bisection): n = 64, N_R = 4096: 49152 k-points 57.29 vs
// 56.84, 57344: 66.71 vs 66.82, 100000: 114.4 vs 115.5; n = 48, N_R = 512: 30000: 5.94 vs 5.79, 40000: 7.55 vs 7.81;
// n = 32, N_R = 256: 16384: 1.45 vs 1.42, 24576: 1.84 vs 1.87.  (The rule was 640 n in round 1 and 384 n between the
// free-running QL and the faster bisection.)  TBK_SMALL_CALL_PER_ORBITAL overrides the factor (measurements only).
constexpr int64_t TBK_SMALL_CALL = 4096;
// (Up to 12 orbitals the QL chain used to be the shorter one -- 61 us at n = 8 -- until small matrices got the idle
// lanes of their wave for multisection: 1000 silicon k-points 59 -> 20 us, so small calls bisect at every size now.)

static int launch_tridiag_eigenvalues(tbk_model* m, hipStream_t s, double* d_de, int64_t nk, double* d_E,
                                      bool beside_ql = false, bool small_call = false, const void* d_band = nullptr) {
    // two-stage reduction: the second stage (band -> tridiagonal) of this chunk runs here, in front of its bisection --
    // on the tridiagonal stream, i.e. next to the first stage of the following chunk
    if (d_band) TBK_CHECK(tbk_launch_band_chase(m, s, d_band, nk, d_de));
    if (tbk_eig_small_supported(m->n_orb) && !small_call) return tbk_launch_ql(m, s, d_de, nk, d_E, beside_ql);
    return tbk_launch_bisect(m, s, d_de, nk, d_E);
}

// k chunks of the pipeline.  The lane-per-matrix QL is a latency chain (~3 ms however few matrices it
// gets), and the QL of the last chunk has nothing to hide behind: the schedule therefore ends on a short
// chunk (one XCD round of k tiles plus the ragged remainder) whose reduction is brief, and that chunk's QL
// runs next to the QL of the chunk before it.  All other chunks are multiples of 4096 k-points.
static std::vector<int64_t> chunk_schedule(tbk_model* m, int64_t nk, int64_t chunk) {
    std::vector<int64_t> out;
    const int64_t unit = 4096;
    if (!tbk_eig_small_supported(m->n_orb) || m->k_chunk > 0 || chunk < 2 * unit || nk < 3 * unit) {
        for (int64_t c0 = 0; c0 < nk; c0 += chunk) out.push_back(std::min(chunk, nk - c0));
        return out;
    }
    const int64_t last = unit + nk % unit;  // in [4096, 8192)
    int64_t rest = nk - last;               // a multiple of 4096
    const int64_t n_big = (rest + chunk - 1) / chunk;
    for (int64_t i = 0; i < n_big; ++i) {
        // as even as whole units allow, larger chunks first
        const int64_t share = round_up((rest + (n_big - i) - 1) / (n_big - i), unit);
        const int64_t take = std::min(std::min(share, chunk), rest);
        out.push_back(take);
        rest -= take;
    }
    out.push_back(last);
    return out;
}

// Fills H for k-points [c0, c0 + nkc) of the call (phase rows + contraction on the main stream).
using HBuilder = std::function<int(int64_t c0, int64_t nkc, double* d_H)>;

// Chunks of a folded call: whole runs (mesh planes) packed up to the chunk size -- every run a chunk cuts costs a second
// fold pass, ragged mesh lines and a handful of small launches on both sides of the cut (the 100^3 mesh in chunks of 30 000
// = three planes instead of 32 768: 150.8 -> 143.6 ms).  Runs longer than a chunk are cut into chunk-sized pieces.
static std::vector<int64_t> run_schedule(const std::vector<int64_t>& runs, int64_t chunk) {
    std::vector<int64_t> out;
    int64_t cur = 0;
    for (size_t r = 0; r + 1 < runs.size(); ++r) {
        int64_t len = runs[r + 1] - runs[r];
        if (cur > 0 && cur + len > chunk) {
            out.push_back(cur);
            cur = 0;
        }
        while (len > chunk) {
            out.push_back(chunk);
            len -= chunk;
        }
        cur += len;
    }
    if (cur > 0) out.push_back(cur);
    return out;
}

static int eigenval_wave_pipeline(tbk_model* m, const double* d_k, int64_t nk, double* d_E, const HBuilder* builder = nullptr,
                                  const std::vector<int64_t>* runs = nullptr) {
    // The direct builder in two halves: the phase rows of a chunk only need the previous contraction to be done with the
    // row buffer (stream order), not the eigensolver to be done with H -- so they are enqueued BEFORE the main stream
    // waits for the previous chunk's reduction and run under it (an HBM-write kernel beside a VALU-bound one: 1.5 ms
    // per 100 k k-points at the headline shape).
    int64_t rows_ready_for = -1;  // c0 of the chunk whose phase rows are in ws_phase
    const auto prepare_rows = [&](int64_t c0, int64_t nkc) -> int {
        if (tbk_hk_inline_phases(m, nkc)) return TBK_OK;
        const int64_t nk_pad = phase_ld(nkc);
        TBK_CHECK(m->ws_phase.reserve((size_t)std::max<int64_t>(m->k2, 1) * nk_pad * sizeof(double)));
        TBK_CHECK(fill_rows(m, d_k + c0 * m->dim, nkc, nk_pad, m->ws_phase.as<double>()));
        rows_ready_for = c0;
        return TBK_OK;
    };
    const HBuilder direct = [&](int64_t c0, int64_t nkc, double* d_H) -> int {
        const int64_t nk_pad = phase_ld(nkc);
        const double* kc = d_k + c0 * m->dim;
        if (tbk_hk_inline_phases(m, nkc)) return build_h(m, nullptr, nkc, nk_pad, HK_TRI, 2, kc, nullptr, d_H);
        if (rows_ready_for != c0) TBK_CHECK(prepare_rows(c0, nkc));
        return build_h(m, m->ws_phase.as<double>(), nkc, nk_pad, HK_TRI, 2, kc, nullptr, d_H);
    };
    const HBuilder& build = builder ? *builder : direct;
    const int64_t chunk = choose_chunk(m, nk, true);
    const size_t n = (size_t)m->n_orb;
    const size_t nn2 = n * n * 2;
    DevBuf* debuf[2] = {&m->ws_E, &m->ws_E2};
    const std::vector<int64_t> sched =
        (runs != nullptr && m->k_chunk == 0 && nk > chunk) ? run_schedule(*runs, chunk) : chunk_schedule(m, nk, chunk);
    const int64_t n_chunks = (int64_t)sched.size();
    const int64_t max_chunk = *std::max_element(sched.begin(), sched.end());
    // a property of the call, not of its chunking: TBK_OPT_K_CHUNK must not change the results
    static const int64_t per_orbital = tbk_exp_env("TBK_SMALL_CALL_PER_ORBITAL") ? atoll(tbk_exp_env("TBK_SMALL_CALL_PER_ORBITAL")) : 768;
    const bool small_call = nk <= std::max<int64_t>(TBK_SMALL_CALL, per_orbital * (int64_t)m->n_orb);
    TBK_CHECK(m->ws_H.reserve((size_t)max_chunk * nn2 * sizeof(double)));
    for (int b = 0; b < (n_chunks > 1 ? 2 : 1); ++b)
        TBK_CHECK(debuf[b]->reserve((size_t)max_chunk * n * 2 * sizeof(double)));
    double* d_H = m->ws_H.as<double>();
    // Folded H(k) (a mesh: ~30 small launches per chunk, 1.05 of the 5.5 ms a 32768-point chunk of cfg4 takes) is
    // built BESIDE the reduction of the previous chunk, into a second H buffer; chunk c then waits for the reduction
    // of chunk c - 2.  (Not for the direct contraction, which fills the chip and shares the FP64 pipe: see above.)
    // Round 4, MEASURED AND LEFT OFF (DESIGN_LOG.md R4.15): the same for the direct H(k) of the two-stage sizes (from 185
    // orbitals) -- that reduction is a chain of short phases which leaves the matrix pipe idle four fifths of the time, and
    // the sparse H(k) is an HBM-write kernel.  Default: one after the other (the round-3 order); TBK_H_OVERLAP_BIG=1 turns
    // the overlap (and the 84 KiB hk_lds_floor below) on for measurements.
    static const bool overlap_on = tbk_exp_env("TBK_H_OVERLAP") == nullptr || atoi(tbk_exp_env("TBK_H_OVERLAP")) != 0;
    static const bool overlap_big = tbk_exp_env("TBK_H_OVERLAP_BIG") != nullptr && atoi(tbk_exp_env("TBK_H_OVERLAP_BIG")) != 0;
    const bool h_overlap = n_chunks > 2 && overlap_on &&
                           ((builder != nullptr && tbk_eig_small_supported(m->n_orb)) ||
                            (overlap_big && builder == nullptr && !tbk_eig_small_supported(m->n_orb) && tbk_eig_two_stage(m)));
    double* d_Hbuf[2] = {d_H, d_H};
    struct LdsFloor {  // (reset on every way out)
        tbk_model* m;
        ~LdsFloor() { m->hk_lds_floor = 0; }
    } lds_floor_guard{m};
    if (h_overlap && builder == nullptr) {
        static const int floor_kib = tbk_exp_env("TBK_H_OVERLAP_LDS") ? atoi(tbk_exp_env("TBK_H_OVERLAP_LDS")) : 84;
        m->hk_lds_floor = (size_t)floor_kib * 1024;
    }
    if (h_overlap) {
        TBK_CHECK(m->ws_H2.reserve((size_t)max_chunk * nn2 * sizeof(double)));
        d_Hbuf[1] = m->ws_H2.as<double>();
    }
    // two-stage reduction in two launches (TBK_BAND_FUSE=0): stage two of a chunk goes to the tridiagonal stream; fused
    // (default) it is part of the reduction kernel and this flag stays off
    const bool two_stage = !tbk_eig_small_supported(m->n_orb) && tbk_eig_two_stage(m) && n_chunks > 1 && !tbk_band_fused(m->n_orb);
    if (two_stage) {
        TBK_CHECK(m->ws_band.reserve((size_t)max_chunk * tbk_band_scratch_per_matrix(m->n_orb)));
        for (int b = 0; b < 2; ++b) TBK_CHECK(m->ws_bandmat[b].reserve((size_t)max_chunk * tbk_band_bytes_per_matrix(m->n_orb)));
        TBK_CHECK(tbk_band_xl_reserve(m, max_chunk));
    }
    if (n_chunks == 1) {
        // one chunk has nothing to overlap: everything in order on the main stream, no cross-stream events (they
        // cost more than the kernels of a single-k call)
        double* d_de = debuf[0]->as<double>();
        TBK_CHECK(build(0, nk, d_H));
        if (tbk_eig_small_supported(m->n_orb))
            TBK_CHECK(tbk_launch_tridiag(m, m->stream, d_H, nk, d_de));
        else
            TBK_CHECK(tbk_launch_tridiag_stream(m, m->stream, d_H, nk, d_de));
        TBK_CHECK(launch_tridiag_eigenvalues(m, m->stream, d_de, nk, d_E, false, small_call));
        if (m->chunk_done) {
            TBK_HIP(hipEventRecord(m->ev_ql[0], m->stream));
            TBK_CHECK(m->chunk_done(0, nk, m->ev_ql[0]));
        }
        return TBK_OK;
    }
    int64_t prev_c0 = 0, prev_nkc = 0, c0 = 0;
    for (int64_t c = 0; c < n_chunks; ++c) {
        const int b = (int)(c & 1);
        const int64_t nkc = sched[c];
        double* d_de = debuf[b]->as<double>();
        d_H = d_Hbuf[b];
        if (c >= 1) {
            if (!builder) TBK_CHECK(prepare_rows(c0, nkc));  // under the previous chunk's reduction
            // H(c) overwrites an H buffer; the direct contraction must not share the chip with the eigensolver either
            if (!h_overlap)
                TBK_HIP(hipStreamWaitEvent(m->stream, m->ev_tri[b ^ 1], 0));
            else if (c >= 2)
                TBK_HIP(hipStreamWaitEvent(m->stream, m->ev_tri[b], 0));
            // (d, e) of chunk c - 2 must have been consumed before the reduction of chunk c overwrites them; beside
            // the previous reduction, H(c) itself need not wait for that
            if (c >= 2 && !h_overlap) TBK_HIP(hipStreamWaitEvent(m->stream, m->ev_ql[b], 0));
        }
        TBK_CHECK(build(c0, nkc, d_H));
        TBK_HIP(hipEventRecord(m->ev_hk[b], m->stream));

        TBK_HIP(hipStreamWaitEvent(m->stream_eig, m->ev_hk[b], 0));
        if (c >= 2 && h_overlap) TBK_HIP(hipStreamWaitEvent(m->stream_eig, m->ev_ql[b], 0));
        if (tbk_eig_small_supported(m->n_orb))
            TBK_CHECK(tbk_launch_tridiag(m, m->stream_eig, d_H, nkc, d_de));
        else if (two_stage)
            TBK_CHECK(tbk_launch_band_reduce(m, m->stream_eig, d_H, nkc, m->ws_band.ptr, m->ws_bandmat[b].ptr));
        else
            TBK_CHECK(tbk_launch_tridiag_stream(m, m->stream_eig, d_H, nkc, d_de));
        TBK_HIP(hipEventRecord(m->ev_tri[b], m->stream_eig));

        if (c >= 1) {  // tridiagonal stage of the previous chunk, alongside this chunk's reduction
            // TBK_CHASE_EARLY=1 (measurements): ... alongside this chunk's H(k) already -- it starts as soon as its own
            // reduction is done
            static const bool early = tbk_exp_env("TBK_CHASE_EARLY") && atoi(tbk_exp_env("TBK_CHASE_EARLY")) != 0;
            if (!(early && two_stage)) TBK_HIP(hipStreamWaitEvent(m->stream_ql, m->ev_hk[b], 0));
            // (d, e) of the previous chunk: implied by ev_hk unless H(c) was built beside that reduction
            if (h_overlap || (early && two_stage)) TBK_HIP(hipStreamWaitEvent(m->stream_ql, m->ev_tri[b ^ 1], 0));
            TBK_CHECK(launch_tridiag_eigenvalues(m, m->stream_ql, debuf[b ^ 1]->as<double>(), prev_nkc,
                                                 d_E + (size_t)prev_c0 * n, false, small_call,
                                                 two_stage ? m->ws_bandmat[b ^ 1].ptr : nullptr));
            TBK_HIP(hipEventRecord(m->ev_ql[b ^ 1], m->stream_ql));
            if (m->chunk_done) TBK_CHECK(m->chunk_done(prev_c0, prev_nkc, m->ev_ql[b ^ 1]));
        }
        prev_c0 = c0;
        prev_nkc = nkc;
        c0 += nkc;
    }
    {  // Eigenvalues of the last chunk, behind its own reduction on the eig stream, i.e. next to QL(last - 1).
       // Nothing is left to hide a 2.7 ms QL chain behind: the (short) last chunk takes the bisection kernel,
       // a wave per matrix for ~0.1 ms (measured: 2.6 ms off the 100k step).
        const int b = (int)((n_chunks - 1) & 1);
        TBK_CHECK(launch_tridiag_eigenvalues(m, m->stream_eig, debuf[b]->as<double>(), prev_nkc,
                                             d_E + (size_t)prev_c0 * n, n_chunks > 1,
                                             small_call || n_chunks > 1, two_stage ? m->ws_bandmat[b].ptr : nullptr));
        TBK_HIP(hipEventRecord(m->ev_ql[b], m->stream_eig));
        if (m->chunk_done) TBK_CHECK(m->chunk_done(prev_c0, prev_nkc, m->ev_ql[b]));
    }
    // later work on the main stream (gather, D2H, the next call) sees the finished eigenvalues
    for (int b = 0; b < (n_chunks > 1 ? 2 : 1); ++b) TBK_HIP(hipStreamWaitEvent(m->stream, m->ev_ql[b], 0));
    TBK_HIP(hipStreamWaitEvent(m->stream, m->ev_tri[(n_chunks - 1) & 1], 0));
    return TBK_OK;
}

// k lists with long runs of one shared component (grids in meshgrid order, stacks of planes): every run is
// evaluated on the model folded along that component (tbk_fold.hip).  Returns TBK_OK with *done = false when the
// list does not qualify.
static int eigenval_folded(tbk_model* m, const double* d_k, const double* h_k, int64_t nk, double* d_E, bool* done) {
    *done = false;
    if (!m->fold_enabled || m->sparse || m->kdotp || m->dim < 2 || m->n_r < 64 || nk < 1024) return TBK_OK;
    // Device-resident lists are never read back (include/tbk.h: the device entry points enqueue and return): the run
    // structure comes from the caller's host copy of the list (tbk_eigenval / tbk_eigenval_device_hint) or not at all.
    if (h_k == nullptr) return TBK_OK;
    std::vector<int64_t> runs;
    const int f = tbk_fold_choose(m, h_k, nk, runs);
    if (f < 0) return TBK_OK;
    const int dim = m->dim;
    TBK_CHECK(m->ws_kfold.reserve((size_t)nk * (dim - 1) * sizeof(double)));
    double* d_k2 = m->ws_kfold.as<double>();
    TBK_CHECK(tbk_fold_drop_component(m, d_k, dim, f, nk, d_k2));
    const size_t nn2 = (size_t)m->n_orb * m->n_orb * 2;
    // the chunk pipeline (schedule, overlap of the tridiagonal stage) runs over the whole list; only the H(k) of a
    // chunk is assembled run by run, each piece on the model folded for its run.
    // Runs are folded a group at a time (one pass over Bt for up to 16 of them); `group_lo` is the first run of the
    // group whose operands are in the plan's buffer.
    const int64_t n_runs = (int64_t)runs.size() - 1;
    const int group = tbk_fold_group_size();
    int64_t group_lo = -1;
    tbk_fold_plan_t& plan1 = m->fold[f];
    std::vector<int> reduced;  // original component of every reduced one
    for (int d = 0; d < dim; ++d)
        if (d != f) reduced.push_back(d);

    // [lo, hi) of one run, `m` folded for that run: phase rows + contraction of the (dim - 1)-dimensional model
    auto piece_plane = [&](int64_t lo, int64_t hi, double* d_Hp) -> int {
        const int64_t len = hi - lo, nk_pad = phase_ld(len);
        TBK_CHECK(m->ws_phase.reserve((size_t)std::max<int64_t>(m->k2, 1) * nk_pad * sizeof(double)));
        TBK_CHECK(fill_rows(m, d_k2 + lo * (dim - 1), len, nk_pad, m->ws_phase.as<double>()));
        return build_h(m, m->ws_phase.as<double>(), len, nk_pad, HK_TRI, 2, d_k2 + lo * (dim - 1), nullptr, d_Hp);
    };

    // Second level (meshes): inside a plane the k-points come in LINES -- equal-length sub-runs of one more shared
    // component whose remaining coordinates repeat from line to line.  Every line is a (dim - 2)-dimensional model
    // (13 instead of 313 lattice vectors at the headline shape); all lines of the piece go through ONE launch with
    // per-line operands and shared phase rows (tbk_launch_hk_dense_lines).  Ragged ends of the piece, and anything
    // that does not have this structure, take piece_plane.
    struct LineInfo {
        bool ok = false;     // the piece has a body of whole mesh lines
        int e2 = -1;         // reduced component shared along a line
        int64_t L = 0;       // points per line
        int64_t body = 0;    // first point of the body
        int64_t n_lines = 0;
    };
    const int dim1 = dim - 1;
    const int line_cap = 512;  // lines per batch (operands: cap x k2'' x row)
    auto same_line = [&](int64_t a0, int64_t b0, int64_t L, int e2) {  // equal remaining coordinates along two lines
        for (int64_t t = 0; t < L; ++t)
            for (int e = 0; e < dim1; ++e)
                if (e != e2 && h_k[(a0 + t) * dim + reduced[e]] != h_k[(b0 + t) * dim + reduced[e]]) return false;
        return true;
    };
    auto analyse = [&](int64_t lo, int64_t hi) -> LineInfo {
        LineInfo li;
        if (dim1 < 2 || hi - lo < 512) return li;
        // the reduced component with the longest sub-runs
        int e2 = -1;
        int64_t best_changes = hi - lo;
        for (int e = 0; e < dim1; ++e) {
            int64_t changes = 0;
            for (int64_t i = lo + 1; i < hi; ++i) changes += h_k[i * dim + reduced[e]] != h_k[(i - 1) * dim + reduced[e]];
            if (changes < best_changes) {
                best_changes = changes;
                e2 = e;
            }
        }
        if (e2 < 0 || best_changes < 4) return li;
        const int c2 = reduced[e2];
        std::vector<int64_t> sb(1, lo);  // sub-run starts
        for (int64_t i = lo + 1; i < hi; ++i)
            if (h_k[i * dim + c2] != h_k[(i - 1) * dim + c2]) sb.push_back(i);
        sb.push_back(hi);
        const size_t n_sub = sb.size() - 1;
        if (n_sub < 6) return li;
        const int64_t L = sb[2] - sb[1];  // an interior line
        if (L < 8 || L > TBK_BM) return li;
        // body: the longest prefix of interior sub-runs (from the second one) that are lines like the first of them
        size_t first = (sb[1] - sb[0] == L && same_line(sb[0], sb[1], L, e2)) ? 0 : 1, last = first;
        while (last < n_sub && sb[last + 1] - sb[last] == L && same_line(sb[first], sb[last], L, e2)) ++last;
        li.n_lines = (int64_t)(last - first);
        if (li.n_lines < 4) return li;
        li.ok = true;
        li.e2 = e2;
        li.L = L;
        li.body = sb[first];
        return li;
    };
    // lines a0, a0 + L, ... (n of them) of the CURRENT first-level model -> slots slot0 ... of the second-level plan
    auto fold_body = [&](tbk_fold_plan_t& plan2, const LineInfo& li, int64_t a0, int64_t n, int slot0) -> int {
        return tbk_fold_lines(m, plan2, d_k2 + a0 * dim1 + li.e2, li.L * dim1, (int)n, slot0);
    };
    // one launch for n lines whose operands are in slots 0 .. n - 1 (shared phase rows: the lines have equal coordinates)
    auto contract_lines = [&](tbk_fold_plan_t& plan2, const LineInfo& li, int64_t a0, int64_t n, double* d_Hp) -> int {
        const int64_t row_len = (int64_t)m->ncol_pad * 2;
        tbk_fold_saved_t saved2;
        TBK_CHECK(tbk_fold_enter(m, plan2, 0, saved2));
        int rc = m->ws_kline.reserve((size_t)li.L * std::max(dim1 - 1, 1) * sizeof(double));
        if (rc == TBK_OK) rc = tbk_fold_drop_component(m, d_k2 + a0 * dim1, dim1, li.e2, li.L, m->ws_kline.as<double>());
        if (rc == TBK_OK) rc = m->ws_phase.reserve((size_t)std::max<int64_t>(m->k2, 1) * TBK_BM * sizeof(double));
        if (rc == TBK_OK) rc = fill_rows(m, m->ws_kline.as<double>(), li.L, TBK_BM, m->ws_phase.as<double>());
        if (rc == TBK_OK)
            rc = tbk_launch_hk_dense_lines(m, m->ws_phase.as<double>(), n, (int)li.L, plan2.k2 * row_len, d_Hp);
        tbk_fold_leave(m, saved2);
        return rc;
    };

    // Second level (meshes): inside a plane the k-points come in LINES -- equal-length sub-runs of one more shared
    // component whose remaining coordinates repeat from line to line.  Every line is a (dim - 2)-dimensional model
    // (13 instead of 313 lattice vectors at the headline shape); all lines of the piece go through ONE launch with
    // per-line operands and shared phase rows (tbk_launch_hk_dense_lines).  Ragged ends of the piece, and anything
    // that does not have this structure, take piece_plane.
    auto piece = [&](int64_t lo, int64_t hi, double* d_Hp) -> int {
        const LineInfo li = analyse(lo, hi);
        if (!li.ok) return piece_plane(lo, hi, d_Hp);
        tbk_fold_plan_t* plan2 = nullptr;
        TBK_CHECK(tbk_fold_subplan(m, plan1, li.e2, line_cap, &plan2));
        if (!plan2 || plan2->n_rho * 3 > plan1.n_rho) return piece_plane(lo, hi, d_Hp);

        if (li.body > lo) TBK_CHECK(piece_plane(lo, li.body, d_Hp));  // ragged head
        for (int64_t l0 = 0; l0 < li.n_lines; l0 += line_cap) {
            const int64_t nl = std::min<int64_t>(line_cap, li.n_lines - l0);
            const int64_t a0 = li.body + l0 * li.L;
            // (the lines' shared-component values are read on the device: first point of every line)
            TBK_CHECK(fold_body(*plan2, li, a0, nl, 0));
            TBK_CHECK(contract_lines(*plan2, li, a0, nl, d_Hp + (size_t)(a0 - lo) * nn2));
        }
        const int64_t body_end = li.body + li.n_lines * li.L;
        if (body_end < hi) TBK_CHECK(piece_plane(body_end, hi, d_Hp + (size_t)(body_end - lo) * nn2));  // ragged tail
        return TBK_OK;
    };

    // first-level model of run r in place (its group folded if it is not in the plan's buffer)
    auto enter_run = [&](size_t r, tbk_fold_saved_t& saved) -> int {
        if (group_lo < 0 || (int64_t)r < group_lo || (int64_t)r >= group_lo + group) {
            group_lo = (int64_t)r;
            const int n_g = (int)std::min<int64_t>(group, n_runs - group_lo);
            double kf[64];
            for (int g = 0; g < n_g; ++g) kf[g] = h_k[runs[(size_t)(group_lo + g)] * dim + f];
            TBK_CHECK(tbk_fold_group(m, plan1, kf, n_g, 0));
        }
        return tbk_fold_enter(m, plan1, (int)((int64_t)r - group_lo), saved);
    };
    // A chunk of WHOLE runs that are nothing but equal mesh lines (the planes of a mesh: run_schedule cuts chunks at
    // run boundaries): the lines of all its planes are folded plane by plane into consecutive slots -- light launches
    // that get through beside the previous chunk's reduction -- and contracted by ONE launch, instead of one contraction
    // (which cannot start before the reduction has left the chip) and four light launches behind it per plane.
    auto batched = [&](int64_t c0, int64_t nkc, double* d_H, bool* done) -> int {
        *done = false;
        const size_t r0 = (size_t)(std::upper_bound(runs.begin(), runs.end(), c0) - runs.begin()) - 1;
        if (runs[r0] != c0) return TBK_OK;
        size_t r1 = r0;
        while (r1 + 1 < runs.size() && runs[r1 + 1] <= c0 + nkc) ++r1;
        if (r1 - r0 < 2 || runs[r1] != c0 + nkc) return TBK_OK;  // fewer than two whole runs, or a cut run
        const LineInfo li0 = analyse(runs[r0], runs[r0 + 1]);
        if (!li0.ok || li0.body != runs[r0] || li0.body + li0.n_lines * li0.L != runs[r0 + 1]) return TBK_OK;
        int64_t total = li0.n_lines;
        for (size_t r = r0 + 1; r < r1; ++r) {
            const LineInfo li = analyse(runs[r], runs[r + 1]);
            if (!li.ok || li.e2 != li0.e2 || li.L != li0.L || li.body != runs[r] ||
                li.body + li.n_lines * li.L != runs[r + 1] || !same_line(runs[r0], runs[r], li0.L, li0.e2))
                return TBK_OK;
            total += li.n_lines;
        }
        if (total > line_cap) return TBK_OK;
        tbk_fold_plan_t* plan2 = nullptr;
        TBK_CHECK(tbk_fold_subplan(m, plan1, li0.e2, line_cap, &plan2));
        if (!plan2 || plan2->n_rho * 3 > plan1.n_rho) return TBK_OK;
        int slot = 0;
        for (size_t r = r0; r < r1; ++r) {
            tbk_fold_saved_t saved;
            TBK_CHECK(enter_run(r, saved));
            const int64_t n = (runs[r + 1] - runs[r]) / li0.L;
            const int rc = fold_body(*plan2, li0, runs[r], n, slot);
            tbk_fold_leave(m, saved);
            TBK_CHECK(rc);
            slot += (int)n;
        }
        {
            // the second-level plan sits on top of a first-level model: any run of the chunk will do for the shapes (the
            // last one is in the buffer whatever groups the chunk straddles)
            tbk_fold_saved_t saved;
            TBK_CHECK(enter_run(r1 - 1, saved));
            const int rc = contract_lines(*plan2, li0, c0, total, d_H);
            tbk_fold_leave(m, saved);
            TBK_CHECK(rc);
        }
        *done = true;
        return TBK_OK;
    };

    const HBuilder folded = [&](int64_t c0, int64_t nkc, double* d_H) -> int {
        {
            bool done = false;
            TBK_CHECK(batched(c0, nkc, d_H, &done));
            if (done) return TBK_OK;
        }
        size_t r = (size_t)(std::upper_bound(runs.begin(), runs.end(), c0) - runs.begin()) - 1;
        for (int64_t lo = c0; lo < c0 + nkc; ++r) {
            const int64_t hi = std::min(runs[r + 1], c0 + nkc);
            tbk_fold_saved_t saved;
            TBK_CHECK(enter_run(r, saved));
            const int rc = piece(lo, hi, d_H + (size_t)(lo - c0) * nn2);
            tbk_fold_leave(m, saved);
            TBK_CHECK(rc);
            lo = hi;
        }
        return TBK_OK;
    };
    TBK_CHECK(eigenval_wave_pipeline(m, d_k, nk, d_E, &folded, &runs));
    m->counters[TBK_CNT_FOLDED_CALLS] += 1;
    m->counters[TBK_CNT_FOLDED_KPOINTS] += nk;
    *done = true;
    return TBK_OK;
}

// the call takes the library's own reduction kernels (the chunk pipeline), not rocSOLVER
static bool eigenval_own_solvers(const tbk_model* m) {
    return m->eigensolver != TBK_EIG_ROCSOLVER &&
           (tbk_eig_small_supported(m->n_orb) || (m->eigensolver == TBK_EIG_AUTO && tbk_eig_stream_supported(m->n_orb)));
}

static int eigenval_device_solve(tbk_model* m, const double* d_k, const double* h_k, int64_t nk, double* d_E) {
    TBK_ARG(m != nullptr, "model is NULL");
    TBK_LOCK(m);
    TBK_ARG(nk >= 0, "nk < 0");
    if (nk == 0) return TBK_OK;
    TBK_ARG(d_k && d_E, "k / E is NULL");
    TBK_HIP(hipSetDevice(m->device));
    m->call_nk = nk;
    m->counters[TBK_CNT_EIGENVAL_CALLS] += 1;
    if (m->eigensolver == TBK_EIG_WAVE && !tbk_eig_small_supported(m->n_orb)) {
        tbk_set_error("TBK_EIG_WAVE handles n_orb <= 64 only (n_orb = %d)", m->n_orb);
        return TBK_ERR_ARGUMENT;
    }
    if (eigenval_own_solvers(m)) {
        bool done = false;
        TBK_CHECK(eigenval_folded(m, d_k, h_k, nk, d_E, &done));
        if (done) return TBK_OK;
        return eigenval_wave_pipeline(m, d_k, nk, d_E);
    }

    int64_t chunk = choose_chunk(m, nk, true);
    const size_t nn2 = (size_t)m->n_orb * m->n_orb * 2;
    // rocsolver_zheevd_strided_batched faulted (memory access fault inside the library) on 2048 matrices of 768 / 1024
    // orbitals -- 1.2e9 / 2.1e9 complex elements in one call -- and ran 2048 x 640 (8.4e8): calls are kept below 2^29
    // elements
    chunk = std::max<int64_t>(1, std::min<int64_t>(chunk, (int64_t(1) << 29) / std::max<int64_t>(1, (int64_t)m->n_orb * m->n_orb)));
    // rocSOLVER path: TBK_EIG_ROCSOLVER, or n_orb above the own solvers' range
    m->counters[TBK_CNT_LIBRARY_CALLS] += 1;
    for (int64_t c0 = 0; c0 < nk; c0 += chunk) {
        const int64_t nkc = std::min(chunk, nk - c0);
        const int64_t nk_pad = phase_ld(nkc);
        TBK_CHECK(m->ws_phase.reserve((size_t)std::max<int64_t>(m->k2, 1) * nk_pad * sizeof(double)));
        TBK_CHECK(m->ws_H.reserve((size_t)nkc * nn2 * sizeof(double)));
        double* d_A = m->ws_phase.as<double>();
        double* d_H = m->ws_H.as<double>();
        const double* kc = d_k + c0 * m->dim;
        TBK_CHECK(fill_rows(m, kc, nkc, nk_pad, d_A));
        TBK_CHECK(build_h(m, d_A, nkc, nk_pad, HK_TRI, 2, kc, nullptr, d_H));
        TBK_CHECK(tbk_eig_batched(m, d_H, nkc, d_E + (size_t)c0 * m->n_orb));
    }
    return TBK_OK;
}

// NaN / Inf anywhere in the hoppings or in k reaches the eigenvalues (the solvers write NaN for a non-finite matrix):
// scipy's eigvalsh(check_finite=True) raises there (_tb_model.py:1147-1150).  One pass over the finished eigenvalues
// on the device raises the flag that tbk_eigenval_check turns into TBK_ERR_NOT_FINITE -- the host-side
// np.isfinite(out).all() it replaces cost 80 ms for 20 M k-points of an 8-orbital model, as much as all the kernels.
__global__ void __launch_bounds__(256) flag_nonfinite_kernel(const double* __restrict__ E, int64_t total, int* __restrict__ flag) {
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) bad |= !isfinite(E[i]);
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicAdd(flag, 1);
}

static int eigenval_device_impl(tbk_model* m, const double* d_k, const double* h_k, int64_t nk, double* d_E) {
    TBK_CHECK(eigenval_device_solve(m, d_k, h_k, nk, d_E));
    // the wave solvers raise the flag themselves (QL and bisection see every non-finite (d, e) and answer NaN);
    // rocSOLVER's eigenvalues get the pass over the output
    const bool own_solvers = m != nullptr && eigenval_own_solvers(m);
    if (nk > 0 && m != nullptr && !own_solvers) {
        const int64_t total = nk * m->n_orb;
        const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 8 * 1024);
        hipLaunchKernelGGL(flag_nonfinite_kernel, dim3(blocks), dim3(256), 0, m->stream, d_E, total, m->ws_flag.as<int>() + 1);
        TBK_HIP(hipGetLastError());
    }
    return TBK_OK;
}

extern "C" int tbk_eigenval_device(tbk_model* m, const double* d_k, int64_t nk, double* d_E) {
    return eigenval_device_impl(m, d_k, nullptr, nk, d_E);
}

extern "C" int tbk_eigenval_device_hint(tbk_model* m, const double* d_k, const double* h_k, int64_t nk, double* d_E) {
    return eigenval_device_impl(m, d_k, h_k, nk, d_E);
}

extern "C" int tbk_model_counter(tbk_model* m, int counter, int64_t* value) {
    TBK_ARG(m != nullptr && value != nullptr, "model / value is NULL");
    TBK_ARG(counter >= 0 && counter < TBK_CNT_COUNT, "unknown counter");
    TBK_LOCK(m);
    *value = m->counters[counter];
    return TBK_OK;
}

// Wait for everything enqueued on the main stream -- through an event.  hipStreamSynchronize on this runtime goes to sleep
// for ~250 us in calls whose GPU work takes a few tens of microseconds (one-k hamilton / eigenval: 290 instead of 50 - 90 us
// per call, tools/trace_single_k.py); hipEventSynchronize on an event recorded at the same point does not.
static int wait_main_stream(tbk_model* m) {
    TBK_HIP(hipEventRecord(m->ev_sync, m->stream));
    TBK_HIP(hipEventSynchronize(m->ev_sync));
    return TBK_OK;
}

extern "C" int tbk_synchronize(tbk_model* m) {
    TBK_ARG(m != nullptr, "model is NULL");
    TBK_LOCK(m);
    TBK_HIP(hipSetDevice(m->device));
    return wait_main_stream(m);
}

extern "C" int tbk_eigenval_check(tbk_model* m) {
    TBK_ARG(m != nullptr, "model is NULL");
    TBK_LOCK(m);
    TBK_HIP(hipSetDevice(m->device));
    int flag[2] = {0, 0};
    TBK_HIP(hipMemcpyAsync(flag, m->ws_flag.ptr, sizeof(flag), hipMemcpyDeviceToHost, m->stream));
    TBK_HIP(hipMemsetAsync(m->ws_flag.ptr, 0, sizeof(flag), m->stream));
    TBK_CHECK(wait_main_stream(m));
    if (flag[1] != 0) {
        tbk_set_error("array must not contain infs or NaNs");  // scipy's message for the same condition
        return TBK_ERR_NOT_FINITE;
    }
    if (flag[0] != 0) {
        tbk_set_error("eigensolver did not converge for %d matrices", flag[0]);
        return TBK_ERR_NO_CONVERGENCE;
    }
    return TBK_OK;
}

// ---- host-buffer entry points -------------------------------------------------------------------
// H leaves the device in chunks through two buffers, chunk c + 1 being computed while chunk c crosses PCIe: 20 000
// k-points at N_orb = 64, N_R = 4096 in 25 ms instead of 46 (52 GB/s) when the caller's array has been written before.
// A FRESH result array (np.empty: no pages behind it yet) is bound by the kernel's page-fault rate instead, ~21 GB/s
// on these hosts whoever takes the faults: populating the pages from helper threads (MADV_HUGEPAGE +
// MADV_POPULATE_WRITE, four threads, ahead of the copy or racing it) did not beat the copy thread faulting by itself
// (66 vs 57 ms for 1.3 GB), so there is no such helper here.
extern "C" int tbk_hamilton(tbk_model* m, const double* k, int64_t nk, int convention,
                            const double* pos, double* H_out) {
    TBK_ARG(m != nullptr, "model is NULL");
    TBK_LOCK(m);
    TBK_ARG(convention == 1 || convention == 2, "convention must be 1 or 2");
    TBK_ARG(nk >= 0, "nk < 0");
    if (nk == 0) return TBK_OK;
    TBK_ARG(k && H_out, "k / H is NULL");
    TBK_ARG(convention == 2 || pos != nullptr, "convention 1 needs pos");
    TBK_HIP(hipSetDevice(m->device));
    const size_t nn2 = (size_t)m->n_orb * m->n_orb * 2;
    {
        const size_t k_bytes = (size_t)nk * m->dim * sizeof(double), h_bytes = (size_t)nk * nn2 * sizeof(double);
        const size_t p_bytes = convention == 1 ? (size_t)m->n_orb * m->dim * sizeof(double) : 0;
        const size_t h_off = (k_bytes + p_bytes + 63) / 64 * 64;  // (16-byte stores of H: keep the block aligned)
        if (m->h_stage != nullptr && h_off + h_bytes <= m->h_stage_bytes) {
            // small result (one k-point: 64 KiB of H at 64 orbitals): [k | pos | H] through the pinned buffer.  The chunked
            // download below -- a blocking copy into pageable memory -- takes 290 us per call for results of 25 - 64 KiB
            // in a loop of one-k calls (tools/trace_single_k.py), this path 50 - 95 us whatever the size
            char* st = static_cast<char*>(m->h_stage);
            TBK_CHECK(m->ws_k.reserve(k_bytes));
            TBK_CHECK(m->ws_out.reserve(h_bytes));
            // ONE k-point of a dense model (the Z2Pack call shape): k goes into the kernel arguments and the positions of
            // convention 1 stay on the device from call to call -- two uploads and one launch less per call
            const bool inline_k = nk == 1 && !m->sparse && !m->kdotp && tbk_hk_inline_phases(m, 1);
            const double* d_pos = nullptr;
            if (inline_k) {
                if (convention == 1) {
                    const size_t n_pos = (size_t)m->n_orb * m->dim;
                    if (m->pos_cache.size() != n_pos || std::memcmp(m->pos_cache.data(), pos, p_bytes) != 0) {
                        TBK_CHECK(m->ws_posraw.reserve(p_bytes));
                        std::memcpy(st + k_bytes, pos, p_bytes);
                        TBK_HIP(hipMemcpyAsync(m->ws_posraw.ptr, st + k_bytes, p_bytes, hipMemcpyHostToDevice, m->stream));
                        m->pos_cache.assign(pos, pos + n_pos);
                    }
                    m->d_pos_inline = m->ws_posraw.as<double>();
                    d_pos = m->d_pos_inline;  // (non-NULL for the argument checks; the kernels read pos_raw)
                }
                m->h_k_inline = k;
            } else {
                std::memcpy(st, k, k_bytes);
                TBK_HIP(hipMemcpyAsync(m->ws_k.ptr, st, k_bytes, hipMemcpyHostToDevice, m->stream));
                if (convention == 1) {
                    TBK_CHECK(m->ws_pos.reserve(p_bytes));
                    std::memcpy(st + k_bytes, pos, p_bytes);
                    TBK_HIP(hipMemcpyAsync(m->ws_pos.ptr, st + k_bytes, p_bytes, hipMemcpyHostToDevice, m->stream));
                    d_pos = m->ws_pos.as<double>();
                }
            }
            // Round 6: up to 1 MiB of H the kernels store straight into the pinned buffer (a host allocation is
            // device-addressable; the stores collect in the L2 and leave with the system-scope release of ev_sync) -- no
            // copy kernel and no dependent boundary in front of it: one-k hamilton 84 -> 68 us at 64 orbitals, 124 -> 112
            // at 256 (sparse).  4 MiB (512 orbitals) written that way take longer than the copy (scattered 16-byte stores
            // across PCIe: 1.69 -> 1.81 ms), so a bigger H still takes the copy; downloading it in two or four pieces, each
            // copied on to the caller's array while the next crosses PCIe, was measured and is within the noise of the
            // 1.3 - 1.4 ms kernel in front of it (1586 / 1661 / 1704 and 1568 / 1613 / 1563 us for 1 / 2 / 4 pieces).
            static const size_t direct_max = tbk_exp_env("TBK_ZERO_COPY_MAX") ? (size_t)atoll(tbk_exp_env("TBK_ZERO_COPY_MAX")) : (size_t(1) << 20);
            const bool direct = h_bytes <= direct_max;
            double* d_out = direct ? reinterpret_cast<double*>(st + h_off) : m->ws_out.as<double>();
            const int rc_inline = tbk_hamilton_device(m, m->ws_k.as<double>(), nk, convention, d_pos, d_out);
            m->h_k_inline = nullptr;
            m->d_pos_inline = nullptr;
            TBK_CHECK(rc_inline);
            if (!direct) TBK_HIP(hipMemcpyAsync(st + h_off, m->ws_out.ptr, h_bytes, hipMemcpyDeviceToHost, m->stream));
            TBK_CHECK(wait_main_stream(m));
            static const bool no_copy = tbk_exp_env("TBK_ABLATE_NO_HOSTCOPY") != nullptr;  // (timing only: what the host copy costs)
            if (!no_copy) std::memcpy(H_out, st + h_off, h_bytes);
            return TBK_OK;
        }
    }
    // H leaves in chunks of 16 to 128 MiB (a quarter of the result) through two device buffers: chunk c + 1 is computed while chunk c crosses PCIe
    // (the copy into pageable memory blocks this thread, not the GPU)
    const size_t total_bytes = (size_t)nk * nn2 * sizeof(double);
    const size_t chunk_bytes = std::min<size_t>(size_t(128) << 20, std::max<size_t>(size_t(16) << 20, total_bytes / 4));
    int64_t out_chunk = std::max<int64_t>(1, (int64_t)(chunk_bytes / (nn2 * sizeof(double))));
    out_chunk = std::min(out_chunk, nk);
    const int64_t n_chunks = (nk + out_chunk - 1) / out_chunk;
    TBK_CHECK(m->ws_k.reserve((size_t)nk * m->dim * sizeof(double)));
    TBK_HIP(hipMemcpyAsync(m->ws_k.ptr, k, (size_t)nk * m->dim * sizeof(double), hipMemcpyHostToDevice, m->stream));
    const double* d_pos = nullptr;
    if (convention == 1) {
        TBK_CHECK(m->ws_pos.reserve((size_t)m->n_orb * m->dim * sizeof(double)));
        TBK_HIP(hipMemcpyAsync(m->ws_pos.ptr, pos, (size_t)m->n_orb * m->dim * sizeof(double),
                               hipMemcpyHostToDevice, m->stream));
        d_pos = m->ws_pos.as<double>();
    }
    DevBuf* obuf[2] = {&m->ws_out, &m->ws_out2};
    for (int b = 0; b < (n_chunks > 1 ? 2 : 1); ++b) TBK_CHECK(obuf[b]->reserve((size_t)out_chunk * nn2 * sizeof(double)));
    auto compute = [&](int64_t c) -> int {
        const int64_t c0 = c * out_chunk, nkc = std::min(out_chunk, nk - c0);
        TBK_CHECK(tbk_hamilton_device(m, m->ws_k.as<double>() + c0 * m->dim, nkc, convention, d_pos, obuf[c & 1]->as<double>()));
        TBK_HIP(hipEventRecord(m->ev_out[c & 1], m->stream));
        return TBK_OK;
    };
    TBK_CHECK(compute(0));
    for (int64_t c = 0; c < n_chunks; ++c) {
        const int64_t c0 = c * out_chunk, nkc = std::min(out_chunk, nk - c0);
        if (c + 1 < n_chunks) TBK_CHECK(compute(c + 1));  // its buffer was drained by the (blocking) copy of chunk c - 1
        TBK_HIP(hipEventSynchronize(m->ev_out[c & 1]));
        TBK_HIP(hipMemcpy(H_out + (size_t)c0 * nn2, obuf[c & 1]->ptr, (size_t)nkc * nn2 * sizeof(double), hipMemcpyDeviceToHost));
    }
    return wait_main_stream(m);
}

// Small eigenvalue calls: the eigenvalues AND the two flag words leave the device through ONE small kernel that stores them
// straight into the pinned buffer (device-addressable host memory; the stores leave with the system-scope release of ev_sync).
// They used to be two hipMemcpyAsync, i.e. two copy kernels of ~4.4 us each with a dependent boundary in front of each: 8.8 of
// the 33 us of GPU work of a one-k eigenval of the silicon model.
__global__ void __launch_bounds__(256) export_small_kernel(const double* __restrict__ E, int64_t count, const int* __restrict__ flag,
                                                           double* __restrict__ host_E, int* __restrict__ host_flag) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) host_E[i] = E[i];
    if (blockIdx.x == 0 && threadIdx.x < 2) host_flag[threadIdx.x] = flag[threadIdx.x];
}

extern "C" int tbk_eigenval(tbk_model* m, const double* k, int64_t nk, double* E_out) {
    TBK_ARG(m != nullptr, "model is NULL");
    TBK_LOCK(m);
    TBK_ARG(nk >= 0, "nk < 0");
    if (nk == 0) return TBK_OK;
    TBK_ARG(k && E_out, "k / E is NULL");
    TBK_HIP(hipSetDevice(m->device));
    const size_t k_bytes = (size_t)nk * m->dim * sizeof(double), e_bytes = (size_t)nk * m->n_orb * sizeof(double);
    TBK_CHECK(m->ws_k.reserve(k_bytes));
    TBK_CHECK(m->ws_out.reserve(e_bytes));
    const size_t e_off = (k_bytes + 63) / 64 * 64;
    if (m->h_stage != nullptr && e_off + e_bytes + 16 <= m->h_stage_bytes) {
        // small call: [k | E | flags] through the pinned buffer, everything enqueued, one synchronisation
        char* st = static_cast<char*>(m->h_stage);
        int* flag = reinterpret_cast<int*>(st + e_off + e_bytes);
        // (one k-point of a dense model on the matrix-vector path: k travels in the kernel arguments, see tbk_hamilton --
        // only the chunk pipeline reads it from there: the rocSOLVER branch fills its phase rows from ws_k, which a call
        // that skipped the upload would leave stale)
        const bool inline_k = nk == 1 && !m->sparse && !m->kdotp && eigenval_own_solvers(m) && tbk_hk_inline_phases(m, 1);
        if (inline_k) {
            m->h_k_inline = k;
        } else {
            std::memcpy(st, k, k_bytes);
            TBK_HIP(hipMemcpyAsync(m->ws_k.ptr, st, k_bytes, hipMemcpyHostToDevice, m->stream));
        }
        // (the eigenvalues stored straight into the pinned buffer, like H in tbk_hamilton: measured, no gain -- 180.2 vs 180.0 us)
        const int rc_inline = eigenval_device_impl(m, m->ws_k.as<double>(), k, nk, m->ws_out.as<double>());
        m->h_k_inline = nullptr;
        TBK_CHECK(rc_inline);
        {
            const int64_t count = nk * m->n_orb;
            const unsigned blocks = (unsigned)std::min<int64_t>((count + 255) / 256, 64);
            hipLaunchKernelGGL(export_small_kernel, dim3(blocks), dim3(256), 0, m->stream, m->ws_out.as<double>(), count,
                               m->ws_flag.as<int>(), reinterpret_cast<double*>(st + e_off), flag);
            TBK_HIP(hipGetLastError());
        }
        TBK_CHECK(wait_main_stream(m));
        std::memcpy(E_out, st + e_off, e_bytes);
        if (flag[0] != 0 || flag[1] != 0) return tbk_eigenval_check(m);  // (rare) the ordinary path reports and resets
        return TBK_OK;
    }
    TBK_HIP(hipMemcpyAsync(m->ws_k.ptr, k, k_bytes, hipMemcpyHostToDevice, m->stream));
    TBK_CHECK(eigenval_device_impl(m, m->ws_k.as<double>(), k, nk, m->ws_out.as<double>()));
    TBK_HIP(hipMemcpyAsync(E_out, m->ws_out.ptr, e_bytes, hipMemcpyDeviceToHost, m->stream));
    return tbk_eigenval_check(m);  // synchronises
}

// ------------------------------------------------------------------------------------------------
// the reduction stage alone, on caller-supplied matrices
// ------------------------------------------------------------------------------------------------
extern "C" int tbk_tridiagonal_reduce(int device, int n_orb, int64_t nk, const double* H, int method, double* d, double* e,
                                      double* H_reduced) {
    TBK_ARG(nk >= 0, "nk < 0");
    TBK_ARG(n_orb >= 1 && (n_orb <= 64 || tbk_eig_stream_supported(n_orb)),
            "n_orb must be in [1, 4096] (larger matrices go through rocSOLVER as a whole)");
    TBK_ARG(method >= TBK_REDUCE_AUTO && method <= TBK_REDUCE_TWO_STAGE, "unknown reduction method");
    TBK_ARG(method != TBK_REDUCE_TWO_STAGE || tbk_eig_band_supported(n_orb), "the two-stage reduction handles 64 < n_orb <= 4096");
    TBK_ARG(method != TBK_REDUCE_ONE_STAGE || n_orb <= 512, "the one-stage reduction handles n_orb <= 512");
    if (nk == 0) return TBK_OK;
    TBK_ARG(H && d && e, "H / d / e is NULL");
    tbk_model* m = nullptr;
    TBK_CHECK(create_common(device, 1, n_orb, 0, nullptr, 2, &m));
    const size_t n = (size_t)n_orb, mat_bytes = n * n * 2 * sizeof(double);
    int rc = [&]() -> int {
        TBK_LOCK(m);
        m->call_nk = nk;
        TBK_CHECK(m->ws_H.reserve((size_t)nk * mat_bytes));
        TBK_CHECK(m->ws_E.reserve((size_t)nk * n * 2 * sizeof(double)));
        TBK_HIP(hipMemcpyAsync(m->ws_H.ptr, H, (size_t)nk * mat_bytes, hipMemcpyHostToDevice, m->stream));
        if (tbk_eig_small_supported(n_orb))
            TBK_CHECK(tbk_launch_tridiag(m, m->stream, m->ws_H.as<double>(), nk, m->ws_E.as<double>()));
        else
            TBK_CHECK(tbk_launch_tridiag_stream(m, m->stream, m->ws_H.as<double>(), nk, m->ws_E.as<double>(), method));
        TBK_HIP(hipMemcpyAsync(d, m->ws_E.ptr, (size_t)nk * n * sizeof(double), hipMemcpyDeviceToHost, m->stream));
        TBK_HIP(hipMemcpyAsync(e, m->ws_E.as<double>() + (size_t)nk * n, (size_t)nk * n * sizeof(double), hipMemcpyDeviceToHost,
                               m->stream));
        if (H_reduced)
            TBK_HIP(hipMemcpyAsync(H_reduced, m->ws_H.ptr, (size_t)nk * mat_bytes, hipMemcpyDeviceToHost, m->stream));
        TBK_HIP(hipStreamSynchronize(m->stream));
        return TBK_OK;
    }();
    tbk_model_destroy(m);
    return rc;
}

// The reduction stage ALONE on the chip, timed with HIP events: what `eig_roofline.standalone` of bench.py quotes beside the
// in-pipeline figure (there the H(k) of the next chunk shares the FP64 pipe).  Random Hermitian matrices are made on the
// device; every repetition works on a fresh copy (the reduction consumes its input), only the reduction is inside the events.
__global__ void __launch_bounds__(256) random_hermitian_kernel(double* __restrict__ H, int n, int64_t nk) {
    const int64_t total = nk * (int64_t)n * n;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t mat = t / ((int64_t)n * n);
        const int r = (int)((t / n) % n), c = (int)(t % n);
        const int i = r < c ? r : c, j = r < c ? c : r;  // the element of the upper triangle this one mirrors
        uint64_t x = (uint64_t)mat * 0x9E3779B97F4A7C15ull + (uint64_t)i * 0xBF58476D1CE4E5B9ull + (uint64_t)j * 0x94D049BB133111EBull + 1;
        x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
        const double re = (double)(int64_t)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5;
        x ^= x >> 29; x *= 0x9E3779B97F4A7C15ull; x ^= x >> 32;
        const double im = (double)(int64_t)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5;
        H[2 * t] = re;
        H[2 * t + 1] = i == j ? 0.0 : (r < c ? im : -im);
    }
}

extern "C" int tbk_reduce_standalone(int device, int n_orb, int64_t nk, int reps, double* us_per_matrix) {
    TBK_ARG(us_per_matrix != nullptr, "us_per_matrix is NULL");
    TBK_ARG(nk >= 1 && reps >= 1, "nk / reps < 1");
    TBK_ARG(n_orb >= 1 && (n_orb <= 64 || tbk_eig_stream_supported(n_orb)), "n_orb must be in [1, 4096]");
    for (int q = 0; q < 3; ++q) us_per_matrix[q] = 0.0;
    tbk_model* m = nullptr;
    TBK_CHECK(create_common(device, 1, n_orb, 0, nullptr, 2, &m));
    const size_t n = (size_t)n_orb, mat_bytes = n * n * 2 * sizeof(double);
    DevBuf pristine;
    hipEvent_t ev[2] = {nullptr, nullptr};
    int rc = [&]() -> int {
        TBK_LOCK(m);
        m->call_nk = nk;
        TBK_CHECK(pristine.reserve((size_t)nk * mat_bytes));
        TBK_CHECK(m->ws_H.reserve((size_t)nk * mat_bytes));
        TBK_CHECK(m->ws_E.reserve((size_t)nk * n * 2 * sizeof(double)));
        TBK_HIP(hipEventCreate(&ev[0]));
        TBK_HIP(hipEventCreate(&ev[1]));
        hipLaunchKernelGGL(random_hermitian_kernel, dim3(4096), dim3(256), 0, m->stream, pristine.as<double>(), n_orb, nk);
        TBK_HIP(hipGetLastError());
        const bool band = !tbk_eig_small_supported(n_orb) && tbk_eig_two_stage(m);
        if (band) {
            TBK_CHECK(m->ws_band.reserve((size_t)nk * tbk_band_scratch_per_matrix(n_orb)));
            TBK_CHECK(m->ws_bandmat[0].reserve((size_t)nk * tbk_band_bytes_per_matrix(n_orb)));
        }
        // what: 0 = the reduction as the pipeline runs it, 1 = first stage alone, 2 = second stage alone (two-stage sizes only).
        // The second stage has no input of its own: every repetition of `what == 2` chases the band the LAST repetition of
        // `what == 1` left in ws_bandmat[0] (the chase reads it and writes only (d, e): the same work every time).
        for (int what = 0; what < (band ? 3 : 1); ++what) {
            float sum = 0.0f;
            for (int r = 0; r <= reps; ++r) {  // (repetition 0 warms up)
                if (what != 2)
                    TBK_HIP(hipMemcpyAsync(m->ws_H.ptr, pristine.ptr, (size_t)nk * mat_bytes, hipMemcpyDeviceToDevice, m->stream));
                TBK_HIP(hipEventRecord(ev[0], m->stream));
                if (what == 0) {
                    if (tbk_eig_small_supported(n_orb))
                        TBK_CHECK(tbk_launch_tridiag(m, m->stream, m->ws_H.as<double>(), nk, m->ws_E.as<double>()));
                    else
                        TBK_CHECK(tbk_launch_tridiag_stream(m, m->stream, m->ws_H.as<double>(), nk, m->ws_E.as<double>(), TBK_REDUCE_AUTO));
                } else if (what == 1) {
                    TBK_CHECK(tbk_launch_band_reduce(m, m->stream, m->ws_H.as<double>(), nk, m->ws_band.ptr, m->ws_bandmat[0].ptr));
                } else {
                    TBK_CHECK(tbk_launch_band_chase(m, m->stream, m->ws_bandmat[0].ptr, nk, m->ws_E.as<double>()));
                }
                TBK_HIP(hipEventRecord(ev[1], m->stream));
                TBK_HIP(hipEventSynchronize(ev[1]));
                float ms = 0.0f;
                TBK_HIP(hipEventElapsedTime(&ms, ev[0], ev[1]));
                if (r > 0) sum += ms;
            }
            us_per_matrix[what] = (double)sum / reps * 1e3 / (double)nk;
        }
        return TBK_OK;
    }();
    for (auto& e : ev)
        if (e) (void)hipEventDestroy(e);
    pristine.release();
    tbk_model_destroy(m);
    return rc;
}

// ------------------------------------------------------------------------------------------------
// k.p models
// ------------------------------------------------------------------------------------------------
extern "C" int tbk_kdotp_create(int device, int dim, int n_orb, int64_t n_p, const int32_t* powers,
                                const double* coeffs, tbk_kdotp** out) {
    TBK_ARG(out != nullptr, "out is NULL");
    *out = nullptr;
    TBK_ARG(n_p == 0 || (powers && coeffs), "powers / coeffs is NULL");
    for (int64_t t = 0; t < n_p * dim; ++t) TBK_ARG(powers[t] >= 0, "negative power");
    tbk_model* m = nullptr;
    TBK_CHECK(create_common(device, dim, n_orb, n_p, nullptr, 1, &m));
    m->kdotp = true;
    double* d_raw = nullptr;
    const size_t raw_bytes = (size_t)n_p * n_orb * n_orb * 2 * sizeof(double);
    int rc = [&]() -> int {
        if (n_p > 0) {
            TBK_HIP(hipMalloc((void**)&m->d_powers, (size_t)n_p * dim * sizeof(int32_t)));
            TBK_HIP(hipMemcpy(m->d_powers, powers, (size_t)n_p * dim * sizeof(int32_t), hipMemcpyHostToDevice));
            TBK_HIP(hipMalloc((void**)&d_raw, raw_bytes));
            TBK_HIP(hipMemcpyAsync(d_raw, coeffs, raw_bytes, hipMemcpyHostToDevice, m->stream));
        }
        TBK_CHECK(tbk_stage_kdotp(m, d_raw));
        TBK_HIP(hipStreamSynchronize(m->stream));
        return TBK_OK;
    }();
    if (d_raw) (void)hipFree(d_raw);
    if (rc != TBK_OK) {
        tbk_model_destroy(m);
        return rc;
    }
    tbk_kdotp* kp = new (std::nothrow) tbk_kdotp();
    if (!kp) {
        tbk_model_destroy(m);
        tbk_set_error("out of host memory");
        return TBK_ERR_MEMORY;
    }
    kp->core = m;
    *out = kp;
    return TBK_OK;
}

extern "C" void tbk_kdotp_destroy(tbk_kdotp* kp) {
    if (!kp) return;
    tbk_model_destroy(kp->core);
    delete kp;
}

extern "C" int tbk_kdotp_hamilton(tbk_kdotp* kp, const double* k, int64_t nk, double* H_out) {
    TBK_ARG(kp != nullptr, "model is NULL");
    return tbk_hamilton(kp->core, k, nk, 2, nullptr, H_out);
}

extern "C" int tbk_kdotp_eigenval(tbk_kdotp* kp, const double* k, int64_t nk, double* E_out) {
    TBK_ARG(kp != nullptr, "model is NULL");
    return tbk_eigenval(kp->core, k, nk, E_out);
}

// ------------------------------------------------------------------------------------------------
// device memory helpers
// ------------------------------------------------------------------------------------------------
extern "C" int tbk_device_malloc(int device, int64_t bytes, void** d_ptr) {
    TBK_ARG(d_ptr != nullptr && bytes >= 0, "bad malloc arguments");
    TBK_CHECK(require_device(device));
    *d_ptr = nullptr;
    if (bytes == 0) return TBK_OK;
    TBK_HIP(hipMalloc(d_ptr, (size_t)bytes));
    return TBK_OK;
}

extern "C" int tbk_device_free(int device, void* d_ptr) {
    if (!d_ptr) return TBK_OK;
    TBK_CHECK(require_device(device));
    TBK_HIP(hipFree(d_ptr));
    return TBK_OK;
}

extern "C" int tbk_memcpy_h2d(int device, void* d_dst, const void* h_src, int64_t bytes) {
    TBK_CHECK(require_device(device));
    if (bytes > 0) {
        TBK_HIP(hipMemcpy(d_dst, h_src, (size_t)bytes, hipMemcpyHostToDevice));
        TBK_HIP(hipDeviceSynchronize());  // pageable H2D may return before the DMA has landed
    }
    return TBK_OK;
}

extern "C" int tbk_memcpy_d2h(int device, void* h_dst, const void* d_src, int64_t bytes) {
    TBK_CHECK(require_device(device));
    if (bytes > 0) TBK_HIP(hipMemcpy(h_dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost));
    return TBK_OK;
}

extern "C" int tbk_device_mem_info(int device, int64_t* free_bytes, int64_t* total_bytes) {
    TBK_CHECK(require_device(device));
    size_t f = 0, t = 0;
    TBK_HIP(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    return TBK_OK;
}

// ------------------------------------------------------------------------------------------------
// timing
// ------------------------------------------------------------------------------------------------
extern "C" int tbk_get_timing(tbk_model* m, double* ms, int64_t* launches, int reset) {
    TBK_ARG(m != nullptr, "model is NULL");
    TBK_LOCK(m);
    TBK_HIP(hipSetDevice(m->device));
    TBK_HIP(hipStreamSynchronize(m->stream));
    TBK_HIP(hipStreamSynchronize(m->stream_eig));
    TBK_HIP(hipStreamSynchronize(m->stream_ql));
    for (auto& ev : m->events) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, ev.start, ev.stop) == hipSuccess) {
            m->t_ms[ev.stage] += (double)t;
            m->t_n[ev.stage] += 1;
        }
        (void)hipEventDestroy(ev.start);
        (void)hipEventDestroy(ev.stop);
    }
    m->events.clear();
    for (int i = 0; i < TBK_T_COUNT; ++i) {
        if (ms) ms[i] = m->t_ms[i];
        if (launches) launches[i] = m->t_n[i];
        if (reset) {
            m->t_ms[i] = 0.0;
            m->t_n[i] = 0;
        }
    }
    return TBK_OK;
}

extern "C" int tbk_mfma_f64_peak(int device, double* tflops) {
    TBK_ARG(tflops != nullptr, "tflops is NULL");
    TBK_CHECK(require_device(device));
    return tbk_run_mfma_f64_peak(tflops);
}
