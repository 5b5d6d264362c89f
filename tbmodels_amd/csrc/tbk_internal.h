// tbk_internal.h -- shared declarations of libtbk.so (not part of the C ABI; see include/tbk.h).
#pragma once

#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>

#include <atomic>
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "tbk.h"

// ------------------------------------------------------------------------------------------------
// Tile geometry of the dense H(k) kernel (tbk_hk_dense.hip).  The staging code pads to these.
// ------------------------------------------------------------------------------------------------
constexpr int TBK_BM = 128;      // k-points per workgroup tile
constexpr int TBK_BNP = 64;      // packed (i <= j) matrix elements per workgroup tile (x2 real columns)
constexpr int TBK_BK = 16;       // depth of one LDS stage in real K rows (= 8 lattice vectors)
constexpr int TBK_CT = 16;       // packed elements per MFMA column tile
constexpr int TBK_MAX_DIM = 8;   // lattice dimension limit of the phase kernel

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
void tbk_set_error(const char* fmt, ...);

#define TBK_HIP(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            tbk_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,        \
                          __LINE__);                                                              \
            return (e_ == hipErrorOutOfMemory) ? TBK_ERR_MEMORY : TBK_ERR_DEVICE;                 \
        }                                                                                         \
    } while (0)

#define TBK_ROCBLAS(expr)                                                                         \
    do {                                                                                          \
        rocblas_status s_ = (expr);                                                               \
        if (s_ != rocblas_status_success) {                                                       \
            tbk_set_error("%s failed: rocblas_status %d (%s:%d)", #expr, (int)s_, __FILE__,       \
                          __LINE__);                                                              \
            return (s_ == rocblas_status_memory_error) ? TBK_ERR_MEMORY : TBK_ERR_DEVICE;         \
        }                                                                                         \
    } while (0)

#define TBK_CHECK(expr)                                                                           \
    do {                                                                                          \
        int r_ = (expr);                                                                          \
        if (r_ != TBK_OK) return r_;                                                              \
    } while (0)

#define TBK_LOCK(m) std::lock_guard<std::recursive_mutex> tbk_lock_((m)->mu)

#define TBK_ARG(cond, msg)                                                                        \
    do {                                                                                          \
        if (!(cond)) {                                                                            \
            tbk_set_error("invalid argument: %s", msg);                                           \
            return TBK_ERR_ARGUMENT;                                                              \
        }                                                                                         \
    } while (0)

// ------------------------------------------------------------------------------------------------
// Environment switches.  The DEFAULT library reads only the switches a test uses as an independent cross-check of the product
// path (plain getenv in the sources: TBK_BAND, TBK_BAND_SPLIT, TBK_BAND_XL, TBK_BAND_XL_FROM, TBK_CHASE_WINDOW, TBK_REG128,
// TBK_REG128_NW2, TBK_GATHER_BLOCK_ROWS -- the table in DESIGN.md section 6).  Every measurement switch of a variant that was
// built, measured and dropped goes through tbk_exp_env(), which is a constant NULL unless the library is built with
// `make EXPERIMENTS=1` (-DTBK_EXPERIMENTS): a user cannot flip a product-path branch by accident, and the dead branches
// fold away at compile time.
// ------------------------------------------------------------------------------------------------
#ifdef TBK_EXPERIMENTS
inline const char* tbk_exp_env(const char* name) { return getenv(name); }
#else
inline const char* tbk_exp_env(const char*) { return nullptr; }
#endif

// ------------------------------------------------------------------------------------------------
// dynamic LDS above the 64 KiB default: hipFuncSetAttribute acts on the CURRENT device's copy of the
// kernel, so it is raised once per (kernel instantiation, device) -- `done` is that instantiation's flag row.
// ------------------------------------------------------------------------------------------------
constexpr int TBK_MAX_DEVICES = 64;

// (the flags are atomics: two handles on one device -- Model.devices = [0, 0] -- launch from two host threads)
using tbk_flag_row = std::atomic<bool>[TBK_MAX_DEVICES];
inline hipError_t tbk_raise_lds_limit(const void* kernel, int bytes, tbk_flag_row& done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < TBK_MAX_DEVICES && done[dev].load(std::memory_order_acquire)) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && dev >= 0 && dev < TBK_MAX_DEVICES) done[dev].store(true, std::memory_order_release);
    return e;
}

// ------------------------------------------------------------------------------------------------
// a grow-only device buffer
// ------------------------------------------------------------------------------------------------
struct DevBuf {
    void* ptr = nullptr;
    size_t bytes = 0;
    int reserve(size_t want);  // keeps contents only if no reallocation happens
    void release();
    template <class T>
    T* as() const {
        return static_cast<T*>(ptr);
    }
};

struct EventPair {
    hipEvent_t start, stop;
    int stage;
};

// ------------------------------------------------------------------------------------------------
// folding along one k component (tbk_fold.hip)
// ------------------------------------------------------------------------------------------------
struct tbk_fold_plan_t {
    bool built = false;
    int dim = 0;                 // dimension of the lattice this plan folds (the folded one has dim - 1)
    int64_t n_r = 0;             // its lattice vectors
    int64_t n_rho = 0, n_rho_pad = 0, k2 = 0;  // folded lattice vectors; K rows of the folded operand
    int capacity = 0;            // folded operands that fit d_B2
    std::vector<int32_t> h_R2;   // host copy of the folded lattice [n_rho][dim - 1] (second-level plans are built on it)
    int32_t* d_R2 = nullptr;     // [n_rho_pad][dim - 1]
    int64_t* d_lptr = nullptr;   // [n_rho_pad + 1] lists of contributing lattice vectors
    int32_t* d_lrec = nullptr;   // r | (negated ? 1 << 31 : 0)
    int32_t* d_rcomp = nullptr;  // [n_r] the folded component of every lattice vector
    double* d_B2 = nullptr;      // [capacity][k2][ncol_pad * 2] folded operands
    double* d_table = nullptr;   // [n_r][slots][2] (cos, sin) of the shared-component phases
    int64_t table_entries = 0;   // its capacity in (r, slot) pairs
    tbk_fold_plan_t* sub = nullptr;  // [dim - 1] second-level plans (mesh lines inside a mesh plane), built on demand
};

struct tbk_fold_saved_t {
    int dim;
    int64_t n_r, n_r_pad, k2;
    int32_t* d_R;
    double* d_B;
};

// ------------------------------------------------------------------------------------------------
// staged model
// ------------------------------------------------------------------------------------------------
struct tbk_model {
    // One host thread at a time per handle: the entry points share workspaces, the fold cache and the stage timers
    // (ctypes drops the GIL for the duration of a call).  Recursive: the host-buffer calls go through the device ones.
    std::recursive_mutex mu;
    int device = 0;
    int64_t call_nk = 0;  // k-points of the eigenvalue call in progress: choices that must not depend on the chunking
    int n_cu = 256;  // compute units of the device (workgroup slots per round = 2 * n_cu for the H(k) kernel)
    int dim = 0;
    int n_orb = 0;
    int64_t n_r = 0;
    bool sparse = false;
    bool kdotp = false;  // K rows are k.p monomials (one per Taylor coefficient) instead of phases

    // --- common staging ---
    int64_t n_r_pad = 0;   // n_r rounded up so that 2 * n_r_pad is a multiple of TBK_BK
    int64_t k2 = 0;        // real K rows of the contraction: 2 * n_r_pad (cos, sin per lattice vector)
    int ncol = 0;          // n_orb (n_orb + 1) / 2 packed upper-triangle elements
    int ncol_pad = 0;      // rounded up to TBK_BNP
    int32_t* d_R = nullptr;       // [n_r_pad][dim] lattice vectors (padding rows are zero)
    int32_t* d_colmap = nullptr;  // [ncol_pad]  (i << 16) | j, or -1 for padding
    int32_t* d_powers = nullptr;  // k.p only: [n_r][dim] monomial exponents

    // --- dense: symmetrised hop planes, tile-interleaved  Bt[K2][ncol_pad / 16][2][16] ---
    double* d_B = nullptr;

    // --- sparse: per packed element, the list of lattice vectors that touch it ---
    int64_t nnz_rec = 0;
    int64_t* d_cptr = nullptr;   // [ncol + 1]
    int32_t* d_rec_r = nullptr;  // [nnz_rec]  (kind << 28) | r   kind: 0 direct, 1 transposed, 2 diagonal
    double* d_rec_v = nullptr;   // [nnz_rec][2]
    // the same records in the order the LDS kernel walks them: per 64 packed elements ("wave round") a number of steps,
    // every step one record (or none) per lane, arranged so that the 16 lanes the LDS serves together read 16 different
    // 16-byte slots of its 256-byte row (tbk_hk_csr.hip: tbk_csr_schedule)
    int sched_kt = 0;             // k-points per phase tile the schedule was built for (0: no schedule)
    int64_t sched_steps = 0;
    int64_t* d_sptr = nullptr;    // [ceil(ncol / 64) + 1] first step of every wave round
    int32_t* d_srec_r = nullptr;  // [sched_steps][64]  (sign code << 30) | byte offset of the phase row, 0x80000000: no record
    double* d_srec_v = nullptr;   // [sched_steps][64][2]

    int64_t staged_bytes = 0;

    // --- folding (k lists with long runs of one shared component) ---
    std::vector<int32_t> h_R;  // host copy of the lattice vectors [n_r][dim]
    tbk_fold_plan_t fold[TBK_MAX_DIM];
    bool fold_enabled = true;
    int64_t counters[TBK_CNT_COUNT] = {0, 0, 0, 0};  // tbk_model_counter

    // --- options ---
    int eigensolver = TBK_EIG_AUTO;
    int64_t k_chunk = 0;
    bool timing = false;

    // --- runtime ---
    hipStream_t stream = nullptr;      // phase rows, H(k), rocSOLVER, collectives
    hipStream_t stream_eig = nullptr;  // wave eigensolver: reduction to tridiagonal form
    hipStream_t stream_ql = nullptr;   // wave eigensolver: tridiagonal QL (latency-bound, overlaps the rest)
    hipStream_t stream_xl[3] = {nullptr, nullptr, nullptr};  // band_xl_* above 1024 orbitals: the other groups of a batch (tbk_eig_band.hip)
    hipEvent_t ev_xl[4] = {nullptr, nullptr, nullptr, nullptr};  // fork, and one join per extra group
    hipEvent_t ev_hk[2] = {nullptr, nullptr};   // H[buf] written
    hipEvent_t ev_tri[2] = {nullptr, nullptr};  // H[buf] consumed, (d, e)[buf] written
    hipEvent_t ev_out[2] = {nullptr, nullptr};  // tbk_hamilton: chunk in ws_out / ws_out2 computed
    hipEvent_t ev_ql[2] = {nullptr, nullptr};   // (d, e)[buf] consumed, eigenvalues written
    hipEvent_t ev_sync = nullptr;               // host waits on the main stream go through this event (tbk_api.hip)
    rocblas_handle blas = nullptr;
    DevBuf ws_phase;  // [K2][nk_pad] cos/sin rows
    DevBuf ws_H;      // [chunk][n_orb][n_orb] complex
    DevBuf ws_H2;     // second H buffer: folded H(k) of a chunk is built beside the previous chunk's reduction
    DevBuf ws_E;      // rocSOLVER: [chunk][n_orb] off-diagonal scratch; wave solver: (d, e) of buffer 0
    DevBuf ws_E2;     // wave solver: (d, e) of buffer 1
    DevBuf ws_info;   // [chunk] int
    DevBuf ws_k;      // host-entry staging of k / pos / E
    DevBuf ws_pos;
    DevBuf ws_orb;    // convention 1: orbital phase table of the current chunk
    DevBuf ws_out;
    DevBuf ws_out2;  // second device buffer of the chunked H(k) download (tbk_hamilton)
    DevBuf ws_flag;   // int[2]: {non-convergence count, non-finite count}
    // small host-buffer calls (one k-point per call is what Z2Pack-style callers do): k, the result and the flags cross
    // PCIe through this PINNED buffer -- asynchronous DMA copies enqueued back to back and ONE synchronisation, where
    // three copies from / to pageable memory each cost a host-side staging round trip (~20 us apiece)
    void* h_stage = nullptr;
    size_t h_stage_bytes = 0;
    DevBuf ws_part;   // split-K partial tiles of the dense H(k) kernel (small k batches)
    DevBuf ws_kfold;  // k-points of a folded run without the folded component
    DevBuf ws_kline;  // one mesh line without both folded components (second-level fold)
    DevBuf ws_band;   // two-stage reduction: pending [V | W] panel of every matrix of a chunk
    DevBuf ws_bandmat[2];  // ... and the band matrices between its stages (one per chunk in flight)
    // one-k host calls (tbk_hamilton / tbk_eigenval, nk == 1, dense): the k-point goes into the kernel arguments (no upload),
    // the convention-1 positions stay on the device between calls (uploaded again only when their bytes change)
    const double* h_k_inline = nullptr;    // the caller's k-point for the duration of the call, else NULL
    const double* d_pos_inline = nullptr;  // raw positions [n_orb][dim] on the device for the call in progress, else NULL
    std::vector<double> pos_cache;         // host copy of what ws_posraw holds
    DevBuf ws_posraw;
    DevBuf ws_split;  // calls of a few matrices: T and the members' partial X between the launches of the first stage
    DevBuf ws_xl;     // the launch chain of band_xl_*: the second matrix buffer (the sweep of a panel reads one, writes the other)
    // Set for the duration of one eigenvalue call by tbk_eigenval_device_gather (tbk_comm.hip): the chunk pipeline calls it
    // whenever the eigenvalues of rows [c0, c0 + nkc) of the call have been enqueued, with an event recorded behind
    // them -- the all-gather of finished rows leaves on the communicator's stream while later chunks compute.
    std::function<int(int64_t c0, int64_t nkc, hipEvent_t done)> chunk_done;
    size_t hk_lds_floor = 0;  // dynamic LDS the dense contraction asks for at least (0: what it needs)
    std::vector<EventPair> events;
    double t_ms[TBK_T_COUNT] = {0, 0, 0, 0};
    int64_t t_n[TBK_T_COUNT] = {0, 0, 0, 0};
};

struct tbk_kdotp {
    tbk_model* core = nullptr;  // the dense pipeline with monomial rows in place of phase rows
};

// roctx range around a stage (no-ops unless a roctx library can be loaded); tbk_api.hip
void tbk_range_push(const char* name);
void tbk_range_pop();

// timing scope helper: records a start/stop pair on the model stream when timing is on
struct StageTimer {
    tbk_model* m;
    EventPair ev;
    bool on;
    hipStream_t stream;
    StageTimer(tbk_model* m_, int stage, hipStream_t s = nullptr);
    ~StageTimer();
};

// ------------------------------------------------------------------------------------------------
// kernels (each .hip file exposes plain launchers)
// ------------------------------------------------------------------------------------------------
enum HkMode { HK_TRI = 0, HK_FULL = 1 };

// tbk_phase.hip
int tbk_launch_phase(tbk_model* m, const double* d_k, int64_t nk, int64_t nk_pad, double* d_A);
int tbk_launch_orbital_phases(tbk_model* m, const double* d_k, const double* d_pos, int64_t nk, double* d_orb);
int tbk_launch_monomials(hipStream_t s, const int32_t* d_powers, int dim, int64_t n_p,
                         int64_t n_p_pad, const double* d_k, int64_t nk, int64_t nk_pad,
                         double* d_A);

// tbk_stage.hip
int tbk_stage_dense(tbk_model* m, const double* d_hop_raw);
int tbk_stage_kdotp(tbk_model* m, const double* d_coeff_raw);

// tbk_hk_dense.hip
int tbk_launch_hk_dense(tbk_model* m, const double* d_A, int64_t nk, int64_t nk_pad, int mode,
                        int convention, const double* d_k, const double* d_pos, double* d_H);

int tbk_launch_hk_dense_lines(tbk_model* m, const double* d_A, int64_t n_lines, int line_len, int64_t b_stride, double* d_H);

// tbk_hk_csr.hip
int tbk_launch_hk_csr(tbk_model* m, const double* d_A, int64_t nk, int64_t nk_pad, int mode,
                      int convention, const double* d_k, const double* d_pos, double* d_H);

// host side of the sparse path: k-points per LDS phase tile for n_r lattice vectors (0: the tile does not fit), and the
// conflict-free walk order of the per-element records for that tile shape
int tbk_csr_tile_kpoints(int64_t n_r);
void tbk_csr_schedule(int ncol, int kt, const std::vector<int64_t>& cptr, const std::vector<int32_t>& rec_r, const std::vector<double>& rec_v,
                      std::vector<int64_t>& sptr, std::vector<int32_t>& srec_r, std::vector<double>& srec_v);

// tbk_eig.hip
int tbk_eig_batched(tbk_model* m, double* d_H, int64_t nk, double* d_E);    // full rocSOLVER zheevd

// tbk_eig_stream.hip
bool tbk_eig_stream_supported(int n);
// method: TBK_REDUCE_AUTO (what eigenval takes), _ONE_STAGE, _TWO_STAGE (tbk.h)
int tbk_launch_tridiag_stream(tbk_model* m, hipStream_t s, double* d_H, int64_t nk, double* d_de, int method = 0);
bool tbk_hk_inline_phases(const tbk_model* m, int64_t nk);  // tbk_hk_dense.hip
bool tbk_hk_gemv_path(const tbk_model* m, int64_t nk);
int tbk_launch_bisect(tbk_model* m, hipStream_t s, const double* d_de, int64_t nk, double* d_E);
size_t tbk_eig_scratch_per_k(const tbk_model* m);
int tbk_band_xl_reserve(tbk_model* m, int64_t max_nk);  // tbk_eig_band.hip: ws_xl for chunks of up to max_nk matrices

// tbk_eig_band.hip: two-stage reduction (dense -> band on the matrix pipe, band -> tridiagonal in LDS)
bool tbk_eig_band_supported(int n);
bool tbk_eig_band_preferred(int n);
size_t tbk_band_scratch_per_matrix(int n);
size_t tbk_band_bytes_per_matrix(int n);
size_t tbk_band_xl_buffer_per_matrix(int n);  // the second matrix buffer of the launch chain above 1024 orbitals
bool tbk_eig_two_stage(const tbk_model* m);  // the band path applies to this model (64 < n_orb <= band_maxn() = 4096 -- 1024 with TBK_BAND_XL=0 --, not TBK_BAND=0)
// d_de_fused != NULL: every workgroup runs the second stage for its matrix too and writes (d, e); d_band is not used
int tbk_launch_band_reduce(tbk_model* m, hipStream_t s, double* d_H, int64_t nk, void* d_vw, void* d_band,
                           double* d_de_fused = nullptr);
bool tbk_band_fused(int n);  // both stages in one kernel (<= 256 orbitals) or two launches, the second one overlappable
// calls of a few matrices: the first stage as a chain of launches, every tile pass on several CUs (never fused with stage two)
bool tbk_band_split(const tbk_model* m, int64_t nk);
bool tbk_band_xl_grouped(int n, int64_t nk);  // above 1024 orbitals, a batch: tbk_launch_band_reduce(..., d_band, d_de) runs both stages group by group
int tbk_launch_band_chase(tbk_model* m, hipStream_t s, const void* d_band, int64_t nk, double* d_de);

// tbk_eig_small.hip
bool tbk_eig_small_supported(int n);
// (above 32 orbitals the head of every matrix in d_H is overwritten with its trailing 32 x 32 block: H is consumed)
int tbk_launch_tridiag(tbk_model* m, hipStream_t s, double* d_H, int64_t nk, double* d_de);
bool tbk_eig_reg128_supported(int n);  // 64 < n <= 128: the eight-wave register kernel does the first n - 64 steps
int tbk_launch_tridiag_reg128(hipStream_t s, double* d_H, int n, int64_t nk, double* d_D, double* d_E, int64_t h_stride, int ldd, int off);
int tbk_launch_tridiag_tail64(hipStream_t s, double* d_H, int64_t nk, double* d_D, double* d_E, int n_full, int64_t call_nk);  // tbk_eig_stream.hip hands over here
// `beside_ql`: this launch shares the chip with another QL launch (the tail of the chunk pipeline): use
// half-size workgroups (32 KiB of LDS) that fit next to two resident 64 KiB ones.
int tbk_launch_ql(tbk_model* m, hipStream_t s, const double* d_de, int64_t nk, double* d_E, bool beside_ql = false);

// tbk_fold.hip
int tbk_fold_choose(tbk_model* m, const double* h_k, int64_t nk, std::vector<int64_t>& run_starts);
int tbk_fold_group_size();
int tbk_fold_group(tbk_model* m, tbk_fold_plan_t& plan, const double* h_kf, int n_g, int slot0);
int tbk_fold_lines(tbk_model* m, tbk_fold_plan_t& plan, const double* d_kf, int64_t stride, int n_lines, int slot0 = 0);
int tbk_fold_enter(tbk_model* m, tbk_fold_plan_t& plan, int slot, tbk_fold_saved_t& saved);
void tbk_fold_leave(tbk_model* m, const tbk_fold_saved_t& saved);
int tbk_fold_drop_component(tbk_model* m, const double* d_k, int dim, int f, int64_t nk, double* d_k2);
int tbk_fold_subplan(tbk_model* m, tbk_fold_plan_t& parent, int f2, int capacity, tbk_fold_plan_t** out);
void tbk_fold_release(tbk_model* m);
int64_t tbk_fold_min_run();

// tbk_peak.hip
int tbk_run_mfma_f64_peak(double* tflops);
