/*
 * tbk.h -- C ABI of libtbk.so: the MI355X (gfx950) k-space evaluation path of a tight-binding model.
 *
 *     H(k) = sum_R exp(2 pi i k.R) hop[R]  +  h.c.        and        eigenvalues of H(k)
 *
 * The reference (Z2PackDev/TBmodels 1.4.4) is pure Python and has NO FFI / plugin interface for
 * this path: the path is two bound methods of tbmodels.Model.  This header is therefore the
 * boundary a maintainer would bind (ctypes stub in INTEGRATION.md) to replace the bodies of
 *
 *     Model.hamilton(k, convention=2)   /root/reference/src/tbmodels/_tb_model.py:1076-1132
 *     Model.eigenval(k)                 /root/reference/src/tbmodels/_tb_model.py:1134-1150
 *     KdotpModel.hamilton / eigenval    /root/reference/src/tbmodels/kdotp.py:51-100
 *
 * Conventions
 *   - plain pointers and sizes only; complex128 is passed as interleaved (re, im) doubles;
 *   - every array is C-contiguous, row-major, in the reference's own layouts:
 *       R    int32   [n_r][dim]          lattice vectors = keys of model.hop (_tb_model.py:206-210)
 *       hop  double  [n_r][n_orb][n_orb][2]   = values of model.hop: half-space blocks, the R = 0
 *                                             block stored HALVED (_tb_model.py:268, :218)
 *       k    double  [nk][dim]           reduced coordinates, NOT reduced mod 1 (_tb_model.py:1103-1108)
 *       pos  double  [n_orb][dim]        orbital positions, only read for convention 1 (:1124-1128)
 *       H    double  [nk][n_orb][n_orb][2]    H[k][i][j] (_tb_model.py:1109)
 *       E    double  [nk][n_orb]         ascending eigenvalues per k (scipy.linalg.eigvalsh, :1149)
 *   - every function returns 0 on success, a tbk_status otherwise; tbk_last_error() gives the
 *     message of the calling thread's last failure;
 *   - "host" entry points take caller-owned host memory and return when the result is in it;
 *     "device" entry points take device pointers on the model's device, enqueue on the model's
 *     streams and return without synchronising and without reading device memory back
 *     (tbk_synchronize waits);
 *   - a handle's staged model is immutable after creation (re-create it when model.hop changes);
 *     host threads calling into ONE handle are serialised by a lock inside it (they share its
 *     workspaces and streams); different handles are independent and run concurrently.
 *
 * There is no CPU implementation behind this interface: without a gfx950 device every compute
 * entry point fails with TBK_ERR_DEVICE.
 */
#ifndef TBK_H
#define TBK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tbk_model tbk_model; /* staged hoppings + workspaces on one device */
typedef struct tbk_kdotp tbk_kdotp; /* staged Taylor coefficients of a k.p model   */
typedef struct tbk_comm tbk_comm;   /* RCCL communicator of one rank               */

typedef enum tbk_status {
    TBK_OK = 0,
    TBK_ERR_ARGUMENT = 1, /* bad size / null pointer / convention not in {1, 2}  -> ValueError   */
    TBK_ERR_DEVICE = 2,   /* no device, HIP / rocBLAS / RCCL failure              -> RuntimeError */
    TBK_ERR_MEMORY = 3,   /* device allocation failed                             -> MemoryError  */
    TBK_ERR_NOT_FINITE = 4, /* NaN / Inf in H(k): scipy check_finite=True         -> ValueError   */
    TBK_ERR_NO_CONVERGENCE = 5 /* eigensolver did not converge                    -> LinAlgError  */
} tbk_status;

/* eigensolver selection for tbk_model_set_option(TBK_OPT_EIGENSOLVER, ...):
 *   WAVE      hand-written register-resident Householder reduction (n_orb <= 64 only) + tridiagonal stage
 *   ROCSOLVER rocsolver_zheevd_strided_batched
 *   AUTO      WAVE when n_orb <= 64; the own one- / two-stage Householder kernels up to n_orb = 4096 (above 1024 orbitals the
 *             first stage is a chain of launches per panel, csrc/tbk_eig_band.hip band_xl_*); ROCSOLVER above.
 * Tridiagonal stage of the two hand-written paths: lane-per-matrix QL for large batches of n_orb <= 64,
 * bisection on Sturm counts otherwise (n_orb > 64, calls of <= max(4096, 768 n_orb) k-points, and the last
 * chunk of a call).  Both are backward stable; they agree to rounding, so eigenvalues are reproducible run to
 * run but depend on the batch size at the 1e-13 level.  The same holds for the reduction above 128 orbitals: calls of
 * a few matrices take wider kernels / a chain of launches (tbk_eig_band.hip: tbk_band_split, `wide`) whose partial sums
 * differ from the one-workgroup kernels' in the last bit.  Bitwise reproducibility is therefore a property of a CALL SHAPE:
 * the same k list on the same number of devices / ranks (tbk_eigenval_multi and ShardedEigenval cut it into
 * ceil(nk / n) slabs, and every slab chooses its kernels by ITS size) gives the same bits every time; the same list on a
 * different device count agrees to rounding only. */
enum { TBK_EIG_AUTO = 0, TBK_EIG_WAVE = 1, TBK_EIG_ROCSOLVER = 2 };
enum {
    TBK_OPT_EIGENSOLVER = 1, /* one of TBK_EIG_*                                           */
    TBK_OPT_K_CHUNK = 2,     /* max k-points per internal chunk (0 = choose from free HBM) */
    TBK_OPT_TIMING = 3,      /* 1: bracket every kernel with HIP events (tbk_get_timing)   */
    TBK_OPT_FOLD = 4         /* 0: never fold k lists with long runs of one shared component (grids) into
                              *    lower-dimensional models (default 1; dense models, eigenval only)          */
};

/* ---- library / device ------------------------------------------------------------------ */
const char* tbk_version(void);
const char* tbk_last_error(void);
int tbk_device_count(int* count);

/* ---- model staging (replaces the per-call walk over model.hop, _tb_model.py:1111) ------- */

/* Dense hoppings.  `hop` = the n_r stored (n_orb x n_orb) complex blocks, in the order of `R`. */
int tbk_model_create_dense(int device, int dim, int n_orb, int64_t n_r, const int32_t* R,
                           const double* hop, tbk_model** out);

/* Sparse hoppings (model._sparse, _sparse_matrix.py:35-37): the per-R scipy CSR matrices
 * concatenated as COO triplets; entries of lattice vector r are [r_ptr[r], r_ptr[r+1]).
 * Duplicate (row, col) entries inside one R are summed, like scipy's toarray(). */
int tbk_model_create_csr(int device, int dim, int n_orb, int64_t n_r, const int32_t* R,
                         const int64_t* r_ptr, const int32_t* row, const int32_t* col,
                         const double* val, tbk_model** out);

void tbk_model_destroy(tbk_model* m);
int tbk_model_set_option(tbk_model* m, int option, int64_t value);
int tbk_model_info(const tbk_model* m, int* device, int* dim, int* n_orb, int64_t* n_r,
                   int* is_sparse, int64_t* staged_bytes);
/* Event counters of a handle since its creation (which path the eigenvalue calls took). */
enum { TBK_CNT_EIGENVAL_CALLS = 0, TBK_CNT_FOLDED_CALLS = 1, TBK_CNT_FOLDED_KPOINTS = 2,
       TBK_CNT_LIBRARY_CALLS = 3, /* eigenvalue calls handed to rocSOLVER (on request, or above the own kernels' range) */
       TBK_CNT_COUNT = 4 };
int tbk_model_counter(tbk_model* m, int counter, int64_t* value);

/* ---- the hot path, host buffers (what Model.hamilton / Model.eigenval call) -------------- */

/* H(k) for nk k-points.  convention in {1, 2}; pos may be NULL for convention 2. */
int tbk_hamilton(tbk_model* m, const double* k, int64_t nk, int convention, const double* pos,
                 double* H_out);

/* Ascending eigenvalues of H(k) (convention 2) for nk k-points. */
int tbk_eigenval(tbk_model* m, const double* k, int64_t nk, double* E_out);

/* ---- the hot path on several devices from ONE process (host buffers) ----------------------
 * handles[0..n_handles) are staged copies of the SAME model, normally one per device (several on one device are
 * allowed).  The k list is cut into contiguous slabs of ceil(nk / n_handles) rows (handle i takes slab i; the last
 * slabs may be short or empty), every slab runs on its handle's device from its own host thread, and every device
 * copies its result straight into its rows of the caller's array: k-points are independent (_tb_model.py:1111-1123)
 * and the result is in caller order (:1147-1150) without any exchange.  Mesh slabs are folded like whole meshes.
 * On failure the status and message of the first failing slab in k order are returned (a NaN k-point gives
 * TBK_ERR_NOT_FINITE exactly once).  n_handles == 1 is tbk_eigenval / tbk_hamilton. */
int tbk_eigenval_multi(tbk_model* const* handles, int n_handles, const double* k, int64_t nk, double* E_out);
int tbk_hamilton_multi(tbk_model* const* handles, int n_handles, const double* k, int64_t nk, int convention,
                       const double* pos, double* H_out);

/* ---- the hot path, device buffers (bench, sharded runs, device-side consumers) ----------- */
int tbk_hamilton_device(tbk_model* m, const double* d_k, int64_t nk, int convention,
                        const double* d_pos, double* d_H);
int tbk_eigenval_device(tbk_model* m, const double* d_k, int64_t nk, double* d_E);
/* The same for a caller that still holds the host array it uploaded: h_k == the contents of d_k (or NULL).
 * Device-resident lists are never read back -- that would synchronise -- so long runs of one shared k component
 * (uniform meshes in meshgrid order, stacks of planes; TBK_OPT_FOLD) are only recognised through h_k: the run
 * STRUCTURE and the shared-component values are taken from h_k, everything else from d_k.  tbk_eigenval_device is
 * this call with h_k = NULL (never folds); tbk_eigenval (host buffers) always has the list. */
int tbk_eigenval_device_hint(tbk_model* m, const double* d_k, const double* h_k, int64_t nk, double* d_E);
/* Check the info flags of the eigenvalue calls since the last check (synchronises). */
int tbk_eigenval_check(tbk_model* m);
int tbk_synchronize(tbk_model* m);

/* ---- the eigensolver's reduction stage alone (scipy.linalg.eigvalsh of _tb_model.py:1149 = this + the tridiagonal stage)
 * nk Hermitian matrices H[nk][n_orb][n_orb][2] (row-major; only the upper triangle i <= j is read) are reduced to real
 * symmetric tridiagonal form with the same eigenvalues: d[nk][n_orb] diagonals, e[nk][n_orb] off-diagonals (e[.][n-1] = 0).
 * n_orb <= 4096.  method: TBK_REDUCE_AUTO = what tbk_eigenval takes for this size (register-resident reduction up to 64
 * orbitals, one-stage reduction up to 188 -- in registers up to 128, streaming above -- two-stage reduction -- dense ->
 * band of half-width 8 on the matrix pipe, band -> tridiagonal by bulge chasing -- from 185 to 4096); _ONE_STAGE / _TWO_STAGE force one of them (one-stage:
 * n_orb <= 512 only; two-stage: 64 < n_orb <= 4096 only).  H_reduced (may be NULL) receives the work copy of the matrices as the reduction left it:
 * after the two-stage reduction its upper triangle holds the band form of stage one.
 * Host buffers; synchronous.  For tests and for callers that bring their own matrices. */
enum { TBK_REDUCE_AUTO = 0, TBK_REDUCE_ONE_STAGE = 1, TBK_REDUCE_TWO_STAGE = 2 };
int tbk_tridiagonal_reduce(int device, int n_orb, int64_t nk, const double* H, int method, double* d, double* e,
                           double* H_reduced);

/* The reduction stage alone on the chip, timed with HIP events on random Hermitian matrices made on the device (nothing
 * crosses PCIe): us_per_matrix[0] = the reduction as tbk_eigenval runs it for this size and call size (both stages of the
 * two-stage path), [1] = its first stage (dense -> band) alone, [2] = its second stage (band -> tridiagonal) alone -- [1],
 * [2] are 0 below the two-stage sizes.  Mean over `reps` repetitions after one warm-up.  Measurement only (bench.py
 * `eig_roofline.standalone`): the in-pipeline stage time shares the FP64 pipe with the next chunk's H(k). */
int tbk_reduce_standalone(int device, int n_orb, int64_t nk, int reps, double* us_per_matrix);

/* ---- k.p models (kdotp.py:51-100): H(k) = sum_p prod_d k_d^powers[p][d] * coeffs[p] ------- */
int tbk_kdotp_create(int device, int dim, int n_orb, int64_t n_p, const int32_t* powers,
                     const double* coeffs, tbk_kdotp** out);
void tbk_kdotp_destroy(tbk_kdotp* m);
int tbk_kdotp_hamilton(tbk_kdotp* m, const double* k, int64_t nk, double* H_out);
int tbk_kdotp_eigenval(tbk_kdotp* m, const double* k, int64_t nk, double* E_out);
/* The same on several devices from one process: staged copies of ONE k.p model, contiguous k slabs, one host thread
 * per non-empty slab -- tbk_eigenval_multi / tbk_hamilton_multi for kdotp.py:51-100. */
int tbk_kdotp_eigenval_multi(tbk_kdotp* const* handles, int n_handles, const double* k, int64_t nk, double* E_out);
int tbk_kdotp_hamilton_multi(tbk_kdotp* const* handles, int n_handles, const double* k, int64_t nk, double* H_out);

/* Model.construct_kdotp (_tb_model.py:942-982): Taylor coefficients of H(k) around k0 for n_p power
 * tuples.  powers: int32 [n_p][dim]; prefactor: double [n_p][2] = (2 pi i)^{|p|} / prod p_d! as (re, im);
 * coeffs_out: double [n_p][n_orb][n_orb][2] (Hermitian matrices).  Host buffers; dense handles only. */
int tbk_kdotp_coefficients(tbk_model* m, const double* k0, int64_t n_p, const int32_t* powers,
                           const double* prefactor, double* coeffs_out);

/* ---- device memory helpers (so a Python host needs no other GPU runtime) ----------------- */
int tbk_device_malloc(int device, int64_t bytes, void** d_ptr);
int tbk_device_free(int device, void* d_ptr);
int tbk_memcpy_h2d(int device, void* d_dst, const void* h_src, int64_t bytes);
int tbk_memcpy_d2h(int device, void* h_dst, const void* d_src, int64_t bytes);
int tbk_device_mem_info(int device, int64_t* free_bytes, int64_t* total_bytes);

/* ---- timing: HIP events around every kernel of the path, on the stream it is launched on -------
 * stages: PHASE phase rows; HK the H(k) contraction; EIG reduction to tridiagonal form (or the whole
 * rocSOLVER call); QL the tridiagonal stage (QL + sort, or bisection).  Stages of different k chunks may overlap. */
enum { TBK_T_PHASE = 0, TBK_T_HK = 1, TBK_T_EIG = 2, TBK_T_QL = 3, TBK_T_COUNT = 4 };
/* ms[i] = summed duration of stage i, launches[i] = number of timed launches; reset = 1 clears. */
int tbk_get_timing(tbk_model* m, double* ms, int64_t* launches, int reset);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI --------------------------------------- */
/* 128-byte RCCL unique id, created on rank 0 and handed to the other ranks by the launcher. */
int tbk_comm_unique_id(void* id128);
int tbk_comm_create(int device, int world_size, int rank, const void* id128, tbk_comm** out);
void tbk_comm_destroy(tbk_comm* c);
/* Size of the communicator and this process' rank in it, as RCCL reports them (ncclCommCount / ncclCommUserRank). */
int tbk_comm_ranks(tbk_comm* c, int* count, int* rank);
/* All-gather of per-rank eigenvalue slabs: every rank contributes `count` doubles from d_send and
 * receives world_size * count doubles in rank order in d_recv.  Enqueued on `m`'s stream; with m == NULL (a rank
 * that could not stage its model still has to take part) on the communicator's own stream (tbk_comm_synchronize). */
int tbk_comm_allgather_f64(tbk_comm* c, tbk_model* m, const double* d_send, double* d_recv,
                           int64_t count);
/* The same gather on the communicator's own stream: it starts once the work enqueued on `m`'s stream so far
 * is complete and overlaps whatever is enqueued on `m` afterwards.  Callers alternate two (send, recv) pairs,
 * slot 0 / 1; tbk_comm_wait_slot makes `m`'s stream wait for the previous gather of a slot before its buffers
 * are written again (stream dependency, the host does not block); tbk_comm_synchronize waits on the host. */
int tbk_comm_allgather_f64_overlapped(tbk_comm* c, tbk_model* m, const double* d_send, double* d_recv,
                                      int64_t count, int slot);
int tbk_comm_wait_slot(tbk_comm* c, tbk_model* m, int slot);
int tbk_comm_synchronize(tbk_comm* c);
/* Agreement in front of a sharded call: every rank contributes `status` (0 = ready to take part), verdict[world]
 * (HOST) receives every rank's word.  One 8-byte all-gather through buffers the communicator owns since its
 * creation; synchronous.  A rank whose staging / allocation failed reports it HERE, and nobody enters the data
 * collectives (a rank raising alone in front of a collective leaves its peers hanging in it). */
int tbk_comm_agree(tbk_comm* c, int status, double* verdict);
/* The landing area of tbk_eigenval_device_gather for slabs of `per` rows of n_orb eigenvalues (grow-only, sized from per,
 * n_orb and the world size alone).  Call it in the step whose outcome tbk_comm_agree exchanges: an allocation failure on one
 * rank then reaches every rank's verdict BEFORE anybody enters the data collectives.  (The gather allocates by itself when
 * this was skipped; a failure there travels in that rank's status word, behind the same sequence of collectives.) */
int tbk_comm_prepare_gather(tbk_comm* c, int n_orb, int64_t per);

/* One sharded eigenvalue call with the gather pipelined behind the k chunks (replaces the per-process body of a
 * multiprocessing farm over Model.eigenval, _tb_model.py:1134-1150; k-points are independent, :1111-1123).
 * Every rank calls it with the SAME `per` (slab length in k-points = ceil(NK / world)) and its own nk <= per k-points
 * (d_k / h_k as in tbk_eigenval_device_hint).  d_all[world][per][n_orb] (device) receives the eigenvalues of ALL ranks in
 * rank = caller order: this rank's rows are computed in place, and while later k chunks compute, finished blocks of
 * rows are all-gathered on the communicator's stream and moved to their places; rows nk..per of a short slab are
 * zero.  host_status != 0: this rank computes nothing and reports that status (a failure in front of the call).
 * d_status_all[world] (device) receives every rank's tbk_status as doubles (the solvers' non-finite / convergence
 * flags included; they are consumed) -- the last collective of the call, so every rank sees the same verdict.
 * Everything is enqueued: tbk_comm_synchronize(c) waits for the result, and `m`'s main stream waits for the gathers
 * before any later call touches the buffers. */
int tbk_eigenval_device_gather(tbk_comm* c, tbk_model* m, const double* d_k, const double* h_k, int64_t nk, int64_t per,
                               int host_status, double* d_all, double* d_status_all);

/* ---- microbenchmark: sustained v_mfma_f64_16x16x4_f64 rate of the device (TFLOP/s) -------- */
int tbk_mfma_f64_peak(int device, double* tflops);

#ifdef __cplusplus
}
#endif
#endif /* TBK_H */
