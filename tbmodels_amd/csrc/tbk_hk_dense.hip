// tbk_hk_dense.hip -- H(k) for a chunk of k-points and dense hoppings: the Fourier sum of
// /root/reference/src/tbmodels/_tb_model.py:1109-1128 as ONE real f64 MFMA contraction.
//
//     H[k][e].{re,im} = sum_kk  A[kk][k] * Bt[kk][e].{re,im}
//
//     A   [k2][nk_pad]                 cos/sin rows from tbk_phase.hip (k contiguous)
//     Bt  [k2][ncol_pad/16][2][16]     symmetrised hoppings from tbk_stage.hip
//     e = packed upper-triangle element (i <= j);  output scattered to H[k][i][j] (and, in FULL
//     mode, the conjugate to H[k][j][i]) -- the `H += H^H` of :1123 is already inside Bt.
//
// Machine mapping (gfx950):
//   * v_mfma_f64_16x16x4_f64: 16 k-points x 16 columns x 4 K rows per instruction, one f64 of A and
//     of B per lane (A[i = lane & 15][kk = lane >> 4], B[kk = lane >> 4][j = lane & 15]), result
//     D[row = (lane >> 4) + 4 reg][col = lane & 15].
//   * workgroup = 4 waves = 128 k-points x 64 packed elements (128 real columns); wave (wm, wn)
//     owns 64 k-points x 32 elements = 4 x 2 x {re, im} = 16 accumulators (128 VGPRs).
//   * K is walked in stages of 16 rows through double-buffered LDS, filled by LDS-DMA
//     (global_load_lds_dwordx4) issued before the MFMAs of the current stage; one barrier per stage.
//     LDS rows are padded by 16 doubles so that the two K rows a 32-lane group reads with
//     ds_read_b64 fall in different halves of the 64-bank row: conflict-free.
//   * both operands are K-major with the fast index contiguous, so every staging load is a
//     full 1 KiB wave row (16 B per lane) and every fragment read is 128 contiguous bytes per
//     16 lanes.
//   * the tile grid is walked so that the 32 workgroups resident on one XCD (block b runs on XCD
//     b % 8) share A row-panels and Bt column-panels in that XCD's private L2.
//
// Roofline: 8 * ncol * n_r flops per k-point against (16 + 16) B of operand traffic per
// (k-tile row + column) K step -- FP64-MFMA bound (78.6 TFLOP/s), not HBM bound; see DESIGN.md.

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "tbk_internal.h"

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int LDA = TBK_BM + 16;       // doubles per K row of the A stage
constexpr int LDB = 2 * TBK_BNP + 16;  // doubles per K row of the B stage
constexpr int STAGE_DOUBLES = TBK_BK * (LDA + LDB);

static_assert(TBK_BM == 128 && TBK_BNP == 64 && TBK_BK == 16, "wave decomposition is written for 128x64x16");

struct HkArgs {
    const double* A;
    const double* Bt;
    const int32_t* colmap;
    const double* kpts;  // [nk][dim]: read by the matrix-vector kernel when it makes its own phase rows (A == nullptr)
    const int32_t* R;    // [n_r_pad][dim] lattice vectors, same use
    int64_t n_r;
    const double* pos;   // convention 1 only: orbital phase table e[k][p] = exp(2 pi i k.pos_p), [nk][n_orb][2]
    double* H;
    int64_t k2;
    int64_t nk;
    int64_t nk_pad;
    int ncol_pad;
    int n_orb;
    int dim;
    int mt_count;  // k tiles
    int nt_count;  // element tiles
    int xcd_rows;  // 0: plain order; > 0: k tiles per XCD super-row
    // split-K (small k batches): the launch covers `splits * unit_grid` units, unit u = split y = u / unit_grid of
    // block b = u % unit_grid of the tile walk; a unit owns a contiguous range of K stages and stores its partial tile
    // to P[y][k][e] (re, im) instead of scattering it; hk_finish_kernel adds the partials in fixed order
    double* P;
    int splits;
    int unit_grid;
    int64_t p_rows;  // k rows per split in P
    // "lines" launches (second-level fold, tbk_fold.hip): k tile t is one mesh line -- its own operand at
    // Bt + t * b_tile_stride, the SAME phase rows for every line (a_tile_stride = 0), rows_per_tile k-points of output
    int64_t a_tile_stride;  // TBK_BM in ordinary launches
    int64_t b_tile_stride;  // 0 in ordinary launches
    int rows_per_tile;      // TBK_BM in ordinary launches
    // "tail" launches: the units of the last, partly filled round of a launch are split along K once more (launch()
    // below): this launch covers units block_offset + blockIdx.x, blockIdx.y < sub_splits takes a part of the unit's
    // stage range, and the partial tile goes to the compact P2[blockIdx.y][blockIdx.x][128][64] (re, im);
    // hk_finish_tiles_kernel adds them in order into H (splits == 1) or into the unit's P[y] (splits > 1)
    int block_offset;
    int p_tiles;  // > 0: a tail launch of that many units
    int sub_splits;
    double* P2;
    // one-k host calls (round 4; Z2Pack evaluates ONE k-point per call, _tb_model.py:1103-1108): the k-point travels in the
    // kernel arguments instead of through an upload the kernels would have to wait for, and the convention-1 phases of the one
    // k-point are formed where they are used from the raw orbital positions (no orbital_phase_kernel launch in front)
    double k_val[TBK_MAX_DIM];
    int k_inline;
    const double* pos_raw;  // [n_orb][dim], with k_inline and convention 1
};

__device__ __forceinline__ double hk_kcomp(const HkArgs& a, int64_t kq, int d) { return a.k_inline ? a.k_val[d] : a.kpts[kq * a.dim + d]; }

// One finished element of the packed tile -> H[k][i][j] (and H[k][j][i] conjugated in FULL mode), with the
// convention-1 orbital phases if asked (_tb_model.py:1124-1128; the table e[k][p] = exp(2 pi i k.pos_p) is
// filled once per chunk by tbk_launch_orbital_phases).
template <int MODE, int CONV>
__device__ __forceinline__ void store_element(const HkArgs& a, int64_t kq, int oi, int oj, double re, double im) {
    if (CONV == 1 && oi != oj) {  // (on the diagonal conj(e_i) e_i = 1: left alone, so that Im H[i][i] stays exactly 0)
        d2 ei, ej;
        if (a.k_inline) {  // the one k-point of a host call: e_p = exp(2 pi i k.pos_p), the arithmetic of orbital_phase_kernel
            double di = 0.0, dj = 0.0;
            for (int d = 0; d < a.dim; ++d) {
                di = fma(a.k_val[d], a.pos_raw[oi * a.dim + d], di);
                dj = fma(a.k_val[d], a.pos_raw[oj * a.dim + d], dj);
            }
            double sn, cs;
            sincospi(2.0 * di, &sn, &cs);
            ei = (d2){cs, sn};
            sincospi(2.0 * dj, &sn, &cs);
            ej = (d2){cs, sn};
        } else {
            ei = *reinterpret_cast<const d2*>(a.pos + ((size_t)kq * a.n_orb + oi) * 2);
            ej = *reinterpret_cast<const d2*>(a.pos + ((size_t)kq * a.n_orb + oj) * 2);
        }
        const double cs = ei[0] * ej[0] + ei[1] * ej[1], sn = ei[0] * ej[1] - ei[1] * ej[0];
        const double t = re * cs - im * sn;
        im = re * sn + im * cs;
        re = t;
    }
    double* hk = a.H + (size_t)kq * a.n_orb * a.n_orb * 2;
    *reinterpret_cast<d2*>(hk + ((size_t)oi * a.n_orb + oj) * 2) = (d2){re, im};
    if (MODE == HK_FULL && oi != oj) *reinterpret_cast<d2*>(hk + ((size_t)oj * a.n_orb + oi) * 2) = (d2){re, -im};
}

__device__ __forceinline__ bool tile_of_block(const HkArgs& a, int b, int& mt, int& nt) {
    if (a.xcd_rows == 0) {
        nt = b % a.nt_count;
        mt = b / a.nt_count;
        return mt < a.mt_count;
    }
    // XCD-aware walk (block b is dispatched to XCD b % 8): XCD x owns a contiguous range of k tiles;
    // inside it consecutive blocks sweep `xcd_rows` k tiles for one element tile, then the next element
    // tile, so the ~64 co-resident workgroups of an XCD form an (xcd_rows x 16) patch of the tile grid and
    // share A row-panels and Bt column-panels in that XCD's private L2.  The last group of an XCD may
    // hold fewer rows; it is walked densely as well (no interleaved empty blocks).
    const int x = b & 7;
    const int t = b >> 3;
    const int base = a.mt_count >> 3, extra = a.mt_count & 7;
    const int rows = base + (x < extra ? 1 : 0);
    if (t >= rows * a.nt_count) return false;
    const int start = x * base + (x < extra ? x : extra);
    const int full = rows / a.xcd_rows;
    const int per_group = a.xcd_rows * a.nt_count;
    int row;
    if (t < full * per_group) {
        const int g = t / per_group, r = t % per_group;
        row = g * a.xcd_rows + r % a.xcd_rows;
        nt = r / a.xcd_rows;
    } else {
        const int tt = t - full * per_group, rr = rows - full * a.xcd_rows;
        row = full * a.xcd_rows + tt % rr;
        nt = tt / rr;
    }
    mt = start + row;
    return true;
}

template <int MODE, int CONV, bool SPLIT>
__global__ void __launch_bounds__(256, 2) hk_dense_kernel(const HkArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];

    const int unit = (int)blockIdx.x + a.block_offset;
    const int split_y = SPLIT ? unit / a.unit_grid : 0;
    int mt_idx, nt_idx;
    if (!tile_of_block(a, SPLIT ? unit - split_y * a.unit_grid : unit, mt_idx, nt_idx)) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // wave-uniform by construction; said so to the compiler, so that everything derived from it -- the K rows this
    // wave stages, their global row pointers, the LDS destinations (M0) -- is scalar-unit arithmetic
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1;  // which 64 k-points
    const int wn = wave & 1;   // which 32 packed elements
    const int l15 = lane & 15;
    const int l4 = lane >> 4;

    const int64_t m0 = (int64_t)mt_idx * a.rows_per_tile;  // first k-point of the tile's output rows
    const int64_t n0 = (int64_t)nt_idx * TBK_BNP;

    d4 acc[4][2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p = 0; p < 2; ++p) acc[i][j][p] = (d4){0.0, 0.0, 0.0, 0.0};

    int s_begin = 0, n_stage = (int)(a.k2 / TBK_BK);
    if (SPLIT) {
        const int per = (n_stage + a.splits - 1) / a.splits;
        s_begin = min(split_y * per, n_stage);
        n_stage = min(s_begin + per, n_stage);
        if (a.p_tiles > 0) {
            const int per2 = (n_stage - s_begin + a.sub_splits - 1) / a.sub_splits;
            s_begin = min(s_begin + (int)blockIdx.y * per2, n_stage);
            n_stage = min(s_begin + per2, n_stage);
        }
    }

    // The K loop holds NO vector-unit instruction besides the MFMAs.  On gfx950 an f64 MFMA and ANY VALU instruction
    // of the SIMD's waves exclude each other (tools/pipe_probe.hip: times add exactly, also for 32-bit integer adds),
    // and a wave issues one VALU instruction per 8 cycles: the 44 address instructions hipcc's first version of this loop
    // carried per stage (64-bit pointer increments in VGPRs, v_readfirstlane for M0, ds_read base adds) cost the matrix
    // pipe ~350 of the 4096 cycles of a stage.  Now:
    //   * staging (global -> LDS, global_load_lds_dwordx4: one 1 KiB K row per wave-instruction, no staging VGPRs):
    //     wave w copies K rows w, w + 4, w + 8, w + 12 of both operands; the row pointers are wave-uniform (SGPR pairs,
    //     advanced by s_add), the lane's 16 bytes are ONE loop-invariant 32-bit VGPR offset, the LDS row base goes to
    //     M0 from scalar registers;
    //   * fragment reads: per-lane LDS base addresses of the two operands in both buffers (four loop-invariant VGPRs),
    //     every (K step, fragment) a compile-time ds_read offset (the loop is written out for buffer 0 and 1).
    const uint32_t lane_bytes = (uint32_t)lane * 16u;
    const int64_t ldgA = a.nk_pad * (int64_t)sizeof(double);                   // bytes per K row
    const int64_t ldgB = (int64_t)a.ncol_pad * 2 * (int64_t)sizeof(double);
    const char* rowA = reinterpret_cast<const char*>(a.A + (int64_t)mt_idx * a.a_tile_stride) + ((int64_t)s_begin * TBK_BK + wave) * ldgA;
    const char* rowB = reinterpret_cast<const char*>(a.Bt + (int64_t)mt_idx * a.b_tile_stride + n0 * 2) + ((int64_t)s_begin * TBK_BK + wave) * ldgB;
    auto issue_stage = [&](auto bufc) {  // the stage rowA / rowB point at -> buffer bufc; advances them by one stage
        constexpr int buf = decltype(bufc)::value;
        double* sA = smem + buf * STAGE_DOUBLES + wave * LDA;
        double* sB = smem + buf * STAGE_DOUBLES + TBK_BK * LDA + wave * LDB;
        // (the offset is re-defined inside the block that uses it: instruction selection works per basic block and only
        // takes the SGPR-base form of the load for `uniform pointer + zext(32-bit VGPR)` it can see whole; hoisted out
        // of the loop the zero-extension became a VGPR pair and every load cost a 64-bit VALU add)
        uint32_t off = lane_bytes;
        asm volatile("" : "+v"(off));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((gptr_t)(rowA + (int64_t)(4 * i) * ldgA + off), (lptr_t)(sA + 4 * i * LDA), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(rowB + (int64_t)(4 * i) * ldgB + off), (lptr_t)(sB + 4 * i * LDB), 16, 0, 0);
        }
        rowA += TBK_BK * ldgA;
        rowB += TBK_BK * ldgB;
    };
    // one opaque per-lane base per (buffer, K step, operand): the four fragments of a step are then two ds_read2_b64 with
    // immediate offsets (left to itself hipcc derives the bases of a stage from one register with a VALU add each)
    typedef __attribute__((address_space(3))) const double* lds_cptr;
    uint32_t fragA[2][TBK_BK / 4], fragB[2][TBK_BK / 4];  // LDS byte addresses
    {
        const uint32_t s0 = (uint32_t)(uintptr_t)(lptr_t)smem;
#pragma unroll
        for (int buf = 0; buf < 2; ++buf)
#pragma unroll
            for (int ks = 0; ks < TBK_BK / 4; ++ks) {
                fragA[buf][ks] = s0 + 8u * (uint32_t)(buf * STAGE_DOUBLES + wm * 64 + l15 + (ks * 4 + l4) * LDA);
                fragB[buf][ks] = s0 + 8u * (uint32_t)(buf * STAGE_DOUBLES + TBK_BK * LDA + wn * 64 + l15 + (ks * 4 + l4) * LDB);
                asm volatile("" : "+v"(fragA[buf][ks]), "+v"(fragB[buf][ks]));
            }
    }
    struct Frags {
        double fa[4], fb[2][2];
    };
    auto read_frags = [&](auto bufc, auto ksc) -> Frags {
        constexpr int buf = decltype(bufc)::value, ks = decltype(ksc)::value;
        const lds_cptr sA = (lds_cptr)(uintptr_t)fragA[buf][ks];
        const lds_cptr sB = (lds_cptr)(uintptr_t)fragB[buf][ks];
        Frags f;
#pragma unroll
        for (int i = 0; i < 4; ++i) f.fa[i] = sA[i * 16];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f.fb[j][0] = sB[j * 32];
            f.fb[j][1] = sB[j * 32 + 16];
        }
        return f;
    };
    auto mfma_step = [&](const Frags& f) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][j][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.fa[i], f.fb[j][0], acc[i][j][0], 0, 0, 0);
                acc[i][j][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.fa[i], f.fb[j][1], acc[i][j][1], 0, 0, 0);
            }
    };
    using buf0 = std::integral_constant<int, 0>;
    using buf1 = std::integral_constant<int, 1>;
    using ks0 = std::integral_constant<int, 0>;
    using ks1 = std::integral_constant<int, 1>;
    using ks2 = std::integral_constant<int, 2>;
    using ks3 = std::integral_constant<int, 3>;
    static_assert(TBK_BK == 16, "four K steps per stage are written out below");
    // One stage: the fragments of K step ks + 1 are requested before the MFMAs of step ks; the stage barrier sits in
    // front of the LAST step's MFMAs, and the first fragments of the NEXT stage are requested right behind it -- under
    // those 16 MFMAs instead of in front of an idle matrix pipe.
    auto stage = [&](auto bufc, auto nextc, Frags& cur, bool more) {
        Frags f1 = read_frags(bufc, ks1{});
        mfma_step(cur);
        Frags f2 = read_frags(bufc, ks2{});
        mfma_step(f1);
        Frags f3 = read_frags(bufc, ks3{});
        mfma_step(f2);
        __syncthreads();  // the next stage has landed (vmcnt(0)) and everyone has READ this one (its last fragments are in registers)
        if (more) cur = read_frags(nextc, ks0{});
        mfma_step(f3);
    };

    // two stages per trip (buffer 0, then 1), so that everything above is a compile-time offset; an odd stage behind it
    const int left = n_stage - s_begin;
    if (left > 0) issue_stage(buf0{});
    __syncthreads();  // drains the LDS-DMA queue (vmcnt) before the barrier
    Frags cur = read_frags(buf0{}, ks0{});
    for (int pair = 0; pair < (left >> 1); ++pair) {
        issue_stage(buf1{});  // lands during this stage's MFMAs
        stage(buf0{}, buf1{}, cur, true);
        const bool more = 2 * pair + 2 < left;
        if (more) issue_stage(buf0{});
        stage(buf1{}, buf0{}, cur, more);
    }
    if (left & 1) stage(buf0{}, buf1{}, cur, false);

    // ---- epilogue: scatter the packed tile into H[k][i][j] (and H[k][j][i]), or park the partial tile ----
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int e = (int)n0 + (wn * 2 + j) * 16 + l15;
        const int32_t ij = a.colmap[e];
        if (ij < 0) continue;
        const int oi = ij >> 16, oj = ij & 0xffff;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int local = wm * 64 + i * 16 + l4 + 4 * r;
                const int64_t kq = m0 + local;
                if (local >= a.rows_per_tile || kq >= a.nk) continue;
                if (SPLIT) {
                    double* part =
                        a.p_tiles > 0
                            ? a.P2 + ((((size_t)blockIdx.y * a.p_tiles + blockIdx.x) * TBK_BM + local) * TBK_BNP +
                                      (wn * 2 + j) * 16 + l15) * 2
                            : a.P + (((size_t)split_y * a.p_rows + kq) * a.ncol_pad + e) * 2;
                    *reinterpret_cast<d2*>(part) = (d2){acc[i][j][0][r], acc[i][j][1][r]};
                } else {
                    store_element<MODE, CONV>(a, kq, oi, oj, acc[i][j][0][r], acc[i][j][1][r]);
                }
            }
        }
    }
}

// Batches of at most 32 k-points (Z2Pack-style callers evaluate one k-point per call): the 128-row MFMA tile would
// spend most of its work on padding (127/128 for one k-point), so this is a plain matrix-vector product on the vector
// unit, bound by reading Bt ONCE (277 MB at N_orb = 64, N_R = 4096; 8.6 GB at N_orb = 512, N_R = 2048).
//
// Round 6: a streaming kernel built from independent WAVES.  A row of Bt is `2 ncol_pad` doubles = ncol_pad / 64 blocks of
// 128 doubles; a wave owns one block (16 B per lane: one global_load_dwordx4 reads 1 KiB of the row -- 8-byte loads reach
// 0.54 - 0.70 of the 16-byte rate on gfx950, MI355X_MICROARCH.md) and a slice of whole lattice vectors (two K rows each), keeps
// two batches of GEMV_U non-temporal loads in flight (row addresses in scalar registers, the lane as the offset), and needs
// nothing from the other waves of its workgroup: no barrier anywhere, consecutive waves take consecutive blocks of the same
// slice (a workgroup reads 4 KiB of every row).  The phase rows of the slice are made by the wave itself into a wave-private
// strip of LDS, AFTER its first two batches have been issued (same arithmetic as tbk_phase.hip; the loads of the lattice
// vectors share the counter of the stream, so the strip is made once, not in the loop); k.p models and slices too long for
// the strip get finished rows (A != nullptr) instead.  A lane holds two `re` or two `im` values of neighbouring elements (Bt
// is [tile][re | im][16]): one exchange with lane ^ 8 turns them into the (re, im) pairs of P[slice][k][e], which
// hk_finish_kernel / hk_finish_wide_kernel add in slice order.  gemv_plan() cuts the slices so that the launch is either ONE
// round of exactly two workgroups per CU (the dynamic LDS size is the occupancy limiter: a CU's share of the bytes is what
// bounds the kernel, so a CU with a third workgroup would finish 1.5x late) or many rounds of workgroups of ~1 MB.
constexpr int GEMV_U = 8;  // rows per batch of loads; two batches in flight

template <int NKV, bool INLINE_PHASES>
__global__ void __launch_bounds__(256) hk_gemv_kernel(const HkArgs a, int strip_doubles) {
    extern __shared__ __attribute__((aligned(16))) double s_rows[];  // [4 waves][strip_doubles]: phase rows [row][NKV] of the wave's slice
    // (32 k-points: 64 accumulators -- with batches of four loads the kernel needs 168 registers instead of 242 plus copies through
    // accumulation registers; the H(k) stage of 32 k-points at the headline shape is 158 us either way, DESIGN_LOG.md R6.11)
    constexpr int U = NKV >= 32 ? 4 : GEMV_U;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nblk = a.ncol_pad >> 6;
    const int task = (int)blockIdx.x * 4 + wave;
    const int sl = task / nblk, cb = task - sl * nblk;
    if (sl >= a.splits) return;  // (whole waves: nothing below synchronises the workgroup)
    const int64_t n_pairs = a.k2 >> 1;  // slices are cut between lattice vectors: a (cos, sin) pair of rows stays together
    const int64_t kk0 = 2 * (sl * n_pairs / a.splits);
    const int n_rows = (int)(2 * ((sl + 1) * n_pairs / a.splits) - kk0);
    const int64_t kbase = (int64_t)blockIdx.z * NKV;  // small models: groups of NKV k-points over blockIdx.z
    const int nk_here = (int)min((int64_t)NKV, a.nk - kbase);
    const int64_t ld2 = a.ncol_pad;  // row stride of Bt in 16-byte units
    // (wave-uniform row pointers + the lane as a 32-bit offset: the loads take their row address from scalar registers)
    const d2* rows = reinterpret_cast<const d2*>(a.Bt) + kk0 * ld2 + cb * 64;
    double* strip = s_rows + wave * strip_doubles;

    d2 buf0[U], buf1[U];
    double acc[NKV][2];
#pragma unroll
    for (int q = 0; q < NKV; ++q) acc[q][0] = acc[q][1] = 0.0;

    // a batch of U rows; rows past the slice's end re-read its last row (a cache hit) and are skipped by consume().  No
    // branch around a batch: behind a merge of control flow the compiler's wait counts assume the SHORTER queue, and the
    // first use of one batch would wait for the whole of the next
    auto fetch = [&](d2 (&buf)[U], int base) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = min(base + u, n_rows - 1);
            buf[u] = __builtin_nontemporal_load(rows + (int64_t)r * ld2 + lane);
        }
    };
    auto consume = [&](const d2 (&buf)[U], int row) {
        // uniform: phase row `row`, k-points 0 .. NKV-1
        const double* arow = INLINE_PHASES ? strip + row * NKV : a.A + (kk0 + row) * a.nk_pad + kbase;
        const int64_t lda = INLINE_PHASES ? NKV : a.nk_pad;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!INLINE_PHASES && row + u >= n_rows) break;  // (the strip holds zeros past the slice's end; finished rows end with A)
#pragma unroll
            for (int q = 0; q < NKV; ++q) {
                const double aq = arow[u * lda + q];
                acc[q][0] = fma(aq, buf[u][0], acc[q][0]);
                acc[q][1] = fma(aq, buf[u][1], acc[q][1]);
            }
        }
    };
    fetch(buf0, 0);
    fetch(buf1, U);
    if (INLINE_PHASES) {
        // n_rows / 2 lattice vectors x NKV k-points, cos and sin of each; zeros up to the next multiple of 16 rows
        for (int idx = lane; idx < ((n_rows + 15) >> 4) * 8 * NKV; idx += 64) {
            const int rr = idx / NKV, q = idx % NKV;
            const int64_t r = (kk0 >> 1) + rr;
            double sn = 0.0, cs = 0.0;
            if (q < nk_here && r < a.n_r && 2 * rr < n_rows) {
                double dot = 0.0;
                for (int d = 0; d < a.dim; ++d) dot = fma(hk_kcomp(a, kbase + q, d), (double)a.R[r * a.dim + d], dot);
                sincospi(2.0 * dot, &sn, &cs);
            }
            strip[(2 * rr) * NKV + q] = cs;
            strip[(2 * rr + 1) * NKV + q] = sn;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // one wave writes and reads the strip: its LDS operations execute in order
    }
    for (int row = 0; row < n_rows; row += 2 * U) {
        // (scheduling fences: left alone, the compiler sinks both batches of loads below both batches of FMAs, and the queue
        // runs empty once per trip)
        consume(buf0, row);
        __builtin_amdgcn_sched_barrier(0);
        fetch(buf0, row + 2 * U);
        __builtin_amdgcn_sched_barrier(0);
        consume(buf1, row + U);
        __builtin_amdgcn_sched_barrier(0);
        fetch(buf1, row + 3 * U);
        __builtin_amdgcn_sched_barrier(0);
    }
    // lanes l and l ^ 8 hold (re, re') and (im, im') of elements e0, e0 + 1: after the exchange lane l writes element e0,
    // lane l ^ 8 element e0 + 1.  (A predicate per k-point, not a `break`: leaving the unrolled loop early made the
    // accumulator index dynamic and the accumulators of the 16- and 32-point instantiations went to scratch memory.)
    const bool hi = (lane & 8) != 0;
    const int e = cb * 64 + (lane >> 4) * 16 + (lane & 7) * 2 + (hi ? 1 : 0);
#pragma unroll
    for (int q = 0; q < NKV; ++q) {
        const double got = __shfl_xor(hi ? acc[q][0] : acc[q][1], 8, 64);
        if (q < nk_here) {
            double* part = a.P + (((size_t)sl * a.p_rows + kbase + q) * a.ncol_pad + e) * 2;
            *reinterpret_cast<d2*>(part) = hi ? (d2){got, acc[q][1]} : (d2){acc[q][0], got};
        }
    }
}

// ONE k-point of a SMALL model (at most 4 blocks of 64 packed elements, i.e. up to 22 orbitals -- most tight-binding models a
// Z2Pack-style caller brings): the whole H(k) in one launch.  A workgroup owns a block, its four waves take a quarter of the K rows
// each (the loop of hk_gemv_kernel: 16-byte loads, the wave's phase rows in its LDS strip), meet once, and wave 0 adds the four
// partial sums in wave order and stores the elements -- hk_gemv_kernel + hk_finish_kernel were two dependent launches of ~3 us each
// around a ~2 us boundary for a call that takes 24 us end to end (silicon).
template <int MODE, int CONV>
__global__ void __launch_bounds__(256) hk_tiny_kernel(const HkArgs a, int strip_doubles) {
    extern __shared__ __attribute__((aligned(16))) double s_tiny[];  // [4][strip_doubles] phase rows, then [4][64] (re, im) partial sums
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cb = blockIdx.x;
    const int64_t n_pairs = a.k2 >> 1;
    const int64_t kk0 = 2 * (wave * n_pairs / 4);
    const int n_rows = (int)(2 * ((wave + 1) * n_pairs / 4) - kk0);  // k2 is a multiple of 16: at least four rows per wave
    const int64_t ld2 = a.ncol_pad;
    const d2* rows = reinterpret_cast<const d2*>(a.Bt) + kk0 * ld2 + cb * 64;
    double* strip = s_tiny + wave * strip_doubles;
    d2* part = reinterpret_cast<d2*>(s_tiny + 4 * strip_doubles);  // [4][64]
    d2 buf0[GEMV_U], buf1[GEMV_U];
    auto fetch = [&](d2 (&buf)[GEMV_U], int base) {
#pragma unroll
        for (int u = 0; u < GEMV_U; ++u) buf[u] = rows[(int64_t)min(base + u, n_rows - 1) * ld2 + lane];
    };
    fetch(buf0, 0);
    fetch(buf1, GEMV_U);
    for (int rr = lane; rr < ((n_rows + 15) >> 4) * 8; rr += 64) {
        const int64_t r = (kk0 >> 1) + rr;
        double sn = 0.0, cs = 0.0;
        if (r < a.n_r && 2 * rr < n_rows) {
            double dot = 0.0;
            for (int d = 0; d < a.dim; ++d) dot = fma(hk_kcomp(a, 0, d), (double)a.R[r * a.dim + d], dot);
            sincospi(2.0 * dot, &sn, &cs);
        }
        strip[2 * rr] = cs;
        strip[2 * rr + 1] = sn;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the wave's own strip: its LDS operations execute in order)
    double acc0 = 0.0, acc1 = 0.0;
    auto consume = [&](const d2 (&buf)[GEMV_U], int row) {
#pragma unroll
        for (int u = 0; u < GEMV_U; ++u) {
            const double aq = strip[row + u];  // (zero past the slice's end)
            acc0 = fma(aq, buf[u][0], acc0);
            acc1 = fma(aq, buf[u][1], acc1);
        }
    };
    for (int row = 0; row < n_rows; row += 2 * GEMV_U) {
        consume(buf0, row);
        __builtin_amdgcn_sched_barrier(0);
        fetch(buf0, row + 2 * GEMV_U);
        __builtin_amdgcn_sched_barrier(0);
        consume(buf1, row + GEMV_U);
        __builtin_amdgcn_sched_barrier(0);
        fetch(buf1, row + 3 * GEMV_U);
        __builtin_amdgcn_sched_barrier(0);
    }
    // (re, re') / (im, im') of elements e0, e0 + 1 in lanes l and l ^ 8 -> (re, im) pairs, as in hk_gemv_kernel
    const bool hi = (lane & 8) != 0;
    const int el = (lane >> 4) * 16 + (lane & 7) * 2 + (hi ? 1 : 0);
    const double got = __shfl_xor(hi ? acc0 : acc1, 8, 64);
    part[wave * 64 + el] = hi ? (d2){got, acc1} : (d2){acc0, got};
    __syncthreads();
    if (wave != 0) return;
    const int e = cb * 64 + lane;
    const int32_t ij = a.colmap[e];
    if (ij < 0) return;
    d2 sum = part[lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        sum[0] += part[w * 64 + lane][0];
        sum[1] += part[w * 64 + lane][1];
    }
    store_element<MODE, CONV>(a, 0, ij >> 16, ij & 0xffff, sum[0], sum[1]);
}

// split-K finish: one thread per (k-point, packed element) adds the partial tiles in split order and stores the
// element like the epilogue above.
template <int MODE, int CONV>
__global__ void __launch_bounds__(256) hk_finish_kernel(const HkArgs a) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int e = (int)(idx % a.ncol_pad);
    const int64_t kq = idx / a.ncol_pad;
    if (kq >= a.nk) return;
    const int32_t ij = a.colmap[e];
    if (ij < 0) return;
    double re = 0.0, im = 0.0;
    for (int sp = 0; sp < a.splits; ++sp) {
        const d2 v = *reinterpret_cast<const d2*>(a.P + (((size_t)sp * a.p_rows + kq) * a.ncol_pad + e) * 2);
        re += v[0];
        im += v[1];
    }
    store_element<MODE, CONV>(a, kq, ij >> 16, ij & 0xffff, re, im);
}

// The same for many splits and few k-points (the matrix-vector path: ~100 K slices, one k-point): 16 threads per
// element, thread j adds splits j, j + 16, ... and the 16 partial sums are combined in a fixed tree -- the one-thread
// loop was a chain of ~100 dependent loads, 31 us of a 117 us single-k hamilton() call.
template <int MODE, int CONV>
__global__ void __launch_bounds__(256) hk_finish_wide_kernel(const HkArgs a) {
    __shared__ double part[4][16][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int el = lane & 15;
    const int j = wave * 4 + (lane >> 4);  // 0 .. 15
    const int64_t idx = (int64_t)blockIdx.x * 16 + el;
    const int e = (int)(idx % a.ncol_pad);
    const int64_t kq = idx / a.ncol_pad;
    double re = 0.0, im = 0.0;
    if (kq < a.nk)
        for (int sp = j; sp < a.splits; sp += 16) {
            const d2 v = *reinterpret_cast<const d2*>(a.P + (((size_t)sp * a.p_rows + kq) * a.ncol_pad + e) * 2);
            re += v[0];
            im += v[1];
        }
    re += __shfl_xor(re, 16, 64);
    im += __shfl_xor(im, 16, 64);
    re += __shfl_xor(re, 32, 64);
    im += __shfl_xor(im, 32, 64);
    if (lane < 16) {
        part[wave][el][0] = re;
        part[wave][el][1] = im;
    }
    __syncthreads();
    if (threadIdx.x < 16 && kq < a.nk) {
        const int32_t ij = a.colmap[e];
        if (ij < 0) return;
        const double sr = (part[0][el][0] + part[1][el][0]) + (part[2][el][0] + part[3][el][0]);
        const double si = (part[0][el][1] + part[1][el][1]) + (part[2][el][1] + part[3][el][1]);
        store_element<MODE, CONV>(a, kq, ij >> 16, ij & 0xffff, sr, si);
    }
}

// Tail launches: one workgroup per tail unit adds its sub-split partial tiles in order and either finishes the
// element (the launch was not split otherwise) or hands the sum to the unit's slot of P for hk_finish_kernel.
template <int MODE, int CONV>
__global__ void __launch_bounds__(256) hk_finish_tiles_kernel(const HkArgs a) {
    const int unit = (int)blockIdx.x + a.block_offset;
    const int split_y = unit / a.unit_grid;
    int mt_idx, nt_idx;
    if (!tile_of_block(a, unit - split_y * a.unit_grid, mt_idx, nt_idx)) return;
    const int el = threadIdx.x & (TBK_BNP - 1);
    const int e = nt_idx * TBK_BNP + el;
    const int32_t ij = a.colmap[e];
    if (ij < 0) return;
    for (int local = threadIdx.x / TBK_BNP; local < a.rows_per_tile; local += 256 / TBK_BNP) {
        const int64_t kq = (int64_t)mt_idx * a.rows_per_tile + local;
        if (kq >= a.nk) break;
        double re = 0.0, im = 0.0;
        for (int sp = 0; sp < a.sub_splits; ++sp) {
            const d2 v = *reinterpret_cast<const d2*>(
                a.P2 + ((((size_t)sp * a.p_tiles + blockIdx.x) * TBK_BM + local) * TBK_BNP + el) * 2);
            re += v[0];
            im += v[1];
        }
        if (a.splits > 1)
            *reinterpret_cast<d2*>(a.P + (((size_t)split_y * a.p_rows + kq) * a.ncol_pad + e) * 2) = (d2){re, im};
        else
            store_element<MODE, CONV>(a, kq, ij >> 16, ij & 0xffff, re, im);
    }
}

template <int MODE, int CONV>
hipError_t launch_gemv(tbk_model* m, const HkArgs& a, size_t lds, hipStream_t s) {
    const int tasks = (a.ncol_pad >> 6) * a.splits;  // waves: (block of 64 packed elements) x (K slice)
    const dim3 grid((unsigned)((tasks + 3) / 4), 1, (unsigned)((a.nk + 31) / 32));
    static std::atomic<bool> raised[2][6][TBK_MAX_DEVICES] = {};
#define TBK_GEMV(N, SLOT)                                                                                               \
    do {                                                                                                                \
        if (a.A == nullptr) {                                                                                           \
            hipError_t e1 = tbk_raise_lds_limit(reinterpret_cast<const void*>(&hk_gemv_kernel<N, true>), 160 * 1024, raised[0][SLOT]); \
            if (e1 != hipSuccess) return e1;                                                                            \
            hipLaunchKernelGGL((hk_gemv_kernel<N, true>), grid, dim3(256), lds, s, a, (int)(lds / 32));                                  \
        } else {                                                                                                        \
            hipError_t e1 = tbk_raise_lds_limit(reinterpret_cast<const void*>(&hk_gemv_kernel<N, false>), 160 * 1024, raised[1][SLOT]); \
            if (e1 != hipSuccess) return e1;                                                                            \
            hipLaunchKernelGGL((hk_gemv_kernel<N, false>), grid, dim3(256), lds, s, a, (int)(lds / 32));                                 \
        }                                                                                                               \
    } while (0)
    if (a.nk <= 1)
        TBK_GEMV(1, 0);
    else if (a.nk <= 2)
        TBK_GEMV(2, 1);
    else if (a.nk <= 4)
        TBK_GEMV(4, 2);
    else if (a.nk <= 8)
        TBK_GEMV(8, 3);
    else if (a.nk <= 16)
        TBK_GEMV(16, 4);
    else
        TBK_GEMV(32, 5);
#undef TBK_GEMV
    (void)m;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int64_t threads = a.nk * a.ncol_pad;
    if (a.splits >= 32)
        hipLaunchKernelGGL((hk_finish_wide_kernel<MODE, CONV>), dim3((unsigned)((threads + 15) / 16)), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((hk_finish_kernel<MODE, CONV>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

// Launches of a few rounds end on a ragged one (4096 k-points at N_orb = 64: 1056 tiles = 2.06 rounds of 512
// workgroup slots take the time of 3; measured 1.26 us per k-point against 0.99 for long launches), split launches
// likewise (1024 k-points: 264 tiles x 5 splits = 2.6 rounds).  The units (block x split) of the last, partly filled
// round are therefore split along K once more among all slots: whole rounds in one ordinary launch, the tail as a
// (tail units) x (sub-splits) launch and a per-unit finish in fixed order.  Above TAIL_MAX_ROUNDS the ragged round
// stops mattering (no difference measured at 16.5 vs 15.98 rounds: the chip is power-limited there).
constexpr int TAIL_MAX_ROUNDS = 12;

static int tail_split_rounds() {
    static const int rounds = [] {
        const char* v = tbk_exp_env("TBK_HK_TAIL_SPLIT");  // measurements only: "0" switches it off, N sets the limit
        return v ? atoi(v) : TAIL_MAX_ROUNDS;
    }();
    return rounds;
}

// `a.splits` >= 1 with a.P / a.p_rows set by the caller when > 1; `grid` blocks of the tile walk.
template <int MODE, int CONV>
int launch(tbk_model* m, const HkArgs& a0, int grid) {
    hipStream_t s = m->stream;
    // 73,728 B: above the 64 KiB default cap.  m->hk_lds_floor (the pipeline of the two-stage sizes, when H(k) of the next chunk
    // runs beside a reduction): ask for more than half a CU's LDS, so that ONE contraction workgroup shares a CU with one
    // reduction workgroup instead of two of a kind
    const size_t lds = std::max<size_t>(2 * STAGE_DOUBLES * sizeof(double), m->hk_lds_floor);
    static std::atomic<bool> raised[TBK_MAX_DEVICES] = {};
    static std::atomic<bool> raised_split[TBK_MAX_DEVICES] = {};
    TBK_HIP(tbk_raise_lds_limit(reinterpret_cast<const void*>(&hk_dense_kernel<MODE, CONV, false>), 160 * 1024, raised));
    TBK_HIP(tbk_raise_lds_limit(reinterpret_cast<const void*>(&hk_dense_kernel<MODE, CONV, true>), 160 * 1024, raised_split));
    HkArgs a = a0;
    a.unit_grid = grid;
    const int slots = 2 * m->n_cu;  // __launch_bounds__(256, 2) and 72 KiB of LDS: two workgroups per CU
    const int n_stage = (int)(a.k2 / TBK_BK);
    const int units = grid * a.splits;
    const int full = units / slots * slots, tail = units - full;
    const int unit_stages = (n_stage + a.splits - 1) / a.splits;
    int sub = 1;
    if (full > 0 && tail > 0 && units < tail_split_rounds() * slots && unit_stages >= 8) {
        // Time of u equal workgroups in rounds of two per CU: one alone on its CU runs at 1.06 ms per tile against
        // 1.97 ms for each of two sharing it, so up to n_cu workgroups cost 0.54.  Sub-splitting the tail s ways
        // costs rounds(tail * s) / s plus the partial tiles (~90 ns per tile and sub-split) and three launches.
        auto rounds = [&](int u) { return u <= slots / 2 ? 0.54 : (double)((u + slots - 1) / slots); };
        const double round_ms = 1.97 * unit_stages / 512.0;
        const double plain = rounds(tail) * round_ms;
        double best = std::min(0.9 * plain, plain - 0.03);
        for (int sp = 2; sp <= std::min(64, unit_stages / 4); ++sp) {
            const double cost = rounds(tail * sp) / sp * round_ms + sp * tail * 9e-5 + 0.01;
            if (cost < best) {
                best = cost;
                sub = sp;
            }
        }
    }
    const size_t base_bytes = a.splits > 1 ? (size_t)a.splits * a.p_rows * a.ncol_pad * 2 * sizeof(double) : 0;
    if (sub >= 2) {
        TBK_CHECK(m->ws_part.reserve(base_bytes + (size_t)sub * tail * TBK_BM * TBK_BNP * 2 * sizeof(double)));
        a.P = a.splits > 1 ? m->ws_part.as<double>() : nullptr;
        if (a.splits > 1)
            hipLaunchKernelGGL((hk_dense_kernel<MODE, CONV, true>), dim3(full), dim3(256), lds, s, a);
        else
            hipLaunchKernelGGL((hk_dense_kernel<MODE, CONV, false>), dim3(full), dim3(256), lds, s, a);
        TBK_HIP(hipGetLastError());
        HkArgs t = a;
        t.block_offset = full;
        t.p_tiles = tail;
        t.sub_splits = sub;
        t.P2 = reinterpret_cast<double*>(m->ws_part.as<char>() + base_bytes);
        hipLaunchKernelGGL((hk_dense_kernel<MODE, CONV, true>), dim3(tail, sub), dim3(256), lds, s, t);
        TBK_HIP(hipGetLastError());
        hipLaunchKernelGGL((hk_finish_tiles_kernel<MODE, CONV>), dim3(tail), dim3(256), 0, s, t);
        TBK_HIP(hipGetLastError());
    } else if (a.splits > 1) {
        hipLaunchKernelGGL((hk_dense_kernel<MODE, CONV, true>), dim3(units), dim3(256), lds, s, a);
        TBK_HIP(hipGetLastError());
    } else {
        hipLaunchKernelGGL((hk_dense_kernel<MODE, CONV, false>), dim3(grid), dim3(256), lds, s, a);
        TBK_HIP(hipGetLastError());
    }
    if (a.splits > 1) {
        const int64_t threads = a.nk * a.ncol_pad;
        hipLaunchKernelGGL((hk_finish_kernel<MODE, CONV>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, a);
        TBK_HIP(hipGetLastError());
    }
    return TBK_OK;
}

// K slices of the matrix-vector path: (blocks of 64 packed elements) x (K slices) independent waves, four to a workgroup.
// The kernel is bound by the bytes a CU pulls, so the workgroups must come out EVEN over the CUs: an operand of up to ~1.5 MB
// per workgroup slot goes in ONE round of at most two workgroups per CU (`*lds_out` = half a CU's LDS keeps a third one away);
// a bigger one in workgroups of ~1 MB (256 rows per wave), many rounds, the ragged end a few per cent.  The partial sums
// (one row of P per slice and k-point) stay below 256 MB.
int gemv_nkv(int64_t nk) {  // the kernel instantiation's k-points per wave
    int nkv = 1;
    while (nkv < std::min<int64_t>(nk, 32)) nkv *= 2;
    return nkv;
}
int64_t gemv_strip_rows(const tbk_model* m, int64_t slices) {  // rows of the longest slice, rounded up to whole trips of the loop
    const int64_t n_pairs = m->k2 / 2;
    return (2 * ((n_pairs + slices - 1) / slices) + 15) / 16 * 16;
}

void gemv_plan(const tbk_model* m, int64_t nk, int* slices_out, size_t* lds_out) {
    const int nblk = m->ncol_pad / 64;
    const int64_t n_pairs = m->k2 / 2;
    const size_t per_split = (size_t)nk * m->ncol_pad * 2 * sizeof(double);
    const int64_t cap = std::max<int64_t>(1, (int64_t)((size_t(256) << 20) / per_split));
    // (one-k hamilton at N_orb = 64, N_R = 4096 with 1 / 2 / 3 / 4 / 5 workgroups per CU: 69.4 / 65.2 / 66.5 / 67.9 / 69.2 us)
    const int64_t wave_slots = (int64_t)m->n_cu * 8;  // two workgroups per CU
    int64_t slices = wave_slots / nblk;               // one round: nblk * slices <= slots
    size_t lds = 80 * 1024;
    const int64_t rows_one_round = slices > 0 ? (m->k2 + slices - 1) / slices : m->k2 + 1;
    if (slices < 1 || nblk * slices * 10 < wave_slots * 9 || rows_one_round > 384 || slices > cap) {
        // many rounds: workgroups of ~1 MB, at least eight per CU while a slice keeps 16 rows
        slices = std::max<int64_t>(m->k2 / 256, std::min<int64_t>(((int64_t)m->n_cu * 32 + nblk - 1) / nblk, n_pairs / 8));
        slices = std::max<int64_t>(1, std::min(cap, slices));
        lds = 16 * 1024;
    }
    slices = std::max<int64_t>(1, std::min(slices, n_pairs / 8));  // at least 16 rows per slice
    if (nblk * slices * 2 <= wave_slots) lds = 16 * 1024;  // (a small model: nothing to balance)
    if (lds < 64 * 1024) {
        // no occupancy limiter: as much LDS as the phase strips of four waves take (up to 64 KiB), so that e.g. groups of 32
        // k-points of a small model still make their own rows (the 1000-point silicon mesh: no phase_rows_kernel launch)
        const size_t need = (size_t)4 * gemv_strip_rows(m, slices) * gemv_nkv(nk) * sizeof(double);
        if (need <= size_t(64) * 1024) lds = std::max(lds, need);
    }
    *slices_out = (int)slices;
    *lds_out = lds;
}

}  // namespace

// True when tbk_launch_hk_dense will take the matrix-vector path AND can make its phase rows itself: the caller then
// skips tbk_launch_phase and passes d_A = nullptr.
bool tbk_hk_inline_phases(const tbk_model* m, int64_t nk) {
    if (!tbk_hk_gemv_path(m, nk) || m->kdotp || m->d_R == nullptr) return false;
    // the phase rows of a wave's slice for its (up to 32) k-points must fit its strip of the workgroup's LDS
    int slices;
    size_t lds;
    gemv_plan(m, nk, &slices, &lds);
    return gemv_strip_rows(m, slices) * gemv_nkv(nk) <= (int64_t)(lds / 32);
}

// The matrix-vector path: up to 32 k-points of any model, and up to 4096 k-points (in groups of 32) of a SMALL model --
// at most 22 orbitals and fewer than 128 lattice vectors, i.e. a handful of MFMA tiles walking a dozen K stages one
// load latency at a time (the 1000-point silicon grid: 31 us in the tile kernel, ~10 us here).
bool tbk_hk_gemv_path(const tbk_model* m, int64_t nk) {
    if (nk < 1 || m->k2 <= 0 || m->sparse) return false;
    if (nk <= 32) return true;
    return nk <= 4096 && m->ncol_pad <= 256 && m->k2 < 16 * TBK_BK;
}

int tbk_launch_hk_dense(tbk_model* m, const double* d_A, int64_t nk, int64_t nk_pad, int mode,
                        int convention, const double* d_k, const double* d_pos, double* d_H) {
    if (nk == 0) return TBK_OK;
    HkArgs a;
    a.A = d_A;
    a.Bt = m->d_B;
    a.colmap = m->d_colmap;
    a.kpts = d_k;
    a.R = m->d_R;
    a.n_r = m->n_r;
    a.pos = d_pos;
    a.H = d_H;
    a.k2 = m->k2;
    a.nk = nk;
    a.nk_pad = nk_pad;
    a.ncol_pad = m->ncol_pad;
    a.n_orb = m->n_orb;
    a.dim = m->dim;
    a.mt_count = (int)((nk + TBK_BM - 1) / TBK_BM);  // nk_pad is only the row stride of A
    a.nt_count = m->ncol_pad / TBK_BNP;
    int grid;
    if (a.mt_count >= 32) {  // below that a plain round-robin over the XCDs balances better
        a.xcd_rows = 4;
        if (const char* v = tbk_exp_env("TBK_HK_XCD_ROWS")) a.xcd_rows = std::max(1, atoi(v));  // measurements only
        const int max_rows = (a.mt_count + 7) / 8;
        grid = max_rows * a.nt_count * 8;
    } else {
        a.xcd_rows = 0;
        grid = a.mt_count * a.nt_count;
    }
    a.P = nullptr;
    a.splits = 1;
    a.p_rows = 0;
    a.a_tile_stride = TBK_BM;
    a.b_tile_stride = 0;
    a.rows_per_tile = TBK_BM;
    a.block_offset = 0;
    a.p_tiles = 0;
    a.sub_splits = 1;
    a.P2 = nullptr;
    a.unit_grid = 1;
    a.k_inline = 0;
    a.pos_raw = nullptr;
    for (int d = 0; d < TBK_MAX_DIM; ++d) a.k_val[d] = 0.0;
    if (tbk_hk_gemv_path(m, nk)) {
        int slices;
        size_t lds;
        gemv_plan(m, nk, &slices, &lds);
        const size_t per_split = (size_t)nk * a.ncol_pad * 2 * sizeof(double);
        if (m->h_k_inline != nullptr && nk == 1 && d_A == nullptr && tbk_hk_inline_phases(m, 1)) {
            a.k_inline = 1;  // (tbk_hamilton / tbk_eigenval on host buffers, one k-point: no upload of k)
            for (int d = 0; d < m->dim; ++d) a.k_val[d] = m->h_k_inline[d];
            a.pos_raw = m->d_pos_inline;  // convention 1: the raw positions (tbk_hamilton keeps them on the device)
        }
        TBK_ARG(d_A != nullptr || (tbk_hk_inline_phases(m, nk) && (d_k != nullptr || a.k_inline)), "phase rows missing");
        TBK_ARG(convention != 1 || mode == HK_TRI || d_pos != nullptr || (a.k_inline && a.pos_raw != nullptr), "convention 1 needs the orbital phases");
        if (nk == 1 && d_A == nullptr && a.ncol_pad <= 256 && m->k2 <= 4096) {  // (<= 36 KiB of LDS)
            // one k-point of a small model: the whole H(k) in ONE launch (hk_tiny_kernel)
            const int strip = (int)(((m->k2 / 2 + 3) / 4 * 2 + 15) / 16 * 16);  // rows of the longest quarter, whole trips of the loop
            const size_t lds_tiny = ((size_t)4 * strip + 4 * 64 * 2) * sizeof(double);
            const dim3 grid_tiny((unsigned)(a.ncol_pad / 64));
            StageTimer t(m, TBK_T_HK);
            if (mode == HK_TRI)
                hipLaunchKernelGGL((hk_tiny_kernel<HK_TRI, 2>), grid_tiny, dim3(256), lds_tiny, m->stream, a, strip);
            else if (convention == 1)
                hipLaunchKernelGGL((hk_tiny_kernel<HK_FULL, 1>), grid_tiny, dim3(256), lds_tiny, m->stream, a, strip);
            else
                hipLaunchKernelGGL((hk_tiny_kernel<HK_FULL, 2>), grid_tiny, dim3(256), lds_tiny, m->stream, a, strip);
            TBK_HIP(hipGetLastError());
            return TBK_OK;
        }
        TBK_CHECK(m->ws_part.reserve(per_split * slices));
        a.P = m->ws_part.as<double>();
        a.splits = slices;
        a.p_rows = nk;
        StageTimer t(m, TBK_T_HK);
        if (mode == HK_TRI) {
            TBK_HIP((launch_gemv<HK_TRI, 2>(m, a, lds, m->stream)));
        } else if (convention == 1) {
            TBK_HIP((launch_gemv<HK_FULL, 1>(m, a, lds, m->stream)));
        } else {
            TBK_HIP((launch_gemv<HK_FULL, 2>(m, a, lds, m->stream)));
        }
        return TBK_OK;
    }
    // Small k batches: a workgroup's K loop is a serial chain (1.06 ms at N_R = 4096 whatever the batch), and fewer
    // tiles than workgroup slots leave CUs idle -- split K into ~1280 units (block x split) so that the launch fills
    // the chip for two to three rounds; launch() cuts the ragged last round once more.  Partial tiles go to a
    // workspace, hk_finish_kernel adds them in fixed order.  Measured at N_orb = 64, N_R = 4096: one k-point
    // 1056 -> 163 us (before the matrix-vector path), 1000 k-points 2137 -> 1264 us.  (Just enough splits for ONE
    // full round plus a sub-split tail was slower up to 500 k-points: three more launches on a 0.2 ms kernel.)
    const int n_stage = (int)(m->k2 / TBK_BK);
    const int tiles = a.mt_count * a.nt_count;
    if (tiles < 2 * m->n_cu && n_stage >= 16) {
        int splits = std::min((1280 + tiles / 2) / tiles, n_stage / 4);
        const size_t per_split = (size_t)nk_pad * a.ncol_pad * 2 * sizeof(double);
        splits = (int)std::min<size_t>((size_t)splits, (size_t(512) << 20) / per_split);
        if (splits > 1) {
            TBK_CHECK(m->ws_part.reserve(per_split * splits));
            a.P = m->ws_part.as<double>();
            a.splits = splits;
            a.p_rows = nk_pad;
        }
    }
    StageTimer t(m, TBK_T_HK);
    if (mode == HK_TRI) {
        TBK_CHECK((launch<HK_TRI, 2>(m, a, grid)));
    } else if (convention == 1) {
        TBK_CHECK((launch<HK_FULL, 1>(m, a, grid)));
    } else {
        TBK_CHECK((launch<HK_FULL, 2>(m, a, grid)));
    }
    return TBK_OK;
}

// H(k) of n_lines mesh lines of line_len (<= 128) k-points each in ONE launch: line t uses the operand at
// m->d_B + t * b_stride (a second-level folded model per line) and all lines share the phase rows d_A[K][128]
// (the k-points of a line differ only in the remaining component, which is the same sequence on every line).
// TRI mode, convention 2 (the eigenvalue path).
int tbk_launch_hk_dense_lines(tbk_model* m, const double* d_A, int64_t n_lines, int line_len, int64_t b_stride,
                              double* d_H) {
    if (n_lines == 0) return TBK_OK;
    TBK_ARG(line_len >= 1 && line_len <= TBK_BM, "a mesh line must fit one k tile");
    HkArgs a;
    a.A = d_A;
    a.Bt = m->d_B;
    a.colmap = m->d_colmap;
    a.kpts = nullptr;
    a.R = m->d_R;
    a.n_r = m->n_r;
    a.pos = nullptr;
    a.H = d_H;
    a.k2 = m->k2;
    a.nk = n_lines * line_len;
    a.nk_pad = TBK_BM;
    a.ncol_pad = m->ncol_pad;
    a.n_orb = m->n_orb;
    a.dim = m->dim;
    a.mt_count = (int)n_lines;
    a.nt_count = m->ncol_pad / TBK_BNP;
    a.P = nullptr;
    a.splits = 1;
    a.p_rows = 0;
    a.a_tile_stride = 0;
    a.b_tile_stride = b_stride;
    a.rows_per_tile = line_len;
    a.block_offset = 0;
    a.p_tiles = 0;
    a.sub_splits = 1;
    a.P2 = nullptr;
    a.unit_grid = 1;
    int grid;
    if (a.mt_count >= 32) {
        a.xcd_rows = 4;
        grid = ((a.mt_count + 7) / 8) * a.nt_count * 8;
    } else {
        a.xcd_rows = 0;
        grid = a.mt_count * a.nt_count;
    }
    StageTimer t(m, TBK_T_HK);
    TBK_CHECK((launch<HK_TRI, 2>(m, a, grid)));
    return TBK_OK;
}
