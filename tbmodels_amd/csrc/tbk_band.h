// tbk_band.h -- what the three translation units of the two-stage reduction share (round 6: tbk_eig_band.hip was one file of
// 3700 lines / 31 kernels):
//   tbk_eig_band.hip        stage 1 in one workgroup per matrix (band_reduce_kernel), the size policy, tbk_launch_band_reduce
//   tbk_eig_band_chase.hip  stage 2 as kernels of its own (band_chase4 / 4g / 4w), tbk_launch_band_chase
//   tbk_eig_band_xl.hip     stage 1 as a chain of launches (band_xl_*: above 1024 orbitals, and calls of a few matrices)
//   tbk_band_chase.h        the body of stage 2 (chase4_body): inlined into the fused stage-1 kernel and into the stage-2 kernels
// Device helpers live in an anonymous namespace (every unit compiles its own copy); the host-side policy functions are defined
// once, in tbk_eig_band.hip.
#pragma once

#include <algorithm>
#include <cstdlib>

#include "tbk_dpp.h"
#include "tbk_internal.h"

namespace {

typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));

// Phase clock for tools/band_phase_clock.hip (compiled out of the library): cycles per phase, accumulated by
// thread 0 of workgroup 0.
#ifdef TBK_PHASE_CLOCK
__device__ unsigned long long tbk_band_clock[32];
// (accumulated in registers and written once at the end: a global update per point would wait for every load in flight)
#define TBK_CLK(k)                                              \
    do {                                                        \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        clk_acc_[k] += now_ - clk_prev_;                        \
        clk_prev_ = now_;                                       \
    } while (0)
#else
#define TBK_CLK(k)
#endif

// Non-temporal tile loads / stores (so that the streaming tiles do not push the re-read [V | W] blocks out of L2) were
// measured: 4.51 -> 4.78 us per matrix at 256 orbitals, 31.5 -> 31.0 at 512 -- nothing either way; off.
#ifndef TBK_TILE_NT
#define TBK_TILE_NT 0
#endif

// Ablation switches for tools/band_ablate.sh (TIMING ONLY: each makes the results wrong; never defined in the library build).
// They price the levers of the tile pass before anything is built on them (round 4):
//   TBK_ABLATE_OPERANDS  every visit reads the partner's [V | W] / Vn operand blocks of ONE fixed block (always cached)
//   TBK_ABLATE_BARRIER   no workgroup barrier per step of the pass
//   TBK_ABLATE_STORES    the updated tiles are never stored (an upper bound for ANY scheme that defers the update)
//   TBK_ABLATE_WIN_IO    band_chase4w_kernel without the global loads / stores of the columns that enter and leave the window
//   TBK_ABLATE_WIN_FORCE the 32-slot window kernel from 257 orbitals on at every call size (against the plain LDS form at 512)
//   TBK_ABLATE_STORES_ALT  ... stored on every second panel only: what "the rank-16 update every second panel" saves in
//                        stores, before any of its costs (a K = 32 update, the corrections of the products)
constexpr int PB = 8;    // panel height = band half-width
constexpr int TS = 16;   // MFMA tile edge

// Round 5: the 8 x 8 Gram-type sums of a panel -- P^H P of the panel QR, V^H V of the T factor, V^H X of the W phase -- on the
// MATRIX pipe: the rows go through a wave-private LDS plane into operand order ([Re | Im] as 16 real columns, 16 MFMAs per 64
// rows), the waves' 16 x 16 partial products meet ONCE, and the panel QR takes ALL its reflectors from that one Gram matrix
// (tools/two_stage_model.py: panel_qr_gram).  Before: one round of vector products, 64-bit DPP wave sums and
// a workgroup barrier PER REFLECTOR (8 per panel) plus two more for T and W.  TBK_PANEL_GRAM=0 builds the round-4 form (A/B).
// One row per thread only (up to 256 orbitals, and every call of a few matrices): with two rows per thread the recurrence's
// tracked block beside both rows did not fit the register file (60 - 340 B of scratch in every arrangement tried).
#ifndef TBK_PANEL_GRAM
#define TBK_PANEL_GRAM 1
#endif
constexpr int GP = 17;   // pitch (doubles) of a wave's transposition plane [64 rows][16 values]
// a column whose remaining norm^2 (a difference of Gram sums) has cancelled below this fraction of its full norm^2 ends the
// round: the rows apply the reflectors found so far and a fresh Gram matrix is formed (errors ~ eps sqrt(1 / fraction))
constexpr double GRAM_THRESH = 1.0 / 64.0;

// bytes of the X (+ V) area at the head of the dynamic LDS: [npad][8] complex once or twice, and at least the waves'
// transposition planes, which live there while X and V are dead
__host__ __device__ inline size_t band_xv_bytes(int npad, bool vn_lds, int nw, int rows) {
    const size_t xv = (size_t)npad * PB * 16 * (vn_lds ? 2 : 1);
    const size_t planes = (TBK_PANEL_GRAM && rows == 1) ? (size_t)nw * 64 * GP * 8 : 0;
    return xv > planes ? xv : planes;
}

__device__ __forceinline__ d2 cmul(d2 a, d2 b) { return (d2){a[0] * b[0] - a[1] * b[1], a[0] * b[1] + a[1] * b[0]}; }
__device__ __forceinline__ d2 cmulc(d2 a, d2 b) { return (d2){a[0] * b[0] + a[1] * b[1], a[1] * b[0] - a[0] * b[1]}; }  // a conj(b)
__device__ __forceinline__ d2 conjd(d2 a) { return (d2){a[0], -a[1]}; }
// acc += a b      /   acc += a conj(b)   /   acc -= a conj(b)
__device__ __forceinline__ void cfma(d2& acc, d2 a, d2 b) {
    acc[0] = fma(a[0], b[0], acc[0]);
    acc[1] = fma(a[0], b[1], acc[1]);
    acc[0] = fma(-a[1], b[1], acc[0]);
    acc[1] = fma(a[1], b[0], acc[1]);
}
__device__ __forceinline__ void cfmac(d2& acc, d2 a, d2 b) {
    acc[0] = fma(a[0], b[0], acc[0]);
    acc[1] = fma(a[1], b[0], acc[1]);
    acc[0] = fma(a[1], b[1], acc[0]);
    acc[1] = fma(-a[0], b[1], acc[1]);
}
__device__ __forceinline__ void cfnmac(d2& acc, d2 a, d2 b) {
    acc[0] = fma(-a[0], b[0], acc[0]);
    acc[1] = fma(-a[1], b[0], acc[1]);
    acc[0] = fma(-a[1], b[1], acc[0]);
    acc[1] = fma(a[0], b[1], acc[1]);
}

// acc -= a conj(b) with a = the value lane T of `a_bc` holds in this lane's row of 16 lanes (tbk_dpp.h); the same four
// FMAs in the same order as cfnmac
template <int T>
__device__ __forceinline__ void cfnmac_bc(d2& acc, d2 a_bc, d2 b) {
    double re = acc[0], im = acc[1];
    fnmac_bc<T>(re, a_bc[0], b[0]);
    fnmac_bc<T>(im, a_bc[1], b[0]);
    fnmac_bc<T>(re, a_bc[1], b[1]);
    fmac_bc<T>(im, a_bc[0], b[1]);
    acc = (d2){re, im};
}

// acc += a b with b = the value lane T of `b_bc` holds in this lane's row of 16 lanes; the products and order of cfma
template <int T>
__device__ __forceinline__ void cfma_bc(d2& acc, d2 a, d2 b_bc) {
    double re = acc[0], im = acc[1];
    fmac_bc<T>(re, b_bc[0], a[0]);
    fmac_bc<T>(im, b_bc[1], a[0]);
    fnmac_bc<T>(re, b_bc[1], a[1]);
    fmac_bc<T>(im, b_bc[0], a[1]);
    acc = (d2){re, im};
}

// acc -= conj(a) b with a = the value lane T of `a_bc` holds in this lane's row of 16 lanes
template <int T>
__device__ __forceinline__ void cfnmacj_bc(d2& acc, d2 a_bc, d2 b) {
    double re = acc[0], im = acc[1];
    fnmac_bc<T>(re, a_bc[0], b[0]);
    fnmac_bc<T>(re, a_bc[1], b[1]);
    fnmac_bc<T>(im, a_bc[0], b[1]);
    fmac_bc<T>(im, a_bc[1], b[0]);
    acc = (d2){re, im};
}

// acc -= a b with b = the value lane T of `b_bc` holds in this lane's row of 16 lanes
template <int T>
__device__ __forceinline__ void cfnma_bc(d2& acc, d2 a, d2 b_bc) {
    double re = acc[0], im = acc[1];
    fnmac_bc<T>(re, b_bc[0], a[0]);
    fmac_bc<T>(re, b_bc[1], a[1]);
    fnmac_bc<T>(im, b_bc[1], a[0]);
    fnmac_bc<T>(im, b_bc[0], a[1]);
    acc = (d2){re, im};
}

// a wave-uniform double, moved to scalar registers
__device__ __forceinline__ double to_scalar(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// the value lane L of the wave holds, as a wave-uniform scalar
template <int L>
__device__ __forceinline__ double lane_value(double v) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), L);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), L);
    return __hiloint2double(hi, lo);
}

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// x <- [sum of x over the lane pair | sum of y over the lane pair] (lower | upper 32 lanes, or even | odd rows of 16)
__device__ __forceinline__ void swap_add(double& x, double y, bool half32) {
    unsigned xl = (unsigned)__double2loint(x), xh = (unsigned)__double2hiint(x);
    unsigned yl = (unsigned)__double2loint(y), yh = (unsigned)__double2hiint(y);
    if (half32) {
        const auto rl = __builtin_amdgcn_permlane32_swap(xl, yl, false, false);
        const auto rh = __builtin_amdgcn_permlane32_swap(xh, yh, false, false);
        x = __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
    } else {
        const auto rl = __builtin_amdgcn_permlane16_swap(xl, yl, false, false);
        const auto rh = __builtin_amdgcn_permlane16_swap(xh, yh, false, false);
        x = __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
    }
}

// four wave-wide sums at once: lane l ends with the total of value (l >> 4); fixed summation tree
__device__ __forceinline__ double reduce4(double p0, double p1, double p2, double p3) {
    swap_add(p0, p2, true);
    swap_add(p1, p3, true);
    swap_add(p0, p1, false);
    double v = p0;
    v += dpp_mov<0x128>(v);  // row_ror 8, 4, 2, 1
    v += dpp_mov<0x124>(v);
    v += dpp_mov<0x122>(v);
    v += dpp_mov<0x121>(v);
    return v;
}

// barrier with explicit waits: LDS traffic (lgkmcnt) and the global stores other waves of this workgroup re-read (vmcnt)
__device__ __forceinline__ void wg_sync() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
}
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// a counter in LDS, written / polled without the vmcnt(0) a volatile access would bring along (the tile loads and stores in
// flight have nothing to do with it); LDS operations of a wave are performed in order
__device__ __forceinline__ void lds_post(int* p, int value) {
    const unsigned at = (unsigned)(size_t)(__attribute__((address_space(3))) int*)p;
    asm volatile("ds_write_b32 %0, %1" ::"v"(at), "v"(value) : "memory");
}
__device__ __forceinline__ int lds_poll(const int* p) {
    const unsigned at = (unsigned)(size_t)(__attribute__((address_space(3))) const int*)p;
    int value;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(value) : "v"(at) : "memory");
    return __builtin_amdgcn_readfirstlane(value);
}

// 1 / x and (sqrt(x), 1 / sqrt(x)) from the hardware estimates plus Newton steps: a dozen instructions less per call
// than IEEE division / sqrt, on the serial path of every Householder step.  x > 0 and well inside the double range
// (the callers' x are squared norms: matrices scaled below 1e-150 would have underflowed there already).
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    return fma(r, e, r);
}
__device__ __forceinline__ void fast_sqrt_rsqrt(double x, double& root, double& rroot) {
    double r = __builtin_amdgcn_rsq(x);
    double g = x * r, h = 0.5 * r;
    double e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    const double d = fma(-g, g, x);
    root = fma(d, h, g);
    rroot = 2.0 * h;
    e = fma(-root, rroot, 1.0);  // one more step for the reciprocal
    rroot = fma(rroot, e, rroot);
}

// Workgroup sums of up to 64 per-thread values, fixed order (transposed butterflies inside a wave, then the waves in
// index order).  Values are handed over four at a time -- wave_partial4(slot, ...) for slots 0, 4, 8, ... -- so that
// a caller never holds more than four of them in registers; wg_finish(nv) makes the totals readable in s_tot[0 .. nv).
__device__ __forceinline__ void wave_partial4(int slot, double p0, double p1, double p2, double p3, double* s_part, int lane, int wave) {
    const double t = reduce4(p0, p1, p2, p3);
    if ((lane & 15) == 0) s_part[wave * 64 + slot + (lane >> 4)] = t;
}
template <int NW>
__device__ __forceinline__ void wg_finish(int nv, double* s_part, double* s_tot, int tid) {
    wg_sync();
    if (tid < nv) {
        double acc = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) acc += s_part[w * 64 + tid];
        s_tot[tid] = acc;
    }
    wg_sync();
}

// ------------------------------------------------------------------------------------------------
// stage 2
// ------------------------------------------------------------------------------------------------
// The band as LOWER diagonals in LDS: element (i, j), 0 <= i - j < 16, at sL[(i - j) * NP + j]; NP % 16 == 9 makes the
// 8 x 8 block accesses of a wave (lane = row a + 8 column b) conflict-free for ds_read_b128.
template <int CTRL>
__device__ __forceinline__ d2 dpp_mov2(d2 v) { return (d2){dpp_mov<CTRL>(v[0]), dpp_mov<CTRL>(v[1])}; }

// sum over the row index a = lane & 7 (lanes that share b): every lane ends with the total
__device__ __forceinline__ double sum_a(double v) {
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);  // row_half_mirror: the other quad of the 8
    return v;
}
__device__ __forceinline__ d2 sum_a2(d2 v) { return (d2){sum_a(v[0]), sum_a(v[1])}; }

// ------------------------------------------------------------------------------------------------
// stage 2: FOUR sweeps per wave.  A chase step works on 8 x 8 blocks; the first version gave a whole wave to a sweep, and
// most of its ~350 instructions per step were cross-lane reductions and scalar work replicated 64 times (2.3 us per
// 256 x 256 matrix; removed).  Here a sweep gets 16 lanes -- lane
// (a = l & 7, h = (l >> 3) & 1) holds row a, columns 4 h .. 4 h + 3 of a block -- so row sums are four local terms plus
// one exchange, the vectors that are needed by column (y, x) cross over through a wave-private LDS scratch, and every
// instruction advances four sweeps at once (~115 instructions per chase step).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double sum_a8(double v) { return sum_a(v); }  // over the 8 lanes that share (slot, h)

// the 16 working diagonals of the second stage, in LDS or in global memory: view[i] is element i either way
template <bool GLOBAL>
struct BandView {
    d2* lds;
    d2* glob;
    __device__ __forceinline__ d2& operator[](size_t i) const {
        if constexpr (GLOBAL)  // uniform base + 32-bit byte offset (a matrix' diagonals are < 4 GiB): scalar-base addressing
            return *reinterpret_cast<d2*>(reinterpret_cast<char*>(glob) + (unsigned)((unsigned)i * 16u));
        else
            return lds[i];
    }
};

// The previous panel's [V | W] rows live in global memory in MFMA-fragment order, so that a 16-row block is four
// contiguous 1 KiB wave loads: entry (row, c) of block I = row / 16 at  ((I * 4 + c / 4) * 64 + (c % 4) * 16 + row % 16).
__device__ __forceinline__ size_t vw_index(int row, int c) {
    return ((size_t)(row >> 4) * 4 + (c >> 2)) * 64 + (size_t)(c & 3) * 16 + (row & 15);
}

// workgroups that share the tile pass of ONE matrix in the launch chain: enough that a wave owns one block (two at 1024
// orbitals), at most eight
__host__ __device__ inline int tbk_band_split_members(int n, int nw) {
    const int nbk = (n + TS - 1) / TS;
    const int want = (nbk + nw - 1) / nw;
    return want < 1 ? 1 : (want > 8 ? 8 : want);
}

struct Frag {  // a 16 x 16 complex operand block in A/B-operand layout: lane l holds [l % 16][l / 16 + 4 s], s = 0..3
    double re[4], im[4];
};
#ifdef TBK_EXPERIMENTS
// The band's way out of the matrix for the launch chain: band[i][dd] = H[i][i + dd], dd = 0..8 (what the tail of the one-launch
// kernel does).
__global__ void __launch_bounds__(256) band_extract_kernel(const double* __restrict__ Hall, int n, d2* __restrict__ band_all, size_t band_stride) {
    const double* H = Hall + (size_t)blockIdx.x * n * n * 2;
    d2* band = band_all + (size_t)blockIdx.x * band_stride;
    for (int idx = threadIdx.x; idx < n * (PB + 1); idx += 256) {
        const int i = idx / (PB + 1), dd = idx - i * (PB + 1);
        band[idx] = (i + dd < n) ? *reinterpret_cast<const d2*>(H + ((size_t)i * n + i + dd) * 2) : (d2){0.0, 0.0};
    }
}
#endif

}  // namespace

// ---- host side: sizes and policy (defined in tbk_eig_band.hip) ---------------------------------------------------------------------
constexpr int BAND_ONE_WG_MAXN = 1024;    // one workgroup per matrix: two rows per thread of 512 threads, X (8 complex per row) is 128 KiB of LDS
constexpr int BAND_LDS_CHASE_MAXN = 512;  // above: the chase keeps its 16 diagonals in global memory
bool tbk_band_is_xl(int n);               // the sizes that ALWAYS take the launch chain of band_xl_* (above 1024 orbitals; TBK_BAND_XL_FROM)
int tbk_band_chase_pitch(int n);          // pitch of a working diagonal of the second stage
bool tbk_band_chase_global_forced(int n);
bool tbk_band_chase_small_window(const tbk_model* m, int n, int64_t nk);
// tbk_eig_band_chase.hip: stage two of nk matrices on stream s, no stage timer (the callers hold one)
int tbk_band_launch_chase(tbk_model* m, hipStream_t s, const void* d_band, int64_t nk, double* d_D, double* d_E);
// tbk_eig_band_xl.hip: the first stage as a chain of launches; d_de != NULL: the second stage of every group behind it
int tbk_band_launch_xl(tbk_model* m, hipStream_t s, double* d_H, int n, int64_t nk, void* d_scratch, void* d_band, double* d_de = nullptr);
