// tbk_eig_band.hip -- two-stage Householder tridiagonalisation for 64 < n_orb <= 1024, one workgroup per matrix:
//
//   stage 1  band_reduce_kernel   dense -> band of half-width 8.  Panels of 8 rows; per panel ONE pass over the
//                                 16 x 16 tiles of the stored (upper) triangle of the trailing matrix, on the matrix
//                                 pipe (v_mfma_f64_16x16x4_f64): the rank-16 update with the previous panel's (V, W)
//                                 and the product with the next panel's V in the same visit of a tile.
//   stage 2  chase4_body          band -> tridiagonal by Householder bulge chasing in LDS, four sweeps per wave,
//                                 sweeps pipelined two steps apart; the tail of the stage-1 kernel up to 256
//                                 orbitals (the band never leaves the LDS), band_chase4_kernel up to 512,
//                                 band_chase4g_kernel (the 16 working diagonals in global memory) up to 1024.
//
// Reference step: scipy.linalg.eigvalsh per k-point (/root/reference/src/tbmodels/_tb_model.py:1147-1150).
// The one-stage reduction of tbk_eig_stream.hip reads the trailing triangle once per Householder step
// (16 n^3 / 6 bytes per matrix: 56 MB at n = 256, BLAS-2 on the vector unit); here the triangle is read and
// written once per 8 steps (11 MB at n = 256) and the O(n^3) work is GEMM-shaped.  tools/two_stage_model.py is
// the NumPy statement of the same data flow (index conventions, phases, formulas); the comments below refer to it.
//
// Conventions: H is the row-major n x n complex matrix of which only the upper triangle (i <= j) is valid
// (the H(k) kernels' TRI output); nothing here reads the lower triangle.  Householder reflectors follow LAPACK
// (zlarfg / zgeqr2 / zlarft): H_c = I - tau_c v_c v_c^H, Q = H_0 ... H_7 = I - V T V^H.

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

// The body of the second stage for the calling workgroup's matrix: `band` = compact band (9 complex per row), or, when
// it is null, the band is read from the upper triangle of the row-major matrix Hm itself (the fused kernel).  `smem` is
// the workgroup's dynamic LDS (16 np complex + NW * 64 complex + n ints); Dm / Em receive the tridiagonal.
// GBAND != nullptr (above 512 orbitals: 16 diagonals of 1024 columns are 264 KiB, more than a CU's LDS): the diagonals live
// in that global buffer instead -- L2-resident, 264 KiB per matrix in flight -- and only the scratch and the schedule
// stay in LDS.  Same code: a wave's own accesses are ordered, the steps of a tick touch disjoint cells, and every tick
// ends on wg_sync (s_waitcnt vmcnt(0) + barrier: the workgroup's stores are visible to its other waves, same CU).
template <int NW, bool FROM_H, int CALLER = 0, bool GLOBAL = false>  // (one instantiation per calling kernel: each is inlined
// into it -- with two callers of one instantiation hipcc keeps a real function call: 248 registers and a stack frame in both)
__device__ inline void chase4_body(const d2* __restrict__ band, const double* __restrict__ Hm, double* smem, int n, int np,
                                            int stagger, double* __restrict__ Dm, double* __restrict__ Em, d2* gband = nullptr) {
    constexpr int NSLOT = 4 * NW;
    // (two differently typed views of the diagonals: the address space is a compile-time fact -- picked at run time the
    // accesses were flat instructions and 24 more registers)
    d2* const sLl = reinterpret_cast<d2*>(smem);
    d2* const sLg = gband;
    const BandView<GLOBAL> sL{sLl, sLg};
    d2* sScr = GLOBAL ? reinterpret_cast<d2*>(smem) : sLl + (size_t)16 * np;  // [NW][4 slots][16]: y (8) and x (8) by row
    int* sStart = reinterpret_cast<int*>(sScr + NW * 64);           // [n] first tick of every sweep
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int a = lane & 7, h = (lane >> 3) & 1, g = lane >> 4;
    d2* scr = sScr + (wave * 4 + g) * 16;

    for (int i = tid; i < 16 * np; i += NW * 64) sL[i] = (d2){0.0, 0.0};
    wg_sync();
    for (int i = tid; i < n * (PB + 1); i += NW * 64) {
        const int j = i / (PB + 1), dd = i % (PB + 1);
        if (j + dd < n) {
            const d2 v = FROM_H ? *reinterpret_cast<const d2*>(Hm + ((size_t)j * n + j + dd) * 2) : band[i];
            sL[(size_t)dd * np + j] = (d2){v[0], -v[1]};
        }
    }
    const int n_sweeps = n - 2;
    auto sweep_len = [&](int j) { return (n - 1 - j + PB - 1) / PB; };
    if (tid == 0) {
        for (int s = 0; s < n_sweeps; ++s) {
            int t0 = 0;
            if (s > 0) t0 = sStart[s - 1] + stagger;
            if (s >= NSLOT) t0 = max(t0, sStart[s - NSLOT] + sweep_len(s - NSLOT));
            sStart[s] = t0;
        }
    }
    wg_sync();
    if (n_sweeps > 0) {
        const int total_ticks = sStart[n_sweeps - 1] + sweep_len(n_sweeps - 1);
        // element (i, j) of the band lives at (i - j) np + j: per lane and column c the part that does not depend on
        // the block position r0
        int dstat[4], bstat[4];
        bool d_low[4];
        double d_imf[4];  // what the stored imaginary part is multiplied by: -1 above the diagonal (conjugate), 0 on it, 1 below
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int b = 4 * h + c;
            dstat[c] = abs(a - b) * np + min(a, b);
            bstat[c] = (PB + a - b) * np + b;
            d_imf[c] = a < b ? -1.0 : (a == b ? 0.0 : 1.0);
            d_low[c] = a >= b;
        }
        const int xstat = (PB + a) * np;  // first column of the block below, row a
        int sw = wave * 4 + g;  // this slot's current / next sweep
        int k = -1, k_len = 0;
        d2 va = (d2){0.0, 0.0}, tau = va;
        d2 vb[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) vb[c] = va;

        // zlarfg from x_a (own row), x_b (the four rows named by this lane's columns) and alpha = x[0]
        auto reflector = [&](d2 xa, const d2 (&xb)[4], d2 alpha, d2& o_va, d2 (&o_vb)[4], d2& o_tau, double& o_beta) {
            const double sigma = sum_a8(a >= 1 ? xa[0] * xa[0] + xa[1] * xa[1] : 0.0);
            o_tau = (d2){0.0, 0.0};
            o_beta = alpha[0];
            o_va = (a == 0) ? (d2){1.0, 0.0} : (d2){0.0, 0.0};
#pragma unroll
            for (int c = 0; c < 4; ++c) o_vb[c] = (4 * h + c == 0) ? (d2){1.0, 0.0} : (d2){0.0, 0.0};
            const bool trivial = (sigma == 0.0 && alpha[1] == 0.0);  // uniform over the slot
            const double norm2 = trivial ? 1.0 : alpha[0] * alpha[0] + alpha[1] * alpha[1] + sigma;
            double root, rroot;
            fast_sqrt_rsqrt(norm2, root, rroot);
            const double beta = -copysign(root, alpha[0]);
            const double rbeta = -copysign(rroot, alpha[0]);
            const double qr_ = alpha[0] - beta, qi_ = alpha[1];
            const double qn = fast_rcp(qr_ * qr_ + qi_ * qi_);
            const d2 scale = (d2){qr_ * qn, -qi_ * qn};
            if (!trivial) {
                o_tau = (d2){(beta - alpha[0]) * rbeta, -alpha[1] * rbeta};
                o_beta = beta;
                if (a != 0) o_va = cmul(xa, scale);
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (4 * h + c != 0) o_vb[c] = cmul(xb[c], scale);
            }
        };

        for (int tick = 0; tick < total_ticks; ++tick) {
            const bool starting = k < 0 && sw < n_sweeps && tick == sStart[min(sw, n_sweeps - 1)];
            if (__any(starting)) {
                // first reflector of a sweep: column sw below the diagonal (slots that do not start read a valid column
                // and drop the result)
                const int j = starting ? sw : 0;
                const d2 xa = sL[(size_t)(1 + a) * np + j];
                d2 xb[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) xb[c] = sL[(size_t)(1 + 4 * h + c) * np + j];
                const d2 alpha = sL[(size_t)np + j];
                d2 n_va, n_vb[4], n_tau;
                double beta;
                reflector(xa, xb, alpha, n_va, n_vb, n_tau, beta);
                if (GLOBAL)
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                else
                    lds_fence();
                if (starting) {
                    va = n_va;
                    tau = n_tau;
#pragma unroll
                    for (int c = 0; c < 4; ++c) vb[c] = n_vb[c];
                    k = 0;
                    k_len = sweep_len(sw);
                    if (h == 0 && j + 1 + a < n) sL[(size_t)(1 + a) * np + j] = (a == 0) ? (d2){beta, 0.0} : (d2){0.0, 0.0};
                }
            }
            const bool active = k >= 0;
            if (__any(active)) {
                const int r0 = active ? sw + 1 + PB * k : 0;
                // all loads of the tick first
                d2 dv[4], bk[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    dv[c] = sL[dstat[c] + r0];
                    bk[c] = sL[bstat[c] + r0];
                }
                const d2 bk0a = sL[xstat + r0];
#pragma unroll
                for (int c = 0; c < 4; ++c) dv[c][1] *= d_imf[c];
                // row sums: y = D v and u = Bk v (four local terms, then the other half of the row)
                d2 ya = (d2){0.0, 0.0}, ua = ya;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    cfma(ya, dv[c], vb[c]);
                    cfma(ua, bk[c], vb[c]);
                }
                ya[0] += dpp_mov<0x128>(ya[0]);
                ya[1] += dpp_mov<0x128>(ya[1]);
                ua[0] += dpp_mov<0x128>(ua[0]);
                ua[1] += dpp_mov<0x128>(ua[1]);
                const d2 tu = cmul(tau, ua);
                const d2 xa = (d2){bk0a[0] - tu[0], bk0a[1] - tu[1]};  // first column of Bk' (v[0] = 1 when tau != 0)
                // D' = H^H D H = D - v w^H - w v^H  with  w = tau y - (|tau|^2 rho / 2) v,  rho = v^H y  (real: D is Hermitian)
                const double rho = sum_a8(va[0] * ya[0] + va[1] * ya[1]);
                const double f = -0.5 * (tau[0] * tau[0] + tau[1] * tau[1]) * rho;
                d2 wa = cmul(tau, ya);
                wa[0] = fma(f, va[0], wa[0]);
                wa[1] = fma(f, va[1], wa[1]);
                // w and x are needed by column too: through the slot's scratch (one wave: its LDS traffic is ordered)
                asm volatile("" ::: "memory");
                if (h == 0) {
                    scr[a] = wa;
                    scr[8 + a] = xa;
                }
                asm volatile("" ::: "memory");
                d2 wb[4], xb[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    wb[c] = scr[4 * h + c];
                    xb[c] = scr[8 + 4 * h + c];
                }
                const d2 alpha = scr[8];
                asm volatile("" ::: "memory");
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    d2 dn = dv[c];
                    cfnmac(dn, va, wb[c]);
                    cfnmac(dn, wa, vb[c]);
                    if (active && d_low[c] && r0 + a < n) sL[dstat[c] + r0] = dn;
                }
                // Bk' = Bk - tau u conj(v_b); next reflector from its first column; Bk'' = Bk' - conj(tau2) v2_a z_b
                d2 bn[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    bn[c] = bk[c];
                    cfnmac(bn[c], tu, vb[c]);
                }
                d2 n_va, n_vb[4], n_tau;
                double beta;
                reflector(xa, xb, alpha, n_va, n_vb, n_tau, beta);
                const d2 ctau2 = conjd(n_tau);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const d2 zc = sum_a2(cmulc(bn[c], n_va));  // conj(v2_a) Bk'[a][b] summed over the rows
                    const d2 f2 = cmul(ctau2, zc);
                    cfma(bn[c], (d2){-n_va[0], -n_va[1]}, f2);
                    if (4 * h + c == 0) bn[c] = (a == 0) ? (d2){beta, 0.0} : (d2){0.0, 0.0};
                    if (active && r0 + PB + a < n && r0 + 4 * h + c < n) sL[bstat[c] + r0] = bn[c];
                }
                // (unconditionally: a slot that is not active holds nothing -- its next sweep starts from the column itself --
                // and a conditional copy is twelve register moves per tick)
                va = n_va;
                tau = n_tau;
#pragma unroll
                for (int c = 0; c < 4; ++c) vb[c] = n_vb[c];
                if (active) {
                    if (++k == k_len) {
                        k = -1;
                        sw += NSLOT;
                    }
                }
            }
            wg_sync();
        }
    }
    wg_sync();
    for (int j = tid; j < n; j += NW * 64) {
        Dm[j] = sL[j][0];
        double e = 0.0;
        if (j + 1 < n) {
            const d2 v = sL[(size_t)np + j];
            e = sqrt(v[0] * v[0] + v[1] * v[1]);
        }
        Em[j] = e;
    }
}

template <int NW>
__global__ void __launch_bounds__(NW * 64)
band_chase4_kernel(const d2* __restrict__ band_all, size_t band_stride, int n, int np, int stagger, double* __restrict__ D,
                   double* __restrict__ E) {
    // (band_stride complex numbers per matrix: the compact band, and from 257 orbitals on the working diagonals of the windowed kernel behind it)
    extern __shared__ __attribute__((aligned(16))) double bc_smem[];
    const size_t mat = blockIdx.x;
    chase4_body<NW, false>(band_all + mat * band_stride, nullptr, bc_smem, n, np, stagger, D + mat * (size_t)n, E + mat * (size_t)n);
}

// above 512 orbitals: the 16 working diagonals in global memory, behind the compact band of the same matrix
// (band_stride complex numbers per matrix: n (PB + 1) compact + 16 np working)
template <int NW>
__global__ void __launch_bounds__(NW * 64)
band_chase4g_kernel(d2* __restrict__ band_all, size_t band_stride, int n, int np, int stagger, double* __restrict__ D, double* __restrict__ E) {
    extern __shared__ __attribute__((aligned(16))) double bc_smem[];
    const size_t mat = blockIdx.x;
    d2* band = band_all + mat * band_stride;
    chase4_body<NW, false, 1, true>(band, nullptr, bc_smem, n, np, stagger, D + mat * (size_t)n, E + mat * (size_t)n, band + (size_t)n * (PB + 1));
}

// Above 512 orbitals (round 5): the working diagonals in a CYCLIC WINDOW of 512 columns in LDS in front of
// the global buffer of band_chase4g_kernel.  With the diagonals in global memory a tick is two memory round trips (the loads,
// then the wait for the stores before the barrier) around a ~1.3 us chain: 3.0 - 3.3 us against the 1.5 us of the LDS form.  But the
// 32 sweeps in flight only ever touch ~490 consecutive columns: sweep s runs in slot s % 32, the sweeps of generation g = s / 32
// follow each other 15 columns apart, and the first sweep of generation g + 1 starts at the top when the first sweep of
// generation g has reached the bottom -- so generation g + 1 sees column j at window column (j + off[g + 1]) mod 512 with
// off[g + 1] = off[g] + (n + 8 - 32 (g + 1)): its top follows the bottom of generation g in the window as it does in time.  A
// column enters the window the tick before the generation's first sweep needs it (one element per thread, fetched at the start
// of the tick, stored to LDS at its end) and leaves the tick after the generation's last sweep touched it; from the first
// generation whose columns all fit (n + 8 - 32 g <= 512) on nothing leaves any more.  A slot of a generation that leaves is taken
// again no sooner than 68 ticks after it started, so that a column is back in global memory before the next generation fetches it.
// tools/two_stage_model.py: stage2_window is this scheme with an occupancy tag per window column (every access finds ITS
// column, a column only enters a free cell; tests/test_two_stage_model.py).  Same arithmetic per sweep as chase4_body: the same bits.
// NW waves = 4 NW sweep slots; CW window columns (>= 15 * 4 NW + 18: what the slots can hold in flight), CWP = pitch of a diagonal
// (= 9 mod 16: bank-conflict free, as in the plain LDS form).  <8, 512, 521>: above 512 orbitals.  <4, 272, 281> (TBK_CHASE_WINDOW_SMALL,
// measurements): 257 - 512 orbitals in 78 KiB instead of the 133 KiB of the plain LDS form.
template <int NW, int CW, int CWP>
__global__ void __launch_bounds__(NW * 64)
band_chase4w_kernel(d2* __restrict__ band_all, size_t band_stride, int n, int np, double* __restrict__ D, double* __restrict__ E) {
    constexpr int NSLOT = 4 * NW, CW_GAP = 2 * NSLOT + 4;
    static_assert(CW >= 15 * NSLOT + 18 && CWP >= CW + PB && CWP % 16 == 9, "window too small for the sweeps in flight / pitch");
    extern __shared__ __attribute__((aligned(16))) double bw_smem[];
    auto modw = [](int x) { return x % CW; };                 // x >= 0
    auto wrapw = [](int x) {  // 0 <= x < 2 CW
        if constexpr ((CW & (CW - 1)) == 0)
            return x & (CW - 1);  // (one instruction instead of compare + subtract + select: nine addresses per tick and sweep)
        else
            return x >= CW ? x - CW : x;
    };
    const size_t mat = blockIdx.x;
    const d2* band = band_all + mat * band_stride;
    d2* gband = band_all + mat * band_stride + (size_t)n * (PB + 1);  // [16][np], element (i, j) at (i - j) np + j
    double* Dm = D + mat * (size_t)n;
    double* Em = E + mat * (size_t)n;
    d2* win = reinterpret_cast<d2*>(bw_smem);            // [16][CWP]
    d2* sScr = win + (size_t)16 * CWP;                   // [NW][4 slots][16]
    int* sStart = reinterpret_cast<int*>(sScr + NW * 64);  // [n]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int a = lane & 7, h = (lane >> 3) & 1, g = lane >> 4;
    d2* scr = sScr + (wave * 4 + g) * 16;
    const int NE = n + PB;
    const int n_sweeps = n - 2;
    const int g_res = NE > CW ? (NE - CW + NSLOT - 1) / NSLOT : 0;  // first generation whose columns all fit the window
    auto sweep_len = [&](int j) { return (n - 1 - j + PB - 1) / PB; };
    auto off_of = [&](int gen) {
        const int m = min(gen, g_res);
        return modw(m * NE - (NSLOT / 2) * m * (m + 1));
    };

    for (int i = tid; i < 16 * np; i += NW * 64) gband[i] = (d2){0.0, 0.0};
    for (int i = tid; i < 16 * CWP; i += NW * 64) win[i] = (d2){0.0, 0.0};
    wg_sync();
    for (int i = tid; i < n * (PB + 1); i += NW * 64) {
        const int j = i / (PB + 1), dd = i % (PB + 1);
        if (j + dd < n) {
            const d2 v = band[i];
            gband[(size_t)dd * np + j] = (d2){v[0], -v[1]};
        }
    }
    if (tid == 0) {
        for (int s = 0; s < n_sweeps; ++s) {
            int t0 = 0;
            if (s > 0) t0 = sStart[s - 1] + 2;
            if (s >= NSLOT) {
                const int prev = s - NSLOT;
                const int len = sweep_len(prev);
                t0 = max(t0, sStart[prev] + (prev / NSLOT < g_res ? max(len, CW_GAP) : len));
            }
            sStart[s] = t0;
        }
    }
    wg_sync();
    if (n_sweeps > 0) {
        // (what the trackers read of the schedule goes through readfirstlane: uniform by construction, and only so does the compiler
        // keep them and everything derived from them in scalar registers -- the tracker arithmetic of every tick on the scalar unit)
        auto sched = [&](int s) { return __builtin_amdgcn_readfirstlane(sStart[s]); };
        const int total_ticks = sched(n_sweeps - 1) + sweep_len(n_sweeps - 1);
        const int n_gen = (n_sweeps + NSLOT - 1) / NSLOT;
        const int last_fetch_gen = min(g_res, n_gen - 1);
        // the columns generation 0 needs at tick 0
        for (int e = tid; e < 9 * 16; e += NW * 64) {
            const int j = e >> 4, dd = e & 15;
            if (j < NE) win[dd * CWP + j] = j < n ? gband[(size_t)dd * np + j] : (d2){0.0, 0.0};
        }
        wg_sync();
        // per lane and column c: diagonal (row of the window) and column offset inside the block
        int wd[4], wb[4];   // window rows (x CWP) of the D and Bk elements
        int cd[4], cb[4];   // their columns relative to r0
        bool d_low[4];
        double d_imf[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int b = 4 * h + c;
            wd[c] = abs(a - b) * CWP;
            cd[c] = min(a, b);
            wb[c] = (PB + a - b) * CWP;
            cb[c] = b;
            d_imf[c] = a < b ? -1.0 : (a == b ? 0.0 : 1.0);
            d_low[c] = a >= b;
        }
        const int wx = (PB + a) * CWP;  // first column of the block below, row a
        int sw = wave * 4 + g;
        int off = 0;  // this slot's generation offset
        int vr0 = 0;  // window column of the slot's block position r0 (kept in [0, CW): + 8 per step)
        int k = -1, k_len = 0;
        d2 va = (d2){0.0, 0.0}, tau = va;
        d2 vb[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) vb[c] = va;
        auto reflector = [&](d2 xa, const d2 (&xb)[4], d2 alpha, d2& o_va, d2 (&o_vb)[4], d2& o_tau, double& o_beta) {
            const double sigma = sum_a8(a >= 1 ? xa[0] * xa[0] + xa[1] * xa[1] : 0.0);
            o_tau = (d2){0.0, 0.0};
            o_beta = alpha[0];
            o_va = (a == 0) ? (d2){1.0, 0.0} : (d2){0.0, 0.0};
#pragma unroll
            for (int c = 0; c < 4; ++c) o_vb[c] = (4 * h + c == 0) ? (d2){1.0, 0.0} : (d2){0.0, 0.0};
            const bool trivial = (sigma == 0.0 && alpha[1] == 0.0);
            const double norm2 = trivial ? 1.0 : alpha[0] * alpha[0] + alpha[1] * alpha[1] + sigma;
            double root, rroot;
            fast_sqrt_rsqrt(norm2, root, rroot);
            const double beta = -copysign(root, alpha[0]);
            const double rbeta = -copysign(rroot, alpha[0]);
            const double qr_ = alpha[0] - beta, qi_ = alpha[1];
            const double qn = fast_rcp(qr_ * qr_ + qi_ * qi_);
            const d2 scale = (d2){qr_ * qn, -qi_ * qn};
            if (!trivial) {
                o_tau = (d2){(beta - alpha[0]) * rbeta, -alpha[1] * rbeta};
                o_beta = beta;
                if (a != 0) o_va = cmul(xa, scale);
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (4 * h + c != 0) o_vb[c] = cmul(xb[c], scale);
            }
        };
        // (uniform) trackers of the columns that enter and leave; what they need of the schedule is read when it changes, not per tick
        constexpr int NEVER = 0x7fffffff;
        const int io_c = tid >> 4, io_dd = tid & 15;  // this thread's column of a chunk and diagonal, when it moves an element
        const int io_row = io_dd * CWP;
        // (uniform base + 32-bit byte offset: scalar-base addressing, two vector instructions per address instead of 64-bit arithmetic;
        // a matrix' diagonals are < 4 GiB)
        const unsigned io_goff = (unsigned)io_dd * (unsigned)np * 16u;
        auto g_at = [&](unsigned byte_off) -> d2& { return *reinterpret_cast<d2*>(reinterpret_cast<char*>(gband) + byte_off); };
        int g_in = 0, t0_in = 0, off_in = 0;   // generation whose first sweep leads, its first tick, its offset
        int t_in_next = last_fetch_gen > 0 ? sched(NSLOT) : NEVER;  // first tick of the generation that leads next
        int s_ev = 0, off_ev = 0;              // next sweep whose own column leaves (generations that leave only)
        int t_ev = g_res > 0 ? sched(0) + 1 : NEVER;
        int g_out = 0, off_out = 0;            // generation whose last sweep trails
        int t_sl = g_res > 0 ? sched(NSLOT - 1) : NEVER;  // first tick of that sweep
        int my_start = sw < n_sweeps ? sStart[sw] : NEVER;  // first tick of this slot's next sweep

        for (int tick = 0; tick < total_ticks; ++tick) {
            // ---- columns that enter for tick + 1: fetched now, stored to the window at the end of this tick ----
            int pf_idx = -1;
            d2 pf_val = (d2){0.0, 0.0};
            {
                const int nt = tick + 1;
                if (nt >= t_in_next) {
                    // (the generation that led until now may have its last columns due at this very tick -- n = 1 mod 8 with 16
                    // slots: they lie behind the matrix, i.e. they are zeros; their cells are free since the last tick)
                    const int j_old = NSLOT * g_in + 1 + PB * (nt - t0_in);
                    if (j_old < NE && tid < 128) {
                        const int j = j_old + io_c;
                        if (j < NE) win[io_row + wrapw(modw(j_old + off_in) + io_c)] = (d2){0.0, 0.0};
                    }
                    ++g_in;
                    t0_in = t_in_next;
                    off_in = off_of(g_in);
                    t_in_next = g_in < last_fetch_gen ? sched(NSLOT * (g_in + 1)) : NEVER;
                }
                const int kk = nt - t0_in;
                if (kk >= 0) {
                    const int base = NSLOT * g_in;
                    const int j_lo = kk == 0 ? base : base + 1 + PB * kk;
                    const int j_hi = min(base + 9 + PB * kk, NE);
                    const int j = j_lo + io_c;
                    if (j < j_hi) {
                        pf_idx = io_row + wrapw(modw(j_lo + off_in) + io_c);
#if !defined(TBK_ABLATE_WIN_IO) && !defined(TBK_ABLATE_WIN_LOADS)
                        // (no `j < n ? ... : 0`: the buffer's columns n .. n + 7 ARE zeros (np >= n + 8, nothing is ever written back there),
                        // and a select would want the loaded value at once -- the whole memory latency at the head of every tick: 12 %)
                        pf_val = g_at(io_goff + (unsigned)j * 16u);
#endif
                    }
                }
            }
            // ---- columns that leave: untouched since the last tick ----
            if (tick == t_ev) {
#if !defined(TBK_ABLATE_WIN_IO) && !defined(TBK_ABLATE_WIN_STORES)
                if (tid >= 128 && tid < 144)  // (io_dd = tid - 128 there)
                    g_at(io_goff + (unsigned)s_ev * 16u) = win[io_row + modw(s_ev + off_ev)];
#endif
                ++s_ev;
                if (s_ev < NSLOT * g_res) {
                    t_ev = sched(s_ev) + 1;
                    if (s_ev % NSLOT == 0) off_ev = off_of(s_ev / NSLOT);
                } else {
                    t_ev = NEVER;
                }
            }
            while (g_out < g_res) {  // (at most twice per tick)
                const int ks = tick - 1 - t_sl;
                if (ks < 0) break;
                const int j_lo = NSLOT * g_out + NSLOT + PB * ks;
                bool through = j_lo >= NE;
                if (!through) {
#if !defined(TBK_ABLATE_WIN_IO) && !defined(TBK_ABLATE_WIN_STORES)
                    if (tid < 128) {
                        const int j = j_lo + io_c;
                        if (j < n) g_at(io_goff + (unsigned)j * 16u) = win[io_row + wrapw(modw(j_lo + off_out) + io_c)];
                    }
#endif
                    if (j_lo + PB < NE) break;
                    // (that was its last chunk: the next generation's first may be due at this very tick)
                }
                ++g_out;
                off_out = off_of(g_out);
                t_sl = g_out < g_res ? sched(NSLOT * g_out + NSLOT - 1) : NEVER;
            }

            const bool starting = k < 0 && tick == my_start;
            if (__any(starting)) {
                const int j = starting ? sw : 0;
                const int vj = modw(j + off);
                const d2 xa = win[(1 + a) * CWP + vj];
                d2 xb[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) xb[c] = win[(1 + 4 * h + c) * CWP + vj];
                const d2 alpha = win[CWP + vj];
                d2 n_va, n_vb[4], n_tau;
                double beta;
                reflector(xa, xb, alpha, n_va, n_vb, n_tau, beta);
                lds_fence();
                if (starting) {
                    va = n_va;
                    tau = n_tau;
#pragma unroll
                    for (int c = 0; c < 4; ++c) vb[c] = n_vb[c];
                    k = 0;
                    k_len = sweep_len(sw);
                    vr0 = wrapw(vj + 1);
                    if (h == 0 && j + 1 + a < n) win[(1 + a) * CWP + vj] = (a == 0) ? (d2){beta, 0.0} : (d2){0.0, 0.0};
                }
            }
            const bool active = k >= 0;
            if (__any(active)) {
                const int r0 = active ? sw + 1 + PB * k : 0;
                int id[4], ib[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    id[c] = wd[c] + wrapw(vr0 + cd[c]);
                    ib[c] = wb[c] + wrapw(vr0 + cb[c]);
                }
                const int ix = wx + vr0;
                d2 dv[4], bk[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    dv[c] = win[id[c]];
                    bk[c] = win[ib[c]];
                }
                const d2 bk0a = win[ix];
#pragma unroll
                for (int c = 0; c < 4; ++c) dv[c][1] *= d_imf[c];
                d2 ya = (d2){0.0, 0.0}, ua = ya;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    cfma(ya, dv[c], vb[c]);
                    cfma(ua, bk[c], vb[c]);
                }
                ya[0] += dpp_mov<0x128>(ya[0]);
                ya[1] += dpp_mov<0x128>(ya[1]);
                ua[0] += dpp_mov<0x128>(ua[0]);
                ua[1] += dpp_mov<0x128>(ua[1]);
                const d2 tu = cmul(tau, ua);
                const d2 xa = (d2){bk0a[0] - tu[0], bk0a[1] - tu[1]};
                const double rho = sum_a8(va[0] * ya[0] + va[1] * ya[1]);
                const double f = -0.5 * (tau[0] * tau[0] + tau[1] * tau[1]) * rho;
                d2 wa = cmul(tau, ya);
                wa[0] = fma(f, va[0], wa[0]);
                wa[1] = fma(f, va[1], wa[1]);
                asm volatile("" ::: "memory");
                if (h == 0) {
                    scr[a] = wa;
                    scr[8 + a] = xa;
                }
                asm volatile("" ::: "memory");
                d2 wbv[4], xb[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    wbv[c] = scr[4 * h + c];
                    xb[c] = scr[8 + 4 * h + c];
                }
                const d2 alpha = scr[8];
                asm volatile("" ::: "memory");
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    d2 dn = dv[c];
                    cfnmac(dn, va, wbv[c]);
                    cfnmac(dn, wa, vb[c]);
                    if (active && d_low[c] && r0 + a < n) win[id[c]] = dn;
                }
                d2 bn[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    bn[c] = bk[c];
                    cfnmac(bn[c], tu, vb[c]);
                }
                d2 n_va, n_vb[4], n_tau;
                double beta;
                reflector(xa, xb, alpha, n_va, n_vb, n_tau, beta);
                const d2 ctau2 = conjd(n_tau);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const d2 zc = sum_a2(cmulc(bn[c], n_va));
                    const d2 f2 = cmul(ctau2, zc);
                    cfma(bn[c], (d2){-n_va[0], -n_va[1]}, f2);
                    if (4 * h + c == 0) bn[c] = (a == 0) ? (d2){beta, 0.0} : (d2){0.0, 0.0};
                    if (active && r0 + PB + a < n && r0 + 4 * h + c < n) win[ib[c]] = bn[c];
                }
                va = n_va;
                tau = n_tau;
#pragma unroll
                for (int c = 0; c < 4; ++c) vb[c] = n_vb[c];
                if (active) {
                    vr0 = wrapw(vr0 + PB);
                    if (++k == k_len) {
                        k = -1;
                        sw += NSLOT;
                        off = off_of(sw / NSLOT);
                        my_start = sw < n_sweeps ? sStart[sw] : NEVER;
                    }
                }
            }
            if (pf_idx >= 0) win[pf_idx] = pf_val;
            wg_sync();
        }
        // what stayed in the window goes back
        {
            const int base = NSLOT * g_res, off_r = off_of(g_res);
            for (int e = tid; e < (n - base) * 16; e += NW * 64) {
                const int j = base + (e >> 4), dd = e & 15;
                gband[(size_t)dd * np + j] = win[dd * CWP + modw(j + off_r)];
            }
        }
    }
    wg_sync();
    for (int j = tid; j < n; j += NW * 64) {
        Dm[j] = gband[j][0];
        double e = 0.0;
        if (j + 1 < n) {
            const d2 v = gband[(size_t)np + j];
            e = sqrt(v[0] * v[0] + v[1] * v[1]);
        }
        Em[j] = e;
    }
}

// ------------------------------------------------------------------------------------------------
// stage 1
// ------------------------------------------------------------------------------------------------
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

// NT threads per workgroup, ROWS rows of the matrix per thread in the thread-per-row phases (n <= NT * ROWS), VN_LDS: the
// next panel's V in LDS beside X (up to 256 orbitals) or in global memory (above: X alone is 64 KiB at 512 orbitals, and
// with V beside it only ONE workgroup fits a CU -- nothing then overlaps its serial phases: the 8-wave / one-row
// instantiation <512, 1, true> measured 31.7 us per 512 x 512 matrix against 29.7 and is no longer built).
//
// PHASE (round 4): 0 = the whole first stage of a matrix in one workgroup (what every call of more than a few matrices
// takes).  For calls of a FEW matrices the first stage is a chain of launches instead -- one matrix' tile pass is bounded
// by the matrix pipe of the ONE CU its workgroup sits on (0.3 ms of a 1.5 ms reduction at 256 orbitals, 2.3 of 7.9 ms at
// 512; DESIGN_LOG.md R4.7), and the tiles of a pass are independent:
//   PHASE 1, panel p, one workgroup per matrix: the W phase of panel p - 1 (from the sum of the members' partial products),
//            then look-ahead, panel QR and T of panel p; V, T go to global memory;
//   PHASE 2, panel p, `members` workgroups per matrix (blockIdx.x = member, blockIdx.y = matrix): the tile pass of panel p,
//            the own blocks dealt out over the waves of ALL members; every member leaves its partial X in global memory.
//            Behind the last panel the same launch applies the last pending update (no products).
// Stream order is the only synchronisation between them (no spinning on flags: nothing can hang).
#ifndef TBK_BAND_WAVES_PER_SIMD
#define TBK_BAND_WAVES_PER_SIMD 2  // register budget of the four-wave kernels (3: 168 registers -- measured: spills)
#endif
#ifdef TBK_ABLATE_BARRIER
#define TBK_PASS_CHAIN 0  // (the ablation removes the meeting point altogether: nobody would announce a finished visit)
#endif
#ifndef TBK_PASS_CHAIN
#define TBK_PASS_CHAIN 1  // 0: a workgroup barrier per step of the tile pass (rounds 2 - 4a)
#endif
#ifndef TBK_PASS_SPLIT
#define TBK_PASS_SPLIT 1  // 0: the left-over blocks of a pass' last round on one wave each, the others idle
#endif
#ifndef TBK_QR_ONE_WAVE
#define TBK_QR_ONE_WAVE 0  // 1: the recurrence of the Gram-form panel QR (one row per thread) on wave 0 alone, its coefficients through LDS
#endif
template <int NT, int ROWS, bool VN_LDS, int PHASE = 0>
__global__ void __launch_bounds__(NT, NT <= 256 ? TBK_BAND_WAVES_PER_SIMD : 1)  // two waves per SIMD: 4 x 128, 2 x 256 or 1 x 512 threads per CU
band_reduce_kernel(double* __restrict__ Hall, int n, d2* __restrict__ VWall, d2* __restrict__ VNall, d2* __restrict__ band_all,
                   size_t band_stride, int np, int stagger, double* __restrict__ D, double* __restrict__ E, int p_fixed = 0,
                   d2* __restrict__ split_all = nullptr) {
    static_assert(PHASE == 0 || !VN_LDS, "the launch chain keeps the next panel's V in global memory");
    constexpr int NW = NT / 64;
    extern __shared__ __attribute__((aligned(16))) double br_smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nbk = (n + TS - 1) / TS;
    const int npad = nbk * TS;
    d2* sX = reinterpret_cast<d2*>(br_smem);               // [npad][8]   A V of the next panel
    d2* sVnL = sX + (size_t)npad * PB;                     // [npad][8]   next panel's V (rows < s are zero), VN_LDS only
    double* sTr = reinterpret_cast<double*>(reinterpret_cast<char*>(br_smem) + band_xv_bytes(npad, VN_LDS, NW, ROWS));  // [NW][16][17] tile transposition planes
    double* sPart = sTr + NW * 16 * 17;                    // [NW][64]
    double* sTot = sPart + NW * 64;                        // [64]
    d2* sRow = reinterpret_cast<d2*>(sTot + 64);           // [2][8] row c of the panel (QR), broadcast; alternating
    // [8][16] the pending [V | W] rows of the look-ahead: only alive between two barriers before the panel's first
    // reduction, so it shares the partial-sum area (2 KiB: with it apart, two workgroups of the 256-orbital kernel
    // left no room on a CU for a bisection workgroup of the previous chunk)
    d2* sS = sRow + 16;                                    // [64]  S = T^H M T
    d2* sT = sS + 64;                                      // [8][8] T of the current panel
    d2* sTau = sT + 64;                                    // [8]
    int* sProg = reinterpret_cast<int*>(sTau + 8);         // [8]  visits finished, per wave (the chain of the tile pass)

    // the launch chain: `members` workgroups share a matrix in PHASE 2; its waves and theirs are numbered through
    const int members = PHASE == 2 ? (int)gridDim.x : 1;
    const int member = PHASE == 2 ? (int)blockIdx.x : 0;
    const int nw_all = NW * members;          // waves that share the tile pass of a matrix
    const int wave_all = member * NW + wave;  // this wave among them
    const size_t mat = PHASE == 2 ? blockIdx.y : blockIdx.x;
    double* H = Hall + mat * (size_t)n * n * 2;
    d2* VW = VWall + mat * (size_t)nbk * 256;
    d2* sVn = VN_LDS ? sVnL : VNall + mat * (size_t)npad * PB;  // the next panel's V, [npad][8], wherever it lives
    // between the launches of the chain, per matrix: T of the panel (64) and the members' partial X ([member][npad][8])
    const size_t split_stride = 64 + (size_t)tbk_band_split_members(n, NW) * npad * PB;
    d2* gT = PHASE != 0 ? split_all + mat * split_stride : nullptr;
    d2* gX = PHASE != 0 ? gT + 64 : nullptr;

    if (tid < 8) sProg[tid] = 0;  // (the first barrier of whatever follows is in front of the first pass)
    int prog_base = 0;            // visits of the passes so far: the chain's counters only ever grow
    // the pending-update buffer starts out empty
    if (PHASE == 0 || (PHASE == 1 && p_fixed == 0))
        for (int i = tid; i < nbk * 256; i += NT) VW[i] = (d2){0.0, 0.0};
    bool have_update = PHASE != 0 && p_fixed > 0;
#ifdef TBK_PHASE_CLOCK
    unsigned long long clk_acc_[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) clk_acc_[k] = 0;
    unsigned long long clk_prev_ = __builtin_readcyclecounter();
#endif

    auto Hat = [&](int i, int j) -> d2* { return reinterpret_cast<d2*>(H + ((size_t)i * n + j) * 2); };

    constexpr bool GRAM = TBK_PANEL_GRAM && ROWS == 1;  // the panel's Gram-type sums on the matrix pipe (see TBK_PANEL_GRAM)
    // Two rows per thread (257 - 1024 orbitals in batches): both rows of the panel beside the recurrence's tracked block do not fit
    // the register file, so the panel's rows live in the X AREA OF THE LDS during the QR ([row][8] complex, X's own layout; X is
    // dead there) and pass through the registers one row at a time.  The matrix instructions read that layout directly -- lane
    // (g, j) takes [row 16 g + rho][Re j] / [Im j - 8] -- so there are no planes: the QR's and T's sums read the rows where they
    // lie, and X itself is an operand of the W phase's sum without being disturbed (V comes from global memory there).
    constexpr bool GRAM2 = TBK_PANEL_GRAM && ROWS > 1 && !VN_LDS;
#if TBK_PANEL_GRAM
    // ---- C = A^H B (8 x 8 complex) of two row-distributed [rows][8] arrays, on the matrix pipe (PHASE 0 / 1 only) ----
    // gram_rows: this wave's 64 rows of the operands go through its LDS plane (which lives at the head of the X / V area: the
    // callers use it only while X and V are dead) into MFMA operand order -- O = [Re | Im] as 16 real columns, lane
    // (g, j) holds O[row 16 g + rho][j] in register rho, A and B operand alike -- and 16 MFMAs add O_a^T O_b to `acc`.
    // gram_finish: the waves' partial products meet ONCE (partials alternate between two areas of the tile-transposition
    // planes, free outside the pass: the next meeting may be written while a slow wave still reads this one); every wave
    // adds them in wave order -- identical bits everywhere -- and leaves C in sG[c][t].
    double* const gplane = reinterpret_cast<double*>(br_smem) + (size_t)wave * (64 * GP);
    d2* const gpart = reinterpret_cast<d2*>(sTr);  // [2][NW][64]
    d2* const sG = reinterpret_cast<d2*>(sPart);   // [8][8]
    int gram_parity = 0;
    auto gram_rows = [&](const d2 (&a)[PB], const d2 (&b)[PB], bool same, d4& acc) {
        const int g_lrow = lane & 15, g_lq = lane >> 4;
        double opa[16], opb[16];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            gplane[lane * GP + j] = a[j][0];
            gplane[lane * GP + PB + j] = a[j][1];
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int rho = 0; rho < 16; ++rho) opa[rho] = gplane[(16 * g_lq + rho) * GP + g_lrow];
        asm volatile("" ::: "memory");
        if (!same) {  // (a wave's LDS operations are performed in order: the plane is reused without a wait)
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                gplane[lane * GP + j] = b[j][0];
                gplane[lane * GP + PB + j] = b[j][1];
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int rho = 0; rho < 16; ++rho) opb[rho] = gplane[(16 * g_lq + rho) * GP + g_lrow];
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int rho = 0; rho < 16; ++rho) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(opa[rho], same ? opa[rho] : opb[rho], acc, 0, 0, 0);
    };
    // GRAM2: acc += O_a^T O_b over the 64 rows from base_row on; the operands [row][8] complex in LDS (a_lds / b_lds) or, when
    // a_glob is set, the A operand in global memory (same layout).  Rows before first_row and from npad on count as zero.
    auto gram_direct = [&](const d2* a_lds, const d2* a_glob, const d2* b_lds, int base_row, int first_row, d4& acc) {
        // (the lane's place is worked out HERE, from a copy of the lane number the compiler cannot see through: hoisted out of
        // the panel loop, three more registers lived across the tile pass -- which sits at 253 with two rows per thread -- and
        // the allocator answered with 100 - 700 B of scratch)
        int lane_here = lane;
        asm volatile("" : "+v"(lane_here));
        const int g_lrow = lane_here & 15, g_lq = lane_here >> 4;
        const int col = g_lrow < 8 ? 2 * g_lrow : 2 * (g_lrow - 8) + 1;  // Re j / Im (j - 8) of a row of 16 doubles
        const bool plain = base_row >= first_row && base_row + 64 <= npad;  // wave-uniform
        double opa[16], opb[16];
#pragma unroll
        for (int rho = 0; rho < 16; ++rho) {
            const int row = base_row + 16 * g_lq + rho;
            const int at = (plain ? row : min(row, npad - 1)) * 16 + col;
            double va = a_glob ? reinterpret_cast<const double*>(a_glob)[at] : reinterpret_cast<const double*>(a_lds)[at];
            if (!plain) va = (row >= first_row && row < npad) ? va : 0.0;
            opa[rho] = va;
            if (b_lds) {
                double vb = reinterpret_cast<const double*>(b_lds)[at];
                if (!plain) vb = (row >= first_row && row < npad) ? vb : 0.0;
                opb[rho] = vb;
            }
        }
#pragma unroll
        for (int rho = 0; rho < 16; ++rho) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(opa[rho], b_lds ? opb[rho] : opa[rho], acc, 0, 0, 0);
    };
    auto gram_finish = [&](const d4& acc) {
        // acc: lane (q, j), register r = M[q + 4 r][j], M = O_a^T O_b;  C[c][t] = M[c][t] + M[8 + c][8 + t] + i (M[c][8 + t] - M[8 + c][t])
        int lane_here = lane;
        asm volatile("" : "+v"(lane_here));
        const int g_lrow = lane_here & 15, g_lq = lane_here >> 4;
        const double sgn = g_lrow < 8 ? 1.0 : -1.0;
        d2 mine;
        mine[0] = fma(dpp_mov<0x128>(acc[2]), sgn, acc[0]);  // c = q:     Re C[c][j] (j < 8) / Im C[c][j - 8]
        mine[1] = fma(dpp_mov<0x128>(acc[3]), sgn, acc[1]);  // c = q + 4
        gpart[(gram_parity * NW + wave) * 64 + lane] = mine;
        lds_fence();
        __syncthreads();
        d2 tot = gpart[(gram_parity * NW) * 64 + lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            const d2 v = gpart[(gram_parity * NW + w) * 64 + lane];
            tot[0] += v[0];
            tot[1] += v[1];
        }
        gram_parity ^= 1;
        // every wave writes the same 128 values (and reads them back behind its own writes)
        double* gd = reinterpret_cast<double*>(sG);
        gd[((g_lq)*PB + (g_lrow & 7)) * 2 + (g_lrow >> 3)] = tot[0];
        gd[((g_lq + 4) * PB + (g_lrow & 7)) * 2 + (g_lrow >> 3)] = tot[1];
        asm volatile("" ::: "memory");
    };
#endif

    // ---- one pass over the tiles of the trailing triangle (model: big_pass) ----
    // s: rows / columns below s are finished (their V / W / Vn rows are zero);  with_hemm: accumulate X = A Vn.
    // A wave walks its visits (own block a = wave + NW q, step t: partner block a + t, cyclically) with the tile and
    // the partner's [V | W] block of the NEXT visit already requested while it works on the current one; the
    // workgroup meets once per step because the partner blocks of different steps overlap in sX.
    auto big_pass = [&](int s, bool with_update, bool with_hemm) {
        const int I0 = s / TS;
        const int na = nbk - I0;
        const int lrow = lane & 15, lq = lane >> 4;
        double* tr = sTr + wave * (16 * 17);
        // this lane's place inside a 16-row block of X / Vn ([row][Re 0..7 | Im 0..7] doubles, row lq + 4 r) and its sign there
        const int lane_x = lq * 16 + 2 * (lrow & 7) + (lrow >> 3);
        const double lane_sgn = (lrow < 8) ? -1.0 : 1.0;
        const int n_q = (na + nw_all - 1) / nw_all;
        const int n_t = na / 2;
        // The last round of own blocks holds r_last = na - nw_all (n_q - 1) of them: fewer than waves unless na is a multiple.
        // When at least two waves are left per block (g_last), the walk of each such block is SPLIT over g_last helper waves
        // -- helper j takes the steps t = j L + 1 .. (j + 1) L after a slot in which everybody fetches the block's operands
        // (helper 0: the diagonal tile) -- so the round lasts L + 1 slots instead of n_t + 1 (240 -> 226 slots over the passes
        // of a 256-orbital matrix).  Every helper keeps its own partial accumulators and flushes them in turn.  L >= r_last
        // keeps the partner blocks of one slot distinct (block offsets i + j L + tau, i < r_last).
        const int q_last = n_q - 1;
        const int r_last = na - nw_all * q_last;
        const int g_last = nw_all / r_last;
        const int l_try = (n_t + g_last - 1) / g_last;
        const bool split = TBK_PASS_SPLIT && g_last >= 2 && n_t >= 1 && l_try >= r_last;
        const int l_split = split ? l_try : 0;
        const int v_last0 = q_last * (n_t + 1);
        const int n_visits = v_last0 + (split ? l_split + 1 : n_t + 1);
        const int h_i = wave_all % r_last, h_j = wave_all / r_last;  // this wave in a split round: block and helper index
        // Partner products are added into sX in a fixed order (results do not depend on timing).  The order used to be kept by a
        // workgroup barrier per step; it is the same when wave w only waits for wave w + 1 to have finished the PREVIOUS step
        // -- block a + t was the partner of wave w + 1 one step earlier, and of nobody else since -- which holds whenever the
        // cyclic walk cannot wrap onto a block of the same round (na >= 2 NW: tools/check_pass_chain.py enumerates the
        // schedules).  The waves then drift apart by what their loads cost them and meet at the end of a round only.
        const bool chain = TBK_PASS_CHAIN && with_hemm && na >= 2 * NW;

        struct Visit {
            bool active, diag, own_is_row;
            bool fetch;      // this visit's "partner" loads are the own block's operands (diagonal tile / first slot of a helper)
            bool own_valid;  // this wave holds (partial) accumulators of an own block in this round
            bool last;       // last slot of a round: the accumulators go to sX behind it
            bool in_split;   // slot of a split last round (barrier per slot, helpers flush in turn)
            int I, I2, Ir, Jc;
            d4 tre, tim;
            Frag par;
            double pb[4];  // the partner's Vn operand ([Re | Im] packed), requested with the rest when it is not in LDS
        };
        auto request = [&](int v, Visit& o) {  // issues the global loads of visit v: no waits, and no branches around
            // the loads (the compiler's wait-count tracking gives up at a merge: it then waits for everything in flight);
            // a record that is not `active` loads some valid tile and is ignored
            const int vq = min(v, n_visits - 1);
            int t, a_raw;
            bool tile, own_ok;
            o.in_split = split && vq >= v_last0;
            if (!o.in_split) {
                const int q = vq / (n_t + 1);
                t = vq - q * (n_t + 1);
                a_raw = wave_all + nw_all * q;
                own_ok = a_raw < na;
                tile = own_ok;
                o.fetch = own_ok && t == 0;
                o.last = t == n_t;
            } else {
                const int tau = vq - v_last0;
                a_raw = nw_all * q_last + h_i;
                own_ok = h_j < g_last;
                t = tau == 0 ? 0 : h_j * l_split + tau;
                tile = own_ok && (tau == 0 ? h_j == 0 : t <= n_t);
                o.fetch = own_ok && tau == 0;
                o.last = tau == l_split;
                t = min(t, n_t);  // (an idle slot of the last helper: some valid block)
            }
            const int a = min(a_raw, na - 1);
            o.own_valid = own_ok;
            o.fetch = o.fetch && v < n_visits;
            o.active = v < n_visits && tile && !((na & 1) == 0 && t == n_t && t > 0 && a_raw >= n_t);
            int a2 = a + t;
            if (a2 >= na) a2 -= na;
            o.I = I0 + a;
            o.I2 = I0 + a2;
            o.diag = (t == 0);
            o.own_is_row = o.I <= o.I2;
            o.Ir = o.own_is_row ? o.I : o.I2;
            o.Jc = o.own_is_row ? o.I2 : o.I;
#pragma unroll
            for (int sg = 0; sg < 4; ++sg) {
#ifdef TBK_ABLATE_OPERANDS
                const d2 v2 = VW[((size_t)I0 * 4 + sg) * 64 + lane];
#else
                const d2 v2 = VW[((size_t)o.I2 * 4 + sg) * 64 + lane];
#endif
                o.par.re[sg] = v2[0];
                o.par.im[sg] = v2[1];
            }
            if (!VN_LDS) {
#pragma unroll
                for (int sg = 0; sg < 4; ++sg)
#ifdef TBK_ABLATE_OPERANDS
                    o.pb[sg] = reinterpret_cast<const double*>(sVn)[(size_t)(I0 * TS + lq + 4 * sg) * 16 + 2 * (lrow & 7) + (lrow >> 3)];
#else
                    o.pb[sg] = (reinterpret_cast<const double*>(sVn) + (size_t)o.I2 * (TS * 16) + lane_x)[sg * 64];  // (uniform base, immediates)
#endif
            }
            // clamped addresses; rows / columns beyond n are masked when the tile is used
            // (32-bit element offsets from the matrix' uniform base: a matrix is at most 16 MiB)
            const unsigned gc = (unsigned)min(o.Jc * TS + lrow, n - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned gr = (unsigned)min(o.Ir * TS + lq + 4 * r, n - 1);
                const d2* at = reinterpret_cast<const d2*>(reinterpret_cast<const char*>(H) + (size_t)((gr * (unsigned)n + gc) * 16u));
                // (tiles stream through once per pass: non-temporal, so that they do not push the [V | W] blocks, which every
                // visit re-reads, out of L2)
                const d2 v2 = TBK_TILE_NT ? __builtin_nontemporal_load(at) : *at;
                o.tre[r] = v2[0];
                o.tim[r] = v2[1];
            }
        };

        Visit va, vb;
        Frag own;
        double own_b[4];
        d4 own1 = (d4){0.0, 0.0, 0.0, 0.0}, own2 = own1;
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) own.re[sg] = own.im[sg] = own_b[sg] = 0.0;
        // one visit: everything between the arrival of its operands and the step's meeting point
        auto visit = [&](const Visit& cur, int v) {
            const bool chain_here = chain && !cur.in_split;
            TBK_CLK(7);
            if (cur.fetch) {  // first slot of an own block: its operands are this visit's "partner" loads
                own = cur.par;
                own1 = (d4){0.0, 0.0, 0.0, 0.0};
                own2 = own1;
#pragma unroll
                for (int sg = 0; sg < 4; ++sg)
                    own_b[sg] = VN_LDS ? (reinterpret_cast<const double*>(sVn) + cur.I * (TS * 16) + lane_x)[sg * 64] : cur.pb[sg];
            }
            if (cur.active) {
                const bool diag = cur.diag, own_is_row = cur.own_is_row;
                const int I2 = cur.I2, Ir = cur.Ir, Jc = cur.Jc;
                double par_b[4];
#pragma unroll
                for (int sg = 0; sg < 4; ++sg)
                    par_b[sg] = diag ? own_b[sg] : (VN_LDS ? (reinterpret_cast<const double*>(sVn) + I2 * (TS * 16) + lane_x)[sg * 64] : cur.pb[sg]);
                d4 tre = cur.tre, tim = cur.tim;
                const int gc = Jc * TS + lrow;
                // (everything the vector unit does here is time the matrix pipe does not get: the masks of the tiles on the
                // matrix' edge and the roles of the two blocks are wave-uniform facts, so they are branches, not 48 selects)
                const bool interior = (Ir + 1) * TS <= n && (Jc + 1) * TS <= n;
                if (!interior) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool inside = Ir * TS + lq + 4 * r < n && gc < n;
                        tre[r] = inside ? tre[r] : 0.0;
                        tim[r] = inside ? tim[r] : 0.0;
                    }
                }
                if (with_update) {
                    // tile -= [V | W]_row . ([W | V]_col)^H : A = row block, k-step sg; B = conj(col block, k-step (sg + 2) % 4)
                    if (own_is_row) {
#pragma unroll
                        for (int sg = 0; sg < 4; ++sg) {
                            const int sb = (sg + 2) & 3;
                            tre = __builtin_amdgcn_mfma_f64_16x16x4f64(own.re[sg], cur.par.re[sb], tre, 0, 0, 1);  // -ar br
                            tre = __builtin_amdgcn_mfma_f64_16x16x4f64(own.im[sg], cur.par.im[sb], tre, 0, 0, 1);  // -ai bi
                            tim = __builtin_amdgcn_mfma_f64_16x16x4f64(own.im[sg], cur.par.re[sb], tim, 0, 0, 1);  // -ai br
                            tim = __builtin_amdgcn_mfma_f64_16x16x4f64(own.re[sg], cur.par.im[sb], tim, 0, 0, 0);  // +ar bi
                        }
                    } else {
#pragma unroll
                        for (int sg = 0; sg < 4; ++sg) {
                            const int sb = (sg + 2) & 3;
                            tre = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.par.re[sg], own.re[sb], tre, 0, 0, 1);
                            tre = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.par.im[sg], own.im[sb], tre, 0, 0, 1);
                            tim = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.par.im[sg], own.re[sb], tim, 0, 0, 1);
                            tim = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.par.re[sg], own.im[sb], tim, 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int gr = Ir * TS + lq + 4 * r;
#ifdef TBK_ABLATE_STORES
                        if (gr < n && gc < n && tre[r] == 1.2345e300) {
#elif defined(TBK_ABLATE_STORES_ALT)
                        if (gr < n && gc < n && ((s / PB) & 1) == 0) {  // stores on every SECOND panel only
#else
                        if (interior || (gr < n && gc < n)) {
#endif
                            d2* at = reinterpret_cast<d2*>(reinterpret_cast<char*>(H) + (size_t)(((unsigned)gr * (unsigned)n + (unsigned)gc) * 16u));
                            if (TBK_TILE_NT)
                                __builtin_nontemporal_store((d2){tre[r], tim[r]}, at);
                            else
                                *at = (d2){tre[r], tim[r]};
                        }
                    }
                }
                TBK_CLK(8);
                if (with_hemm) {
                    // transposed copy [lrow][lq + 4 sg] through this wave's LDS plane, real part then imaginary part: a
                    // wave's LDS operations execute in order, so the plane is reused without waiting in between
                    // (compiler fences only: no hardware wait between the groups; `volatile` accesses would each get an
                    // s_waitcnt vmcnt(0), i.e. wait for this visit's tile stores)
                    double ttre[4], ttim[4];
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int r = 0; r < 4; ++r) tr[(lq + 4 * r) * 17 + lrow] = tre[r];
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int sg = 0; sg < 4; ++sg) ttre[sg] = tr[lrow * 17 + lq + 4 * sg];
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int r = 0; r < 4; ++r) tr[(lq + 4 * r) * 17 + lrow] = tim[r];
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int sg = 0; sg < 4; ++sg) ttim[sg] = tr[lrow * 17 + lq + 4 * sg];
                    asm volatile("" ::: "memory");
                    TBK_CLK(9);
                    if (diag) {
                        // Hermitian tile of which only the upper part is valid: operand element [i = lrow][j = lq + 4 sg]
                        // is the transposed copy where i <= j, the conjugate of the accumulator element otherwise
#pragma unroll
                        for (int sg = 0; sg < 4; ++sg) {
                            const bool upper = lrow <= lq + 4 * sg;
                            const double ar = upper ? ttre[sg] : tre[sg];
                            const double ai = upper ? ttim[sg] : -tim[sg];
                            own1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, own_b[sg], own1, 0, 0, 0);
                            own2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, own_b[sg], own2, 0, 0, 0);
                        }
                    } else {
                        // row part  X_Ir += tile Vn_Jc ;  column part  X_Jc += tile^H Vn_Ir.  The own block's part is
                        // accumulated by the MFMAs themselves (round 4: separate product registers added afterwards were
                        // 16 more live registers and 8 more vector-unit adds per visit); the partner's part goes to LDS
                        d4 o1 = (d4){0.0, 0.0, 0.0, 0.0}, o2 = o1;
                        if (own_is_row) {
#pragma unroll
                            for (int sg = 0; sg < 4; ++sg) {
                                own1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ttre[sg], par_b[sg], own1, 0, 0, 0);
                                own2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ttim[sg], par_b[sg], own2, 0, 0, 0);
                                o1 = __builtin_amdgcn_mfma_f64_16x16x4f64(tre[sg], own_b[sg], o1, 0, 0, 0);
                                o2 = __builtin_amdgcn_mfma_f64_16x16x4f64(tim[sg], own_b[sg], o2, 0, 0, 1);  // conj
                            }
                        } else {
#pragma unroll
                            for (int sg = 0; sg < 4; ++sg) {
                                o1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ttre[sg], own_b[sg], o1, 0, 0, 0);
                                o2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ttim[sg], own_b[sg], o2, 0, 0, 0);
                                own1 = __builtin_amdgcn_mfma_f64_16x16x4f64(tre[sg], par_b[sg], own1, 0, 0, 0);
                                own2 = __builtin_amdgcn_mfma_f64_16x16x4f64(tim[sg], par_b[sg], own2, 0, 0, 1);  // conj
                            }
                        }
                        // partner block: lane (row lq + 4 r, c = lrow) adds Re X[row][c] (c < 8) or Im X[row][c - 8]
                        double* xs = reinterpret_cast<double*>(sX) + I2 * (TS * 16) + lane_x;
                        if (chain_here && wave + 1 < NW) {  // wave + 1 is done with this block (its partner one step ago)
                            const int need = prog_base + v;
                            while (lds_poll(sProg + wave + 1) < need) __builtin_amdgcn_s_sleep(1);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const double rot = dpp_mov<0x128>(o2[r]);  // the other half of the 16-lane row
                            xs[r * 64] += fma(rot, lane_sgn, o1[r]);   // o1 - rot (Re columns) / o1 + rot (Im columns): exact either way
                        }
                    }
                }
            }
            TBK_CLK(10);
            // the step's meeting point: LDS only (this wave's tile stores drain in the background)
#ifndef TBK_ABLATE_BARRIER
            if (chain_here && !cur.last) {
                // (a wave's LDS operations are performed in order: the counter lands behind the sums it announces)
                lds_post(sProg + wave, prog_base + v + 1);  // (all lanes, one address, one value)
            } else {
                lds_fence();
                __syncthreads();
            }
#endif
            TBK_CLK(11);
            if (with_hemm && cur.last) {  // last slot of a round: the own blocks' accumulators go to sX
                // (a split round: the helpers of a block one after the other, in helper order)
                const int turns = cur.in_split ? g_last : 1;
                for (int turn = 0; turn < turns; ++turn) {
                    if (cur.own_valid && (!cur.in_split || h_j == turn)) {
                        double* xs = reinterpret_cast<double*>(sX) + cur.I * (TS * 16) + lane_x;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const double rot = dpp_mov<0x128>(own2[r]);
                            xs[r * 64] += fma(rot, lane_sgn, own1[r]);
                        }
                    }
                    lds_fence();
                    __syncthreads();
                }
            }
        };
        // two visit records in turn (no register copies: a copy of a record waits for its loads where it stands)
        request(0, va);
        for (int v = 0; v < n_visits; v += 2) {
            request(v + 1, vb);
            visit(va, v);
            if (v + 1 < n_visits) {
                request(v + 2, va);
                visit(vb, v + 1);
            }
        }
        prog_base += n_visits;
        // The pass ends on an LDS-only meeting (X is complete).  Its tile stores drain under whatever follows: nobody reads
        // the matrix before the next meeting that waits for the stores -- the end of the W phase in the panel loop, the
        // explicit one behind the last pass, the end of the kernel in the launch chain (round 5: the wait for the stores
        // here was a memory round trip per panel with nothing to do).
        lds_fence();
        __syncthreads();
    };

    // thread t <-> global rows / columns t + rr NT, rr < ROWS, in the thread-per-row phases
    auto row_of = [&](int rr) { return tid + rr * NT; };
    if (PHASE == 2) {
        // the tile pass of panel p_fixed on this member's share of the own blocks; its partial X goes to global memory.
        // (Behind the last panel: the last pending update, no products.)
        const int s = PB * (p_fixed + 1);
        if (n - s >= 2) {
            for (int i = tid; i < npad * PB; i += NT) sX[i] = (d2){0.0, 0.0};
            wg_sync();
            big_pass(s, have_update, true);
            d2* mine = gX + (size_t)member * npad * PB;
            for (int i = tid; i < npad * PB; i += NT) mine[i] = sX[i];
        } else {
            big_pass(PB * p_fixed, true, false);
        }
        return;
    }
    // PHASE 1 enters the loop in the MIDDLE of iteration p_fixed - 1 (its W phase) and leaves in front of the pass of p_fixed
    bool resume_w = PHASE == 1 && p_fixed > 0;
    int p = PHASE == 1 ? max(p_fixed - 1, 0) : 0;
    for (;; ++p) {
        const int g0 = PB * p;       // first row of the panel
        const int s = g0 + PB;       // start of the trailing matrix behind it
        const int m = n - s;
        if (m < 2) break;
        bool qr_row[ROWS];           // rows of the trailing matrix behind the panel
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) qr_row[rr] = row_of(rr) >= s && row_of(rr) < n;
        TBK_CLK(6);
        if (!resume_w) {  // (look-ahead .. tile pass: not indented)
        // ---- look-ahead: block row p (8 rows, columns >= 8 p) brought up to date with the pending (V, W) ----
        // the 8 pending rows [V | W][g0 + r][0 .. 15]: lane t of every row of 16 lanes holds entry t, and the FMAs below
        // take it from there (row_newbcast) -- through LDS they were 128 broadcast reads per thread and panel, and two
        // more workgroup barriers (the staging area is the partial-sum area of the reductions).
        // (No meeting here: the W phase of the previous panel ended on one that waited for its stores of VW and for the
        // pass' stores of the tiles, and the first panel reads nothing anybody wrote.)
        asm volatile("" ::: "memory");  // (compiler fence: the loads below stay here)
        int lane_la = lane;  // (opaque copy: what is derived from the lane number here does not live across the tile pass)
        asm volatile("" : "+v"(lane_la));
        d2 pend[PB];
#pragma unroll
        for (int r = 0; r < PB; ++r) pend[r] = have_update ? VW[vw_index(min(g0 + r, n - 1), lane_la & 15)] : (d2){0.0, 0.0};
        d2 x[ROWS][PB];
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            const int i_row = row_of(rr);
#pragma unroll
            for (int r = 0; r < PB; ++r) x[rr][r] = (d2){0.0, 0.0};
            const bool in_rows = i_row >= g0 && i_row < n;
            if (in_rows) {
#pragma unroll
                for (int r = 0; r < PB; ++r) {
                    const int g = g0 + r;
                    if (g < n) {  // uniform; one unconditional load of the stored (upper) element either way
                        const bool upper = i_row >= g;
                        const d2 v = *Hat(upper ? g : i_row, upper ? i_row : g);
                        x[rr][r] = upper ? v : conjd(v);
                    }
                }
            }
            // (every lane of the wave takes part: a row_newbcast operand is read from lane t whatever this lane's row is,
            // and a lane that is switched off supplies nothing; waves without a row of the panel's range skip the lot)
            if (have_update && __any(in_rows)) {
                d2 vw[16];
                const int i_clamped = min(max(i_row, g0), n - 1);
#pragma unroll
                for (int c = 0; c < 16; ++c) vw[c] = VW[vw_index(i_clamped, c)];
                static_for<0, PB>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    static_for<0, PB>([&](auto tc) {
                        constexpr int t = decltype(tc)::value;
                        cfnmac_bc<t>(x[rr][r], pend[r], vw[PB + t]);       // - V[g][t] conj(W[i][t])
                        cfnmac_bc<PB + t>(x[rr][r], pend[r], vw[t]);       // - W[g][t] conj(V[i][t])
                    });
                });
            }
            // the diagonal block is final
            if (in_rows && i_row < s) {
#pragma unroll
                for (int r = 0; r < PB; ++r)
                    if (g0 + r <= i_row) *Hat(g0 + r, i_row) = x[rr][r];
            }
        }
        TBK_CLK(0);
        // ---- Householder QR of the panel on rows i >= s: y = conj(x) (model: panel_qr) ----
        d2 y[ROWS][PB], vn[ROWS][PB];
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
#pragma unroll
            for (int c = 0; c < PB; ++c) {
                y[rr][c] = qr_row[rr] ? conjd(x[rr][c]) : (d2){0.0, 0.0};
                vn[rr][c] = (d2){0.0, 0.0};
            }
        }
        if (tid < PB) sTau[tid] = (d2){0.0, 0.0};  // (read after the barriers of the Gram sums below)
        if constexpr (GRAM2) {
#if TBK_PANEL_GRAM
            // The panel's rows go to the X area (every thread its own rows; zero outside the trailing rows), the registers are
            // free for the recurrence.  One round = Gram matrix of the rows where they lie, the recurrence on every wave (as in the
            // one-row form, but the coefficients of ALL its reflectors are kept: f per lane, the wave-uniform scale and beta in
            // scalar registers), then every row passes through the registers once: its reflectors are applied, its entry of R
            // goes to the matrix, its row of V to global memory AND back into the X area -- the T factor's sum reads it there.
#pragma unroll
            for (int rr = 0; rr < ROWS; ++rr) {
                if (row_of(rr) < npad) {
#pragma unroll
                    for (int c = 0; c < PB; ++c) sX[(size_t)row_of(rr) * PB + c] = y[rr][c];
                }
            }
            const int last = min(PB, m - 1);
            int lane_here = lane;  // (see gram_direct: nothing derived from the lane number may live across the tile pass)
            asm volatile("" : "+v"(lane_here));
            const int t8 = lane_here & 7;
            int c0 = 0;
            while (c0 < last) {
                d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int rr = 0; rr < ROWS; ++rr) {  // this wave's own rows (its threads wrote them: no meeting in front)
                    const int base_row = wave * 64 + rr * NT;
                    if (base_row + 64 <= s + c0 || base_row >= n) continue;  // wave-uniform: nothing of the sum here
                    gram_direct(sX, nullptr, nullptr, base_row, s + c0, acc);
                }
                TBK_CLK(12);
                gram_finish(acc);
                TBK_CLK(13);
                if (c0 == 0 && have_update && tid < 128) VW[vw_index(g0 + (tid >> 4), tid & 15)] = (d2){0.0, 0.0};
                d2 top[PB];
#pragma unroll
                for (int c = 0; c < PB; ++c) top[c] = (c >= c0 && c < m) ? sX[(size_t)min(s + c, npad - 1) * PB + t8] : (d2){0.0, 0.0};
                d2 g_next = sG[min(c0, PB - 1) * PB + t8];
                bool stopped = false;
                int c1 = last;
                // the coefficients of the round's reflectors wait in LDS for the rows (every wave writes the same values and
                // reads its own writes back): f per column in the S area, (scale, beta, applied?) in the row buffer
                d2* const sF = sS;
                unsigned has_mask = 0;
                static_for<0, PB>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    if (c >= c0 && c < last && !stopped) {  // uniform
                        const d2 g_row = g_next;
                        g_next = sG[min(c + 1, PB - 1) * PB + t8];
                        d2 g = g_row;
                        static_for<0, c>([&](auto ic) {
                            constexpr int i = decltype(ic)::value;
                            cfnmacj_bc<c>(g, top[i], top[i]);
                        });
                        const double gcc = lane_value<c>(g[0]);
                        const double Gcc = lane_value<c>(g_row[0]);
                        if (c > c0 && !(gcc >= GRAM_THRESH * Gcc)) {
                            stopped = true;
                            c1 = c;
                        } else {
                            const d2 alpha = (d2){lane_value<c>(top[c][0]), lane_value<c>(top[c][1])};
                            const d2 rowv = top[c];
                            const double sigma = gcc - (alpha[0] * alpha[0] + alpha[1] * alpha[1]);
                            if (!(gcc == 0.0 || (sigma == 0.0 && alpha[1] == 0.0))) {  // uniform
                                double root, rroot;
                                fast_sqrt_rsqrt(gcc, root, rroot);
                                const double beta = -copysign(root, alpha[0]);
                                const double rbeta = -copysign(rroot, alpha[0]);
                                const d2 tau_c = (d2){(beta - alpha[0]) * rbeta, -alpha[1] * rbeta};
                                if (tid == 0) sTau[c] = tau_c;
                                const double qr_ = alpha[0] - beta, qi_ = alpha[1];
                                const double qn = fast_rcp(qr_ * qr_ + qi_ * qi_);
                                const d2 scale = (d2){qr_ * qn, -qi_ * qn};
                                d2 tz = g;
                                cfnmac(tz, rowv, alpha);
                                d2 z = cmulc(tz, scale);
                                z[0] += rowv[0];
                                z[1] += rowv[1];
                                d2 f = cmul(conjd(tau_c), z);
                                if (t8 <= c) f = (d2){0.0, 0.0};
                                top[c] = t8 > c ? (d2){rowv[0] - f[0], rowv[1] - f[1]} : (t8 == c ? (d2){beta, 0.0} : (d2){0.0, 0.0});
                                static_for<c + 1, PB>([&](auto ic) {
                                    constexpr int i = decltype(ic)::value;
                                    d2 vt = (d2){0.0, 0.0};
                                    cfma_bc<c>(vt, scale, top[i]);
                                    cfma(top[i], (d2){-vt[0], -vt[1]}, f);
                                });
                                sF[c * PB + t8] = f;
                                sRow[c] = scale;
                                sRow[PB + c] = (d2){beta, 0.0};
                                has_mask |= 1u << c;
                            }
                        }
                    }
                });
                TBK_CLK(14);
                // the rows, one at a time through the registers
#pragma unroll
                for (int rr = 0; rr < ROWS; ++rr) {
                    const int i_row = row_of(rr);
                    // (wave-uniform.  Rows of the block the trailing matrix starts in are rewritten although they are finished:
                    // the pass reads whole blocks of V, and their rows of the previous panel's V must become zero)
                    if (!__any(i_row < npad && i_row >= (s & ~(TS - 1)))) continue;
                    const int i_at = min(i_row, npad - 1);
                    d2 yr[PB], vrow[PB];
#pragma unroll
                    for (int c = 0; c < PB; ++c) {
                        yr[c] = sX[(size_t)i_at * PB + c];
                        vrow[c] = (d2){0.0, 0.0};
                    }
                    static_for<0, PB>([&](auto cc) {
                        constexpr int c = decltype(cc)::value;
                        if (c >= c0 && c < c1 && (has_mask >> c & 1u)) {  // uniform
                            const bool below = qr_row[rr] && i_row >= s + c;
                            const bool head = i_row == s + c;
                            const d2 f_c = sF[c * PB + t8];
                            const d2 sc_c = sRow[c];
                            const double beta_c = sRow[PB + c][0];
                            d2 v = cmul(yr[c], sc_c);
                            v = below ? (head ? (d2){1.0, 0.0} : v) : (d2){0.0, 0.0};
                            vrow[c] = v;
                            static_for<c + 1, PB>([&](auto cpc) {
                                constexpr int cp = decltype(cpc)::value;
                                cfnma_bc<cp>(yr[cp], v, f_c);  // (every lane takes part: v is zero outside the rows)
                            });
                            if (below) yr[c] = head ? (d2){beta_c, 0.0} : (d2){0.0, 0.0};
                        }
                    });
                    // the thread of row s + c holds row c of R (final once reflector c is through): column s + c of the block row
                    // (rows s + c with c >= last have no reflector of their own: they are final once the round's are through)
                    if (qr_row[rr] && i_row - s < PB && ((i_row - s >= c0 && i_row - s < c1) || (c1 >= last && i_row - s >= last))) {
                        const int c = i_row - s;
#pragma unroll
                        for (int r = 0; r < PB; ++r)
                            if (g0 + r < n) *Hat(g0 + r, i_row) = (r >= c) ? conjd(yr[r]) : (d2){0.0, 0.0};
                    }
                    if (i_row < npad) {
                        // columns of this round: the row of V (global memory for the pass, the X area for the T factor's sum);
                        // behind them what is left of the panel for the next round -- or, behind the last column with a row
                        // below the diagonal, zeros
#pragma unroll
                        for (int c = 0; c < PB; ++c) {
                            if (c >= c0 && c < c1) {
                                sX[(size_t)i_row * PB + c] = vrow[c];
                                sVn[(size_t)i_row * PB + c] = vrow[c];
                            } else if (c >= c1) {
                                const d2 keep = c1 < last ? yr[c] : (d2){0.0, 0.0};
                                sX[(size_t)i_row * PB + c] = keep;
                                if (c1 >= last) sVn[(size_t)i_row * PB + c] = (d2){0.0, 0.0};
                            }
                        }
                    }
                }
                c0 = c1;
                if (c0 < last) {
                    lds_fence();
                    __syncthreads();
                }
            }
            lds_fence();
            __syncthreads();
#endif
        } else if constexpr (GRAM) {
#if TBK_PANEL_GRAM
            // All reflectors of a round from ONE Gram matrix (model: panel_qr_gram).  Every 16-lane row of every wave runs the
            // recurrence for itself -- lane t holds column t % 8 of G and of the tracked top rows, values every lane needs come
            // out of lane c as scalars (v_readlane) or as row_newbcast operands -- and each reflector is applied to the rows as
            // soon as its coefficients exist: no meeting, no LDS traffic inside a round.
            const int last = min(PB, m - 1);  // columns c < last have a row below the diagonal
            d2* const sTop = sS;              // [8][8] rows s + i of the panel (S only lives inside the W phase)
            const int t8 = lane & 7;
            int c0 = 0;
            while (c0 < last) {  // rounds: one, unless a column's remaining norm cancels (structured matrices)
                d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int rr = 0; rr < ROWS; ++rr) {
                    const int i_top = row_of(rr) - s;
                    if (qr_row[rr] && i_top >= c0 && i_top < PB) {
#pragma unroll
                        for (int cp = 0; cp < PB; ++cp) sTop[i_top * PB + cp] = y[rr][cp];
                    }
                    const bool in_sum = qr_row[rr] && i_top >= c0;  // (rows above c0 are finished rows of R)
                    if (!__any(in_sum)) continue;                  // wave-uniform
                    d2 op[PB];
#pragma unroll
                    for (int cp = 0; cp < PB; ++cp) op[cp] = in_sum ? y[rr][cp] : (d2){0.0, 0.0};
                    gram_rows(op, op, true, acc);
                }
                TBK_CLK(12);  // QR: transposition + Gram products
                gram_finish(acc);
                // everybody is through the look-ahead: the pending rows it consumed are zeroed HERE, so that the stores are
                // long done when the hand-over waits for them (issued there, their round trip was exposed)
                if (c0 == 0 && have_update && tid < 128) VW[vw_index(g0 + (tid >> 4), tid & 15)] = (d2){0.0, 0.0};
                TBK_CLK(13);  // QR: the meeting
                // (row c of G is read one step ahead of its use: all eight rows held from the start were 14 more live registers)
                d2 top[PB];
#pragma unroll
                for (int c = 0; c < PB; ++c) top[c] = (c >= c0 && c < m) ? sTop[c * PB + t8] : (d2){0.0, 0.0};
#if TBK_QR_ONE_WAVE
                // Round 6: the recurrence on wave 0 ALONE -- its coefficients (f per column in the T area,
                // which is dead until the T block below; scale and beta in the row buffer; the round's end and the mask of live
                // reflectors in the totals area) wait in LDS behind one more meeting, then every wave applies the reflectors to
                // its rows.  The other three waves' SIMDs are free for the co-resident workgroup meanwhile.  Same operations on
                // the same values as the every-wave form: the same bits.
                d2* const sF = sT;
                int* const sCtl = reinterpret_cast<int*>(sTot);
                int c1 = last;
                unsigned has_mask = 0;
                if (wave == 0) {
                    d2 g_next = sG[min(c0, PB - 1) * PB + t8];
                    bool stopped = false;
                    static_for<0, PB>([&](auto cc) {
                        constexpr int c = decltype(cc)::value;
                        if (c >= c0 && c < last && !stopped) {  // uniform
                            const d2 g_row = g_next;
                            g_next = sG[min(c + 1, PB - 1) * PB + t8];
                            d2 g = g_row;
                            static_for<0, c>([&](auto ic) {
                                constexpr int i = decltype(ic)::value;
                                cfnmacj_bc<c>(g, top[i], top[i]);
                            });
                            const double gcc = lane_value<c>(g[0]);
                            const double Gcc = lane_value<c>(g_row[0]);
                            if (c > c0 && !(gcc >= GRAM_THRESH * Gcc)) {
                                stopped = true;
                                c1 = c;
                            } else {
                                const d2 alpha = (d2){lane_value<c>(top[c][0]), lane_value<c>(top[c][1])};
                                const d2 rowv = top[c];
                                const double sigma = gcc - (alpha[0] * alpha[0] + alpha[1] * alpha[1]);
                                if (!(gcc == 0.0 || (sigma == 0.0 && alpha[1] == 0.0))) {  // uniform
                                    double root, rroot;
                                    fast_sqrt_rsqrt(gcc, root, rroot);
                                    const double beta = -copysign(root, alpha[0]);
                                    const double rbeta = -copysign(rroot, alpha[0]);
                                    const d2 tau_c = (d2){(beta - alpha[0]) * rbeta, -alpha[1] * rbeta};
                                    if (tid == 0) sTau[c] = tau_c;
                                    const double qr_ = alpha[0] - beta, qi_ = alpha[1];
                                    const double qn = fast_rcp(qr_ * qr_ + qi_ * qi_);
                                    const d2 scale = (d2){qr_ * qn, -qi_ * qn};
                                    d2 tz = g;
                                    cfnmac(tz, rowv, alpha);
                                    d2 z = cmulc(tz, scale);
                                    z[0] += rowv[0];
                                    z[1] += rowv[1];
                                    d2 f = cmul(conjd(tau_c), z);
                                    if (t8 <= c) f = (d2){0.0, 0.0};
                                    top[c] = t8 > c ? (d2){rowv[0] - f[0], rowv[1] - f[1]} : (t8 == c ? (d2){beta, 0.0} : (d2){0.0, 0.0});
                                    static_for<c + 1, PB>([&](auto ic) {
                                        constexpr int i = decltype(ic)::value;
                                        d2 vt = (d2){0.0, 0.0};
                                        cfma_bc<c>(vt, scale, top[i]);
                                        cfma(top[i], (d2){-vt[0], -vt[1]}, f);
                                    });
                                    sF[c * PB + t8] = f;  // (the eight 16-lane rows of the wave write the same values)
                                    sRow[c] = scale;
                                    sRow[PB + c] = (d2){beta, 0.0};
                                    has_mask |= 1u << c;
                                }
                            }
                        }
                    });
                    if (lane == 0) {
                        sCtl[0] = c1;
                        sCtl[1] = (int)has_mask;
                    }
                }
                lds_fence();
                __syncthreads();
                c1 = __builtin_amdgcn_readfirstlane(sCtl[0]);
                has_mask = (unsigned)__builtin_amdgcn_readfirstlane(sCtl[1]);
                static_for<0, PB>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    if (c >= c0 && c < c1 && (has_mask >> c & 1u)) {  // uniform
                        const d2 f = sF[c * PB + t8];
                        const d2 scale = sRow[c];
                        const double beta = sRow[PB + c][0];
#pragma unroll
                        for (int rr = 0; rr < ROWS; ++rr) {
                            const bool below = qr_row[rr] && row_of(rr) >= s + c;
                            const bool head = row_of(rr) == s + c;
                            d2 v = cmul(y[rr][c], scale);
                            v = below ? (head ? (d2){1.0, 0.0} : v) : (d2){0.0, 0.0};
                            vn[rr][c] = v;
                            static_for<c + 1, PB>([&](auto cpc) {
                                constexpr int cp = decltype(cpc)::value;
                                cfnma_bc<cp>(y[rr][cp], v, f);
                            });
                            if (below) y[rr][c] = head ? (d2){beta, 0.0} : (d2){0.0, 0.0};
                        }
                    }
                });
                TBK_CLK(14);
                c0 = c1;
                // (another round, or the T block's Gram matrix: both write areas read above -- the meeting below the loop /
                // the one here orders them)
                if (c0 < last) {
                    lds_fence();
                    __syncthreads();
                }
#else
                d2 g_next = sG[min(c0, PB - 1) * PB + t8];
                bool stopped = false;
                int c1 = last;
                static_for<0, PB>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    if (c >= c0 && c < last && !stopped) {  // uniform
                        // g[t] = sum over the rows from s + c on of conj(y_c) y_t = G[c][t] - sum_{i < c} conj(R[i][c]) R[i][t]
                        const d2 g_row = g_next;
                        g_next = sG[min(c + 1, PB - 1) * PB + t8];
                        d2 g = g_row;
                        static_for<0, c>([&](auto ic) {
                            constexpr int i = decltype(ic)::value;
                            cfnmacj_bc<c>(g, top[i], top[i]);  // (rows above c0: zero)
                        });
                        const double gcc = lane_value<c>(g[0]);
                        const double Gcc = lane_value<c>(g_row[0]);
                        if (c > c0 && !(gcc >= GRAM_THRESH * Gcc)) {  // cancelled: the round ends in front of this column
                            stopped = true;
                            c1 = c;
                        } else {
                            const d2 alpha = (d2){lane_value<c>(top[c][0]), lane_value<c>(top[c][1])};
                            const d2 rowv = top[c];
                            const double sigma = gcc - (alpha[0] * alpha[0] + alpha[1] * alpha[1]);
                            if (!(gcc == 0.0 || (sigma == 0.0 && alpha[1] == 0.0))) {  // uniform
                                double root, rroot;
                                fast_sqrt_rsqrt(gcc, root, rroot);
                                const double beta = -copysign(root, alpha[0]);
                                const double rbeta = -copysign(rroot, alpha[0]);
                                const d2 tau_c = (d2){(beta - alpha[0]) * rbeta, -alpha[1] * rbeta};
                                if (tid == 0) sTau[c] = tau_c;
                                const double qr_ = alpha[0] - beta, qi_ = alpha[1];
                                const double qn = fast_rcp(qr_ * qr_ + qi_ * qi_);
                                const d2 scale = (d2){qr_ * qn, -qi_ * qn};  // 1 / (alpha - beta)
                                // z = v^H y_t = conj(scale) (g - conj(alpha) row) + row;  f = conj(tau) z, columns t > c
                                d2 tz = g;
                                cfnmac(tz, rowv, alpha);
                                d2 z = cmulc(tz, scale);
                                z[0] += rowv[0];
                                z[1] += rowv[1];
                                d2 f = cmul(conjd(tau_c), z);
                                if (t8 <= c) f = (d2){0.0, 0.0};
                                // row c of R; the tracked rows below it
                                top[c] = t8 > c ? (d2){rowv[0] - f[0], rowv[1] - f[1]} : (t8 == c ? (d2){beta, 0.0} : (d2){0.0, 0.0});
                                static_for<c + 1, PB>([&](auto ic) {
                                    constexpr int i = decltype(ic)::value;
                                    d2 vt = (d2){0.0, 0.0};
                                    cfma_bc<c>(vt, scale, top[i]);  // v[i] = scale y[i][c]
                                    cfma(top[i], (d2){-vt[0], -vt[1]}, f);
                                });
                                // the rows' side, every row for itself: v = scale y_c, y_t -= v f[t]
#pragma unroll
                                for (int rr = 0; rr < ROWS; ++rr) {
                                    const bool below = qr_row[rr] && row_of(rr) >= s + c;
                                    const bool head = row_of(rr) == s + c;
                                    d2 v = cmul(y[rr][c], scale);
                                    v = below ? (head ? (d2){1.0, 0.0} : v) : (d2){0.0, 0.0};
                                    vn[rr][c] = v;
                                    static_for<c + 1, PB>([&](auto cpc) {
                                        constexpr int cp = decltype(cpc)::value;
                                        cfnma_bc<cp>(y[rr][cp], v, f);  // (every lane takes part: v is zero outside the rows)
                                    });
                                    if (below) y[rr][c] = head ? (d2){beta, 0.0} : (d2){0.0, 0.0};
                                }
                            }
                        }
                    }
                });
                TBK_CLK(14);  // QR: recurrence + rows
                c0 = c1;
                if (c0 < last) {  // (rare) another round: the top rows and partial areas are written again
                    lds_fence();
                    __syncthreads();
                }
#endif
            }
            // (sTop = the S area and sG = the partial-sum area are written again by the T block's Gram matrix below)
            lds_fence();
            __syncthreads();
#endif
        } else {
#pragma unroll
        for (int c = 0; c < PB; ++c) {
            if (c <= m - 2) {  // uniform: a row below the diagonal exists
                bool below[ROWS];
                double pv[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) pv[k] = 0.0;
                d2* row_buf = sRow + (c & 1) * PB;  // (a fast thread writes row c + 1 while a slow one still reads row c)
#pragma unroll
                for (int rr = 0; rr < ROWS; ++rr) {
                    below[rr] = qr_row[rr] && row_of(rr) >= s + c;
                    if (below[rr]) {
                        pv[0] += y[rr][c][0] * y[rr][c][0] + y[rr][c][1] * y[rr][c][1];
#pragma unroll
                        for (int cp = c + 1; cp < PB; ++cp) {
                            const d2 t = cmulc(y[rr][cp], y[rr][c]);  // conj(y_c) y_cp
                            pv[1 + 2 * (cp - c - 1)] += t[0];
                            pv[2 + 2 * (cp - c - 1)] += t[1];
                        }
                    }
                    if (row_of(rr) == s + c) {
#pragma unroll
                        for (int cp = 0; cp < PB; ++cp) row_buf[cp] = y[rr][cp];
                    }
                }
                TBK_CLK(12);  // QR: products
                // ONE meeting per step: every wave leaves its 16 partial sums (alternating halves of its row of the
                // partial-sum area), and every thread adds the waves' partials itself, in wave order
                double* part = sPart + (c & 1) * 16;
                // (reflector c needs 1 + 2 (7 - c) of the sixteen sums: whole groups of four beyond them are not reduced --
                // 20 instead of 32 four-value reductions per panel)
#pragma unroll
                for (int k4 = 0; k4 < 16; k4 += 4)
                    if (k4 < 1 + 2 * (PB - 1 - c)) wave_partial4(k4, pv[k4], pv[k4 + 1], pv[k4 + 2], pv[k4 + 3], part, lane, wave);
                TBK_CLK(13);  // QR: wave sums
                lds_fence();
                __syncthreads();
                TBK_CLK(14);  // QR: barrier
                // lane k < 16 of every wave adds the waves' partials of sum k (one read per wave of partials, different
                // addresses), and a total reaches the other lanes as a scalar (`v_readlane`): with every thread adding
                // all sixteen totals itself a step was 32 broadcast reads per wave on the LDS pipe two workgroups share
                double mine = 0.0;
#pragma unroll
                for (int w = 0; w < NW; ++w) mine += part[w * 64 + (lane & 15)];
                auto total = [&](int k) {
                    const int lo = __builtin_amdgcn_readlane(__double2loint(mine), k);
                    const int hi = __builtin_amdgcn_readlane(__double2hiint(mine), k);
                    return __hiloint2double(hi, lo);
                };
                const double gcc = total(0);
                const d2 alpha = row_buf[c];
                const double sigma = gcc - (alpha[0] * alpha[0] + alpha[1] * alpha[1]);
                if (!(gcc == 0.0 || (sigma == 0.0 && alpha[1] == 0.0))) {  // uniform
                    double root, rroot;
                    fast_sqrt_rsqrt(gcc, root, rroot);
                    const double beta = -copysign(root, alpha[0]);
                    const double rbeta = -copysign(rroot, alpha[0]);
                    const d2 tau_c = (d2){(beta - alpha[0]) * rbeta, -alpha[1] * rbeta};
                    if (tid == 0) sTau[c] = tau_c;
                    const double qr_ = alpha[0] - beta, qi_ = alpha[1];
                    const double qn = fast_rcp(qr_ * qr_ + qi_ * qi_);
                    const d2 scale = (d2){qr_ * qn, -qi_ * qn};  // 1 / (alpha - beta)
#pragma unroll
                    for (int rr = 0; rr < ROWS; ++rr) {
                        d2 v = (d2){0.0, 0.0};
                        if (below[rr]) v = (row_of(rr) == s + c) ? (d2){1.0, 0.0} : cmul(y[rr][c], scale);
                        vn[rr][c] = v;
                    }
                    const d2 ctau = conjd(tau_c);
#pragma unroll
                    for (int cp = c + 1; cp < PB; ++cp) {
                        const d2 g = (d2){total(1 + 2 * (cp - c - 1)), total(2 + 2 * (cp - c - 1))};
                        const d2 rowv = row_buf[cp];
                        // z = conj(scale) (g - conj(alpha) row) + row
                        d2 t = g;
                        cfnmac(t, rowv, alpha);  // g - row conj(alpha)
                        d2 z = cmulc(t, scale);  // t conj(scale)
                        z[0] += rowv[0];
                        z[1] += rowv[1];
                        const d2 f = cmul(ctau, z);
#pragma unroll
                        for (int rr = 0; rr < ROWS; ++rr) {
                            if (below[rr]) {
                                y[rr][cp][0] -= vn[rr][c][0] * f[0] - vn[rr][c][1] * f[1];
                                y[rr][cp][1] -= vn[rr][c][0] * f[1] + vn[rr][c][1] * f[0];
                            }
                        }
                    }
#pragma unroll
                    for (int rr = 0; rr < ROWS; ++rr)
                        if (below[rr]) y[rr][c] = (row_of(rr) == s + c) ? (d2){beta, 0.0} : (d2){0.0, 0.0};
                }
            }
        }
        // The last step's totals are read from the partial-sum area as they are used; the Gram sums below write it
        // again: without this meeting a wave that had run ahead overwrote partials another wave was still adding up
        // (1 wrong matrix in ~250 000; tools/race_check.py --model).
        lds_fence();
        __syncthreads();
        }
        TBK_CLK(15);  // QR: reflector + update (and whatever follows the last step)
        // the thread of row s + c holds row c of R: column s + c of the block row is conj(R[c][r]) for r >= c
        // (GRAM2: stored where the row passed through the registers)
#pragma unroll
        for (int rr = 0; rr < (GRAM2 ? 0 : ROWS); ++rr) {
            const int i_row = row_of(rr);
            if (qr_row[rr] && i_row < s + PB) {
                const int c = i_row - s;
#pragma unroll
                for (int r = 0; r < PB; ++r)
                    if (g0 + r < n) *Hat(g0 + r, i_row) = (r >= c) ? conjd(y[rr][r]) : (d2){0.0, 0.0};
            }
        }
        TBK_CLK(1);
        // ---- T of the compact WY form from the Gram matrix of V (model: t_factor); kept in LDS over the big pass ----
        {
            if constexpr (GRAM2) {
#if TBK_PANEL_GRAM
                // G = V^H V from the rows of V where the QR left them (the X area), this wave's own rows
                d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int rr = 0; rr < ROWS; ++rr) {
                    const int base_row = wave * 64 + rr * NT;
                    if (base_row + 64 <= s || base_row >= n) continue;  // wave-uniform
                    gram_direct(sX, nullptr, nullptr, base_row, 0, acc);
                }
                gram_finish(acc);
#endif
            } else if constexpr (GRAM) {
#if TBK_PANEL_GRAM
                // G = V^H V on the matrix pipe; the planes sit on the X / V area, so V is handed over BEHIND the meeting
                d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int rr = 0; rr < ROWS; ++rr)
                    if (__any(qr_row[rr])) gram_rows(vn[rr], vn[rr], true, acc);  // (vn is zero outside the trailing rows)
                gram_finish(acc);
#endif
            } else {
            // G[c2][c] = sum_i conj(v_c2) v_c for c2 < c: 28 complex sums, pair (c, c2) at slot 2 (c (c - 1) / 2 + c2)
            double hold[4];
            int k = 0;
#pragma unroll
            for (int c = 1; c < PB; ++c) {
#pragma unroll
                for (int c2 = 0; c2 < c; ++c2) {
                    d2 t = (d2){0.0, 0.0};
#pragma unroll
                    for (int rr = 0; rr < ROWS; ++rr) cfmac(t, vn[rr][c], vn[rr][c2]);  // conj(v_c2) v_c
                    hold[k & 3] = t[0];
                    hold[(k + 1) & 3] = t[1];
                    k += 2;
                    if ((k & 3) == 0) wave_partial4(k - 4, hold[0], hold[1], hold[2], hold[3], sPart, lane, wave);
                }
            }
            }
            // hand-over of Vn (LDS or global) HERE, in front of the T block: nobody reads it before the pass, and with the
            // rows of V dead the eight lanes that build T (28 Gram sums + tau + a row of T: 176 registers) fit
            // (GRAM2: every row of V went out where it passed through the registers)
#pragma unroll
            for (int rr = 0; rr < (GRAM2 ? 0 : ROWS); ++rr) {
                const int i_row = row_of(rr);
                if (i_row < npad) {
#pragma unroll
                    for (int c = 0; c < PB; ++c) sVn[(size_t)i_row * PB + c] = vn[rr][c];
                }
            }
            if constexpr (!GRAM && !GRAM2) wg_finish<NW>(56, sPart, sTot, tid);
            // lane a of the first wave builds row a of T: T[a][c] = -tau_c sum_{c2 = a}^{c - 1} T[a][c2] G[c2][c]
            if (tid < PB) {
                // (the three other waves wait for these eight lanes: all Gram sums and tau first, in flight together, then
                // the recurrence on registers, then the row -- read where they are used, between the stores of T, every
                // one of the 64 reads was a round trip on the chain)
                int a = tid;  // (opaque: the 64 comparisons with it below are not to be hoisted out of the panel loop as masks)
                asm volatile("" : "+v"(a));
                d2 gm[28], tauv[PB], trow[PB];
                if constexpr (GRAM || GRAM2) {
#if TBK_PANEL_GRAM
                    static_for<1, PB>([&](auto cc) {
                        constexpr int c = decltype(cc)::value;
                        static_for<0, c>([&](auto c2c) {
                            constexpr int c2 = decltype(c2c)::value;
                            gm[c * (c - 1) / 2 + c2] = sG[c2 * PB + c];  // conj(v_c2) v_c
                        });
                    });
#endif
                } else {
#pragma unroll
                    for (int k = 0; k < 28; ++k) gm[k] = *reinterpret_cast<const d2*>(sTot + 2 * k);
                }
#pragma unroll
                for (int c = 0; c < PB; ++c) tauv[c] = sTau[c];
#pragma unroll
                for (int c = 0; c < PB; ++c) {
                    d2 acc = (d2){0.0, 0.0};
#pragma unroll
                    for (int c2 = 0; c2 < c; ++c2)
                        if (c2 >= a) cfma(acc, trow[c2], gm[c * (c - 1) / 2 + c2]);
                    const d2 t = cmul(tauv[c], acc);
                    trow[c] = (c == a) ? tauv[c] : (c > a ? (d2){-t[0], -t[1]} : (d2){0.0, 0.0});
                }
#pragma unroll
                for (int c = 0; c < PB; ++c) sT[a * PB + c] = trow[c];
            }
        }
        TBK_CLK(2);
        // ---- hand over: X cleared, the consumed pending rows zeroed (Vn went out in front of the T block) ----
        // (X is cleared in linear order: a thread clearing its own row of 128 bytes shares its banks with every second
        // lane -- the V stores above pay that, the rows being the threads' own)
        for (int i = tid; i < npad * PB; i += NT) sX[i] = (d2){0.0, 0.0};
        if constexpr (!GRAM && !GRAM2) {
            if (have_update && tid < 128) VW[vw_index(g0 + (tid >> 4), tid & 15)] = (d2){0.0, 0.0};
        }
        wg_sync();
        TBK_CLK(3);
        if (PHASE == 1) {  // the pass of this panel is the next launch; T waits for the W phase in global memory
            if (tid < 64) gT[tid] = sT[tid];
            return;
        }
        big_pass(s, have_update, true);
        TBK_CLK(4);
        } else {
            // PHASE 1, first trip: X = the sum of the members' partial products (in member order), T of that panel
            const int mem_n = tbk_band_split_members(n, NW);
            for (int i = tid; i < npad * PB; i += NT) {
                d2 acc = (d2){0.0, 0.0};
                for (int g = 0; g < mem_n; ++g) {
                    const d2 v = gX[(size_t)g * npad * PB + i];
                    acc[0] += v[0];
                    acc[1] += v[1];
                }
                sX[i] = acc;
            }
            if (tid < 64) sT[tid] = gT[tid];
            wg_sync();
            resume_w = false;
        }
        // ---- W = X T - V S / 2,  S = T^H (V^H X) T  (model: stage1_band) ----
        d2 xr[ROWS][PB], vr[ROWS][PB];  // this thread's rows of A V and of V, read back (nothing lives in registers over the pass)
        // (GRAM2: the sum below reads X and V where they lie; the rows are read one at a time further down)
#pragma unroll
        for (int rr = 0; rr < (GRAM2 ? 0 : ROWS); ++rr) {
            const int i_row = min(row_of(rr), npad - 1);
#pragma unroll
            for (int c = 0; c < PB; ++c) {
                xr[rr][c] = qr_row[rr] ? sX[(size_t)i_row * PB + c] : (d2){0.0, 0.0};
                vr[rr][c] = qr_row[rr] ? sVn[(size_t)i_row * PB + c] : (d2){0.0, 0.0};
            }
        }
        {
            if constexpr (GRAM2) {
#if TBK_PANEL_GRAM
                // M = V^H (A V): X from the LDS where the pass left it (undisturbed), V from global memory, both in the operands'
                // own layout.  Rows in front of the trailing matrix are masked: X holds products of finished rows there, and
                // the rows of V in blocks the pass no longer reads are not rewritten (they hold an earlier panel's V).
                d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int rr = 0; rr < ROWS; ++rr) {
                    const int base_row = wave * 64 + rr * NT;
                    if (base_row + 64 <= s || base_row >= n) continue;  // wave-uniform
                    gram_direct(nullptr, sVn, sX, base_row, s, acc);
                }
                gram_finish(acc);
#endif
            } else if constexpr (GRAM) {
#if TBK_PANEL_GRAM
                // M = V^H (A V) on the matrix pipe.  Every thread holds its row of X and V now: behind this meeting the
                // waves' planes may overwrite the X / V area
                lds_fence();
                __syncthreads();
                d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
                if (__any(qr_row[0])) gram_rows(vr[0], xr[0], false, acc);
                gram_finish(acc);
#endif
            } else {
            // M = V^H (A V) is Hermitian: upper triangle, row a at slot a (16 - a): the real diagonal entry, then
            // (re, im) of M[a][b] for b > a -- 64 values
            double hold[4];
            int k = 0;
#pragma unroll
            for (int a = 0; a < PB; ++a) {
#pragma unroll
                for (int b = a; b < PB; ++b) {
                    if (b == a) {
                        double t = 0.0;
#pragma unroll
                        for (int rr = 0; rr < ROWS; ++rr) t += vr[rr][a][0] * xr[rr][a][0] + vr[rr][a][1] * xr[rr][a][1];
                        hold[k & 3] = t;
                        k += 1;
                        if ((k & 3) == 0) wave_partial4(k - 4, hold[0], hold[1], hold[2], hold[3], sPart, lane, wave);
                    } else {
                        d2 t = (d2){0.0, 0.0};
#pragma unroll
                        for (int rr = 0; rr < ROWS; ++rr) cfmac(t, xr[rr][b], vr[rr][a]);  // conj(v_a) x_b
                        hold[k & 3] = t[0];
                        k += 1;
                        if ((k & 3) == 0) wave_partial4(k - 4, hold[0], hold[1], hold[2], hold[3], sPart, lane, wave);
                        hold[k & 3] = t[1];
                        k += 1;
                        if ((k & 3) == 0) wave_partial4(k - 4, hold[0], hold[1], hold[2], hold[3], sPart, lane, wave);
                    }
                }
            }
            wg_finish<NW>(64, sPart, sTot, tid);
            }
            // 64 threads: S[i][j] = sum_ab conj(T[a][i]) M[a][b] T[b][j]
            if (tid < 64) {
                int tid_s = tid;  // (opaque: the comparisons and addresses below are not to live across the panel loop)
                asm volatile("" : "+v"(tid_s));
                const int si = tid_s >> 3, sj = tid_s & 7;
                if constexpr (GRAM || GRAM2) {
#if TBK_PANEL_GRAM
                    // in two steps through the wave's own LDS queue (round 5): (M T)[a][j] once per entry instead of once per
                    // (i, j) -- 16 instead of 72 complex products per thread, on the one wave the other three wait for.
                    // (M as the matrix pipe delivered it, both triangles; the diagonal real)
                    d2* const sMT = sG + 64;  // (behind M in the partial-sum area: 2 KiB with four waves)
                    d2 inner = (d2){0.0, 0.0};  // (M T)[si][sj]
#pragma unroll
                    for (int b = 0; b < PB; ++b) {
                        d2 mab = sG[si * PB + b];
                        if (b == si) mab[1] = 0.0;
                        cfma(inner, mab, sT[b * PB + sj]);
                    }
                    sMT[si * PB + sj] = inner;
                    asm volatile("" ::: "memory");  // (one wave: its LDS operations are performed in order)
                    d2 acc = (d2){0.0, 0.0};
#pragma unroll
                    for (int a = 0; a < PB; ++a) cfmac(acc, sMT[a * PB + sj], sT[a * PB + si]);  // conj(T[a][si]) (M T)[a][sj]
                    sS[tid_s] = acc;
#endif
                } else {
                d2 acc = (d2){0.0, 0.0};
#pragma unroll
                for (int a = 0; a < PB; ++a) {
                    d2 inner = (d2){0.0, 0.0};  // (M T)[a][sj]
#pragma unroll
                    for (int b = 0; b < PB; ++b) {
                        d2 mab;
                        if (b == a) {
                            mab = (d2){sTot[a * (16 - a)], 0.0};
                        } else if (b > a) {
                            const int at = a * (16 - a) + 1 + 2 * (b - a - 1);
                            mab = (d2){sTot[at], sTot[at + 1]};
                        } else {
                            const int at = b * (16 - b) + 1 + 2 * (a - b - 1);
                            mab = (d2){sTot[at], -sTot[at + 1]};
                        }
                        cfma(inner, mab, sT[b * PB + sj]);
                    }
                    cfmac(acc, inner, sT[a * PB + si]);  // conj(T[a][si]) inner
                }
                sS[tid] = acc;
                }
            }
            wg_sync();
        }
        {
            // T and S (8 x 8 each): entry 16 k + t in lane t of register k of every row of 16 lanes, and the FMAs take
            // them from there (row_newbcast: every lane of a wave takes part; xr, vr are zero outside the trailing rows)
            // -- 100 LDS broadcast reads per thread before
            d2 tb[4], sb[4];
            int lane_w = lane;
            asm volatile("" : "+v"(lane_w));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                tb[k] = sT[16 * k + (lane_w & 15)];
                sb[k] = sS[16 * k + (lane_w & 15)];
            }
            if (ROWS > 1) asm volatile("" ::: "memory");  // (the re-reads below stay behind the Gram sums above)
#pragma unroll
            for (int rr = 0; rr < ROWS; ++rr) {
                if (!__any(qr_row[rr])) continue;  // wave-uniform
                const int i_row = row_of(rr);
                if (ROWS > 1) {
                    // two rows per thread: the rows of X and V are read again, one row at a time -- held over the Gram
                    // sums and the S block, both rows' 32 complex numbers (128 registers) beside T, S and the products
                    // pushed 36 dwords of loop invariants into scratch memory
                    const int i_read = min(i_row, npad - 1);
#pragma unroll
                    for (int c = 0; c < PB; ++c) {
                        xr[rr][c] = qr_row[rr] ? sX[(size_t)i_read * PB + c] : (d2){0.0, 0.0};
                        vr[rr][c] = qr_row[rr] ? sVn[(size_t)i_read * PB + c] : (d2){0.0, 0.0};
                    }
                }
                d2 xt[PB], vs[PB];
                static_for<0, PB>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    d2 acc = (d2){0.0, 0.0};
                    static_for<0, c + 1>([&](auto c2c) {
                        constexpr int c2 = decltype(c2c)::value;
                        cfma_bc<8 * (c2 & 1) + c>(acc, xr[rr][c2], tb[c2 >> 1]);  // T[c2][c]
                    });
                    xt[c] = acc;
                    d2 acc2 = (d2){0.0, 0.0};
                    static_for<0, PB>([&](auto c2c) {
                        constexpr int c2 = decltype(c2c)::value;
                        cfma_bc<8 * (c2 & 1) + c>(acc2, vr[rr][c2], sb[c2 >> 1]);  // S[c2][c]
                    });
                    vs[c] = acc2;
                });
                if (qr_row[rr]) {
#pragma unroll
                    for (int c = 0; c < PB; ++c) {
                        const d2 w = (d2){xt[c][0] - 0.5 * vs[c][0], xt[c][1] - 0.5 * vs[c][1]};
                        VW[vw_index(i_row, c)] = vr[rr][c];
                        VW[vw_index(i_row, PB + c)] = w;
                    }
                }
            }
        }
        have_update = true;
        wg_sync();
        TBK_CLK(5);
    }
    // the last pending update (no look-ahead consumed any of its rows)
    if constexpr (PHASE == 0) {  // (in the launch chain the last pending update and the band's way out are launches of their own)
    if (have_update) big_pass(PB * p, true, false);
    // the band leaves in compact form -- band[i][dd] = H[i][i + dd], dd = 0..8 -- so that the matrix buffer is free for
    // the next chunk's H(k) while the second stage still works on this one
    wg_sync();
    if (band_all == nullptr) {
        // fused: this workgroup goes straight on to the second stage of its matrix, in the same LDS -- on a CU the
        // other workgroup is then in some phase of ITS matrix, and the matrix-pipe, memory and vector-issue phases of
        // the two overlap (two chase workgroups side by side are both limited by instruction issue)
        chase4_body<NW, true, NT * 16 + ROWS * 2 + (VN_LDS ? 1 : 0)>(nullptr, H, br_smem, n, np, stagger, D + mat * (size_t)n, E + mat * (size_t)n);
        return;
    }
    {
        d2* band = band_all + mat * band_stride;
        for (int idx = tid; idx < n * (PB + 1); idx += NT) {
            const int i = idx / (PB + 1), dd = idx - i * (PB + 1);
            band[idx] = (i + dd < n) ? *Hat(i, i + dd) : (d2){0.0, 0.0};
        }
    }
#ifdef TBK_PHASE_CLOCK
    // (the last wave: its rows stay in the trailing matrix until the end, wave 0 idles at the barriers from panel 8 on)
    if (blockIdx.x == 0 && threadIdx.x == NT - 64)
        for (int k = 0; k < 16; ++k) tbk_band_clock[k] = clk_acc_[k];
#endif
    }  // PHASE == 0
}

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


// ================================================================================================
// stage 1 ABOVE 1024 orbitals (round 5): the same algorithm as a chain of launches with nothing per row in
// registers or LDS.  The kernels above keep a row of the panel per thread (two at most) and X = A V in LDS
// ([npad][8] complex: 128 KiB at 1024 orbitals) -- neither scales.  Here every panel is two launches:
//   band_xl_serial_kernel   one workgroup per matrix: the W phase of the previous panel, then look-ahead, panel QR from
//                           ONE Gram matrix (the GRAM2 form: the panel's rows in the X / Y buffer in GLOBAL memory, one
//                           row at a time through the registers, matrix instructions reading the [row][8] layout where
//                           it lies) and the T factor; the finished entries of the block row go straight to the compact band;
//   band_xl_sweep_kernel    a workgroup per block row I of the trailing matrix walks ALL tiles of that row -- the ones left of
//                           the diagonal as the transposed stored tiles -- reads them from the OLD matrix buffer, applies the
//                           pending update tile(I, J) -= [V | W]_I ([W | V]_J)^H in registers, adds tile Vn_J to ITS block of X
//                           (one owner per block of X, complete in registers: no partner sums, no LDS for X, any number of
//                           rows) and writes the stored orientation to the NEW buffer.  Every tile is read twice and written
//                           once per panel.
// (band_xl_update_kernel + band_xl_product_kernel: the same as two sweeps on ONE buffer -- the first form, TBK_BAND_XL_SWEEPS=2,
// and the last pending update of the chain; band_xl_sweep4_kernel + band_xl_xsum_kernel: every tile read once, measured, off.)
// Stream order is the only synchronisation between them; a batch goes in two groups of matrices on two streams.  Same
// arithmetic as the kernels above (tools/two_stage_model.py: panel_qr_gram, stage1_band); the global-memory chase and the
// bisection follow.
// ================================================================================================
// YL: the panel's rows live in LDS ([npad][8] complex of dynamic LDS: up to 1024 orbitals) instead of the X / Y buffer in global
// memory -- what the calls of a few matrices take: for ONE matrix every hand-over of the rows through global memory (look-ahead ->
// sums -> reflectors -> T) is a round trip with nothing else on the CU to hide it.
template <int NT, bool YL>
__global__ void __launch_bounds__(NT, 1)
band_xl_serial_kernel(double* __restrict__ Hall, int n, d2* __restrict__ VWall, d2* __restrict__ VNall, d2* __restrict__ XYall,
                      d2* __restrict__ Tall, int p, d2* __restrict__ band_all, size_t band_stride) {
    constexpr int NW = NT / 64;
    extern __shared__ __attribute__((aligned(16))) double xl_smem[];
    __shared__ d2 sPartG[2 * NW * 64];  // the waves' partial Gram products, two areas in turn
    __shared__ d2 sG[128];              // C of the Gram routine; (M T) behind it in the W phase
    __shared__ d2 sS[64], sT[64], sF[64], sTau[PB], sCo[2 * PB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nbk = (n + TS - 1) / TS, npad = nbk * TS;
    const size_t mat = blockIdx.x;
    double* H = Hall + mat * (size_t)n * n * 2;
    d2* VW = VWall + mat * (size_t)nbk * 256;   // pending [V | W] rows, fragment order
    d2* VN = VNall + mat * (size_t)npad * PB;   // the panel's V, [npad][8]
    d2* XY = XYall + mat * (size_t)npad * PB;   // X = A V between the sweep and the W phase; the panel's rows in the QR (unless YL)
    d2* const Yp = YL ? reinterpret_cast<d2*>(xl_smem) : XY;  // the panel's rows during the QR, then its V (for the T factor's sum)
    d2* gT = Tall + mat * 64;                   // T of the panel, for the W phase in the next launch
    auto Hat = [&](int i, int j) -> d2* { return reinterpret_cast<d2*>(H + ((size_t)i * n + j) * 2); };
    // The finished entries of the panel's block row (its diagonal block, its rows of R) ARE band entries: with one sweep per
    // panel (old -> new matrix buffers, band_all != NULL) they go straight to the compact band -- the matrix buffers only ever
    // hold the trailing matrix there -- otherwise into the matrix, from where band_extract_kernel takes them at the end.
    d2* const band = band_all ? band_all + mat * band_stride : nullptr;
    auto put_final = [&](int i, int j, d2 v) {  // entry (i, j), i <= j, of the block row
        if (band) {
            if (j - i <= PB) band[(size_t)i * (PB + 1) + (j - i)] = v;
        } else {
            *Hat(i, j) = v;
        }
    };
    const int lane15 = lane & 15, t8 = lane & 7;

    int gram_parity = 0;
    // acc += O_a^T O_b over the rows [base_row, base_row + 64) that lie in [first_row, npad); a, b: [npad][8] complex in global memory
    auto gram_direct = [&](const d2* a, const d2* b, int base_row, int first_row, d4& acc) {
        const int g_lq = lane >> 4;
        const int col = lane15 < 8 ? 2 * lane15 : 2 * (lane15 - 8) + 1;
        const bool plain = base_row >= first_row && base_row + 64 <= npad;  // wave-uniform
        double opa[16], opb[16];
#pragma unroll
        for (int rho = 0; rho < 16; ++rho) {
            const int row = base_row + 16 * g_lq + rho;
            const size_t at = (size_t)(plain ? row : min(row, npad - 1)) * 16 + col;
            double va = reinterpret_cast<const double*>(a)[at];
            if (!plain) va = (row >= first_row && row < npad) ? va : 0.0;
            opa[rho] = va;
            if (b != a) {
                double vb = reinterpret_cast<const double*>(b)[at];
                if (!plain) vb = (row >= first_row && row < npad) ? vb : 0.0;
                opb[rho] = vb;
            }
        }
#pragma unroll
        for (int rho = 0; rho < 16; ++rho) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(opa[rho], b != a ? opb[rho] : opa[rho], acc, 0, 0, 0);
    };
    // the workgroup's total -> sG[c][t]; the meeting also waits for this wave's global stores (rows other waves read next)
    auto gram_finish = [&](const d4& acc) {
        const int g_lq = lane >> 4;
        const double sgn = lane15 < 8 ? 1.0 : -1.0;
        d2 mine;
        mine[0] = fma(dpp_mov<0x128>(acc[2]), sgn, acc[0]);
        mine[1] = fma(dpp_mov<0x128>(acc[3]), sgn, acc[1]);
        sPartG[(gram_parity * NW + wave) * 64 + lane] = mine;
        wg_sync();
        d2 tot = sPartG[(gram_parity * NW) * 64 + lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            const d2 v = sPartG[(gram_parity * NW + w) * 64 + lane];
            tot[0] += v[0];
            tot[1] += v[1];
        }
        gram_parity ^= 1;
        double* gd = reinterpret_cast<double*>(sG);
        gd[((g_lq)*PB + (lane15 & 7)) * 2 + (lane15 >> 3)] = tot[0];
        gd[((g_lq + 4) * PB + (lane15 & 7)) * 2 + (lane15 >> 3)] = tot[1];
        asm volatile("" ::: "memory");
    };

    if (p == 0)
        for (int i = tid; i < nbk * 256; i += NT) VW[i] = (d2){0.0, 0.0};

    // ---- W of panel p - 1:  W = X T - V S / 2,  S = T^H (V^H X) T ----
    if (p > 0) {
        const int s = PB * p;            // start of that panel's trailing matrix
        const int lo = s & ~(TS - 1);
        if (tid < 64) sT[tid] = gT[tid];
        d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
        for (int q = wave; lo + 64 * q < n; q += NW) gram_direct(VN, XY, lo + 64 * q, s, acc);
        gram_finish(acc);
        if (tid < 64) {
            const int si = tid >> 3, sj = tid & 7;
            d2* const sMT = sG + 64;
            d2 inner = (d2){0.0, 0.0};
#pragma unroll
            for (int b = 0; b < PB; ++b) {
                d2 mab = sG[si * PB + b];
                if (b == si) mab[1] = 0.0;
                cfma(inner, mab, sT[b * PB + sj]);
            }
            sMT[si * PB + sj] = inner;
            asm volatile("" ::: "memory");
            d2 sacc = (d2){0.0, 0.0};
#pragma unroll
            for (int a = 0; a < PB; ++a) cfmac(sacc, sMT[a * PB + sj], sT[a * PB + si]);
            sS[tid] = sacc;
        }
        wg_sync();
        d2 tb[4], sb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            tb[k] = sT[16 * k + lane15];
            sb[k] = sS[16 * k + lane15];
        }
        for (int q = wave; lo + 64 * q < n; q += NW) {
            const int i_row = lo + 64 * q + lane;
            const bool qr = i_row >= s && i_row < n;
            const int ic = min(i_row, npad - 1);
            d2 xr[PB], vr[PB];
#pragma unroll
            for (int c = 0; c < PB; ++c) {
                xr[c] = qr ? XY[(size_t)ic * PB + c] : (d2){0.0, 0.0};
                vr[c] = qr ? VN[(size_t)ic * PB + c] : (d2){0.0, 0.0};
            }
            d2 xt[PB], vs[PB];
            static_for<0, PB>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                d2 a1 = (d2){0.0, 0.0};
                static_for<0, c + 1>([&](auto c2c) {
                    constexpr int c2 = decltype(c2c)::value;
                    cfma_bc<8 * (c2 & 1) + c>(a1, xr[c2], tb[c2 >> 1]);  // T[c2][c]
                });
                xt[c] = a1;
                d2 a2 = (d2){0.0, 0.0};
                static_for<0, PB>([&](auto c2c) {
                    constexpr int c2 = decltype(c2c)::value;
                    cfma_bc<8 * (c2 & 1) + c>(a2, vr[c2], sb[c2 >> 1]);  // S[c2][c]
                });
                vs[c] = a2;
            });
            if (qr) {
#pragma unroll
                for (int c = 0; c < PB; ++c) {
                    VW[vw_index(i_row, c)] = vr[c];
                    VW[vw_index(i_row, PB + c)] = (d2){xt[c][0] - 0.5 * vs[c][0], xt[c][1] - 0.5 * vs[c][1]};
                }
            }
        }
        wg_sync();
    }

    const int g0 = PB * p, s = g0 + PB, m = n - s;
    if (m < 2) return;  // (behind the last panel: only its W phase)
    const bool have_update = p > 0;
    const int lo = g0 & ~(TS - 1);  // first row of the row chunks of this panel: wave w has the rows lo + 64 q + lane, q = w, w + NW, ...

    // ---- look-ahead: block row p brought up to date with the pending (V, W); the panel's rows y = conj(x) go to XY ----
    {
        d2 pend[PB];
#pragma unroll
        for (int r = 0; r < PB; ++r) pend[r] = have_update ? VW[vw_index(g0 + r, lane15)] : (d2){0.0, 0.0};
        for (int q = wave; lo + 64 * q < npad; q += NW) {
            const int i_row = lo + 64 * q + lane;
            const bool in_rows = i_row >= g0 && i_row < n;
            const int ic = min(max(i_row, g0), n - 1);
            d2 x[PB];
#pragma unroll
            for (int r = 0; r < PB; ++r) {
                const int g = g0 + r;
                const bool upper = ic >= g;
                const d2 v = *Hat(upper ? g : ic, upper ? ic : g);
                x[r] = in_rows ? (upper ? v : conjd(v)) : (d2){0.0, 0.0};
            }
            if (have_update && __any(in_rows)) {
                d2 vw[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) vw[c] = VW[vw_index(ic, c)];
                static_for<0, PB>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    static_for<0, PB>([&](auto tc) {
                        constexpr int t = decltype(tc)::value;
                        cfnmac_bc<t>(x[r], pend[r], vw[PB + t]);       // - V[g][t] conj(W[i][t])
                        cfnmac_bc<PB + t>(x[r], pend[r], vw[t]);       // - W[g][t] conj(V[i][t])
                    });
                });
            }
            if (in_rows && i_row < s) {
#pragma unroll
                for (int r = 0; r < PB; ++r)
                    if (g0 + r <= i_row) put_final(g0 + r, i_row, x[r]);
            }
            if (i_row < npad) {
                const bool below = in_rows && i_row >= s;
#pragma unroll
                for (int c = 0; c < PB; ++c) Yp[(size_t)i_row * PB + c] = below ? conjd(x[c]) : (d2){0.0, 0.0};
            }
        }
    }
    if (tid < PB) sTau[tid] = (d2){0.0, 0.0};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's rows of the panel are in memory: its sums below read them

    // ---- panel QR: all reflectors of a round from ONE Gram matrix (model: panel_qr_gram) ----
    {
        const int last = min(PB, m - 1);
        int c0 = 0;
        while (c0 < last) {
            d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
            for (int q = wave; lo + 64 * q < n; q += NW) {
                if (lo + 64 * q + 64 <= s + c0) continue;  // wave-uniform
                gram_direct(Yp, Yp, lo + 64 * q, s + c0, acc);
            }
            gram_finish(acc);
            if (c0 == 0 && have_update && tid < 128) VW[vw_index(g0 + (tid >> 4), tid & 15)] = (d2){0.0, 0.0};
            d2 top[PB];
#pragma unroll
            for (int c = 0; c < PB; ++c) top[c] = (c >= c0 && c < m) ? Yp[(size_t)min(s + c, npad - 1) * PB + t8] : (d2){0.0, 0.0};
            d2 g_next = sG[min(c0, PB - 1) * PB + t8];
            bool stopped = false;
            int c1 = last;
            unsigned has_mask = 0;
            static_for<0, PB>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                if (c >= c0 && c < last && !stopped) {  // uniform
                    const d2 g_row = g_next;
                    g_next = sG[min(c + 1, PB - 1) * PB + t8];
                    d2 g = g_row;
                    static_for<0, c>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        cfnmacj_bc<c>(g, top[i], top[i]);
                    });
                    const double gcc = lane_value<c>(g[0]);
                    const double Gcc = lane_value<c>(g_row[0]);
                    if (c > c0 && !(gcc >= GRAM_THRESH * Gcc)) {
                        stopped = true;
                        c1 = c;
                    } else {
                        const d2 alpha = (d2){lane_value<c>(top[c][0]), lane_value<c>(top[c][1])};
                        const d2 rowv = top[c];
                        const double sigma = gcc - (alpha[0] * alpha[0] + alpha[1] * alpha[1]);
                        if (!(gcc == 0.0 || (sigma == 0.0 && alpha[1] == 0.0))) {  // uniform
                            double root, rroot;
                            fast_sqrt_rsqrt(gcc, root, rroot);
                            const double beta = -copysign(root, alpha[0]);
                            const double rbeta = -copysign(rroot, alpha[0]);
                            const d2 tau_c = (d2){(beta - alpha[0]) * rbeta, -alpha[1] * rbeta};
                            if (tid == 0) sTau[c] = tau_c;
                            const double qr_ = alpha[0] - beta, qi_ = alpha[1];
                            const double qn = fast_rcp(qr_ * qr_ + qi_ * qi_);
                            const d2 scale = (d2){qr_ * qn, -qi_ * qn};
                            d2 tz = g;
                            cfnmac(tz, rowv, alpha);
                            d2 z = cmulc(tz, scale);
                            z[0] += rowv[0];
                            z[1] += rowv[1];
                            d2 f = cmul(conjd(tau_c), z);
                            if (t8 <= c) f = (d2){0.0, 0.0};
                            top[c] = t8 > c ? (d2){rowv[0] - f[0], rowv[1] - f[1]} : (t8 == c ? (d2){beta, 0.0} : (d2){0.0, 0.0});
                            static_for<c + 1, PB>([&](auto ic) {
                                constexpr int i = decltype(ic)::value;
                                d2 vt = (d2){0.0, 0.0};
                                cfma_bc<c>(vt, scale, top[i]);
                                cfma(top[i], (d2){-vt[0], -vt[1]}, f);
                            });
                            sF[c * PB + t8] = f;
                            sCo[c] = scale;
                            sCo[PB + c] = (d2){beta, 0.0};
                            has_mask |= 1u << c;
                        }
                    }
                }
            });
            // the rows, one at a time through the registers
            for (int q = wave; lo + 64 * q < npad; q += NW) {
                const int i_row = lo + 64 * q + lane;
                const bool in_mat = i_row < npad;
                const bool qr = i_row >= s && i_row < n;
                const int ic = min(i_row, npad - 1);
                d2 yr[PB], vrow[PB];
#pragma unroll
                for (int c = 0; c < PB; ++c) {
                    yr[c] = Yp[(size_t)ic * PB + c];
                    vrow[c] = (d2){0.0, 0.0};
                }
                static_for<0, PB>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    if (c >= c0 && c < c1 && (has_mask >> c & 1u)) {  // uniform
                        const bool below = qr && i_row >= s + c;
                        const bool head = i_row == s + c;
                        const d2 f_c = sF[c * PB + t8];
                        const d2 sc_c = sCo[c];
                        const double beta_c = sCo[PB + c][0];
                        d2 v = cmul(yr[c], sc_c);
                        v = below ? (head ? (d2){1.0, 0.0} : v) : (d2){0.0, 0.0};
                        vrow[c] = v;
                        static_for<c + 1, PB>([&](auto cpc) {
                            constexpr int cp = decltype(cpc)::value;
                            cfnma_bc<cp>(yr[cp], v, f_c);
                        });
                        if (below) yr[c] = head ? (d2){beta_c, 0.0} : (d2){0.0, 0.0};
                    }
                });
                if (qr && i_row - s < PB && ((i_row - s >= c0 && i_row - s < c1) || (c1 >= last && i_row - s >= last))) {
                    const int c = i_row - s;
#pragma unroll
                    for (int r = 0; r < PB; ++r) put_final(g0 + r, i_row, (r >= c) ? conjd(yr[r]) : (d2){0.0, 0.0});
                }
                if (in_mat) {
#pragma unroll
                    for (int c = 0; c < PB; ++c) {
                        if (c >= c0 && c < c1) {
                            Yp[(size_t)i_row * PB + c] = vrow[c];
                            VN[(size_t)i_row * PB + c] = vrow[c];
                        } else if (c >= c1) {
                            Yp[(size_t)i_row * PB + c] = c1 < last ? yr[c] : (d2){0.0, 0.0};
                            if (c1 >= last) VN[(size_t)i_row * PB + c] = (d2){0.0, 0.0};
                        }
                    }
                }
            }
            c0 = c1;
            if (c0 < last) wg_sync();
        }
    }
    // ---- T of the compact WY form from G = V^H V ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (this wave's rows of V)
    {
        d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
        for (int q = wave; lo + 64 * q < n; q += NW) {
            if (lo + 64 * q + 64 <= s) continue;  // wave-uniform
            if (YL)
                gram_direct(Yp, Yp, lo + 64 * q, s, acc);  // (the rows of V where the reflectors left them)
            else
                gram_direct(VN, VN, lo + 64 * q, s, acc);
        }
        gram_finish(acc);
        if (tid < PB) {
            const int a = tid;
            d2 gm[28], tauv[PB], trow[PB];
            static_for<1, PB>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                static_for<0, c>([&](auto c2c) {
                    constexpr int c2 = decltype(c2c)::value;
                    gm[c * (c - 1) / 2 + c2] = sG[c2 * PB + c];
                });
            });
#pragma unroll
            for (int c = 0; c < PB; ++c) tauv[c] = sTau[c];
#pragma unroll
            for (int c = 0; c < PB; ++c) {
                d2 tacc = (d2){0.0, 0.0};
#pragma unroll
                for (int c2 = 0; c2 < c; ++c2)
                    if (c2 >= a) cfma(tacc, trow[c2], gm[c * (c - 1) / 2 + c2]);
                const d2 t = cmul(tauv[c], tacc);
                trow[c] = (c == a) ? tauv[c] : (c > a ? (d2){-t[0], -t[1]} : (d2){0.0, 0.0});
            }
#pragma unroll
            for (int c = 0; c < PB; ++c) gT[a * PB + c] = trow[c];
        }
    }
}

// tile(I, J) -= [V | W]_I ([W | V]_J)^H for the block row I = i0 + blockIdx.x, J = I .. nbk - 1 (the waves take every NW-th tile)
template <int NT>
__global__ void __launch_bounds__(NT, 2)
band_xl_update_kernel(double* __restrict__ Hall, int n, const d2* __restrict__ VWall, int i0) {
    constexpr int NW = NT / 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nbk = (n + TS - 1) / TS;
    const size_t mat = blockIdx.y;
    double* H = Hall + mat * (size_t)n * n * 2;
    const d2* VW = VWall + mat * (size_t)nbk * 256;
    const int I = i0 + (int)blockIdx.x;
    const int lrow = lane & 15, lq = lane >> 4;
    Frag own;
#pragma unroll
    for (int sg = 0; sg < 4; ++sg) {
        const d2 v2 = VW[((size_t)I * 4 + sg) * 64 + lane];
        own.re[sg] = v2[0];
        own.im[sg] = v2[1];
    }
    for (int J = I + wave; J < nbk; J += NW) {
        Frag par;
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) {
            const d2 v2 = VW[((size_t)J * 4 + sg) * 64 + lane];
            par.re[sg] = v2[0];
            par.im[sg] = v2[1];
        }
        const bool interior = (I + 1) * TS <= n && (J + 1) * TS <= n;
        const unsigned gc = (unsigned)min(J * TS + lrow, n - 1);
        d4 tre, tim;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned gr = (unsigned)min(I * TS + lq + 4 * r, n - 1);
            const d2 v2 = *reinterpret_cast<const d2*>(reinterpret_cast<const char*>(H) + (size_t)gr * (size_t)n * 16 + (size_t)gc * 16);
            const bool inside = interior || (I * TS + lq + 4 * r < n && J * TS + lrow < n);
            tre[r] = inside ? v2[0] : 0.0;
            tim[r] = inside ? v2[1] : 0.0;
        }
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) {
            const int sb = (sg + 2) & 3;
            tre = __builtin_amdgcn_mfma_f64_16x16x4f64(own.re[sg], par.re[sb], tre, 0, 0, 1);  // -ar br
            tre = __builtin_amdgcn_mfma_f64_16x16x4f64(own.im[sg], par.im[sb], tre, 0, 0, 1);  // -ai bi
            tim = __builtin_amdgcn_mfma_f64_16x16x4f64(own.im[sg], par.re[sb], tim, 0, 0, 1);  // -ai br
            tim = __builtin_amdgcn_mfma_f64_16x16x4f64(own.re[sg], par.im[sb], tim, 0, 0, 0);  // +ar bi
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gr = I * TS + lq + 4 * r;
            if (interior || (gr < n && J * TS + lrow < n))
                *reinterpret_cast<d2*>(reinterpret_cast<char*>(H) + (size_t)gr * (size_t)n * 16 + (size_t)(J * TS + lrow) * 16) = (d2){tre[r], tim[r]};
        }
    }
}

// X_I = sum_J tile(I, J) Vn_J over J = i0 .. nbk - 1 for the block row I = i0 + blockIdx.x: the tiles right of the diagonal
// as stored, those left of it as the transposed stored ones, the diagonal tile completed from its upper part.  The waves
// take every NW-th tile and add their partial blocks in wave order.
template <int NT>
__global__ void __launch_bounds__(NT, 2)
band_xl_product_kernel(const double* __restrict__ Hall, int n, const d2* __restrict__ VNall, d2* __restrict__ XYall, int i0) {
    constexpr int NW = NT / 64;
    __shared__ double sTr[NW * 16 * 17];
    __shared__ double sRed[NW * 4 * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nbk = (n + TS - 1) / TS, npad = nbk * TS;
    const size_t mat = blockIdx.y;
    const double* H = Hall + mat * (size_t)n * n * 2;
    const double* VNd = reinterpret_cast<const double*>(VNall + mat * (size_t)npad * PB);
    double* XYd = reinterpret_cast<double*>(XYall + mat * (size_t)npad * PB);
    const int I = i0 + (int)blockIdx.x;
    const int lrow = lane & 15, lq = lane >> 4;
    const int lane_x = lq * 16 + 2 * (lrow & 7) + (lrow >> 3);
    const double lane_sgn = (lrow < 8) ? -1.0 : 1.0;
    double* tr = sTr + wave * (16 * 17);
    d4 own1 = (d4){0.0, 0.0, 0.0, 0.0}, own2 = own1;
    for (int J = i0 + wave; J < nbk; J += NW) {
        const int Ir = min(I, J), Jc = max(I, J);
        const bool interior = (Ir + 1) * TS <= n && (Jc + 1) * TS <= n;
        const unsigned gc = (unsigned)min(Jc * TS + lrow, n - 1);
        d4 tre, tim;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned gr = (unsigned)min(Ir * TS + lq + 4 * r, n - 1);
            const d2 v2 = *reinterpret_cast<const d2*>(reinterpret_cast<const char*>(H) + (size_t)gr * (size_t)n * 16 + (size_t)gc * 16);
            const bool inside = interior || (Ir * TS + lq + 4 * r < n && Jc * TS + lrow < n);
            tre[r] = inside ? v2[0] : 0.0;
            tim[r] = inside ? v2[1] : 0.0;
        }
        double pb[4];
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) pb[sg] = (VNd + (size_t)J * (TS * 16) + lane_x)[sg * 64];
        if (J >= I) {
            // the own block is the row block of the stored tile: the operand is the transposed copy [lrow][lq + 4 sg]
            double ttre[4], ttim[4];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int r = 0; r < 4; ++r) tr[(lq + 4 * r) * 17 + lrow] = tre[r];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int sg = 0; sg < 4; ++sg) ttre[sg] = tr[lrow * 17 + lq + 4 * sg];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int r = 0; r < 4; ++r) tr[(lq + 4 * r) * 17 + lrow] = tim[r];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int sg = 0; sg < 4; ++sg) ttim[sg] = tr[lrow * 17 + lq + 4 * sg];
            asm volatile("" ::: "memory");
            if (J == I) {  // Hermitian tile of which only the upper part is valid
#pragma unroll
                for (int sg = 0; sg < 4; ++sg) {
                    const bool upper = lrow <= lq + 4 * sg;
                    const double ar = upper ? ttre[sg] : tre[sg];
                    const double ai = upper ? ttim[sg] : -tim[sg];
                    own1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, pb[sg], own1, 0, 0, 0);
                    own2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, pb[sg], own2, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int sg = 0; sg < 4; ++sg) {
                    own1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ttre[sg], pb[sg], own1, 0, 0, 0);
                    own2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ttim[sg], pb[sg], own2, 0, 0, 0);
                }
            }
        } else {
            // the own block is the column block: X_I += tile^H Vn_J, the stored tile is the operand as it is (conjugated)
#pragma unroll
            for (int sg = 0; sg < 4; ++sg) {
                own1 = __builtin_amdgcn_mfma_f64_16x16x4f64(tre[sg], pb[sg], own1, 0, 0, 0);
                own2 = __builtin_amdgcn_mfma_f64_16x16x4f64(tim[sg], pb[sg], own2, 0, 0, 1);  // conj
            }
        }
    }
    // lane (row lq + 4 r, c = lrow): Re X[row][c] (c < 8) or Im X[row][c - 8]; the waves' partial blocks in wave order
#pragma unroll
    for (int r = 0; r < 4; ++r) sRed[(wave * 4 + r) * 64 + lane] = fma(dpp_mov<0x128>(own2[r]), lane_sgn, own1[r]);
    lds_fence();
    __syncthreads();
    for (int r = wave; r < 4; r += NW) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) tot += sRed[(w * 4 + r) * 64 + lane];
        (XYd + (size_t)I * (TS * 16) + lane_x)[r * 64] = tot;
    }
}

// ONE sweep per panel (default): the workgroup of block row I walks ALL tiles of that row -- those left of the diagonal as the
// transposed stored ones -- reads every tile from the OLD matrix buffer, applies the pending rank-16 update in registers
// (either orientation: the row block's [V | W] is the A operand), adds tile Vn_J to its block of X, and writes the updated
// tile to the NEW buffer when it is the stored orientation (J >= I).  Nobody reads what this launch writes, so there is no
// order to keep: every tile crosses HBM three times per panel (read twice, written once) instead of four with the two sweeps
// above, and a panel is two launches instead of three.  The buffers change roles from panel to panel; finished rows never
// enter them (band_xl_serial_kernel writes those to the compact band).
template <int NT>
__global__ void __launch_bounds__(NT, 2)
band_xl_sweep_kernel(const double* __restrict__ Hsrc_all, double* __restrict__ Hdst_all, int n, const d2* __restrict__ VWall,
                     const d2* __restrict__ VNall, d2* __restrict__ XYall, int i0, int flags) {
    constexpr int NW = NT / 64;
    const int with_update = flags & 1, walk = flags & 2;
    __shared__ double sTr[NW * 16 * 17];
    __shared__ double sRed[NW * 4 * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nbk = (n + TS - 1) / TS, npad = nbk * TS;
    const size_t mat = blockIdx.y;
    const double* Hs = Hsrc_all + mat * (size_t)n * n * 2;
    double* Hd = Hdst_all + mat * (size_t)n * n * 2;
    const d2* VW = VWall + mat * (size_t)nbk * 256;
    const double* VNd = reinterpret_cast<const double*>(VNall + mat * (size_t)npad * PB);
    double* XYd = reinterpret_cast<double*>(XYall + mat * (size_t)npad * PB);
    const int I = i0 + (int)blockIdx.x;
    const int lrow = lane & 15, lq = lane >> 4;
    const int lane_x = lq * 16 + 2 * (lrow & 7) + (lrow >> 3);
    const double lane_sgn = (lrow < 8) ? -1.0 : 1.0;
    double* tr = sTr + wave * (16 * 17);
    Frag own;
#pragma unroll
    for (int sg = 0; sg < 4; ++sg) {
        const d2 v2 = with_update ? VW[((size_t)I * 4 + sg) * 64 + lane] : (d2){0.0, 0.0};
        own.re[sg] = v2[0];
        own.im[sg] = v2[1];
    }
    d4 own1 = (d4){0.0, 0.0, 0.0, 0.0}, own2 = own1;
    // The walk: at step t block row I visits its partner (t - I) mod m -- whose workgroup visits I at the same step, so the two
    // reads of a tile (one per orientation) leave their workgroups at about the same time and the second one finds the tile in a
    // cache (L2 when both sit on one XCD, the memory-side cache otherwise) instead of in HBM.  (walk == 0: every block row walks
    // J = i0, i0 + 1, ... -- the two reads of a tile are |I - J| / NW steps apart.)
    const int m_rows = nbk - i0, I_loc = I - i0;
    for (int t = wave; t < m_rows; t += NW) {
        int J_loc = walk ? t - I_loc : t;
        if (J_loc < 0) J_loc += m_rows;
        const int J = i0 + J_loc;
        const int Ir = min(I, J), Jc = max(I, J);
        const bool interior = (Ir + 1) * TS <= n && (Jc + 1) * TS <= n;
        const unsigned gc = (unsigned)min(Jc * TS + lrow, n - 1);
        d4 tre, tim;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned gr = (unsigned)min(Ir * TS + lq + 4 * r, n - 1);
            const d2 v2 = *reinterpret_cast<const d2*>(reinterpret_cast<const char*>(Hs) + (size_t)gr * (size_t)n * 16 + (size_t)gc * 16);
            const bool inside = interior || (Ir * TS + lq + 4 * r < n && Jc * TS + lrow < n);
            tre[r] = inside ? v2[0] : 0.0;
            tim[r] = inside ? v2[1] : 0.0;
        }
        double pb[4];
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) pb[sg] = (VNd + (size_t)J * (TS * 16) + lane_x)[sg * 64];
        if (with_update) {  // (uniform)
            Frag par;
#pragma unroll
            for (int sg = 0; sg < 4; ++sg) {
                const d2 v2 = VW[((size_t)J * 4 + sg) * 64 + lane];
                par.re[sg] = v2[0];
                par.im[sg] = v2[1];
            }
            if (J >= I) {  // tile(I, J) -= [V | W]_I ([W | V]_J)^H
#pragma unroll
                for (int sg = 0; sg < 4; ++sg) {
                    const int sb = (sg + 2) & 3;
                    tre = __builtin_amdgcn_mfma_f64_16x16x4f64(own.re[sg], par.re[sb], tre, 0, 0, 1);
                    tre = __builtin_amdgcn_mfma_f64_16x16x4f64(own.im[sg], par.im[sb], tre, 0, 0, 1);
                    tim = __builtin_amdgcn_mfma_f64_16x16x4f64(own.im[sg], par.re[sb], tim, 0, 0, 1);
                    tim = __builtin_amdgcn_mfma_f64_16x16x4f64(own.re[sg], par.im[sb], tim, 0, 0, 0);
                }
            } else {       // tile(J, I) -= [V | W]_J ([W | V]_I)^H
#pragma unroll
                for (int sg = 0; sg < 4; ++sg) {
                    const int sb = (sg + 2) & 3;
                    tre = __builtin_amdgcn_mfma_f64_16x16x4f64(par.re[sg], own.re[sb], tre, 0, 0, 1);
                    tre = __builtin_amdgcn_mfma_f64_16x16x4f64(par.im[sg], own.im[sb], tre, 0, 0, 1);
                    tim = __builtin_amdgcn_mfma_f64_16x16x4f64(par.im[sg], own.re[sb], tim, 0, 0, 1);
                    tim = __builtin_amdgcn_mfma_f64_16x16x4f64(par.re[sg], own.im[sb], tim, 0, 0, 0);
                }
            }
        }
        if (J >= I) {
            // the stored orientation: the updated tile goes to the new buffer (also without an update: the buffers change roles)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gr = I * TS + lq + 4 * r;
                if (interior || (gr < n && J * TS + lrow < n))
                    *reinterpret_cast<d2*>(reinterpret_cast<char*>(Hd) + (size_t)gr * (size_t)n * 16 + (size_t)(J * TS + lrow) * 16) = (d2){tre[r], tim[r]};
            }
            double ttre[4], ttim[4];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int r = 0; r < 4; ++r) tr[(lq + 4 * r) * 17 + lrow] = tre[r];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int sg = 0; sg < 4; ++sg) ttre[sg] = tr[lrow * 17 + lq + 4 * sg];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int r = 0; r < 4; ++r) tr[(lq + 4 * r) * 17 + lrow] = tim[r];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int sg = 0; sg < 4; ++sg) ttim[sg] = tr[lrow * 17 + lq + 4 * sg];
            asm volatile("" ::: "memory");
            if (J == I) {
#pragma unroll
                for (int sg = 0; sg < 4; ++sg) {
                    const bool upper = lrow <= lq + 4 * sg;
                    const double ar = upper ? ttre[sg] : tre[sg];
                    const double ai = upper ? ttim[sg] : -tim[sg];
                    own1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, pb[sg], own1, 0, 0, 0);
                    own2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, pb[sg], own2, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int sg = 0; sg < 4; ++sg) {
                    own1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ttre[sg], pb[sg], own1, 0, 0, 0);
                    own2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ttim[sg], pb[sg], own2, 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int sg = 0; sg < 4; ++sg) {
                own1 = __builtin_amdgcn_mfma_f64_16x16x4f64(tre[sg], pb[sg], own1, 0, 0, 0);
                own2 = __builtin_amdgcn_mfma_f64_16x16x4f64(tim[sg], pb[sg], own2, 0, 0, 1);  // conj
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) sRed[(wave * 4 + r) * 64 + lane] = fma(dpp_mov<0x128>(own2[r]), lane_sgn, own1[r]);
    lds_fence();
    __syncthreads();
    for (int r = wave; r < 4; r += NW) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) tot += sRed[(w * 4 + r) * 64 + lane];
        (XYd + (size_t)I * (TS * 16) + lane_x)[r * 64] = tot;
    }
}

// The sweep of a BATCH (launch_band_xl: enough matrices to fill the chip): every tile crosses HBM TWICE per panel -- read once,
// written once -- instead of three times.  A workgroup takes FOUR block rows (wave w: row I = i0 + 4 blockIdx.x + w) and walks the
// block columns J together; a wave reads only the stored orientation tile(I, J), J >= I, updates it, writes it to the new buffer
// and forms BOTH products from it: X_I += tile Vn_J in its registers (as above) and the part tile^H Vn_I of X_J, which the four
// waves add up through LDS (one barrier per block column) and leave as this workgroup's partial of X_J in P[blockIdx.x][J].
// band_xl_xsum_kernel then adds the partials to X in a fixed order (workgroup 0, 1, ...): the same sums on every run.  (PMC, 64
// matrices of 1536 orbitals: the one-row sweep above reads 1.27 x the two reads of every tile its walk asks for and writes 1 x --
// 276 GB per call against 155 GB of read-once + write-once; a walk that pairs the two reads of a tile in time -- block row I at
// step t visits (t - I) mod m -- was slower, 1.248 -> 1.366 ms per k-point: the partner's operand blocks are then different for
// every workgroup of a matrix.  One-row workgroups stay for calls of a few matrices: four times as many, a quarter as long.)
template <int NT>
__global__ void __launch_bounds__(NT, 2)
band_xl_sweep4_kernel(const double* __restrict__ Hsrc_all, double* __restrict__ Hdst_all, int n, const d2* __restrict__ VWall,
                      const d2* __restrict__ VNall, d2* __restrict__ XYall, double* __restrict__ Pall, size_t p_stride, int i0,
                      int with_update) {
    constexpr int NW = NT / 64;
    static_assert(NW == 4, "four block rows per workgroup, one per wave; the partial sums are four registers per lane");
    __shared__ double sTr[NW * 16 * 17];
    __shared__ double sRed[2 * NW * 4 * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nbk = (n + TS - 1) / TS, npad = nbk * TS;
    const size_t mat = blockIdx.y;
    const double* Hs = Hsrc_all + mat * (size_t)n * n * 2;
    double* Hd = Hdst_all + mat * (size_t)n * n * 2;
    const d2* VW = VWall + mat * (size_t)nbk * 256;
    const double* VNd = reinterpret_cast<const double*>(VNall + mat * (size_t)npad * PB);
    double* XYd = reinterpret_cast<double*>(XYall + mat * (size_t)npad * PB);
    double* Pd = Pall + mat * p_stride + (size_t)blockIdx.x * nbk * 256;  // this workgroup's partials: [block column][16 rows][8 complex]
    const int I0 = i0 + NW * (int)blockIdx.x;
    const int I = I0 + wave;
    const bool row_ok = I < nbk;  // (uniform per wave)
    const int Ic = min(I, nbk - 1);
    const int lrow = lane & 15, lq = lane >> 4;
    const int lane_x = lq * 16 + 2 * (lrow & 7) + (lrow >> 3);
    const double lane_sgn = (lrow < 8) ? -1.0 : 1.0;
    double* tr = sTr + wave * (16 * 17);
    Frag own;
    double pbi[4];
#pragma unroll
    for (int sg = 0; sg < 4; ++sg) {
        const d2 v2 = with_update ? VW[((size_t)Ic * 4 + sg) * 64 + lane] : (d2){0.0, 0.0};
        own.re[sg] = v2[0];
        own.im[sg] = v2[1];
        pbi[sg] = (VNd + (size_t)Ic * (TS * 16) + lane_x)[sg * 64];
    }
    d4 own1 = (d4){0.0, 0.0, 0.0, 0.0}, own2 = own1;
    // what a step needs from memory: fetched ONE STEP AHEAD (the tile of step J + 1 is on its way while step J computes -- with
    // three or fewer waves per SIMD nothing else covers the latency of HBM)
    struct StepIn {
        d4 tre, tim;
        double pb[4];
        Frag par;
    };
    auto fetch = [&](int J, StepIn& in) {
        const bool interior = (I + 1) * TS <= n && (J + 1) * TS <= n;
        const unsigned gc = (unsigned)min(J * TS + lrow, n - 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned gr = (unsigned)min(I * TS + lq + 4 * r, n - 1);
            const d2 v2 = *reinterpret_cast<const d2*>(reinterpret_cast<const char*>(Hs) + (size_t)gr * (size_t)n * 16 + (size_t)gc * 16);
            const bool inside = interior || (I * TS + lq + 4 * r < n && J * TS + lrow < n);
            in.tre[r] = inside ? v2[0] : 0.0;
            in.tim[r] = inside ? v2[1] : 0.0;
        }
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) in.pb[sg] = (VNd + (size_t)J * (TS * 16) + lane_x)[sg * 64];
        if (with_update) {
#pragma unroll
            for (int sg = 0; sg < 4; ++sg) {
                const d2 v2 = VW[((size_t)J * 4 + sg) * 64 + lane];
                in.par.re[sg] = v2[0];
                in.par.im[sg] = v2[1];
            }
        }
    };
    auto step = [&](int J, StepIn& cur, StepIn& nxt) {
        const int buf = (J - I0) & 1;
        if (row_ok && J + 1 >= I && J + 1 < nbk) fetch(J + 1, nxt);  // (uniform per wave)
        d4 t1 = (d4){0.0, 0.0, 0.0, 0.0}, t2 = t1;
        if (row_ok && J >= I) {  // (uniform per wave)
            const bool interior = (I + 1) * TS <= n && (J + 1) * TS <= n;
            d4 tre = cur.tre, tim = cur.tim;
            if (with_update) {  // tile(I, J) -= [V | W]_I ([W | V]_J)^H
#pragma unroll
                for (int sg = 0; sg < 4; ++sg) {
                    const int sb = (sg + 2) & 3;
                    tre = __builtin_amdgcn_mfma_f64_16x16x4f64(own.re[sg], cur.par.re[sb], tre, 0, 0, 1);
                    tre = __builtin_amdgcn_mfma_f64_16x16x4f64(own.im[sg], cur.par.im[sb], tre, 0, 0, 1);
                    tim = __builtin_amdgcn_mfma_f64_16x16x4f64(own.im[sg], cur.par.re[sb], tim, 0, 0, 1);
                    tim = __builtin_amdgcn_mfma_f64_16x16x4f64(own.re[sg], cur.par.im[sb], tim, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gr = I * TS + lq + 4 * r;
                if (interior || (gr < n && J * TS + lrow < n))
                    *reinterpret_cast<d2*>(reinterpret_cast<char*>(Hd) + (size_t)gr * (size_t)n * 16 + (size_t)(J * TS + lrow) * 16) = (d2){tre[r], tim[r]};
            }
            double ttre[4], ttim[4];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int r = 0; r < 4; ++r) tr[(lq + 4 * r) * 17 + lrow] = tre[r];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int sg = 0; sg < 4; ++sg) ttre[sg] = tr[lrow * 17 + lq + 4 * sg];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int r = 0; r < 4; ++r) tr[(lq + 4 * r) * 17 + lrow] = tim[r];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int sg = 0; sg < 4; ++sg) ttim[sg] = tr[lrow * 17 + lq + 4 * sg];
            asm volatile("" ::: "memory");
            if (J == I) {
#pragma unroll
                for (int sg = 0; sg < 4; ++sg) {
                    const bool upper = lrow <= lq + 4 * sg;
                    const double ar = upper ? ttre[sg] : tre[sg];
                    const double ai = upper ? ttim[sg] : -tim[sg];
                    own1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, cur.pb[sg], own1, 0, 0, 0);
                    own2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, cur.pb[sg], own2, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int sg = 0; sg < 4; ++sg) {
                    own1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ttre[sg], cur.pb[sg], own1, 0, 0, 0);
                    own2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ttim[sg], cur.pb[sg], own2, 0, 0, 0);
                    // ... and this tile's part of X_J: tile^H Vn_I (the registers as they were loaded ARE the transposed operand)
                    t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(tre[sg], pbi[sg], t1, 0, 0, 0);
                    t2 = __builtin_amdgcn_mfma_f64_16x16x4f64(tim[sg], pbi[sg], t2, 0, 0, 1);  // conj
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sRed[((buf * NW + wave) * 4 + r) * 64 + lane] = fma(dpp_mov<0x128>(t2[r]), lane_sgn, t1[r]);
        lds_fence();
        __syncthreads();  // (one per block column: a wave that runs ahead writes the OTHER area, and cannot pass the next barrier alone)
        {
            double tot = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) tot += sRed[((buf * NW + w) * 4 + wave) * 64 + lane];
            (Pd + (size_t)J * 256 + lane_x)[wave * 64] = tot;
        }
    };
    StepIn in_a, in_b;
    if (row_ok && I0 >= I) fetch(I0, in_a);  // (wave 0; the others fetch their first tile in the step before it)
    for (int J = I0; J < nbk; J += 2) {
        step(J, in_a, in_b);
        if (J + 1 < nbk) step(J + 1, in_b, in_a);
    }
    if (row_ok) {
#pragma unroll
        for (int r = 0; r < 4; ++r) (XYd + (size_t)I * (TS * 16) + lane_x)[r * 64] = fma(dpp_mov<0x128>(own2[r]), lane_sgn, own1[r]);
    }
}

// X_J += the partials of the workgroups 0 .. (J - i0) / 4 of band_xl_sweep4_kernel, in that order (grid: block columns x matrices)
__global__ void __launch_bounds__(256)
band_xl_xsum_kernel(d2* __restrict__ XYall, const double* __restrict__ Pall, size_t p_stride, int n, int i0) {
    const int nbk = (n + TS - 1) / TS, npad = nbk * TS;
    const size_t mat = blockIdx.y;
    const int J = i0 + (int)blockIdx.x;
    double* X = reinterpret_cast<double*>(XYall + mat * (size_t)npad * PB) + (size_t)J * 256 + threadIdx.x;
    const double* P = Pall + mat * p_stride + (size_t)J * 256 + threadIdx.x;
    double acc = *X;
    const int last = (J - i0) / 4;
    for (int g = 0; g <= last; ++g) acc += P[(size_t)g * nbk * 256];
    *X = acc;
}

// the band rows from row0 on out of a matrix buffer (the one-sweep chain: the rows behind the last panel), and the whole band back
// INTO the caller's matrix buffer (tbk_tridiagonal_reduce hands that buffer out as the work copy of the reduction)
__global__ void __launch_bounds__(256) band_extract_from_kernel(const double* __restrict__ Hall, int n, d2* __restrict__ band_all, size_t band_stride, int row0) {
    const double* H = Hall + (size_t)blockIdx.x * n * n * 2;
    d2* band = band_all + (size_t)blockIdx.x * band_stride;
    for (int idx = row0 * (PB + 1) + threadIdx.x; idx < n * (PB + 1); idx += 256) {
        const int i = idx / (PB + 1), dd = idx - i * (PB + 1);
        band[idx] = (i + dd < n) ? *reinterpret_cast<const d2*>(H + ((size_t)i * n + i + dd) * 2) : (d2){0.0, 0.0};
    }
}
__global__ void __launch_bounds__(256) band_deposit_kernel(double* __restrict__ Hall, int n, const d2* __restrict__ band_all, size_t band_stride) {
    double* H = Hall + (size_t)blockIdx.x * n * n * 2;
    const d2* band = band_all + (size_t)blockIdx.x * band_stride;
    for (int idx = threadIdx.x; idx < n * (PB + 1); idx += 256) {
        const int i = idx / (PB + 1), dd = idx - i * (PB + 1);
        if (i + dd < n) *reinterpret_cast<d2*>(H + ((size_t)i * n + i + dd) * 2) = band[idx];
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
// per matrix: the pending [V | W] rows in fragment order (16 complex per row) + the next panel's V (8 complex per row; used
// when it does not live in LDS)
// the sizes that take the launch chain of band_xl_*: above 1024 orbitals (TBK_BAND_XL_FROM=n: above n -- tests run the chain
// at sizes the NumPy model is quick at, and A/B it against the one-workgroup kernels)
static bool band_xl(int n) {
    static const int from = getenv("TBK_BAND_XL_FROM") ? atoi(getenv("TBK_BAND_XL_FROM")) : 1024;
    return n > from;
}
// the second matrix buffer of the chain (ws_xl), per matrix of a chunk: only the sizes that ALWAYS take the chain count for the chunk size
// (+ the partial sums of band_xl_sweep4_kernel: [workgroups = block rows / 4][block columns][16 x 8 complex])
static size_t xl_partial_doubles(int n) {
    const size_t nbk = (size_t)((n + TS - 1) / TS);
    return (nbk + 3) / 4 * nbk * 256;
}
static bool xl_sweep4() {
    static const bool on = tbk_exp_env("TBK_BAND_XL_SWEEP4") && atoi(tbk_exp_env("TBK_BAND_XL_SWEEP4")) != 0;
    return on;
}
size_t tbk_band_xl_buffer_per_matrix(int n) {
    return band_xl(n) ? (size_t)n * n * sizeof(d2) + (xl_sweep4() ? xl_partial_doubles(n) * sizeof(double) : 0) : 0;
}
bool tbk_band_split(const tbk_model* m, int64_t nk);
// The chain's second matrix buffer for calls / chunks of up to max_nk matrices, reserved where the callers reserve ws_band and
// ws_bandmat -- in front of the pipeline, not inside a launch (a grow there is a free + malloc, i.e. a device synchronisation
// between the chunks of a call whose later chunk is the larger one).  Calls of a few matrices take the chain at
// every size (tbk_band_split): up to 96 matrices x 16 n^2 bytes, 1.6 GB at 1024 orbitals.
int tbk_band_xl_reserve(tbk_model* m, int64_t max_nk) {
    const int n = m->n_orb;
    if (!(band_xl(n) || tbk_band_split(m, max_nk))) return TBK_OK;
    return m->ws_xl.reserve((size_t)max_nk * ((size_t)n * n * sizeof(d2) + (xl_sweep4() ? xl_partial_doubles(n) * sizeof(double) : 0)));
}
// (+ for the launch chain of band_xl_*: X / the panel's rows [npad][8] and T of the panel)
size_t tbk_band_scratch_per_matrix(int n) {
    const size_t nbk = (size_t)((n + TS - 1) / TS);
    return (nbk * (256 + TS * PB) + nbk * TS * PB + 64) * sizeof(d2);  // (calls of a few matrices take that chain at every size)
}

static int chase_pitch(int n) {
    // TBK_CHASE_PITCH=r (measurements): pitch = r mod 16.  Bank model of the four-sweeps-per-wave layout (DESIGN_LOG R4.2): 148 LDS
    // cycles per tick at 9, 138 at 3 or 11 -- reads of two sweeps that share a 16-lane group collide at every pitch
    static const int want = tbk_exp_env("TBK_CHASE_PITCH") ? (atoi(tbk_exp_env("TBK_CHASE_PITCH")) & 15) | 1 : 9;
    int np = n + PB;
    while (np % 16 != want) ++np;
    return np;
}

bool tbk_band_fused(int n);
constexpr int BAND_ONE_WG_MAXN = 1024;  // one workgroup per matrix: two rows per thread of 512 threads, X (8 complex per row) is 128 KiB of LDS
// above: every panel as three launches with nothing per row in registers or LDS (band_xl_*).  The limit is what has been
// validated (tests/test_gpu_parity.py: 1030 / 1536 / 2048 / 2050 / 3000 / 4096); nothing in the kernels depends on it.  TBK_BAND_XL=0: rocSOLVER above
// 1024 orbitals, as until round 4 (measurements).
static int band_maxn() {
    static const bool xl = !(getenv("TBK_BAND_XL") && atoi(getenv("TBK_BAND_XL")) == 0);
    return xl ? 4096 : BAND_ONE_WG_MAXN;
}
#define BAND_MAXN band_maxn()
constexpr int BAND_LDS_CHASE_MAXN = 512;  // above: the chase keeps its 16 diagonals in global memory
// TBK_CHASE_GLOBAL=1 (measurements): the global-memory chase at every size that runs it as its own launch -- 9 KiB of LDS
// and 158 registers per wave instead of 133 KiB at 512 orbitals, so its workgroups fit beside those of other kernels
static bool chase_global_forced(int n) {
    static const bool forced = tbk_exp_env("TBK_CHASE_GLOBAL") && atoi(tbk_exp_env("TBK_CHASE_GLOBAL")) != 0;
    return forced && !tbk_band_fused(n);
}
// 257 - 768 orbitals, calls of more matrices than the chip has CUs: the windowed kernel with 16 sweep slots and 272 columns -- 78 KiB
// of LDS instead of the 133 KiB of the plain LDS form at 512 orbitals, so two of its workgroups share a CU, or one sits beside a
// first-stage workgroup of the next chunk (76 KiB).  A matrix takes more and slower ticks (1293 x ~2.2 us instead of 1088 x 1.55 at
// 512 orbitals), the chip holds twice as many: cfg5 16.04 -> 16.63 k k-points/s, whole eigenval of 2048 k-points 12.93 -> 11.74 us per
// k-point at 320 orbitals, 18.82 -> 17.66 at 384, 34.57 -> 33.64 at 512; the same bits.  TBK_CHASE_WINDOW_SMALL=0: off.
static bool chase_small_window(const tbk_model* m, int n, int64_t nk) {
    static const bool on = !(tbk_exp_env("TBK_CHASE_WINDOW_SMALL") && atoi(tbk_exp_env("TBK_CHASE_WINDOW_SMALL")) == 0);
    // Up to 768 orbitals (TBK_CHASE_WINDOW_SMALL_MAXN, measurements): above 512 against the 32-slot window -- whole eigenval of 2048
    // k-points 41.1 -> 39.1 us per k-point at 520 orbitals, 64.8 -> 62.9 at 640, 100.1 -> 97.4 at 768, 206.2 -> 215.6 at 1000.
    static const int maxn = tbk_exp_env("TBK_CHASE_WINDOW_SMALL_MAXN") ? atoi(tbk_exp_env("TBK_CHASE_WINDOW_SMALL_MAXN")) : 768;
    return on && n > 256 && n <= maxn && !tbk_band_fused(n) && std::max<int64_t>(m->call_nk, nk) > 256;
}
// does a matrix' band buffer carry the 16 working diagonals behind the compact band (by the size alone: any call may need them)
static bool chase_has_buffer(int n) { return n > BAND_LDS_CHASE_MAXN || chase_global_forced(n) || (n > 256 && !tbk_band_fused(n)); }

// The kernels handle 64 < n <= 512; the two-stage path is TAKEN from 189 orbitals on (129 until round 3): up to 128 the one-stage kernel of
// tbk_eig_stream.hip (four waves per matrix, rows of two 64-column chunks) is faster -- 0.65 vs 0.84 us per matrix at 65
// orbitals, 1.73 vs 2.14 at 128; from 129 on the one-stage rows grow a third chunk and the order flips (3.8 vs 3.3 us at 160).
// Both stages in ONE kernel (the workgroup goes straight on to the bulge chasing of its matrix, in the same LDS) or in
// two launches with the second one on the tridiagonal stream next to the following chunk's first stage.  Per matrix
// the two cost the same -- a workgroup's critical path is the sum of its phases either way -- and in the chunk pipeline
// fused is 1 % ahead at 256 orbitals (cfg3 134.1 vs 132.3 k k-points/s), 4 % behind at 512 (cfg5 13.5 vs 14.0 k: one
// workgroup per CU there, and the separate launch fills the gaps of the next chunk's first stage).  TBK_BAND_FUSE=0 / 1
// forces one (measurements).
bool tbk_band_fused(int n) {
    static const int forced = tbk_exp_env("TBK_BAND_FUSE") ? atoi(tbk_exp_env("TBK_BAND_FUSE")) : -1;
    if (band_xl(n)) return false;  // (the launch chain ends in the band's way out; the second stage is a launch of its own)
    return forced >= 0 ? forced != 0 : n <= 256;
}

bool tbk_eig_band_supported(int n) { return n > 64 && n <= BAND_MAXN; }
bool tbk_eig_band_preferred(int n) {
    // (round 3: the one-stage kernel hands its last 128 steps to the register-resident kernels -- eight waves per matrix
    // from 128 to 64, tbk_eig_small.hip -- which moved the crossover up: 1.34 vs 2.37 us per matrix at 130 orbitals, 1.96 vs
    // 2.57 at 144, 2.56 vs 3.02 at 160; reduction stage of 4096 matrices 12.2 vs 12.7 ms at 176, 13.4 vs 13.9 at 184,
    // 14.9 vs 14.4 at 192.  Round 4, after the trims of both stages, whole eigenval per k-point, one-stage vs two-stage:
    // 2.85 vs 3.16 us at 168, 3.36 vs 3.34 at 176, 3.70 vs 3.70 at 184, 3.88 vs 3.78 at 188 -- 177 .. 192 orbitals pad to
    // the same twelve blocks of 16, so the two-stage path takes over where the one-stage time reaches that: from 185)
    static const int from = tbk_exp_env("TBK_BAND_FROM") ? atoi(tbk_exp_env("TBK_BAND_FROM")) : 185;  // measurements only
    return n >= from && n <= BAND_MAXN;
}

// the compact band between the stages (9 complex per row) and, above 512 orbitals, the second stage's 16 working
// diagonals behind it
size_t tbk_band_bytes_per_matrix(int n) {
    return ((size_t)n * (PB + 1) + (chase_has_buffer(n) ? (size_t)16 * chase_pitch(n) : 0)) * sizeof(d2);
}

// Calls of a few matrices (Z2Pack-style lines and single k-points, _tb_model.py:1103-1108; band-structure paths of a few dozen
// points): the first stage as a chain of launches (PHASE 1 / 2 of band_reduce_kernel), so that every tile pass runs on
// `members` CUs per matrix instead of one.  By the size of the CALL (TBK_OPT_K_CHUNK must not change a result: the partial
// sums of the members differ from one workgroup's in the last bit).  TBK_BAND_SPLIT=0: off (measurements).
bool tbk_band_split(const tbk_model* m, int64_t nk) {
    static const bool on = !(getenv("TBK_BAND_SPLIT") && atoi(getenv("TBK_BAND_SPLIT")) == 0);
    static const int64_t forced_limit = tbk_exp_env("TBK_BAND_SPLIT_MAX") ? atoll(tbk_exp_env("TBK_BAND_SPLIT_MAX")) : 0;
    const int n = m->n_orb;
    if (!on || n <= 128 || n > BAND_ONE_WG_MAXN || band_xl(n)) return false;
    // as long as every member workgroup of every matrix finds a CU of its own: n_cu / members matrices (on 256 CUs: 64 up to
    // 512 orbitals, 32 at 1024).  Measured (one k-point per call, reduction stage): 256 orbitals 2.11 -> 2.04 ms, 384: 4.62 -> 3.80, 512: 8.31 ->
    // 6.01, 1024: 49.0 -> 24.4
    // (up to 256 orbitals the serial launches dominate and 64 matrices in one launch are as fast: 2.49 vs 2.40 ms -- 8 there)
    // Round 5: the chain these calls take is the one of band_xl_* (three launches per panel, sweeps on a workgroup per block row
    // = every CU for ONE matrix; TBK_BAND_SPLIT=2: the round-4 chain, PHASE 1 / 2 of band_reduce_kernel with 4 - 8 member
    // workgroups per matrix).  One-k eigenval, round-4 chain -> band_xl chain: 2.05 -> 1.94 ms at 256 orbitals, 3.99 -> 3.50 at 384,
    // 6.05 -> 5.03 at 512, 13.98 -> 10.28 at 768, 24.35 -> 16.84 at 1024; 64 matrices: 2.23 -> 2.31 / 4.39 -> 4.43 / 6.80 -> 7.16 /
    // 22.5 -> 17.6 / 46.3 -> 35.1; 64 matrices of 512 orbitals in ONE launch of the eight-wave kernel: 8.09 ms -- so calls of up to 8
    // matrices up to 256 orbitals, 64 up to 512, 96 above.
#ifdef TBK_EXPERIMENTS  // (TBK_BAND_SPLIT=2: round 4's chain, PHASE 1 / 2 of band_reduce_kernel -- dropped in round 5, experiments build only)
    static const bool old_chain = getenv("TBK_BAND_SPLIT") && atoi(getenv("TBK_BAND_SPLIT")) == 2;
#else
    constexpr bool old_chain = false;
#endif
    const int64_t limit = forced_limit > 0 ? forced_limit
                          : old_chain      ? (n <= 256 ? 8 : std::max(1, m->n_cu) / tbk_band_split_members(n, 8))
                                           : (n <= 256 ? 8 : n <= 512 ? 64 : 96);
    return std::max<int64_t>(m->call_nk, nk) <= limit;
}

#ifdef TBK_EXPERIMENTS
template <int NT, int ROWS>
static int launch_split(tbk_model* m, hipStream_t s, double* d_H, int n, int64_t nk, d2* d_VW, d2* d_VN, d2* d_band, size_t lds) {
    constexpr int NW = NT / 64;
    const int nbk = (n + TS - 1) / TS, npad = nbk * TS;
    const int members = tbk_band_split_members(n, NW);
    const size_t split_stride = 64 + (size_t)members * npad * PB;
    TBK_CHECK(m->ws_split.reserve((size_t)nk * split_stride * sizeof(d2)));
    d2* d_split = m->ws_split.as<d2>();
    static std::atomic<bool> raised1[TBK_MAX_DEVICES] = {}, raised2[TBK_MAX_DEVICES] = {};
    TBK_HIP(tbk_raise_lds_limit(reinterpret_cast<const void*>(&band_reduce_kernel<NT, ROWS, false, 1>), 160 * 1024, raised1));
    TBK_HIP(tbk_raise_lds_limit(reinterpret_cast<const void*>(&band_reduce_kernel<NT, ROWS, false, 2>), 160 * 1024, raised2));
    const size_t stride = tbk_band_bytes_per_matrix(n) / sizeof(d2);
    const int np = chase_pitch(n);
    int p_end = 0;  // first panel without a trailing matrix behind it: n - 8 (p + 1) < 2
    while (n - PB * (p_end + 1) >= 2) ++p_end;
    for (int p = 0; p <= p_end; ++p) {
        // PHASE 1 of panel p (p_end: only the W phase of the last panel); PHASE 2: its pass (p_end: the last pending update)
        hipLaunchKernelGGL((band_reduce_kernel<NT, ROWS, false, 1>), dim3((unsigned)nk), dim3(NT), lds, s, d_H, n, d_VW, d_VN, d_band, stride,
                           np, 2, (double*)nullptr, (double*)nullptr, p, d_split);
        if (p == p_end && p_end == 0) break;  // (n < 10: nothing was ever pending)
        hipLaunchKernelGGL((band_reduce_kernel<NT, ROWS, false, 2>), dim3((unsigned)members, (unsigned)nk), dim3(NT), lds, s, d_H, n, d_VW,
                           d_VN, d_band, stride, np, 2, (double*)nullptr, (double*)nullptr, p, d_split);
    }
    hipLaunchKernelGGL(band_extract_kernel, dim3((unsigned)nk), dim3(256), 0, s, d_H, n, d_band, stride);
    TBK_HIP(hipGetLastError());
    return TBK_OK;
}

#endif  // TBK_EXPERIMENTS
// The first stage above 1024 orbitals: three launches per panel (serial phases / update sweep / product sweep), one more update
// sweep for the last pending update, then the band's way out.
static int launch_chase(tbk_model* m, hipStream_t s, const void* d_band, int64_t nk, double* d_D, double* d_E);
static int xl_groups(int n, int64_t nk) {
    // TBK_BAND_XL_GROUPS=g (1 - 4; measurements): default 2
    static const int groups_env = tbk_exp_env("TBK_BAND_XL_GROUPS") ? std::min(4, std::max(1, atoi(tbk_exp_env("TBK_BAND_XL_GROUPS")))) : 2;
    return (band_xl(n) && nk >= 4 * groups_env) ? groups_env : 1;
}
bool tbk_band_xl_grouped(int n, int64_t nk) { return xl_groups(n, nk) > 1; }

// d_de != NULL: the second stage of every group runs behind its first stage on the group's stream and (d, e) are written there
static int launch_band_xl(tbk_model* m, hipStream_t s, double* d_H, int n, int64_t nk, d2* d_scratch, d2* d_band, double* d_de = nullptr) {
    const int nbk = (n + TS - 1) / TS, npad = nbk * TS;
    d2* d_VW = d_scratch;
    d2* d_VN = d_VW + (size_t)nk * nbk * 256;
    d2* d_XY = d_VN + (size_t)nk * npad * PB;
    d2* d_T = d_XY + (size_t)nk * npad * PB;
    const size_t stride = tbk_band_bytes_per_matrix(n) / sizeof(d2);
    int p_end = 0;  // first panel without a trailing matrix behind it
    while (n - PB * (p_end + 1) >= 2) ++p_end;
    constexpr int NTS = 512, NTP = 256;
    // TBK_BAND_XL_SWEEPS=2 (measurements): the update sweep and the product sweep as two launches on ONE matrix buffer (the first
    // form of the chain: every tile crosses HBM four times per panel)
    static const bool two_sweeps = tbk_exp_env("TBK_BAND_XL_SWEEPS") && atoi(tbk_exp_env("TBK_BAND_XL_SWEEPS")) == 2;
#ifdef TBK_EXPERIMENTS
    if (two_sweeps) {
        for (int p = 0; p <= p_end; ++p) {
            hipLaunchKernelGGL((band_xl_serial_kernel<NTS, false>), dim3((unsigned)nk), dim3(NTS), 0, s, d_H, n, d_VW, d_VN, d_XY, d_T, p,
                               (d2*)nullptr, (size_t)0);
            if (p == p_end) break;
            const int i0 = PB * (p + 1) / TS, na = nbk - i0;
            if (p > 0)
                hipLaunchKernelGGL((band_xl_update_kernel<NTP>), dim3((unsigned)na, (unsigned)nk), dim3(NTP), 0, s, d_H, n, d_VW, i0);
            hipLaunchKernelGGL((band_xl_product_kernel<NTP>), dim3((unsigned)na, (unsigned)nk), dim3(NTP), 0, s, d_H, n, d_VN, d_XY, i0);
        }
        if (p_end > 0) {  // the last pending update (no look-ahead consumed any of its rows)
            const int i0 = PB * p_end / TS;
            hipLaunchKernelGGL((band_xl_update_kernel<NTP>), dim3((unsigned)(nbk - i0), (unsigned)nk), dim3(NTP), 0, s, d_H, n, d_VW, i0);
        }
        hipLaunchKernelGGL(band_extract_kernel, dim3((unsigned)nk), dim3(256), 0, s, d_H, n, d_band, stride);
        TBK_HIP(hipGetLastError());
        return TBK_OK;
    }
#else
    (void)two_sweeps;
#endif
    // One sweep per panel between two matrix buffers (the caller's and ws_xl) that change roles; the finished rows go to the band
    // as the serial phases produce them, the rows behind the last panel come out of the buffer the last update leaves them in,
    // and the band is put back into the caller's buffer (the work copy tbk_tridiagonal_reduce hands out).
    // TBK_BAND_XL_SWEEP4=1 (measurements): band_xl_sweep4_kernel -- every tile read once, four block rows per workgroup -- for every
    // call of the process.  Built in round 5 and not faster (DESIGN_LOG.md R5.12: 64 / 256 matrices of 1536 orbitals 1.256 -> 1.288 /
    // 0.940 -> 0.893 ms per k-point, of 2048 orbitals 2.585 -> 2.637 / 2.195 -> 2.214): the one-row sweep stays.
    const bool sweep4 = xl_sweep4();
    const size_t p_stride = xl_partial_doubles(n);
    TBK_CHECK(m->ws_xl.reserve((size_t)nk * n * n * 2 * sizeof(double) + (sweep4 ? (size_t)nk * p_stride * sizeof(double) : 0)));
    double* buf[2] = {d_H, m->ws_xl.as<double>()};
    double* d_P = m->ws_xl.as<double>() + (size_t)nk * n * n * 2;  // (partial sums of the read-once sweep: experiments build)
    (void)d_P;
    // up to 1024 orbitals (calls of a few matrices): the panel's rows in LDS (TBK_BAND_XL_YLDS=0: in global memory, as above 1024)
    static const bool y_lds_env = !(tbk_exp_env("TBK_BAND_XL_YLDS") && atoi(tbk_exp_env("TBK_BAND_XL_YLDS")) == 0);
    const bool y_lds = y_lds_env && n <= BAND_ONE_WG_MAXN;
    const size_t y_bytes = (size_t)npad * PB * sizeof(d2);
    // up to 256 orbitals the rows fill four waves only: a workgroup of four (TBK_BAND_XL_SERIAL4=0: eight, measurements) meets faster
    static const bool serial4_env = !(tbk_exp_env("TBK_BAND_XL_SERIAL4") && atoi(tbk_exp_env("TBK_BAND_XL_SERIAL4")) == 0);
    const bool four_waves = serial4_env && y_lds && n <= 256;
    if (y_lds) {
        static std::atomic<bool> raised[TBK_MAX_DEVICES] = {};
        TBK_HIP(tbk_raise_lds_limit(reinterpret_cast<const void*>(&band_xl_serial_kernel<NTS, true>), 132 * 1024, raised));
        static std::atomic<bool> raised4[TBK_MAX_DEVICES] = {};
        TBK_HIP(tbk_raise_lds_limit(reinterpret_cast<const void*>(&band_xl_serial_kernel<256, true>), 132 * 1024, raised4));
    }
    // A batch above 1024 orbitals goes in GROUPS of matrices on streams of their own: the serial phases of a panel occupy one
    // workgroup per matrix (a latency chain on a quarter of the CUs at 64 matrices) while the sweep is bound by HBM, and the second
    // stage is one workgroup per matrix for 2 n ticks -- one group's chains run under the other groups' sweeps.  Per matrix nothing
    // changes (same launches, same order, same bits).
    const int groups = xl_groups(n, nk);
    // TBK_BAND_XL_WALK=1 (measurements): the pairing walk of band_xl_sweep_kernel
    static const int walk_flag = (tbk_exp_env("TBK_BAND_XL_WALK") && atoi(tbk_exp_env("TBK_BAND_XL_WALK")) != 0) ? 2 : 0;
    auto chain = [&](hipStream_t st, int64_t k0, int64_t nkg) {
        double* b[2] = {buf[0] + (size_t)k0 * n * n * 2, buf[1] + (size_t)k0 * n * n * 2};
        d2* vw = d_VW + (size_t)k0 * nbk * 256;
        d2* vn = d_VN + (size_t)k0 * npad * PB;
        d2* xy = d_XY + (size_t)k0 * npad * PB;
        d2* tt = d_T + (size_t)k0 * 64;
        d2* bd = d_band + (size_t)k0 * stride;
        int cur = 0;
        for (int p = 0; p <= p_end; ++p) {
            if (y_lds)
                if (four_waves)
                    hipLaunchKernelGGL((band_xl_serial_kernel<256, true>), dim3((unsigned)nkg), dim3(256), y_bytes, st, b[cur], n, vw, vn, xy, tt, p,
                                       bd, stride);
                else
                    hipLaunchKernelGGL((band_xl_serial_kernel<NTS, true>), dim3((unsigned)nkg), dim3(NTS), y_bytes, st, b[cur], n, vw, vn, xy, tt, p,
                                       bd, stride);
            else
                hipLaunchKernelGGL((band_xl_serial_kernel<NTS, false>), dim3((unsigned)nkg), dim3(NTS), 0, st, b[cur], n, vw, vn, xy, tt, p, bd,
                                   stride);
            if (p == p_end) break;
            const int i0 = PB * (p + 1) / TS, na = nbk - i0;
#ifdef TBK_EXPERIMENTS
            if (sweep4) {
                hipLaunchKernelGGL((band_xl_sweep4_kernel<NTP>), dim3((unsigned)((na + 3) / 4), (unsigned)nkg), dim3(NTP), 0, st, b[cur], b[cur ^ 1],
                                   n, vw, vn, xy, d_P + (size_t)k0 * p_stride, p_stride, i0, p > 0 ? 1 : 0);
                hipLaunchKernelGGL(band_xl_xsum_kernel, dim3((unsigned)na, (unsigned)nkg), dim3(256), 0, st, xy, d_P + (size_t)k0 * p_stride,
                                   p_stride, n, i0);
            } else
#endif
            {
                hipLaunchKernelGGL((band_xl_sweep_kernel<NTP>), dim3((unsigned)na, (unsigned)nkg), dim3(NTP), 0, st, b[cur], b[cur ^ 1], n, vw, vn,
                                   xy, i0, (p > 0 ? 1 : 0) | walk_flag);
            }
            cur ^= 1;
        }
        if (p_end > 0) {  // the last pending update, in place (nobody reads tiles in this launch)
            const int i0 = PB * p_end / TS;
            hipLaunchKernelGGL((band_xl_update_kernel<NTP>), dim3((unsigned)(nbk - i0), (unsigned)nkg), dim3(NTP), 0, st, b[cur], n, vw, i0);
        }
        hipLaunchKernelGGL(band_extract_from_kernel, dim3((unsigned)nkg), dim3(256), 0, st, b[cur], n, bd, stride, PB * p_end);
        hipLaunchKernelGGL(band_deposit_kernel, dim3((unsigned)nkg), dim3(256), 0, st, b[0], n, bd, stride);
        if (d_de) return launch_chase(m, st, bd, nkg, d_de + (size_t)k0 * n, d_de + (size_t)(nk + k0) * n);
        return (int)TBK_OK;
    };
    if (groups == 1) {
        TBK_CHECK(chain(s, 0, nk));
    } else {
        // the side streams and their events exist from the first batch that uses them (not for every model: the temporary
        // models of tbk_tridiagonal_reduce / tbk_reduce_standalone and every small model never get here)
        for (int g = 1; g < groups; ++g)
            if (m->stream_xl[g - 1] == nullptr) TBK_HIP(hipStreamCreateWithFlags(&m->stream_xl[g - 1], hipStreamNonBlocking));
        for (int g = 0; g < groups; ++g)
            if (m->ev_xl[g] == nullptr) TBK_HIP(hipEventCreateWithFlags(&m->ev_xl[g], hipEventDisableTiming));
        TBK_HIP(hipEventRecord(m->ev_xl[0], s));
        const int64_t per = (nk + groups - 1) / groups;
        for (int g = 1; g < groups; ++g) TBK_HIP(hipStreamWaitEvent(m->stream_xl[g - 1], m->ev_xl[0], 0));
        // (the host enqueues group after group; the streams run side by side from the first launch on)
        // A failing group does not end the function: `s` first waits for every side stream that has work -- the caller's stream
        // must not go on to reuse ws_H / ws_xl / ws_band under kernels still running there
        int rc = TBK_OK;
        for (int g = 0; g < groups; ++g) {
            const int64_t k0 = g * per, nkg = std::min(per, nk - k0);
            if (nkg <= 0) break;
            hipStream_t st = g == 0 ? s : m->stream_xl[g - 1];
            if (rc == TBK_OK) rc = chain(st, k0, nkg);
            if (g > 0) {
                hipError_t e = hipEventRecord(m->ev_xl[g], st);
                if (e == hipSuccess) e = hipStreamWaitEvent(s, m->ev_xl[g], 0);
                if (e != hipSuccess && rc == TBK_OK) {
                    tbk_set_error("joining a side stream of the launch chain failed: %s", hipGetErrorString(e));
                    rc = TBK_ERR_DEVICE;
                }
            }
        }
        if (rc != TBK_OK) return rc;
    }
    TBK_HIP(hipGetLastError());
    return TBK_OK;
}

// Stage one: the upper triangle of every d_H matrix is overwritten; d_vw: scratch of tbk_band_scratch_per_matrix(n)
// bytes per matrix; d_band receives the band, tbk_band_bytes_per_matrix(n) bytes per matrix.
int tbk_launch_band_reduce(tbk_model* m, hipStream_t s, double* d_H, int64_t nk, void* d_vw, void* d_band, double* d_de_fused) {
    const int n = m->n_orb;
    if (nk == 0) return TBK_OK;
    StageTimer t(m, TBK_T_EIG, s);
    if (band_xl(n)) return launch_band_xl(m, s, d_H, n, nk, static_cast<d2*>(d_vw), static_cast<d2*>(d_band), d_de_fused);
    const int nbk = (n + TS - 1) / TS, npad = nbk * TS;
    // up to 256 orbitals a row per thread, V and X in LDS; above, TWO rows per thread and V in global memory, so that two
    // workgroups still fit a CU (76 KiB each at 512 orbitals) -- with 512 threads / V in LDS only one did and nothing
    // overlapped its serial phases (31.7 instead of 29.7 us per 512 x 512 matrix; that instantiation is gone)
    const bool vn_lds = n <= 256;
    // Calls of a few matrices (one k-point per call is what Z2Pack-style callers do, _tb_model.py:1103-1108): every matrix has
    // a CU to itself anyway, so it gets EIGHT waves and one row per thread -- twice the waves on the tile pass, half the
    // rows per thread in the thread-per-row phases.  By the size of the CALL (TBK_OPT_K_CHUNK must not change a result:
    // the partial sums of eight waves differ from those of four in the last bit).  TBK_BAND_WIDE=0: off (measurements).
    static const bool wide_env = !(tbk_exp_env("TBK_BAND_WIDE") && atoi(tbk_exp_env("TBK_BAND_WIDE")) == 0);
    static const bool wide_all = tbk_exp_env("TBK_BAND_WIDE") && atoi(tbk_exp_env("TBK_BAND_WIDE")) == 2;  // (measurements: every call size)
    const bool wide = wide_env && n <= 512 && (wide_all || std::max<int64_t>(m->call_nk, nk) <= 128);
    const int nw = (n > 512 || wide) ? 8 : 4;
    const int rows_per_thread = (n > 512 || (n > 256 && !wide)) ? 2 : 1;  // (the instantiation chosen below)
    size_t lds = band_xv_bytes(npad, vn_lds, nw, rows_per_thread) + (size_t)(nw * 16 * 17 + nw * 64 + 64) * 8 + (16 + 64 + 64 + 8 + 2) * 16;
    // d_de_fused: the workgroup runs the second stage too (same LDS) and writes (d, e) itself; d_band is not used
    const int np = chase_pitch(n);
    double* d_D = d_de_fused;
    double* d_E = d_de_fused ? d_de_fused + (size_t)nk * n : nullptr;
    if (d_de_fused) {
        lds = std::max(lds, (size_t)16 * np * 16 + (size_t)nw * 64 * 16 + (size_t)n * sizeof(int) + 16);
        d_band = nullptr;
    }
    d2* d_VW = static_cast<d2*>(d_vw);
    d2* d_VN = d_VW + (size_t)nk * nbk * 256;
#ifdef TBK_EXPERIMENTS  // (TBK_BAND_SPLIT=2: round 4's chain, PHASE 1 / 2 of band_reduce_kernel -- dropped in round 5, experiments build only)
    static const bool old_chain = getenv("TBK_BAND_SPLIT") && atoi(getenv("TBK_BAND_SPLIT")) == 2;
#else
    constexpr bool old_chain = false;
#endif
    if (d_de_fused == nullptr && tbk_band_split(m, nk) && !old_chain)
        return launch_band_xl(m, s, d_H, n, nk, static_cast<d2*>(d_vw), static_cast<d2*>(d_band));
#ifdef TBK_EXPERIMENTS
    if (d_de_fused == nullptr && tbk_band_split(m, nk)) {
        // the launch chain: one row per thread where the rows allow it (the serial phases are thread-per-row)
        auto lds_for = [&](int waves, int rows) { return band_xv_bytes(npad, false, waves, rows) + (size_t)(waves * 16 * 17 + waves * 64 + 64) * 8 + (16 + 64 + 64 + 8 + 2) * 16; };
        if (n <= 256) return launch_split<256, 1>(m, s, d_H, n, nk, d_VW, d_VN, static_cast<d2*>(d_band), lds_for(4, 1));
        if (n <= 512) return launch_split<512, 1>(m, s, d_H, n, nk, d_VW, d_VN, static_cast<d2*>(d_band), lds_for(8, 1));
        return launch_split<512, 2>(m, s, d_H, n, nk, d_VW, d_VN, static_cast<d2*>(d_band), lds_for(8, 2));
    }
#endif
    static std::atomic<bool> raised[6][TBK_MAX_DEVICES] = {};
#define TBK_REDUCE(NTV, ROWSV, VNL, SLOT)                                                                                       \
    do {                                                                                                                        \
        TBK_HIP(tbk_raise_lds_limit(reinterpret_cast<const void*>(&band_reduce_kernel<NTV, ROWSV, VNL>), 160 * 1024, raised[SLOT])); \
        hipLaunchKernelGGL((band_reduce_kernel<NTV, ROWSV, VNL>), dim3((unsigned)nk), dim3(NTV), lds, s, d_H, n, d_VW, d_VN,      \
                           static_cast<d2*>(d_band), tbk_band_bytes_per_matrix(n) / sizeof(d2), np, 2, d_D, d_E);               \
    } while (0)
    // TBK_BAND_NARROW=1 (measurement, round 4): TWO waves per matrix and two rows per thread up to 256 orbitals -- four matrices
    // per CU instead of two, the per-wave overhead of the serial phases (reductions, scalar chains) paid half as often per
    // matrix; 38 KiB of LDS, second stage in its own launch
    static const bool narrow_env = tbk_exp_env("TBK_BAND_NARROW") && atoi(tbk_exp_env("TBK_BAND_NARROW")) != 0;
#ifdef TBK_EXPERIMENTS
    if (narrow_env && !wide && n <= 256 && d_de_fused == nullptr) {
        lds = band_xv_bytes(npad, false, 2, 2) + (size_t)(2 * 16 * 17 + 2 * 64 + 64) * 8 + (16 + 64 + 64 + 8 + 2) * 16;
        TBK_REDUCE(128, 2, false, 5);
    } else
#else
    (void)narrow_env;
#endif
    if (wide && vn_lds)
        TBK_REDUCE(512, 1, true, 3);
    else if (wide)
        TBK_REDUCE(512, 1, false, 4);
    else if (vn_lds)
        TBK_REDUCE(256, 1, true, 0);
    else if (n <= 512)
        TBK_REDUCE(256, 2, false, 1);
    else  // 513 .. 1024 orbitals (round 4): eight waves, two rows per thread, one workgroup per CU (X alone is 128 KiB)
        TBK_REDUCE(512, 2, false, 2);
#undef TBK_REDUCE
    TBK_HIP(hipGetLastError());
    return TBK_OK;
}

// Stage two: d_band -> d_de = d[nk][n] followed by e[nk][n]
static int launch_chase(tbk_model* m, hipStream_t s, const void* d_band, int64_t nk, double* d_D, double* d_E) {
    const int n = m->n_orb;
    // TBK_CHASE_WINDOW=0 (measurements): no windowed kernel -- above 512 orbitals the global-memory form, the plain LDS form below
    static const bool window_env = !(getenv("TBK_CHASE_WINDOW") && atoi(getenv("TBK_CHASE_WINDOW")) == 0);
    const bool small_window = window_env && chase_small_window(m, n, nk);
#ifdef TBK_ABLATE_WIN_FORCE  // (timing: the 32-slot window from 257 orbitals on, at every call size)
    const bool win_force = n > 256 && !tbk_band_fused(n);
#else
    const bool win_force = false;
#endif
    if (n > BAND_LDS_CHASE_MAXN || chase_global_forced(n) || small_window || win_force) {
        const int np = chase_pitch(n);
        // The working diagonals in a cyclic LDS window in front of the global buffer (band_chase4w_kernel; the same bits as the
        // global-memory form below).  One workgroup per CU (158 KiB of LDS) and still ahead at every call size: whole eigenval of
        // 2048 k-points 44.7 -> 41.0 us per k-point at 520 orbitals, 110.6 -> 99.6 at 768, 245.5 -> 216.1 at 1024; one k-point 15.2 ->
        // 13.0 ms at 1024, 32.1 -> 27.2 at 1536, 53.7 -> 44.4 at 2048.
        if (window_env && !chase_global_forced(n)) {
            d2* d_b = static_cast<d2*>(const_cast<void*>(d_band));
            const size_t stride = tbk_band_bytes_per_matrix(n) / sizeof(d2);
            if (small_window && !win_force) {
                const size_t ldsw = (size_t)16 * 281 * 16 + (size_t)4 * 64 * 16 + (size_t)n * sizeof(int) + 16;
                static std::atomic<bool> raised_s[TBK_MAX_DEVICES] = {};
                TBK_HIP(tbk_raise_lds_limit(reinterpret_cast<const void*>(&band_chase4w_kernel<4, 272, 281>), 160 * 1024, raised_s));
                hipLaunchKernelGGL((band_chase4w_kernel<4, 272, 281>), dim3((unsigned)nk), dim3(256), ldsw, s, d_b, stride, n, np, d_D, d_E);
                TBK_HIP(hipGetLastError());
                return TBK_OK;
            }
            const size_t ldsw = (size_t)16 * 521 * 16 + (size_t)8 * 64 * 16 + (size_t)n * sizeof(int) + 16;
            static std::atomic<bool> raised_w[TBK_MAX_DEVICES] = {};
            TBK_HIP(tbk_raise_lds_limit(reinterpret_cast<const void*>(&band_chase4w_kernel<8, 512, 521>), 160 * 1024, raised_w));
            hipLaunchKernelGGL((band_chase4w_kernel<8, 512, 521>), dim3((unsigned)nk), dim3(512), ldsw, s, d_b, stride, n, np, d_D, d_E);
            TBK_HIP(hipGetLastError());
            return TBK_OK;
        }
        // 32 sweeps in flight, two steps apart, from 512 orbitals on (a sweep is n / 8 >= 64 steps long); 16 below
        static const int env_nwg = tbk_exp_env("TBK_CHASE_NW") ? atoi(tbk_exp_env("TBK_CHASE_NW")) : 0;
        // (TBK_CHASE_NW=12, round 5: twelve waves = 48 sweeps in flight for calls of a few matrices -- measured: one-k eigenval
        // 13.88 -> 14.09 ms at 768 orbitals, 24.90 -> 25.17 at 1024, the same bits: the ticks' global-memory round trips, not the
        // 32 slots, bound it.  Eight stay.)
        const int nwg = env_nwg ? env_nwg : (n <= 256 ? 4 : 8);
        const size_t ldsg = (size_t)nwg * 64 * 16 + (size_t)n * sizeof(int) + 16;
        d2* d_b = static_cast<d2*>(const_cast<void*>(d_band));
        const size_t stride = tbk_band_bytes_per_matrix(n) / sizeof(d2);
        if (nwg <= 4)
            hipLaunchKernelGGL(band_chase4g_kernel<4>, dim3((unsigned)nk), dim3(256), ldsg, s, d_b, stride, n, np, 2, d_D, d_E);
#ifdef TBK_EXPERIMENTS
        else if (nwg <= 8)
            hipLaunchKernelGGL(band_chase4g_kernel<8>, dim3((unsigned)nk), dim3(512), ldsg, s, d_b, stride, n, np, 2, d_D, d_E);
        else
            hipLaunchKernelGGL(band_chase4g_kernel<12>, dim3((unsigned)nk), dim3(768), ldsg, s, d_b, stride, n, np, 2, d_D, d_E);
#else
        else
            hipLaunchKernelGGL(band_chase4g_kernel<8>, dim3((unsigned)nk), dim3(512), ldsg, s, d_b, stride, n, np, 2, d_D, d_E);
#endif
        TBK_HIP(hipGetLastError());
        return TBK_OK;
    }
    {
        const int np = chase_pitch(n);
        // Consecutive sweeps run `stagger` chase steps apart: 2 is the closest that keeps the steps of one tick on
        // disjoint cells (tools/two_stage_model.py: check_pipeline).  Waves per workgroup: enough sweeps in flight to
        // fill that pipeline (a sweep is ~n / 8 steps long).  TBK_CHASE_NW / TBK_CHASE_STAGGER: measurements only.
        static const int env_nw = tbk_exp_env("TBK_CHASE_NW") ? atoi(tbk_exp_env("TBK_CHASE_NW")) : 0;
        static const int env_stagger = tbk_exp_env("TBK_CHASE_STAGGER") ? atoi(tbk_exp_env("TBK_CHASE_STAGGER")) : 0;
        const int stagger = env_stagger >= 2 ? env_stagger : 2;
        // four sweeps per wave: a sweep is ~n / 8 steps long and sweeps start two ticks apart
        const int nw4 = env_nw ? env_nw : (n <= 128 ? 2 : n <= 256 ? 4 : 8);
        const size_t lds4 = (size_t)16 * np * 16 + (size_t)nw4 * 64 * 16 + (size_t)n * sizeof(int) + 16;
        static std::atomic<bool> raised4[3][TBK_MAX_DEVICES] = {};
#define TBK_CHASE4(NWV, SLOT)                                                                                             \
    do {                                                                                                                  \
        TBK_HIP(tbk_raise_lds_limit(reinterpret_cast<const void*>(&band_chase4_kernel<NWV>), 160 * 1024, raised4[SLOT]));   \
        hipLaunchKernelGGL(band_chase4_kernel<NWV>, dim3((unsigned)nk), dim3(NWV * 64), lds4, s, static_cast<const d2*>(d_band), tbk_band_bytes_per_matrix(n) / sizeof(d2), n, np, stagger, d_D, d_E); \
    } while (0)
        if (nw4 <= 2)
            TBK_CHASE4(2, 0);
        else if (nw4 <= 4)
            TBK_CHASE4(4, 1);
        else
            TBK_CHASE4(8, 2);
#undef TBK_CHASE4
        TBK_HIP(hipGetLastError());
    }
    return TBK_OK;
}

int tbk_launch_band_chase(tbk_model* m, hipStream_t s, const void* d_band, int64_t nk, double* d_de) {
    if (nk == 0) return TBK_OK;
    StageTimer t(m, TBK_T_EIG, s);
    return launch_chase(m, s, d_band, nk, d_de, d_de + (size_t)nk * m->n_orb);
}
