// tbk_eig_stream.hip -- Householder tridiagonalisation for 64 < n_orb <= 512: one workgroup per matrix,
// the matrix stays in HBM / L2; blocked and symmetric, so that per Householder step only the stored
// (upper) triangle of the trailing matrix is READ, and it is written back once every NB steps.
//
// Reference step: scipy.linalg.eigvalsh per k-point (/root/reference/src/tbmodels/_tb_model.py:1147-1150).
// Above 64 orbitals the matrix no longer fits the registers of a workgroup (tbk_eig_small.hip); rocSOLVER's
// batched zhetrd needs 46 us per 256x256 matrix (hundreds of her2 / hemv launches) and 260 us at n = 512.
// An unblocked reduction that streams the full matrix once per step is HBM bound at 2 * 16 n^3 / 3 bytes
// per matrix (measured: 5.1 TB/s, 35 us at n = 256 -- no better than the vendor path).  So, LAPACK-latrd
// style, the rank-2 updates are kept as a panel (V, W) of up to NB pairs in LDS and applied lazily:
//
//     1. column j of the up-to-date matrix = conj(row j of the stored triangle) - panel terms        O(n NB)
//     2. reflector from it: beta, tau, v'                                                             O(n)
//     3. ONE pass over the stored triangle of the trailing matrix, A[r][c], c >= r:
//          (only when the panel is full: A[r][c] -= sum_b V_b[r] conj(W_b[c]) + W_b[r] conj(V_b[c]); store)
//          u_r += A[r][c] v'_c          (row part: wave reduction)
//          u_c += conj(A[r][c]) v'_r    (column part, c > r: per-lane accumulators, added wave by wave)
//        and, while the panel is not applied, u -= V (W^H v') + W (V^H v')                          O(n NB)
//     4. rho = v'^H u (real), w' = tau u - (|tau|^2 rho / 2) v';  append (v', w') to the panel        O(n)
//
// Traffic per matrix: 16 n^3 / 6 bytes of reads plus 2 * 16 n^3 / (6 NB) for the flushes -- 56 MB at
// n = 256, NB = 8 instead of 178 MB.  Rows are contiguous in the stored triangle (the H(k) kernels' TRI
// output is used as is), a wave owns whole rows (16 B per lane, coalesced), per-row scalars come from LDS,
// summation order is fixed (no floating-point atomics): results are reproducible.

#include <cstdlib>

#include <algorithm>

#include "tbk_internal.h"

namespace {

typedef double d2 __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_mov<0x128>(v);
    v += dpp_mov<0x124>(v);
    v += dpp_mov<0x122>(v);
    v += dpp_mov<0x121>(v);
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
    }
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
    }
    return v;
}

// Several wave-wide sums at once: 64 lanes x 4 values -> lane l ends with the total of value (l >> 4).
// Each stage halves the number of live values while it doubles the lanes summed (a transposing butterfly):
// 21 VALU issues instead of 4 x 18 for four separate wave_sum calls.  The summation tree of every value is
// fixed, so results are reproducible.
__device__ __forceinline__ void swap_add(double& x, double y, bool half32) {
    // x <- [sum of x over the lane pair | sum of y over the lane pair] (lower | upper half, or even | odd rows)
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

// four values: lane l ends with the total of value (l >> 4)
__device__ __forceinline__ double reduce4(double (&p)[4]) {
    swap_add(p[0], p[2], true);   // lane bit 5 <- value bit 1
    swap_add(p[1], p[3], true);
    swap_add(p[0], p[1], false);  // lane bit 4 <- value bit 0
    double v = p[0];
    v += dpp_mov<0x128>(v);  // row_ror:8, 4, 2, 1: the sum over the row of 16 lanes
    v += dpp_mov<0x124>(v);
    v += dpp_mov<0x122>(v);
    v += dpp_mov<0x121>(v);
    return v;
}

// barrier with explicit waits: LDS stores (lgkmcnt) and the global row stores other waves will re-read (vmcnt)
__device__ __forceinline__ void wg_sync() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
}

constexpr int ST_MAXN = 512;

// Phase clock for tools/stream_phase_clock.hip (compiled out of the library): cycles per phase of a Householder
// step, accumulated by thread 0 of workgroup 0.
#ifdef TBK_PHASE_CLOCK
__device__ unsigned long long tbk_phase_clock[16];
#define TBK_CLK(k)                                                  \
    do {                                                            \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                  \
            const unsigned long long now_ = clock64();         \
            tbk_phase_clock[k] += now_ - clk_prev_;                 \
            clk_prev_ = now_;                                       \
        }                                                           \
    } while (0)
#else
#define TBK_CLK(k)
#endif

__device__ __forceinline__ d2 cmul(d2 a, d2 b) { return (d2){a[0] * b[0] - a[1] * b[1], a[0] * b[1] + a[1] * b[0]}; }
// a * conj(b)
__device__ __forceinline__ d2 cmulc(d2 a, d2 b) { return (d2){a[0] * b[0] + a[1] * b[1], a[1] * b[0] - a[0] * b[1]}; }

// NU: 64-column chunks per row (n <= 64 NU); NB: panel width; ST_THREADS: workgroup size.  Up to 128 orbitals a
// step is a chain of barriers and short phases (17 k cycles per step at n = 80 with eight waves, of which the
// pass over the triangle needs ~2 k): four waves per matrix, twice as many matrices per CU.
template <int NU, int NB, int ST_THREADS>
__global__ void __launch_bounds__(ST_THREADS)
herm_tridiag_stream_kernel(double* __restrict__ H, int n, double* __restrict__ D, double* __restrict__ E, int n_steps) {
    // n_steps = n - 1: the whole reduction.  n_steps = n - T, T = 64 or 128 (round 3): the first n - T Householder steps
    // only; the trailing T x T block, brought up to date with the pending panel, is written over the head of this matrix'
    // storage (row-major, leading dimension T, upper triangle) and the register-resident kernels of tbk_eig_small.hip take
    // over: a step costs ~17 k cycles here (a chain of barriers around a pass over memory), ~7 k in the eight-wave
    // kernel (128 -> 64) and ~2 k in the kernels below 64.
    constexpr int ST_WAVES = ST_THREADS / 64;
    extern __shared__ __attribute__((aligned(16))) double st_smem[];
    constexpr int NP = 64 * NU;  // padded vector length
    d2* sV = reinterpret_cast<d2*>(st_smem);  // [NB][NP] pending v
    d2* sW = sV + NB * NP;                    // [NB][NP] pending w
    d2* sx = sW + NB * NP;                    // [NP] column j
    d2* su = sx + NP;                         // [NP] u = A v'
    d2* svn = su + NP;                        // [NP] the new reflector v' (its own buffer: no barrier between x and v')
    // per-wave column sums, added up in fixed order after ONE barrier -- where the LDS budget allows it (two
    // workgroups per CU must still fit at 3 and 4 chunks): otherwise the waves add one after the other
    constexpr bool COLBUF = (NU >= 5);
    d2* scol = svn + NP;                      // [ST_WAVES][NP] (COLBUF only)
    __shared__ d2 stau;
    __shared__ d2 sg[NB], sh[NB];             // W_b^H v', V_b^H v'
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t mat = blockIdx.x;
    double* A = H + mat * (size_t)n * n * 2;
    double* Dm = D + mat * (size_t)n;
    double* Em = E + mat * (size_t)n;

    for (int i = tid; i < (2 * NB + 3) * NP; i += ST_THREADS) sV[i] = (d2){0.0, 0.0};
    int p = 0;  // pending (v, w) pairs in the panel
    wg_sync();
#ifdef TBK_PHASE_CLOCK
    unsigned long long clk_prev_ = clock64();
#endif

    for (int j = 0; j < n_steps; ++j) {
        // ---- 1. column j of the up-to-date matrix, from row j of the stored triangle ----
        for (int i = tid; i < NP; i += ST_THREADS) {
            d2 x = (d2){0.0, 0.0};
            if (i >= j && i < n) {
                const d2 a = *reinterpret_cast<const d2*>(A + ((size_t)j * n + i) * 2);
                x = (d2){a[0], -a[1]};
                for (int b = 0; b < p; ++b) {
                    const d2 t1 = cmulc(sV[b * NP + i], sW[b * NP + j]);
                    const d2 t2 = cmulc(sW[b * NP + i], sV[b * NP + j]);
                    x[0] -= t1[0] + t2[0];
                    x[1] -= t1[1] + t2[1];
                }
            }
            sx[i] = x;
            su[i] = (d2){0.0, 0.0};
        }
        wg_sync();
        TBK_CLK(1);

        // ---- 2. reflector (every wave computes the scalars; everyone writes its share of v') ----
        {
            double part = 0.0;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int i = lane + 64 * u;
                if (i > j + 1 && i < n) {
                    const d2 x = sx[i];
                    part = fma(x[0], x[0], part);
                    part = fma(x[1], x[1], part);
                }
            }
            const double sigma = wave_sum(part);
            const d2 al = sx[j + 1];
            const double alr = al[0], ali = al[1];
            double tr = 0.0, ti = 0.0, scr = 0.0, sci = 0.0, one = 0.0, ej = alr;
            if (!(sigma == 0.0 && ali == 0.0)) {
                const double beta = -copysign(sqrt(alr * alr + ali * ali + sigma), alr);
                const double rbeta = 1.0 / beta;
                tr = (beta - alr) * rbeta;
                ti = -ali * rbeta;
                const double qr = alr - beta, qi = ali;
                const double qn = 1.0 / (qr * qr + qi * qi);
                scr = qr * qn;
                sci = -qi * qn;
                one = 1.0;
                ej = beta;
            }
            const d2 dj = sx[j];
            if (tid == 0) {
                Dm[j] = dj[0];
                Em[j] = ej;
                stau = (d2){tr, ti};
            }
            for (int i = tid; i < NP; i += ST_THREADS) {
                d2 v = (d2){0.0, 0.0};
                if (i > j + 1 && i < n) {
                    v = cmul(sx[i], (d2){scr, sci});
                } else if (i == j + 1) {
                    v[0] = one;
                }
                svn[i] = v;
            }
        }
        wg_sync();
        TBK_CLK(2);
        d2 nv[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) nv[u] = svn[lane + 64 * u];

        // ---- 3. one pass over the stored triangle of the trailing matrix ----
        const bool flush = (p == NB);
        d2 colacc[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) colacc[u] = (d2){0.0, 0.0};
        // Row r of the stored triangle: columns r .. n-1, 16 B per lane, NU loads.  The loads are UNCONDITIONAL
        // (lanes left of the diagonal or beyond n re-read the row's diagonal element; masked when used): with
        // predicated loads hipcc waited for each one where it was issued, which serialised the pass on
        // memory latency.  A wave takes RB of its rows at a time: RB NU loads in flight, and the 2 RB row sums
        // (re, im) share ONE transposed reduction -- the pass is VALU-issue bound, and two wave_sum calls per
        // row were a third of its instructions.
        // rows per group: 2 everywhere (4 and 8 were measured for rows of <= 2 chunks: more registers, fewer
        // workgroups per CU, no faster)
        constexpr int RB = 2;
        auto load_group = [&](int r0, d2 (&a)[RB][NU]) {
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const int r = min(r0 + q * ST_WAVES, n - 1);
                const double* row = A + (size_t)r * n * 2;
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int c = min(max(lane + 64 * u, r), n - 1);
                    a[q][u] = *reinterpret_cast<const d2*>(row + (size_t)c * 2);
                }
            }
        };
        // (requesting the next group before this one is reduced was measured with tools/stream_phase_clock.hip:
        // 5 % slower at n = 80 .. 128, neutral above)
        for (int r0 = j + 1 + wave; r0 < n; r0 += ST_WAVES * RB) {
            d2 a[RB][NU];
            load_group(r0, a);
            double part[2 * RB];
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const int r = r0 + q * ST_WAVES;
                d2 rowsum = (d2){0.0, 0.0};
                if (r < n) {  // uniform
                    const d2 vr = svn[r];
                    double* row = A + (size_t)r * n * 2;
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        if (64 * (u + 1) <= r) continue;  // uniform: the whole chunk lies left of the diagonal
                        const int c = lane + 64 * u;
                        const bool valid = c >= r && c < n;
                        d2 av = a[q][u];
                        if (flush) {
                            const int cc = min(c, NP - 1);
#pragma unroll
                            for (int b = 0; b < NB; ++b) {
                                // av -= V_b[r] conj(W_b[c]) + W_b[r] conj(V_b[c]): eight FMAs straight into av
                                const d2 vr_b = sV[b * NP + r], wr_b = sW[b * NP + r];
                                const d2 vc_b = sV[b * NP + cc], wc_b = sW[b * NP + cc];
                                av[0] = fma(-vr_b[0], wc_b[0], av[0]);
                                av[1] = fma(-vr_b[1], wc_b[0], av[1]);
                                av[0] = fma(-vr_b[1], wc_b[1], av[0]);
                                av[1] = fma(vr_b[0], wc_b[1], av[1]);
                                av[0] = fma(-wr_b[0], vc_b[0], av[0]);
                                av[1] = fma(-wr_b[1], vc_b[0], av[1]);
                                av[0] = fma(-wr_b[1], vc_b[1], av[0]);
                                av[1] = fma(wr_b[0], vc_b[1], av[1]);
                            }
                            if (valid) *reinterpret_cast<d2*>(row + (size_t)c * 2) = av;
                        }
                        av[0] = valid ? av[0] : 0.0;
                        av[1] = valid ? av[1] : 0.0;
                        rowsum[0] = fma(av[0], nv[u][0], rowsum[0]);  // a * v'_c
                        rowsum[1] = fma(av[0], nv[u][1], rowsum[1]);
                        rowsum[0] = fma(-av[1], nv[u][1], rowsum[0]);
                        rowsum[1] = fma(av[1], nv[u][0], rowsum[1]);
                        if (c > r) {  // conj(a) * v'_r  (av = 0 on invalid lanes)
                            colacc[u][0] = fma(vr[0], av[0], colacc[u][0]);
                            colacc[u][1] = fma(vr[1], av[0], colacc[u][1]);
                            colacc[u][0] = fma(vr[1], av[1], colacc[u][0]);
                            colacc[u][1] = fma(-vr[0], av[1], colacc[u][1]);
                        }
                    }
                }
                part[2 * q] = rowsum[0];
                part[2 * q + 1] = rowsum[1];
            }
            {
                const double total = reduce4(part);  // lane l: value l >> 4 = (row slot l >> 5, re / im bit 4)
                const int r_mine = r0 + (lane >> 5) * ST_WAVES;
                if ((lane & 15) == 0 && r_mine < n) reinterpret_cast<double*>(su)[2 * r_mine + ((lane >> 4) & 1)] = total;
            }
        }
        if constexpr (COLBUF) {
#pragma unroll
            for (int u = 0; u < NU; ++u) scol[wave * NP + lane + 64 * u] = colacc[u];
            wg_sync();
        TBK_CLK(flush ? 7 : 3);
            for (int i = tid; i < NP; i += ST_THREADS) {  // column parts of all waves, fixed summation order
                d2 acc = su[i];
#pragma unroll
                for (int w = 0; w < ST_WAVES; ++w) {
                    const d2 t = scol[w * NP + i];
                    acc[0] += t[0];
                    acc[1] += t[1];
                }
                su[i] = acc;
            }
            wg_sync();
        TBK_CLK(4);
        } else {
            wg_sync();
        TBK_CLK(flush ? 7 : 3);
            // column parts, one wave after the other: fixed summation order
            for (int w = 0; w < ST_WAVES; ++w) {
                if (wave == w) {
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        const int c = lane + 64 * u;
                        const d2 t = su[c];
                        su[c] = (d2){t[0] + colacc[u][0], t[1] + colacc[u][1]};
                    }
                }
                wg_sync();
            }
        }
        // panel not applied to memory: u -= V (W^H v') + W (V^H v');  wave b reduces pair b
        if (!flush && p > 0) {
            for (int b = wave; b < p; b += ST_WAVES) {
                d2 g = (d2){0.0, 0.0}, h = (d2){0.0, 0.0};
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int i = lane + 64 * u;
                    const d2 tg = cmulc(nv[u], sW[b * NP + i]);  // conj(W) v'
                    const d2 th = cmulc(nv[u], sV[b * NP + i]);
                    g[0] += tg[0];
                    g[1] += tg[1];
                    h[0] += th[0];
                    h[1] += th[1];
                }
                g[0] = wave_sum(g[0]);
                g[1] = wave_sum(g[1]);
                h[0] = wave_sum(h[0]);
                h[1] = wave_sum(h[1]);
                if (lane == 0) {
                    sg[b] = g;
                    sh[b] = h;
                }
            }
            wg_sync();
            for (int i = tid; i < NP; i += ST_THREADS) {
                d2 acc = su[i];
                for (int b = 0; b < p; ++b) {
                    const d2 t1 = cmul(sV[b * NP + i], sg[b]);
                    const d2 t2 = cmul(sW[b * NP + i], sh[b]);
                    acc[0] -= t1[0] + t2[0];
                    acc[1] -= t1[1] + t2[1];
                }
                su[i] = acc;
            }
            wg_sync();
        TBK_CLK(5);
        }
        if (flush) p = 0;

        // ---- 4. w' = tau u - (|tau|^2 rho / 2) v', rho = v'^H u real; append (v', w') to the panel ----
        {
            double part = 0.0;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int i = lane + 64 * u;
                if (i > j && i < n) {
                    const d2 uu = su[i];
                    part = fma(nv[u][0], uu[0], part);
                    part = fma(nv[u][1], uu[1], part);
                }
            }
            const double rho = wave_sum(part);
            const d2 tau = stau;
            const double a2 = -0.5 * (tau[0] * tau[0] + tau[1] * tau[1]) * rho;
            for (int i = tid; i < NP; i += ST_THREADS) {
                const d2 v = svn[i];
                d2 w = (d2){0.0, 0.0};
                if (i > j && i < n) {
                    const d2 t = cmul(su[i], tau);
                    w[0] = fma(a2, v[0], t[0]);
                    w[1] = fma(a2, v[1], t[1]);
                }
                sV[p * NP + i] = v;
                sW[p * NP + i] = w;
            }
            ++p;
        }
        wg_sync();
        TBK_CLK(6);
    }
    if (n_steps < n - 1) {
        // hand-over: entry (i, c), i <= c, of the trailing T x T block = stored element - the pending panel terms, written
        // row-major with leading dimension T over the head of the matrix' storage.  The target overlaps rows that are
        // still to be read, but target index t = i T + c always lies below its source index (s0 + i) n + s0 + c: in
        // batches of ascending t, each read into registers before it is written, nothing is overwritten before it is read.
        const int s0 = n_steps;
        const int T = n - n_steps;
        const int lgT = 31 - __builtin_clz((unsigned)T);  // T = 64 or 128
        constexpr int PER = 8;
        for (int base = 0; base < T * T; base += PER * ST_THREADS) {
            d2 keep[PER];
#pragma unroll
            for (int t = 0; t < PER; ++t) {
                const int idx = base + tid + t * ST_THREADS;
                const int i = idx >> lgT, c = idx & (T - 1);
                d2 a = (d2){0.0, 0.0};
                if (i <= c && i < T) {
                    a = *reinterpret_cast<const d2*>(A + ((size_t)(s0 + i) * n + s0 + c) * 2);
                    for (int b = 0; b < p; ++b) {
                        const d2 t1 = cmulc(sV[b * NP + s0 + i], sW[b * NP + s0 + c]);
                        const d2 t2 = cmulc(sW[b * NP + s0 + i], sV[b * NP + s0 + c]);
                        a[0] -= t1[0] + t2[0];
                        a[1] -= t1[1] + t2[1];
                    }
                }
                keep[t] = a;
            }
            wg_sync();
#pragma unroll
            for (int t = 0; t < PER; ++t) {
                const int idx = base + tid + t * ST_THREADS;
                const int i = idx >> lgT, c = idx & (T - 1);
                if (i <= c && i < T) *reinterpret_cast<d2*>(A + (size_t)idx * 2) = keep[t];
            }
            // (no barrier here: the sources of later batches lie above every target written so far)
        }
        return;
    }
    // last diagonal element, with whatever is still pending in the panel
    if (tid == 0) {
        const int i = n - 1;
        double d = A[((size_t)i * n + i) * 2];
        for (int b = 0; b < p; ++b) {
            const d2 v = sV[b * NP + i], w = sW[b * NP + i];
            d -= 2.0 * (v[0] * w[0] + v[1] * w[1]);
        }
        Dm[i] = d;
        Em[i] = 0.0;
    }
}


// ------------------------------------------------------------------------------------------------
// Eigenvalues of the tridiagonal (d, e) by bisection on Sturm counts: one workgroup per matrix, one
// lane per eigenvalue (lane m brackets the m-th smallest, so the output is ascending by construction).
// The lane-per-matrix QL of tbk_eig_small.hip is a serial chain of O(n^2) dependent steps per matrix:
// at n = 256 it ran 71 ms per 8192 matrices on half-empty SIMDs and held 64 KiB of LDS per block, which
// kept a second reduction workgroup off every CU.  Here the n * ~55 Sturm sweeps of a matrix run in
// parallel lanes, (d, e^2) are broadcast from 8 KiB of LDS, and there is no division in the sweep:
// the characteristic polynomials p_i(x) = (d_i - x) p_{i-1} - e_{i-1}^2 p_{i-2} are carried directly and
// rescaled by a power of two every 8 steps; the number of sign changes is the number of eigenvalues < x.
// Accuracy: the bracket is halved until it is below 2 ulp of the spectrum's scale (backward stable).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

// LPE lanes per eigenvalue (1, 4 or 16): with LPE > 1 every sweep tests LPE interior points of the bracket at once and
// keeps the one of the LPE + 1 sub-intervals that holds the eigenvalue (multisection): log2(LPE + 1) bits per sweep
// instead of one.  It spends LPE times the lanes to cut the serial chain, so it is for calls of a few matrices only
// (a single 64 x 64 matrix: 57 sweeps = 116 us with one lane per eigenvalue, 14 sweeps with 16).
template <int LPE>
__global__ void __launch_bounds__(1024)
tridiag_bisect_kernel(const double* __restrict__ D, const double* __restrict__ E, int n, double* __restrict__ out,
                      int* __restrict__ flags) {
    // 16 n bytes of LDS (dynamic): at n = 64 a block must fit beside the 64 KiB QL blocks of the previous chunk
    extern __shared__ __attribute__((aligned(16))) double bs_smem[];
    const int n_pad = (n + 63) & ~63;
    // (d_i, e_{i-1}^2) side by side -- e_{i-1}^2 couples i - 1 and i: one 16-byte broadcast read per Sturm step (two
    // 8-byte reads per step made the LDS pipe, not the vector unit, the bound once a step was down to 5 VALU issues)
    double* sde = bs_smem;
    __shared__ double sred[2][16];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int nwave = blockDim.x >> 6;
    const size_t mat = blockIdx.x;
    double lo_g = 1e300, hi_g = -1e300;
    int bad = 0;
    // (round 5: above 1024 orbitals a thread holds more than one row of (d, e) and gridDim.y workgroups share the eigenvalues)
    for (int i = tid; i < n; i += (int)blockDim.x) {
        const double d = D[mat * n + i];
        const double e = (i < n - 1) ? E[mat * n + i] : 0.0;
        const double em = (i > 0) ? E[mat * n + i - 1] : 0.0;
        const double rad = fabs(e) + fabs(em);  // Gershgorin disc
        lo_g = fmin(lo_g, d - rad);
        hi_g = fmax(hi_g, d + rad);
        bad |= !(isfinite(d) && isfinite(e));
    }
    lo_g = wave_min(lo_g);
    hi_g = wave_max(hi_g);
    if (lane == 0) {
        sred[0][wave] = lo_g;
        sred[1][wave] = hi_g;
    }
    // NaN / Inf anywhere in H(k) reaches (d, e); fmin / fmax and the sign tests below would quietly drop it.
    // The caller maps non-finite eigenvalues to the ValueError of scipy's check_finite (_tb_model.py:1147-1150).
    if (__syncthreads_or(bad)) {
        if (blockIdx.y == 0) {
            for (int i = tid; i < n; i += (int)blockDim.x) out[mat * n + i] = __builtin_nan("");
            if (tid == 0) atomicAdd(flags + 1, 1);  // flags[1]: non-finite input (tbk_eigenval_check -> TBK_ERR_NOT_FINITE)
        }
        return;
    }
    double gl = sred[0][0], gu = sred[1][0];
    for (int w = 1; w < nwave; ++w) {
        gl = fmin(gl, sred[0][w]);
        gu = fmax(gu, sred[1][w]);
    }
    // Work in units of the spectrum's scale (an exact power of two, undone at the end), with the coupling e_i^2 kept
    // above 1e-60: a block-diagonal (d, e) -- sparse on-site blocks, decoupled orbitals -- has e_i = 0 exactly, and
    // at a midpoint that IS an eigenvalue of a leading block (x = 0 for any spectrum symmetric about 0) two
    // consecutive zeros p_{i-1} = p_i = 0 would zero the whole rest of the recurrence.  With e_i^2 > 0 a zero is
    // followed by -e_i^2 p_{i-1} != 0; the perturbation of the spectrum is below 1e-30 of its scale.
    const double scale_raw = fmax(fabs(gl), fabs(gu));
    const int sc_exp = (scale_raw > 0.0) ? -ilogb(scale_raw) : 0;
    for (int i = tid; i < n; i += (int)blockDim.x) {
        const double d = D[mat * n + i];
        const double e = (i < n - 1) ? E[mat * n + i] : 0.0;
        const double es = ldexp(e, sc_exp);
        sde[2 * i] = ldexp(d, sc_exp);
        if (i + 1 < n) sde[2 * (i + 1) + 1] = fmax(es * es, 1e-60);
        if (i == 0) sde[1] = 0.0;
    }
    __syncthreads();
    gl = ldexp(gl, sc_exp);
    gu = ldexp(gu, sc_exp);
    const double scale = fmax(fabs(gl), fabs(gu));  // in [1, 2), or 0 for the zero matrix
    const double slack = 4.0 * 2.220446049250313e-16 * scale * n + 1e-300;
    double lo = gl - slack, hi = gu + slack;
    // Brackets are halved down to 2 ulp of the spectrum's scale up to 64 orbitals (the QL kernel of those sizes and this
    // one must agree to 1e-13); above, down to n / 16 ulp -- the reduction in front of this kernel has rounded the
    // matrix by more than that already, and the last four or five halvings of 57 bought nothing.
    const double tol = (n > 64 ? 0.0625 * n : 2.0) * 2.220446049250313e-16 * scale + 1e-300;

    // Sturm count at x: sign changes along p_0 = 1, p_1, ..., p_n, a zero taking the sign opposite to its
    // predecessor.  Blocks of eight steps; in a block every step appends the sign bit of p to a history word (one
    // issue) and notes an exact zero (one compare into a lane mask), and the changes are counted once per block with a
    // population count: 5 VALU issues per step + 10 per block, against 9 per step + 8 when every step also applied the
    // zero rule (the bool / select form had compiled to 37).  A block in which ANY lane of the wave met an exact zero
    // (block-diagonal matrices with a spectrum symmetric about 0: tools/fuzz_parity.py) is redone with the rule in
    // every step; the other lanes get the same count from either form.
    double st_p = 0.0;  // p_n(x) of the last count = st_p * 2^st_e (the secant steps of the main loop use it)
    int st_e = 0;
    auto sturm_count = [&](double x) -> int {
        double pp = 1.0, p = sde[0] - x;
        int etot = 0;
        int sgn = (p <= 0.0) ? 1 : 0;  // p_1 against p_0 = 1 > 0
        int cnt = sgn;
        auto step = [&](double d_i, double e2_prev) {
            const double pn = fma(d_i - x, p, -e2_prev * pp);
            pp = p;
            p = pn;
            const int neg_bit = (int)((unsigned)__double2hiint(p) >> 31);
            const int s_new = (p == 0.0) ? (sgn ^ 1) : neg_bit;
            cnt += s_new ^ sgn;
            sgn = s_new;
        };
        auto rescale = [&]() {  // (p, pp) by a common power of two: signs and the recurrence are unaffected
            const double big = fmax(fabs(p), fabs(pp));
            if (big > 0.0) {
                const int ex = -ilogb(big);
                p = ldexp(p, ex);
                pp = ldexp(pp, ex);
                etot -= ex;
            }
        };
        int i0 = 1;
        for (; i0 + 8 <= n; i0 += 8) {
            // the eight (d, e^2) pairs of the block are fetched together (uniform addresses: LDS broadcasts); read
            // one by one inside the recurrence they put an LDS round trip into every step of the serial chain
            double dv[8], ev[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const d2 de = *reinterpret_cast<const d2*>(sde + 2 * (i0 + t));
                dv[t] = de[0];
                ev[t] = de[1];
            }
            const double p_in = p, pp_in = pp;
            unsigned hist = (unsigned)sgn;
            bool zero = false;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const double pn = fma(dv[t] - x, p, -ev[t] * pp);
                pp = p;
                p = pn;
                hist = __builtin_amdgcn_alignbit(hist, (unsigned)__double2hiint(p), 31);  // (hist << 1) | sign bit
                zero = zero || (p == 0.0);
            }
            if (__any(zero)) {
                p = p_in;
                pp = pp_in;
#pragma unroll
                for (int t = 0; t < 8; ++t) step(dv[t], ev[t]);
            } else {
                cnt += __popc((hist ^ (hist >> 1)) & 0xffu);
                sgn = (int)(hist & 1u);
            }
            rescale();
        }
        for (; i0 < n; ++i0) step(sde[2 * i0], sde[2 * i0 + 1]);  // ragged tail (< 8 steps)
        st_p = p;
        st_e = etot;
        return cnt;
    };

    // (calls of a few matrices, LPE > 1: gridDim.y workgroups share a matrix, part p takes the eigenvalues p * per ..., so
    // that the waves of ONE matrix sit on several CUs -- sixteen waves on one CU take turns on its four SIMDs, and a sweep
    // is a serial chain per wave: 66 -> 40 us for a single 64 x 64 matrix with four parts)
    const int m_per = (n + (int)gridDim.y - 1) / (int)gridDim.y;
    const int m = (int)blockIdx.y * m_per + tid / LPE;  // this lane's eigenvalue index
    const int sub = tid % LPE;  // ... and which interior point of the bracket it tests
    const int m_end = min(n, ((int)blockIdx.y + 1) * m_per);
    int cnt_lo = 0, cnt_hi = n;  // Sturm counts at the ends of the bracket
    if (n > 64) {  // (with LPE > 1 too: one sweep for log2(n + 1) halvings, against log2(LPE + 1) of a multisection sweep)
        // First round shared by the whole matrix: the n lanes count at n evenly spaced points of the Gershgorin
        // interval, and every lane reads ITS bracket off the (monotone) counts -- log2(n + 1) halvings for one sweep.
        int* scnt = reinterpret_cast<int*>(sde + 2 * n_pad);
        const double width = hi - lo;
        const double step_w = width / (n + 1);
        for (int t = tid; t < n; t += (int)blockDim.x) scnt[t] = sturm_count(lo + (t + 1) * step_w);
        __syncthreads();
        // smallest point index t with count(x_t) > m: the m-th eigenvalue lies in (x_{t-1}, x_t]
        int first = 0, len = n;  // binary search over t in [0, n): first t with scnt[t] > m, n if none
        while (len > 0) {
            const int half = len >> 1;
            const int probe = min(first + half, n - 1);
            const bool go_right = scnt[probe] <= m;
            first = go_right ? probe + 1 : first;
            len = go_right ? len - half - 1 : half;
        }
        const double new_lo = (first == 0) ? lo : lo + first * step_w;
        const double new_hi = (first >= n) ? hi : lo + (first + 1) * step_w;
        cnt_lo = (first == 0) ? 0 : scnt[min(first, n) - 1];
        cnt_hi = (first >= n) ? n : scnt[first];
        lo = new_lo;
        hi = new_hi;
    }
    // same trip count for every lane: cut the bracket by LPE + 1 per sweep down to `tol`
    int iters = 2;
    {
        const double w0 = (n > 64) ? (gu - gl + 2.0 * slack) / (n + 1) : hi - lo;
        for (double w = w0; w > tol && iters < 1100; w *= 1.0 / (LPE + 1)) ++iters;
    }

    if (LPE == 1) {
        // One lane per eigenvalue.  The bracket is ALWAYS moved by the Sturm count (count(lo) <= m < count(hi)); once it
        // isolates the eigenvalue (the counts differ by one) and p_n is known at both ends with opposite signs, the next
        // point is the secant point of p_n instead of the midpoint (Illinois variant: an end that stays put has its value
        // halved; a step that does not at least halve the bracket is followed by a midpoint step).  Clusters, exact
        // degeneracies (no sign change) and rounding noise at the end fall back to plain bisection, whose trip count
        // `iters` remains the bound; a wave leaves as soon as all its brackets are down to `tol`: ~20 sweeps instead of
        // ~45 at 256 orbitals on H(k) data.
        double f_lo = 0.0, f_hi = 0.0;
        int e_lo = 0, e_hi = 0;
        bool k_lo = false, k_hi = false, force_mid = false;
        int last_side = -1;
        for (int it = 0; it < 2 * iters; ++it) {  // (a secant step that disappoints is followed by a halving one)
            const double width = hi - lo;
            if (!__any(width > tol && m < n)) break;  // (lanes past the last eigenvalue only keep their wave company)
            double x = 0.5 * (lo + hi);
            bool sec = k_lo && k_hi && (cnt_hi - cnt_lo == 1) && !force_mid;
            if (sec) {
                const int de = min(max(e_hi - e_lo, -1000), 1000);
                const double r = ldexp(f_hi / f_lo, de);  // p_n(hi) / p_n(lo): negative across a simple root
                const double t = 1.0 / (1.0 - r);
                // never closer than tol / 2 to an end (Brent's minimal step): iterates that close in on the root from
                // one side would leave the other end where it is -- the step of tol / 2 lands beyond the root and the
                // bracket collapses to that size
                const double xs = fmin(fmax(lo + t * width, lo + 0.5 * tol), hi - 0.5 * tol);
                sec = (r < 0.0) && (xs > lo) && (xs < hi);
                if (sec) x = xs;
            }
            const int cnt = sturm_count(x);
            int side;
            if (cnt > m) {  // more than m eigenvalues below x: the m-th lies left of x
                hi = x;
                cnt_hi = cnt;
                f_hi = st_p;
                e_hi = st_e;
                k_hi = true;
                side = 1;
            } else {
                lo = x;
                cnt_lo = cnt;
                f_lo = st_p;
                e_lo = st_e;
                k_lo = true;
                side = 0;
            }
            if (sec && side == last_side) {
                if (side == 1)
                    e_lo -= 1;
                else
                    e_hi -= 1;
            }
            last_side = side;
            force_mid = sec && (hi - lo > 0.5 * width);
        }
    }
    for (int it = 0; LPE > 1 && it < iters; ++it) {
        const double width = hi - lo;
        const double x = lo + width * ((sub + 1) * (1.0 / (LPE + 1)));
        const int cnt = sturm_count(x);
        {
            // z = how many of the group's points have the m-th eigenvalue to their right (cnt <= m is monotone in
            // sub): the eigenvalue lies between point z - 1 (or lo) and point z (or hi); the same arithmetic in every
            // lane of the group, so the group keeps one common bracket
            int z = (cnt > m) ? 0 : 1;
#pragma unroll
            for (int off = 1; off < LPE; off <<= 1) z += __shfl_xor(z, off, 64);
            const double new_lo = (z == 0) ? lo : lo + width * (z * (1.0 / (LPE + 1)));
            const double new_hi = (z == LPE) ? hi : lo + width * ((z + 1) * (1.0 / (LPE + 1)));
            lo = new_lo;
            hi = new_hi;
        }
    }
    if (sub == 0 && m < m_end) out[mat * n + m] = ldexp(0.5 * (lo + hi), -sc_exp);
}

template <int NU, int NB, int ST_THREADS>
hipError_t launch_stream(hipStream_t s, unsigned nk, double* d_H, int n, double* d_D, double* d_E, int n_steps) {
    const bool colbuf = (NU >= 5);  // see COLBUF in the kernel
    const size_t lds = (size_t)(2 * NB + 3 + (colbuf ? ST_THREADS / 64 : 0)) * 64 * NU * sizeof(d2);
    static std::atomic<bool> raised[TBK_MAX_DEVICES] = {};
    {
        hipError_t e = tbk_raise_lds_limit(reinterpret_cast<const void*>(&herm_tridiag_stream_kernel<NU, NB, ST_THREADS>), (int)lds,
                                           raised);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((herm_tridiag_stream_kernel<NU, NB, ST_THREADS>), dim3(nk), dim3(ST_THREADS), lds, s, d_H, n, d_D,
                       d_E, n_steps);
    return hipGetLastError();
}

}  // namespace

// sizes the own solvers of this file and of tbk_eig_band.hip cover between them: the one-stage kernel here up to 512
// orbitals, the two-stage reduction up to band_maxn() = 4096 (round 5: the launch chain of band_xl_* above 1024; with
// TBK_BAND_XL=0 the range ends at 1024 again); rocSOLVER above
bool tbk_eig_stream_supported(int n) { return n > 64 && (n <= ST_MAXN || tbk_eig_band_supported(n)); }

// two-stage reduction (tbk_eig_band.hip) unless TBK_BAND=0 asks for the one-stage kernel of this file
bool tbk_eig_two_stage(const tbk_model* m) {
    static const bool band = [] {
        const char* v = getenv("TBK_BAND");
        return v == nullptr || atoi(v) != 0;
    }();
    if (m->n_orb > ST_MAXN) return tbk_eig_band_supported(m->n_orb);  // above 512 orbitals there is no one-stage kernel
    return band && tbk_eig_band_preferred(m->n_orb);
}

// d_de: d[nk][n] followed by e[nk][n]; d_H (upper triangle of the row-major H) is overwritten
int tbk_launch_tridiag_stream(tbk_model* m, hipStream_t s, double* d_H, int64_t nk, double* d_de, int method) {
    const int n = m->n_orb;
    if (nk == 0) return TBK_OK;
    if (n > ST_MAXN && method == TBK_REDUCE_ONE_STAGE) {
        tbk_set_error("the one-stage reduction handles n_orb <= %d (n_orb = %d)", ST_MAXN, n);
        return TBK_ERR_ARGUMENT;
    }
    if (method == TBK_REDUCE_TWO_STAGE || (method == TBK_REDUCE_AUTO && tbk_eig_two_stage(m))) {  // both stages in order on this stream (single-chunk calls, tbk_tridiagonal_reduce)
        TBK_CHECK(m->ws_band.reserve((size_t)nk * tbk_band_scratch_per_matrix(n)));
        TBK_CHECK(tbk_band_xl_reserve(m, nk));
        if (tbk_band_fused(n) && !tbk_band_split(m, nk)) return tbk_launch_band_reduce(m, s, d_H, nk, m->ws_band.ptr, nullptr, d_de);
        TBK_CHECK(m->ws_bandmat[0].reserve((size_t)nk * tbk_band_bytes_per_matrix(n)));
        if (tbk_band_xl_grouped(n, nk)) return tbk_launch_band_reduce(m, s, d_H, nk, m->ws_band.ptr, m->ws_bandmat[0].ptr, d_de);
        TBK_CHECK(tbk_launch_band_reduce(m, s, d_H, nk, m->ws_band.ptr, m->ws_bandmat[0].ptr));
        return tbk_launch_band_chase(m, s, m->ws_bandmat[0].ptr, nk, d_de);
    }
    double* d_D = d_de;
    double* d_Eo = d_de + (size_t)nk * n;
    StageTimer t(m, TBK_T_EIG, s);
    // One instantiation per 64 columns of padded row length (work and LDS scale with the padding).  Measured
    // (tools/bench_sizes.py): four waves and a panel of 4 up to 128 orbitals (n = 80: 1.08 us per matrix against
    // 1.42 with eight waves and a panel of 8); above that panel / workgroup variants are within 2 % of each other
    // Round 3: the kernel stops after the first n - 64 steps and the register-resident kernels finish the trailing 64 x 64
    // block (tbk_launch_tridiag_tail64); TBK_STREAM_SPLIT=0: the whole reduction here (measurements).
    static const bool split_on = !(tbk_exp_env("TBK_STREAM_SPLIT") && atoi(tbk_exp_env("TBK_STREAM_SPLIT")) == 0);
    // above 128 orbitals: this kernel goes down to the trailing 128 x 128 block, the eight-wave register kernel to
    // 64 x 64 (TBK_REG128=0: this kernel down to 64)
    const bool via128 = split_on && n > 128 && tbk_eig_reg128_supported(128);
    const int n_steps = via128 ? n - 128 : split_on ? n - 64 : n - 1;
    if (split_on && tbk_eig_reg128_supported(n))  // round 3: 65 .. 128 orbitals never leave the registers
        TBK_CHECK(tbk_launch_tridiag_reg128(s, d_H, n, nk, d_D, d_Eo, (int64_t)n * n * 2, n, 0));
    else if (n <= 128)
        TBK_HIP((launch_stream<2, 4, 256>(s, (unsigned)nk, d_H, n, d_D, d_Eo, n_steps)));
    else if (n <= 192)
        TBK_HIP((launch_stream<3, 8, 512>(s, (unsigned)nk, d_H, n, d_D, d_Eo, n_steps)));
    else if (n <= 256)
        TBK_HIP((launch_stream<4, 8, 512>(s, (unsigned)nk, d_H, n, d_D, d_Eo, n_steps)));
    else if (n <= 384)  // (5 and 7 chunks were measured too: no better than 6 and 8)
        TBK_HIP((launch_stream<6, 4, 512>(s, (unsigned)nk, d_H, n, d_D, d_Eo, n_steps)));
    else
        TBK_HIP((launch_stream<8, 4, 512>(s, (unsigned)nk, d_H, n, d_D, d_Eo, n_steps)));
    if (via128) TBK_CHECK(tbk_launch_tridiag_reg128(s, d_H, 128, nk, d_D, d_Eo, (int64_t)n * n * 2, n, n - 128));
    if (split_on) TBK_CHECK(tbk_launch_tridiag_tail64(s, d_H, nk, d_D, d_Eo, n, m->call_nk));
    return TBK_OK;
}

// eigenvalues of the tridiagonals d_de = (d[nk][n], e[nk][n]) -> d_E[nk][n], ascending
int tbk_launch_bisect(tbk_model* m, hipStream_t s, const double* d_de, int64_t nk, double* d_E) {
    const int n = m->n_orb;
    if (nk == 0) return TBK_OK;
    StageTimer t(m, TBK_T_QL, s);
    const int n_pad = (n + 63) / 64 * 64;
    const size_t lds = 2 * (size_t)n_pad * sizeof(double) + (size_t)n_pad * sizeof(int);  // (d, e^2) + the shared round's counts
    const double* d_e = d_de + (size_t)nk * n;
    // a few matrices cannot fill the chip with one lane per eigenvalue: spend lanes on shorter chains instead.  By the
    // size of the CALL, not of this chunk: TBK_OPT_K_CHUNK must not change results, and the variants differ in the
    // last bit.  Small matrices get the lanes their first wave would leave idle anyway (8 orbitals: 8 per eigenvalue).
    const int64_t call_nk = std::max(m->call_nk, nk);
    int lpe = call_nk <= 32 ? 16 : call_nk <= 512 ? 4 : 1;
    unsigned threads, parts = 1;
    static const int lpe_env = tbk_exp_env("TBK_BISECT_LPE") ? atoi(tbk_exp_env("TBK_BISECT_LPE")) : 0;  // (measurements: 1, 4 or 16 above 64 orbitals)
    if (n > 64) {
        // Above 64 orbitals (round 5): 16 or 4 lanes per eigenvalue at every size, over as many workgroups as that takes (until
        // round 4 the lanes had to fit ONE workgroup: 4 at 256 orbitals, 2 at 512 -- 31 sweeps where one lane with its secant steps
        // needs ~20 --, 1 above), while the call stays a few waves per CU: the sweeps are latency chains and idle lanes are free,
        // busy ones are not.  Measured (tools/bench_single_k.py, us of this stage, 1 / 4 / 16 lanes): one k-point at 256 orbitals
        // 188 / 160 / 114, at 512 414 / 370 / 244, at 1024 1348 / 1118 / 772; 64 k-points at 512 orbitals 462 / 404 / 800, at 1024
        // 1410 / 1300 / 3230, at 1536 1.8 / 3.5 ms / --.
        const int64_t eigenvalues = call_nk * (int64_t)n;
        if (lpe == 16 && eigenvalues * 16 > (int64_t(1) << 17)) lpe = 4;
        if (lpe == 4 && eigenvalues * 4 > (int64_t(1) << 18)) lpe = 1;
        if (lpe_env == 1 || lpe_env == 4 || lpe_env == 16) lpe = lpe_env;
        if (lpe > 1) {
            threads = (unsigned)std::min(1024, n_pad);  // (the shared first round: one sweep per thread up to 1024 orbitals)
            const unsigned per = threads / (unsigned)lpe;  // eigenvalues per workgroup (>= the kernel's m_per = ceil(n / parts))
            parts = ((unsigned)n + per - 1) / per;
        } else {
            threads = (unsigned)n_pad;
            if (threads > 1024) {  // above 1024 orbitals: the eigenvalues in `parts` workgroups of <= 1024 lanes
                parts = (threads + 1023) / 1024;
                const unsigned per = ((unsigned)n + parts - 1) / parts;  // eigenvalues per workgroup (the kernel's m_per)
                threads = (per + 63) / 64 * 64;
            }
        }
    } else {
        while (lpe < 16 && n * lpe * 2 <= 64) lpe *= 2;
        while (lpe > 1 && n * lpe > 1024) lpe /= 2;
        threads = (unsigned)((n * lpe + 63) / 64 * 64);
        // a few matrices with several lanes per eigenvalue: up to four workgroups per matrix (each still loads all of (d, e),
        // so at least n threads), their waves on different CUs
        if (lpe > 1)
            while (parts < 4 && (threads / (parts * 2)) % 64 == 0 && threads / (parts * 2) >= (unsigned)n_pad && n % (int)(parts * 2) == 0) parts *= 2;
        threads /= parts;
    }
    // (above ~3270 orbitals the (d, e^2) table and the shared round's counts pass 64 KiB -- 96 KiB: 20 bytes per orbital up to 4096
    // orbitals; the kernel has static LDS beside it, so not the whole 160 KiB)
    static std::atomic<bool> raised[5][TBK_MAX_DEVICES] = {};
#define TBK_BISECT(L, SLOT)                                                                                                        \
    do {                                                                                                                           \
        if (lds > (size_t(64) << 10))                                                                                              \
            TBK_HIP(tbk_raise_lds_limit(reinterpret_cast<const void*>(&tridiag_bisect_kernel<L>), 96 * 1024, raised[SLOT]));       \
        hipLaunchKernelGGL(tridiag_bisect_kernel<L>, dim3((unsigned)nk, parts), dim3(threads), lds, s, d_de, d_e, n, d_E,         \
                           m->ws_flag.as<int>());                                                                                  \
    } while (0)
    switch (lpe) {
        case 16: TBK_BISECT(16, 4); break;
        case 8: TBK_BISECT(8, 3); break;
        case 4: TBK_BISECT(4, 2); break;
        case 2: TBK_BISECT(2, 1); break;
        default: TBK_BISECT(1, 0); break;
    }
#undef TBK_BISECT
    TBK_HIP(hipGetLastError());
    return TBK_OK;
}
