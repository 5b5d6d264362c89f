// tbk_band_chase.h -- the body of the second stage (band -> tridiagonal by bulge chasing), inlined into its callers: the fused
// stage-1 kernel of tbk_eig_band.hip (up to 256 orbitals) and the kernels of tbk_eig_band_chase.hip.
#pragma once

#include "tbk_band.h"

namespace {

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

}  // namespace
