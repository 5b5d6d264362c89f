// tbk_eig_band_xl.hip -- stage 1 of the two-stage reduction as a CHAIN OF LAUNCHES (round 5): above 1024 orbitals, and for calls of a
// few matrices at every two-stage size.  Reference step: scipy.linalg.eigvalsh per k-point
// (/root/reference/src/tbmodels/_tb_model.py:1147-1150).  Split out of tbk_eig_band.hip in round 6; the comments at the kernels are
// unchanged.

#include "tbk_band.h"

namespace {

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
    return tbk_band_is_xl(n) ? (size_t)n * n * sizeof(d2) + (xl_sweep4() ? xl_partial_doubles(n) * sizeof(double) : 0) : 0;
}
bool tbk_band_split(const tbk_model* m, int64_t nk);
// The chain's second matrix buffer for calls / chunks of up to max_nk matrices, reserved where the callers reserve ws_band and
// ws_bandmat -- in front of the pipeline, not inside a launch (a grow there is a free + malloc, i.e. a device synchronisation
// between the chunks of a call whose later chunk is the larger one).  Calls of a few matrices take the chain at
// every size (tbk_band_split): up to 96 matrices x 16 n^2 bytes, 1.6 GB at 1024 orbitals.
int tbk_band_xl_reserve(tbk_model* m, int64_t max_nk) {
    const int n = m->n_orb;
    if (!(tbk_band_is_xl(n) || tbk_band_split(m, max_nk))) return TBK_OK;
    return m->ws_xl.reserve((size_t)max_nk * ((size_t)n * n * sizeof(d2) + (xl_sweep4() ? xl_partial_doubles(n) * sizeof(double) : 0)));
}
// The first stage above 1024 orbitals: three launches per panel (serial phases / update sweep / product sweep), one more update
// sweep for the last pending update, then the band's way out.
static int xl_groups(int n, int64_t nk) {
    // TBK_BAND_XL_GROUPS=g (1 - 4; measurements): default 2
    static const int groups_env = tbk_exp_env("TBK_BAND_XL_GROUPS") ? std::min(4, std::max(1, atoi(tbk_exp_env("TBK_BAND_XL_GROUPS")))) : 2;
    return (tbk_band_is_xl(n) && nk >= 4 * groups_env) ? groups_env : 1;
}
bool tbk_band_xl_grouped(int n, int64_t nk) { return xl_groups(n, nk) > 1; }

// d_de != NULL: the second stage of every group runs behind its first stage on the group's stream and (d, e) are written there
int tbk_band_launch_xl(tbk_model* m, hipStream_t s, double* d_H, int n, int64_t nk, void* d_scratch_v, void* d_band_v, double* d_de) {
    d2* d_scratch = static_cast<d2*>(d_scratch_v);
    d2* d_band = static_cast<d2*>(d_band_v);
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
        if (d_de) return tbk_band_launch_chase(m, st, bd, nkg, d_de + (size_t)k0 * n, d_de + (size_t)(nk + k0) * n);
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
