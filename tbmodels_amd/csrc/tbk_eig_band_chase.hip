// tbk_eig_band_chase.hip -- stage 2 of the two-stage reduction as kernels of its own: band -> tridiagonal by Householder bulge
// chasing (chase4_body, tbk_band_chase.h) with the 16 working diagonals in LDS (band_chase4_kernel, up to 512 orbitals), in global
// memory (band_chase4g_kernel) or in a cyclic LDS window in front of the global buffer (band_chase4w_kernel: above 512 orbitals, and
// 257 - 768 in big calls).  Reference step: scipy.linalg.eigvalsh per k-point (/root/reference/src/tbmodels/_tb_model.py:1147-1150).
// Split out of tbk_eig_band.hip in round 6; the comments at the kernels are unchanged.

#include "tbk_band.h"
#include "tbk_band_chase.h"

namespace {

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

}  // namespace

// Stage two: d_band -> d_de = d[nk][n] followed by e[nk][n]
int tbk_band_launch_chase(tbk_model* m, hipStream_t s, const void* d_band, int64_t nk, double* d_D, double* d_E) {
    const int n = m->n_orb;
    // TBK_CHASE_WINDOW=0 (measurements): no windowed kernel -- above 512 orbitals the global-memory form, the plain LDS form below
    static const bool window_env = !(getenv("TBK_CHASE_WINDOW") && atoi(getenv("TBK_CHASE_WINDOW")) == 0);
    const bool small_window = window_env && tbk_band_chase_small_window(m, n, nk);
#ifdef TBK_ABLATE_WIN_FORCE  // (timing: the 32-slot window from 257 orbitals on, at every call size)
    const bool win_force = n > 256 && !tbk_band_fused(n);
#else
    const bool win_force = false;
#endif
    if (n > BAND_LDS_CHASE_MAXN || tbk_band_chase_global_forced(n) || small_window || win_force) {
        const int np = tbk_band_chase_pitch(n);
        // The working diagonals in a cyclic LDS window in front of the global buffer (band_chase4w_kernel; the same bits as the
        // global-memory form below).  One workgroup per CU (158 KiB of LDS) and still ahead at every call size: whole eigenval of
        // 2048 k-points 44.7 -> 41.0 us per k-point at 520 orbitals, 110.6 -> 99.6 at 768, 245.5 -> 216.1 at 1024; one k-point 15.2 ->
        // 13.0 ms at 1024, 32.1 -> 27.2 at 1536, 53.7 -> 44.4 at 2048.
        if (window_env && !tbk_band_chase_global_forced(n)) {
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
        const int np = tbk_band_chase_pitch(n);
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
    return tbk_band_launch_chase(m, s, d_band, nk, d_de, d_de + (size_t)nk * m->n_orb);
}
