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

#include "tbk_band.h"
#include "tbk_band_chase.h"

namespace {

// ------------------------------------------------------------------------------------------------
// stage 1
// ------------------------------------------------------------------------------------------------

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

}  // namespace

// ------------------------------------------------------------------------------------------------
// sizes, policy, launchers
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
// per matrix: the pending [V | W] rows in fragment order (16 complex per row) + the next panel's V (8 complex per row; used
// when it does not live in LDS)
// the sizes that take the launch chain of band_xl_*: above 1024 orbitals (TBK_BAND_XL_FROM=n: above n -- tests run the chain
// at sizes the NumPy model is quick at, and A/B it against the one-workgroup kernels)
bool tbk_band_is_xl(int n) {
    static const int from = getenv("TBK_BAND_XL_FROM") ? atoi(getenv("TBK_BAND_XL_FROM")) : 1024;
    return n > from;
}
// (+ for the launch chain of band_xl_*: X / the panel's rows [npad][8] and T of the panel)
size_t tbk_band_scratch_per_matrix(int n) {
    const size_t nbk = (size_t)((n + TS - 1) / TS);
    return (nbk * (256 + TS * PB) + nbk * TS * PB + 64) * sizeof(d2);  // (calls of a few matrices take that chain at every size)
}

int tbk_band_chase_pitch(int n) {
    // TBK_CHASE_PITCH=r (measurements): pitch = r mod 16.  Bank model of the four-sweeps-per-wave layout (DESIGN_LOG R4.2): 148 LDS
    // cycles per tick at 9, 138 at 3 or 11 -- reads of two sweeps that share a 16-lane group collide at every pitch
    static const int want = tbk_exp_env("TBK_CHASE_PITCH") ? (atoi(tbk_exp_env("TBK_CHASE_PITCH")) & 15) | 1 : 9;
    int np = n + PB;
    while (np % 16 != want) ++np;
    return np;
}

bool tbk_band_fused(int n);
// above: every panel as three launches with nothing per row in registers or LDS (band_xl_*).  The limit is what has been
// validated (tests/test_gpu_parity.py: 1030 / 1536 / 2048 / 2050 / 3000 / 4096); nothing in the kernels depends on it.  TBK_BAND_XL=0: rocSOLVER above
// 1024 orbitals, as until round 4 (measurements).
static int band_maxn() {
    static const bool xl = !(getenv("TBK_BAND_XL") && atoi(getenv("TBK_BAND_XL")) == 0);
    return xl ? 4096 : BAND_ONE_WG_MAXN;
}
#define BAND_MAXN band_maxn()
// TBK_CHASE_GLOBAL=1 (measurements): the global-memory chase at every size that runs it as its own launch -- 9 KiB of LDS
// and 158 registers per wave instead of 133 KiB at 512 orbitals, so its workgroups fit beside those of other kernels
bool tbk_band_chase_global_forced(int n) {
    static const bool forced = tbk_exp_env("TBK_CHASE_GLOBAL") && atoi(tbk_exp_env("TBK_CHASE_GLOBAL")) != 0;
    return forced && !tbk_band_fused(n);
}
// 257 - 768 orbitals, calls of more matrices than the chip has CUs: the windowed kernel with 16 sweep slots and 272 columns -- 78 KiB
// of LDS instead of the 133 KiB of the plain LDS form at 512 orbitals, so two of its workgroups share a CU, or one sits beside a
// first-stage workgroup of the next chunk (76 KiB).  A matrix takes more and slower ticks (1293 x ~2.2 us instead of 1088 x 1.55 at
// 512 orbitals), the chip holds twice as many: cfg5 16.04 -> 16.63 k k-points/s, whole eigenval of 2048 k-points 12.93 -> 11.74 us per
// k-point at 320 orbitals, 18.82 -> 17.66 at 384, 34.57 -> 33.64 at 512; the same bits.  TBK_CHASE_WINDOW_SMALL=0: off.
bool tbk_band_chase_small_window(const tbk_model* m, int n, int64_t nk) {
    static const bool on = !(tbk_exp_env("TBK_CHASE_WINDOW_SMALL") && atoi(tbk_exp_env("TBK_CHASE_WINDOW_SMALL")) == 0);
    // Up to 768 orbitals (TBK_CHASE_WINDOW_SMALL_MAXN, measurements): above 512 against the 32-slot window -- whole eigenval of 2048
    // k-points 41.1 -> 39.1 us per k-point at 520 orbitals, 64.8 -> 62.9 at 640, 100.1 -> 97.4 at 768, 206.2 -> 215.6 at 1000.
    static const int maxn = tbk_exp_env("TBK_CHASE_WINDOW_SMALL_MAXN") ? atoi(tbk_exp_env("TBK_CHASE_WINDOW_SMALL_MAXN")) : 768;
    return on && n > 256 && n <= maxn && !tbk_band_fused(n) && std::max<int64_t>(m->call_nk, nk) > 256;
}
// does a matrix' band buffer carry the 16 working diagonals behind the compact band (by the size alone: any call may need them)
static bool chase_has_buffer(int n) { return n > BAND_LDS_CHASE_MAXN || tbk_band_chase_global_forced(n) || (n > 256 && !tbk_band_fused(n)); }

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
    if (tbk_band_is_xl(n)) return false;  // (the launch chain ends in the band's way out; the second stage is a launch of its own)
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
    return ((size_t)n * (PB + 1) + (chase_has_buffer(n) ? (size_t)16 * tbk_band_chase_pitch(n) : 0)) * sizeof(d2);
}

// Calls of a few matrices (Z2Pack-style lines and single k-points, _tb_model.py:1103-1108; band-structure paths of a few dozen
// points): the first stage as a chain of launches (PHASE 1 / 2 of band_reduce_kernel), so that every tile pass runs on
// `members` CUs per matrix instead of one.  By the size of the CALL (TBK_OPT_K_CHUNK must not change a result: the partial
// sums of the members differ from one workgroup's in the last bit).  TBK_BAND_SPLIT=0: off (measurements).
bool tbk_band_split(const tbk_model* m, int64_t nk) {
    static const bool on = !(getenv("TBK_BAND_SPLIT") && atoi(getenv("TBK_BAND_SPLIT")) == 0);
    static const int64_t forced_limit = tbk_exp_env("TBK_BAND_SPLIT_MAX") ? atoll(tbk_exp_env("TBK_BAND_SPLIT_MAX")) : 0;
    const int n = m->n_orb;
    if (!on || n <= 128 || n > BAND_ONE_WG_MAXN || tbk_band_is_xl(n)) return false;
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
    const int np = tbk_band_chase_pitch(n);
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

// Stage one: the upper triangle of every d_H matrix is overwritten; d_vw: scratch of tbk_band_scratch_per_matrix(n)
// bytes per matrix; d_band receives the band, tbk_band_bytes_per_matrix(n) bytes per matrix.
int tbk_launch_band_reduce(tbk_model* m, hipStream_t s, double* d_H, int64_t nk, void* d_vw, void* d_band, double* d_de_fused) {
    const int n = m->n_orb;
    if (nk == 0) return TBK_OK;
    StageTimer t(m, TBK_T_EIG, s);
    if (tbk_band_is_xl(n)) return tbk_band_launch_xl(m, s, d_H, n, nk, static_cast<d2*>(d_vw), static_cast<d2*>(d_band), d_de_fused);
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
    const int np = tbk_band_chase_pitch(n);
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
        return tbk_band_launch_xl(m, s, d_H, n, nk, static_cast<d2*>(d_vw), static_cast<d2*>(d_band));
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
