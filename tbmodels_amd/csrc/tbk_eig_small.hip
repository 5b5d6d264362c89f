// tbk_eig_small.hip -- batched Hermitian eigenvalues for n_orb <= 64, hand-written for gfx950.
//
// Reference step: one `scipy.linalg.eigvalsh` (LAPACK zheevr) per k-point from a Python loop
// (/root/reference/src/tbmodels/_tb_model.py:1147-1150).  A vendor batched zheevd spends ~6 us per
// 64x64 matrix here -- 5x the time of building H(k) -- so small matrices get their own path:
//
//   kernel 1a herm_tridiag_packed<NRP>  n <= 32: 64 / NRP matrices per wave (NRP = 8, 16, 32 lanes per matrix), the
//             matrices live in registers: a lane holds one row (NRP complex, statically indexed).  Householder
//             reduction to a real symmetric tridiagonal (d, e), LAPACK zhetd2/zlarfg arithmetic, full
//             (both-triangle) rank-2 updates so that every lane does the same straight-line work:
//                 p = tau A v;  w = p - (tau/2)(p^H v) v;  A -= v w^H + w v^H
//             v and w are broadcast through LDS per segment, reductions stay inside the segment (DPP), the next
//             Householder column is captured out of the update pass (no dynamic register index).
//   kernel 1b herm_tridiag4<64>         32 < n <= 64: four waves per matrix (see below).
//   kernel 2  tridiag_ql         ONE LANE PER MATRIX: implicit-shift QL on (d, e) held in LDS as
//             [index][lane] (conflict-free), then an in-LDS insertion sort; ascending output like
//             eigvalsh.  The serial chain is short (O(n^2) steps) and 64 matrices share a wave.
//
// Work per matrix at n = 64: ~2.1 Mflop of f64 VALU in kernel 1 (12 FMA per (step, column) pair per
// lane), ~0.3 Mflop in kernel 2; H is read once (16 n^2 B), eigenvalues written once (8 n B).
// Accuracy: backward stable, |dE| ~ n eps ||H||; the parity tests hold it to 1e-10 absolute.

#include <cstdlib>
#include <type_traits>

#include "tbk_dpp.h"
#include "tbk_internal.h"

namespace {

typedef double d2 __attribute__((ext_vector_type(2)));

// 64-lane all-reduce without LDS: rotate-adds inside each row of 16 lanes (DPP row_ror), then the two
// gfx950 row / half swaps.  Every lane ends with the bitwise-identical sum (each stage adds a commutative
// pair), which the wave-uniform branches on the result rely on.
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);  // bound_ctrl: no 'old' to materialise
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_mov<0x128>(v);  // row_ror:8
    v += dpp_mov<0x124>(v);  // row_ror:4
    v += dpp_mov<0x122>(v);  // row_ror:2
    v += dpp_mov<0x121>(v);  // row_ror:1
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

// Workgroup barrier with the LDS wait spelled out.  hipcc (ROCm 7.2) emitted the barrier at the head of
// the Householder loop without an s_waitcnt for the LDS store issued at the end of the previous
// iteration (the store reaches the barrier over the loop back edge): other waves then read the old
// column whenever LDS was slow -- wrong eigenvalues for ~50 of 100 000 matrices, and only while another
// kernel loaded the LDS pipe.  `tests/test_gpu_fullsize.py` catches it.
__device__ __forceinline__ void wg_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
}

__device__ __forceinline__ double bcast(double v, int lane) { return __shfl(v, lane, 64); }
// the same from a wave-uniform lane index: two v_readlane instead of an LDS round trip (ds_bpermute)
__device__ __forceinline__ double bcast_uniform(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// 1 / x and (sqrt(x), 1 / sqrt(x)) from the hardware estimates plus Newton steps: the reflector scalars sit on the serial
// path of every Householder step (the owner wave computes them while the other three wait), and IEEE division / sqrt
// cost a dozen instructions more per call.  x > 0 and well inside the double range (squared norms).
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
    e = fma(-root, rroot, 1.0);
    rroot = fma(rroot, e, rroot);
}


// one wave writes and reads an LDS array: its LDS operations execute in order, the fence keeps the compiler from
// moving them across each other
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// ------------------------------------------------------------------------------------------------
// kernel 1a: up to 32 orbitals -- SEVERAL matrices per wave.  With one matrix per wave an 8 x 8 matrix keeps 8 of 64
// lanes busy; here lane l works on row (l % NRP) of matrix (l / NRP) of the block (NRP = 8, 16 or 32 lanes per
// matrix).  Same arithmetic as above; what changes: reductions and broadcasts stay inside the NRP-lane segment, the
// LDS copies of v and w are addressed per segment, and the "column already reduced" case (tau = 0) is handled with
// selects instead of a wave-uniform branch, because the matrices of a wave need not agree on it.
// ------------------------------------------------------------------------------------------------
template <int SEG>
__device__ __forceinline__ double seg_sum(double v) {
    if (SEG == 8) {
        v += dpp_mov<0x141>(v);  // row_half_mirror: i <-> 7 - i
        v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
        v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
        return v;
    }
    v += dpp_mov<0x128>(v);
    v += dpp_mov<0x124>(v);
    v += dpp_mov<0x122>(v);
    v += dpp_mov<0x121>(v);
    if (SEG == 32) {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
    }
    return v;
}

template <int NRP>
__global__ void __launch_bounds__(64)
herm_tridiag_packed_kernel(const double* __restrict__ H, int n, int64_t nk, double* __restrict__ D,
                           double* __restrict__ E, int64_t h_stride, int ldd, int off) {
    // h_stride: doubles between consecutive matrices (n * n * 2 when they are packed back to back; larger when the
    // matrices are the trailing blocks herm_tridiag4_kernel left at the head of bigger ones); (d, e) of matrix m go to
    // D / E + m * ldd + off
    constexpr int G = 64 / NRP;  // matrices per wave
    constexpr int CG = 4;        // columns per uniform-branch group
    constexpr int NGR = NRP / CG;
    __shared__ d2 sv[64];
    __shared__ d2 sw[64];
    const int lane = threadIdx.x;
    const int row = lane % NRP;
    const int base = lane - row;  // first lane of this matrix' segment
    const int64_t mat = (int64_t)blockIdx.x * G + lane / NRP;
    const bool live = mat < nk;
    const int64_t mc = live ? mat : nk - 1;
    const double* Hm = H + (size_t)mc * h_stride;
    double* Dm = D + (size_t)mc * ldd + off;
    double* Em = E + (size_t)mc * ldd + off;

    // lane <- row `row` of the Hermitian matrix whose upper triangle is stored (unconditional clamped loads)
    double ar[NRP], ai[NRP];
    {
        // (at most 16 loads in flight together.  Three waves per SIMD -- a 168-register bound -- were measured for the
        // 32-lane instantiation: 60 B of scratch in the loop and 1.24 instead of 1.11 ms per 65536 matrices; no bound)
        constexpr int LB = NRP <= 16 ? NRP : 16;
        const int li = min(row, n - 1);
        static_for<0, NRP / LB>([&](auto hc) {
            constexpr int c0 = decltype(hc)::value * LB;
            d2 raw[LB];
#pragma unroll
            for (int u = 0; u < LB; ++u) {
                const int cc = min(c0 + u, n - 1);
                const int lo = min(li, cc), hi = max(li, cc);
                raw[u] = *reinterpret_cast<const d2*>(Hm + ((size_t)lo * n + hi) * 2);
            }
#pragma unroll
            for (int u = 0; u < LB; ++u) {
                const int c = c0 + u;
                const bool inside = row < n && c < n;
                ar[c] = inside ? raw[u][0] : 0.0;
                ai[c] = inside ? (c >= row ? raw[u][1] : -raw[u][1]) : 0.0;
            }
        });
    }
    double xr = ar[0], xi = ai[0];

    // v[c] and w[c] of a lane's own matrix reach the FMAs as DPP operands (row_newbcast: lane t of the lane's row of 16)
    // wherever a matrix spans whole rows of 16 lanes: NRP = 16 -- the vectors themselves are the operand registers;
    // NRP = 32 -- ONE permlane16 swap of a vector with itself leaves [r0 r0 r2 r2] and [r1 r1 r3 r3] (r_k = row k of 16
    // lanes), i.e. entries 0..15 and 16..31 of both matrices of the wave in every row of their segments.  The first
    // version read them as LDS broadcasts: 96 ds_read_b128 per step at 32 orbitals -- the LDS pipe was the bound.
    // (NRP = 8: two matrices share a row of 16 lanes; LDS as before.)
    constexpr bool BC = NRP >= 16;
    constexpr int NB = NRP >= 16 ? NRP / 16 : 1;
    auto spread = [&](double x, double (&o)[NB]) {
        if constexpr (NRP == 32) {
            const unsigned xl = (unsigned)__double2loint(x), xh = (unsigned)__double2hiint(x);
            const auto rl = __builtin_amdgcn_permlane16_swap(xl, xl, false, false);
            const auto rh = __builtin_amdgcn_permlane16_swap(xh, xh, false, false);
            o[0] = __hiloint2double((int)rh[0], (int)rl[0]);
            o[NB - 1] = __hiloint2double((int)rh[1], (int)rl[1]);
        } else {
            o[0] = x;
        }
    };

    for (int j = 0; j < n - 1; ++j) {
        if (live && row == j) Dm[j] = xr;
        const double alr = __shfl(xr, base + j + 1, 64), ali = __shfl(xi, base + j + 1, 64);
        const bool below = (row > j + 1) && (row < n);
        const double sigma = seg_sum<NRP>(below ? (xr * xr + xi * xi) : 0.0);
        // zlarfg: tau = 0 (H = I) when the column is already reduced and its pivot real
        const bool done = (sigma == 0.0) && (ali == 0.0);
        double root, rroot;
        fast_sqrt_rsqrt(done ? 1.0 : alr * alr + ali * ali + sigma, root, rroot);
        const double beta = done ? alr : -copysign(root, alr);
        const double rbeta = done ? 0.0 : -copysign(rroot, alr);
        const double tr = (beta - alr) * rbeta, ti = -ali * rbeta;
        const double qr = alr - beta, qi = ali;
        const double qn = done ? 0.0 : fast_rcp(qr * qr + qi * qi);
        const double scr = qr * qn, sci = -qi * qn;
        if (live && row == 0) Em[j] = beta;

        double vr = 0.0, vi = 0.0;
        if (below) {
            vr = xr * scr - xi * sci;
            vi = xr * sci + xi * scr;
        } else if (row == j + 1) {
            vr = 1.0;
        }
        double vbr[NB], vbi[NB];
        if constexpr (BC) {
            spread(vr, vbr);
            spread(vi, vbi);
        } else {
            wave_lds_fence();
            sv[lane] = (d2){vr, vi};
            wave_lds_fence();
        }

        // u = A v over the columns of this lane's matrix
        double pr = 0.0, pi = 0.0;
        {
            double par[CG], pai[CG];
#pragma unroll
            for (int cc = 0; cc < CG; ++cc) par[cc] = pai[cc] = 0.0;
            static_for<0, NGR>([&](auto grc) {
                constexpr int gr = decltype(grc)::value;
                if (gr * CG + CG - 1 > j) {  // uniform (all matrices of the launch have n orbitals)
                    if constexpr (BC) {
                        static_for<0, CG>([&](auto ccc) {
                            constexpr int cc = decltype(ccc)::value, c = gr * CG + cc;
                            fmac_bc<c & 15>(par[cc], vbr[c >> 4], ar[c]);
                            fmac_bc<c & 15>(pai[cc], vbi[c >> 4], ar[c]);
                            fnmac_bc<c & 15>(par[cc], vbi[c >> 4], ai[c]);
                            fmac_bc<c & 15>(pai[cc], vbr[c >> 4], ai[c]);
                        });
                    } else {
                        d2 vb[CG];
#pragma unroll
                        for (int cc = 0; cc < CG; ++cc) vb[cc] = sv[base + gr * CG + cc];
#pragma unroll
                        for (int cc = 0; cc < CG; ++cc) {
                            par[cc] = fma(ar[gr * CG + cc], vb[cc][0], par[cc]);
                            pai[cc] = fma(ar[gr * CG + cc], vb[cc][1], pai[cc]);
                        }
#pragma unroll
                        for (int cc = 0; cc < CG; ++cc) {
                            par[cc] = fma(-ai[gr * CG + cc], vb[cc][1], par[cc]);
                            pai[cc] = fma(ai[gr * CG + cc], vb[cc][0], pai[cc]);
                        }
                    }
                }
            });
#pragma unroll
            for (int cc = 0; cc < CG; ++cc) {
                pr += par[cc];
                pi += pai[cc];
            }
        }
        if (!((row > j) && (row < n))) pr = pi = 0.0;
        // A is Hermitian, so rho = v^H u is real: ONE reduction instead of the complex dot product p^H v of zhetd2
        // (p = tau u, p^H v = conj(tau) rho), and  w = p - (tau / 2)(p^H v) v = tau u - (|tau|^2 rho / 2) v
        const double rho = seg_sum<NRP>(pr * vr + pi * vi);
        const double a2 = -0.5 * (tr * tr + ti * ti) * rho;
        const double wr = fma(a2, vr, pr * tr - pi * ti);
        const double wi = fma(a2, vi, pr * ti + pi * tr);
        double wbr[NB], wbi[NB];
        if constexpr (BC) {
            spread(wr, wbr);
            spread(wi, wbi);
        } else {
            sw[lane] = (d2){wr, wi};
            wave_lds_fence();
        }

        // A -= v w^H + w v^H; the next Householder column is captured on the way
        double nxr = 0.0, nxi = 0.0;
        static_for<0, NGR>([&](auto grc) {
            constexpr int gr = decltype(grc)::value;
            if (gr * CG + CG - 1 > j) {
                if constexpr (BC) {
                    static_for<0, CG>([&](auto ccc) {
                        constexpr int cc = decltype(ccc)::value, c = gr * CG + cc;
                        double r = ar[c], mm = ai[c];
                        fnmac_bc<c & 15>(r, wbr[c >> 4], vr);
                        fnmac_bc<c & 15>(mm, wbr[c >> 4], vi);
                        fnmac_bc<c & 15>(r, wbi[c >> 4], vi);
                        fmac_bc<c & 15>(mm, wbi[c >> 4], vr);
                        fnmac_bc<c & 15>(r, vbr[c >> 4], wr);
                        fnmac_bc<c & 15>(mm, vbr[c >> 4], wi);
                        fnmac_bc<c & 15>(r, vbi[c >> 4], wi);
                        fmac_bc<c & 15>(mm, vbi[c >> 4], wr);
                        ar[c] = r;
                        ai[c] = mm;
                        if (c == j + 1) {  // uniform
                            nxr = r;
                            nxi = mm;
                        }
                    });
                } else {
                    d2 vb[CG], wb[CG];
#pragma unroll
                    for (int cc = 0; cc < CG; ++cc) {
                        vb[cc] = sv[base + gr * CG + cc];
                        wb[cc] = sw[base + gr * CG + cc];
                    }
#pragma unroll
                    for (int cc = 0; cc < CG; ++cc) {
                        const int c = gr * CG + cc;
                        ar[c] = fma(-vr, wb[cc][0], ar[c]);
                        ai[c] = fma(-vi, wb[cc][0], ai[c]);
                        ar[c] = fma(-vi, wb[cc][1], ar[c]);
                        ai[c] = fma(vr, wb[cc][1], ai[c]);
                        ar[c] = fma(-wr, vb[cc][0], ar[c]);
                        ai[c] = fma(-wi, vb[cc][0], ai[c]);
                        ar[c] = fma(-wi, vb[cc][1], ar[c]);
                        ai[c] = fma(wr, vb[cc][1], ai[c]);
                        if (c == j + 1) {  // uniform
                            nxr = ar[c];
                            nxi = ai[c];
                        }
                    }
                }
            }
        });
        xr = nxr;
        xi = nxi;
    }
    if (live && row == n - 1) Dm[n - 1] = xr;
    if (live && row == 0) Em[n - 1] = 0.0;
}

// ------------------------------------------------------------------------------------------------
// kernel 1b: the same reduction with FOUR waves per matrix (256 threads): wave q holds columns
// c = 4 t + q of every row (16 complex = 64 VGPRs per lane instead of 256), so the kernel fits
// beside the MFMA contraction's waves on a SIMD and its f64 VALU work runs in the shadow of their
// matrix-pipe time.  Per Householder step the four waves exchange two things through LDS -- the new
// column x (written by the wave that owns column j) and their partial products A v -- at the cost of
// two workgroup barriers; v and w are recomputed by every wave (each has all rows) and broadcast
// from a wave-private LDS copy, which needs no barrier.
// ------------------------------------------------------------------------------------------------
// Householder scalars of one step, computed ONCE per matrix by the wave that owns the column and passed to
// the other three through LDS (the f64 sqrt / divisions and the norm reduction cost ~100 VALU issues).
struct HhScalars {
    d2 tau;    // (beta - alpha) / beta
    d2 flag;   // [0] != 0: column already reduced (H = I), [1] unused
};

template <int NR, int NW>  // padded rows, waves per matrix
__global__ void __launch_bounds__(NW * 64, NW == 2 ? 2 : 4)
herm_tridiag4_kernel(double* H, int n, double* __restrict__ D, double* __restrict__ E, int n_steps, int64_t h_stride, int ldd,
                     int off) {
    // h_stride: doubles between consecutive matrices (n * n * 2 when packed back to back; larger when they are the trailing
    // 64 x 64 blocks the streaming kernel of tbk_eig_stream.hip left at the head of bigger matrices); (d, e) of matrix m go
    // to D / E + m * ldd + off
    // n_steps = n - 1: the whole reduction.  n_steps = n - 32 (split mode): only the first n - 32 Householder steps; the
    // trailing 32 x 32 block, fully updated, is then written over the head of this matrix' own storage (row-major,
    // leading dimension 32, upper triangle) for herm_tridiag_packed_kernel<32>, which finishes two such blocks per wave:
    // there the per-step overhead (reductions, scalar chain) is paid once per TWO matrices instead of four times per
    // matrix, and no lane idles on a retired row -- steps 32 .. 62 of a 64 x 64 matrix cost 1.4 ms per 32768 matrices
    // here and 0.7 there.
    const bool split = n_steps < n - 1;
    constexpr int NT = NR / NW;  // columns per lane
    constexpr int NB = (NT + 15) / 16;  // registers that hold a vector for the row_newbcast operands (16 columns each)
    constexpr int TB = 2;        // column groups per skip block
    // register arrays with a (uniform) dynamic index (`s_set_gpr_idx`): at most 16 columns each -- hipcc indexes a longer
    // vector through scratch memory
    constexpr int HT = NT / NB;
    static_assert(NT % NB == 0, "columns per lane must split evenly over the register arrays");
    typedef double dcol __attribute__((ext_vector_type(HT)));
    // 7 KiB of LDS per matrix: four resident workgroups leave room for two 64 KiB QL workgroups on the CU
    // (the QL of the previous chunk runs beside this kernel and must fit in one round).
    // reflector of step j in sx[j & 1]: written by the owner of column j after B2(j - 1), read by every wave between
    // B1(j) and B1(j + 1) -- while the owner of column j + 1 may already be writing the other copy
    __shared__ d2 sx[2][NR];
    __shared__ HhScalars ssc;    // its scalars
    __shared__ d2 sp[NW][NR];    // per-wave partial products
    // w: every wave computes the same bits and stores them to the SAME slots, then reads back through its own LDS
    // queue (ordered behind its own store) -- no barrier, one copy.  The next step's stores are behind B1 / B2, i.e.
    // after every wave's reads of this step.
    __shared__ d2 sw[NR];
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t mat = blockIdx.x;
    double* Hm = H + mat * (size_t)h_stride;
    double* Dm = D + mat * (size_t)ldd + off;
    double* Em = E + mat * (size_t)ldd + off;

    // Lane i <- row i of the Hermitian matrix whose upper triangle is stored: element (i, c) comes from
    // (min, max) of the pair, conjugated below the diagonal.  Unconditional loads from clamped addresses, all
    // NT in flight (hipcc waits for a predicated load where it is issued), masked afterwards.
    dcol ar0, ai0, ar1, ai1;  // columns t < HT, t >= HT (the second pair only with NB == 2); named, not an array:
                              // hipcc keeps an array of vectors in scratch memory
    static_assert(NB <= 2, "at most 32 columns per lane");
    auto AR = [&](auto bc) -> dcol& { if constexpr (decltype(bc)::value == 0) return ar0; else return ar1; };
    auto AI = [&](auto bc) -> dcol& { if constexpr (decltype(bc)::value == 0) return ai0; else return ai1; };
#define TBK_AR(t) AR(std::integral_constant<int, (t) / HT>{})[(t) % HT]
#define TBK_AI(t) AI(std::integral_constant<int, (t) / HT>{})[(t) % HT]
    {
        constexpr int LB = NT <= 16 ? NT : NT / 2;  // loads in flight together (a second set of NT pairs would spill)
        const int li = min(lane, n - 1);
        static_for<0, NT / LB>([&](auto t0c) {
            constexpr int t0 = decltype(t0c)::value * LB;
            d2 raw[LB];
#pragma unroll
            for (int u = 0; u < LB; ++u) {
                const int c = min(NW * (t0 + u) + q, n - 1);
                const int lo = min(li, c), hi = max(li, c);
                raw[u] = *reinterpret_cast<const d2*>(Hm + ((size_t)lo * n + hi) * 2);
            }
            static_for<0, LB>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                const int c = NW * (t0 + u) + q;
                const bool inside = lane < n && c < n;
                TBK_AR(t0 + u) = inside ? raw[u][0] : 0.0;
                TBK_AI(t0 + u) = inside ? (c >= lane ? raw[u][1] : -raw[u][1]) : 0.0;
            });
        });
    }

    // The owner of column jc publishes the reflector that annihilates it below the sub-diagonal -- v (lane i = v[i]) and
    // tau -- and stores d[jc], e[jc].  (It used to publish the column and the scale factor, and every wave formed v:
    // ~15 VALU issues per wave and step that three of the four waves now skip.)
    auto publish = [&](int jc) {
        const int tsel = jc / NW;
        double xr, xi;  // uniform dynamic index
        if (NB == 1 || tsel < HT) {
            xr = ar0[tsel % HT];
            xi = ai0[tsel % HT];
        } else {
            xr = ar1[tsel % HT];
            xi = ai1[tsel % HT];
        }
        if (lane == jc) Dm[jc] = xr;
        if (jc >= n - 1) {
            if (lane == 0) Em[n - 1] = 0.0;
            return;
        }
        const double alr = bcast_uniform(xr, jc + 1), ali = bcast_uniform(xi, jc + 1);
        const bool below = (lane > jc + 1) && (lane < n);
        const double sigma = wave_sum(below ? (xr * xr + xi * xi) : 0.0);
        HhScalars sc;
        double e_out;
        double vr = 0.0, vi = 0.0;
        if (sigma == 0.0 && ali == 0.0) {
            sc.tau = (d2){0.0, 0.0};
            sc.flag = (d2){1.0, 0.0};
            e_out = alr;
        } else {
            double root, rroot;
            fast_sqrt_rsqrt(alr * alr + ali * ali + sigma, root, rroot);
            const double beta = -copysign(root, alr);
            const double rbeta = -copysign(rroot, alr);
            const double qr = alr - beta, qi = ali;
            const double qn = fast_rcp(qr * qr + qi * qi);
            const double sr = qr * qn, si = -qi * qn;  // 1 / (alpha - beta)
            sc.tau = (d2){(beta - alr) * rbeta, -ali * rbeta};
            sc.flag = (d2){0.0, 0.0};
            e_out = beta;
            if (below) {
                vr = xr * sr - xi * si;
                vi = xr * si + xi * sr;
            } else if (lane == jc + 1) {
                vr = 1.0;
            }
        }
        if (lane < NR) sx[jc & 1][lane] = (d2){vr, vi};
        if (lane == 0) {
            ssc = sc;
            Em[jc] = e_out;
        }
    };
    if (q == 0) publish(0);
    int bc_slot[NB];  // the columns whose v / w this lane holds for the broadcasts: NW * (16 b + lane % 16) + q
#pragma unroll
    for (int b = 0; b < NB; ++b) bc_slot[b] = min(NW * (16 * b + (lane & 15)) + q, NR - 1);

    for (int j = 0; j < n_steps; ++j) {
#ifdef TBK_ABLATE_CASCADE
        // TIMING ONLY (results wrong by construction): what repacking the live rows could buy at most.  From step 16 of a
        // 64-row matrix on, 48 of the 64 lanes hold live rows: a cascade 64 -> 48 would keep four matrices on three waves.  Here
        // every fourth matrix simply stops at step 16 -- perfect packing, no repacking cost.  =2: the same at step 8 (56 rows:
        // one matrix in eight) on top.
        // (workgroup b runs on XCD b % 8 and, to begin with, on CU (b / 8) % 32 of it: the matrices that stop are picked by a
        // hash of b -- `b % 4 == 3` put them all on two XCDs, whose early finish bought nothing)
        const unsigned pick = (blockIdx.x * 2654435761u) >> 20;
        if (NR == 64 && j >= 16 && (pick & 3) == 3) break;
        if (TBK_ABLATE_CASCADE == 2 && NR == 64 && j >= 8 && (pick & 7) == 6) break;
        if (TBK_ABLATE_CASCADE == 3 && NR == 64 && j >= 16) break;  // (calibration: EVERY matrix stops at step 16 / at step 1)
        if (TBK_ABLATE_CASCADE == 4 && NR == 64 && j >= 1) break;
#endif
        wg_sync();  // B1: sx, ssc describe the reflector of column j
        const d2 vme = (lane < NR) ? sx[j & 1][lane] : (d2){0.0, 0.0};
        const HhScalars sc = ssc;
        const int jn = j + 1;  // next column, owned by wave jn % 4
        // (split mode: column n_steps is the first one of the trailing block -- its reflector is the next kernel's)
        const bool own_next = (jn & (NW - 1)) == q && !(split && jn == n_steps);
        if (__builtin_amdgcn_readfirstlane(__double2hiint(sc.flag[0])) != 0) {
            wg_sync();  // B2 (keeps the barrier count of both branches equal; orders the reads above)
            if (own_next) publish(jn);
            continue;
        }
        const double tr = sc.tau[0], ti = sc.tau[1];
        const double vr = vme[0], vi = vme[1];
        d2 vb[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) vb[b] = sx[j & 1][bc_slot[b]];

        // partial p = A v over this wave's columns.  (Retired columns carry v = w = 0, so skipping them is only an
        // optimisation, done per block of TB column groups.)  vb: lane t of every row of 16 lanes holds v[4 t + q].
        double par[2] = {0.0, 0.0}, pai[2] = {0.0, 0.0};
        static_for<0, NT / TB>([&](auto tbc) {
            constexpr int tb = decltype(tbc)::value * TB;
            if (NW * (tb + TB) - 1 > j) {  // uniform: any column of this block still active
                static_for<0, TB>([&](auto uc) {
                    constexpr int t = tb + decltype(uc)::value;
                    double a_re = TBK_AR(t), a_im = TBK_AI(t);
                    fmac_bc<t & 15>(par[t & 1], vb[t >> 4][0], a_re);
                    fmac_bc<t & 15>(pai[t & 1], vb[t >> 4][1], a_re);
                    fnmac_bc<t & 15>(par[t & 1], vb[t >> 4][1], a_im);
                    fmac_bc<t & 15>(pai[t & 1], vb[t >> 4][0], a_im);
                });
            }
        });
        if (lane < NR) sp[q][lane] = (d2){par[0] + par[1], pai[0] + pai[1]};
        wg_sync();  // B2: partial products of all four waves
        double pr = 0.0, pi = 0.0;
        {  // the four reads in flight together (inside a lane predicate hipcc waited for each before the next)
            d2 t[NW];
#pragma unroll
            for (int w = 0; w < NW; ++w) t[w] = sp[w][lane < NR ? lane : NR - 1];
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                pr += t[w][0];
                pi += t[w][1];
            }
            if (!(lane > j && lane < n)) pr = pi = 0.0;
        }
        // u = A v is in (pr, pi).  A is Hermitian, so rho = v^H u is real: one reduction instead of the
        // complex dot product p^H v of zhetd2 (p = tau u, p^H v = conj(tau) rho), and
        // w = p - (tau/2)(p^H v) v = tau u - (|tau|^2 rho / 2) v.
        const double rho = wave_sum(pr * vr + pi * vi);
        const double a2 = -0.5 * (tr * tr + ti * ti) * rho;
        const double wr = fma(a2, vr, pr * tr - pi * ti);
        const double wi = fma(a2, vi, pr * ti + pi * tr);
        if (lane < NR) sw[lane] = (d2){wr, wi};
        wave_lds_fence();
        d2 wb[NB];  // lane t of every row: w[NW (16 b + t) + q]  (this wave's own store, read back in order)
#pragma unroll
        for (int b = 0; b < NB; ++b) wb[b] = sw[bc_slot[b]];

        // A -= v w^H + w v^H on this wave's columns
        static_for<0, NT / TB>([&](auto tbc) {
            constexpr int tb = decltype(tbc)::value * TB;
            if (NW * (tb + TB) - 1 > j) {
                static_for<0, TB>([&](auto uc) {
                    constexpr int t = tb + decltype(uc)::value;
                    double r = TBK_AR(t), m = TBK_AI(t);
                    fnmac_bc<t & 15>(r, wb[t >> 4][0], vr);
                    fnmac_bc<t & 15>(m, wb[t >> 4][0], vi);
                    fnmac_bc<t & 15>(r, wb[t >> 4][1], vi);
                    fmac_bc<t & 15>(m, wb[t >> 4][1], vr);
                    fnmac_bc<t & 15>(r, vb[t >> 4][0], wr);
                    fnmac_bc<t & 15>(m, vb[t >> 4][0], wi);
                    fnmac_bc<t & 15>(r, vb[t >> 4][1], wi);
                    fmac_bc<t & 15>(m, vb[t >> 4][1], wr);
                    TBK_AR(t) = r;
                    TBK_AI(t) = m;
                });
            }
        });
        if (own_next) publish(jn);
    }
    if (split) {
        // Every wave has consumed its loads of H long ago (they are behind the first step's barriers), so the head of
        // the matrix' storage is free.  Element (i, c), i >= c, of the trailing block goes to the UPPER-triangle slot
        // (c, i) as its conjugate -- the matrix is Hermitian and both triangles are kept up to date here -- so that the
        // lanes of a column write consecutive addresses.
        const int s0 = n_steps;
        static_for<0, NT>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            const int c = NW * t + q;
            if (c >= s0 && c < n && lane >= c && lane < n)
                *reinterpret_cast<d2*>(Hm + ((size_t)(c - s0) * 32 + (lane - s0)) * 2) = (d2){TBK_AR(t), -TBK_AI(t)};
        });
    }
#undef TBK_AR
#undef TBK_AI
}

// ------------------------------------------------------------------------------------------------
// kernel 1c: 64 < n <= 128 -- EIGHT waves per matrix, the matrix in registers (round 3).  Wave (q, rh) holds columns
// c = 4 t + q of rows 64 rh + lane: 32 complex = 128 VGPRs per lane at 128 orbitals; one workgroup is one matrix.
// It runs the first n - 64 Householder steps and leaves the trailing 64 x 64 block at the head of the matrix' storage
// for the kernels above (tbk_launch_tridiag_tail64), like the streaming kernel of tbk_eig_stream.hip it replaces at
// these sizes -- there a step is a pass over memory between barriers (~17 k cycles), here ~5 k.
//
// What differs from kernel 1b, where every wave has all rows: sums over the rows span two waves, and a third barrier
// per step would cost more than redundant arithmetic.
//   * sigma (the norm of the new column below the sub-diagonal): the two waves that own the column publish the RAW
//     column and their partial sums; behind the step's first barrier every wave forms the reflector scalars and scales
//     its own rows (and the columns its broadcast registers stand for) itself;
//   * rho = v^H A v: every wave reduces its rows x its columns' share of it BEFORE the second barrier and publishes it
//     beside its partial products; behind the barrier everybody adds the eight numbers in the same order;
//   * w at the columns of a wave's broadcast registers comes from the partial products in LDS (any wave can add the four
//     partials of any row), not from a copy of w another wave would have to publish first.
// ------------------------------------------------------------------------------------------------
// the largest divisor of nt that is at most 12
constexpr int load_batch(int nt) {
    int best = 1;
    for (int d = 2; d <= 12; ++d)
        if (nt % d == 0) best = d;
    return best;
}

template <int NC, int NW>  // padded columns (a multiple of 8, <= 128); column residues = waves per row half (4; 2 up to 96 columns)
__global__ void __launch_bounds__(NW * 128, 2)
herm_tridiag8_kernel(double* H, int n, double* __restrict__ D, double* __restrict__ E, int n_steps, int64_t h_stride, int ldd, int off) {
    constexpr int NR = 128;      // rows: two waves of 64
    constexpr int NT = NC / NW;  // columns per lane
    constexpr int NB = (NT + 15) / 16;
    constexpr int TB = 2;
    static_assert(NC % (NW * TB) == 0 && NB <= 3 && (NW == 2 || NW == 4), "column layout");
    __shared__ d2 sx[2][NR];       // raw column of step j in sx[j & 1] (same turn-taking as kernel 1b)
    __shared__ double ssig[2][2];  // its partial norms, one per row half
    __shared__ d2 sp[NW][NR];      // partial products A v per column residue
    __shared__ double srho[2 * NW];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = wv & (NW - 1), rh = wv / NW;
    const int row = rh * 64 + lane;
    const size_t mat = blockIdx.x;
    double* Hm = H + mat * (size_t)h_stride;
    double* Dm = D + mat * (size_t)ldd + off;
    double* Em = E + mat * (size_t)ldd + off;

    // statically indexed only (the next column is caught on its way through the update, and a column that needs no
    // reflector takes the same path with v = w = 0): plain registers, no padding of the column count to 16
    double ar[NT], ai[NT];
    {
        constexpr int LB = load_batch(NT);  // loads in flight together
        const int li = min(row, n - 1);
        static_for<0, NT / LB>([&](auto t0c) {
            constexpr int t0 = decltype(t0c)::value * LB;
            d2 raw[LB];
#pragma unroll
            for (int u = 0; u < LB; ++u) {
                const int c = min(NW * (t0 + u) + q, n - 1);
                const int lo = min(li, c), hi = max(li, c);
                raw[u] = *reinterpret_cast<const d2*>(Hm + ((size_t)lo * n + hi) * 2);
            }
            static_for<0, LB>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                const int c = NW * (t0 + u) + q;
                const bool inside = row < n && c < n;
                ar[t0 + u] = inside ? raw[u][0] : 0.0;
                ai[t0 + u] = inside ? (c >= row ? raw[u][1] : -raw[u][1]) : 0.0;
            });
        });
    }

    // the two waves that hold column jc hand it out as it is, with their share of sum |x_i|^2 over the rows below the
    // sub-diagonal; d[jc] is the diagonal entry
    auto publish = [&](int jc, double xr, double xi) {
        if (row == jc) Dm[jc] = xr;
        const bool below = (row > jc + 1) && (row < n);
        const double part = wave_sum(below ? (xr * xr + xi * xi) : 0.0);
        sx[jc & 1][row] = (d2){xr, xi};
        if (lane == 0) ssig[jc & 1][rh] = part;
    };
    if (q == 0) publish(0, ar[0], ai[0]);
    int bc_col[NB];  // the columns this lane's broadcast registers stand for: NW * (16 b + lane % 16) + q
#pragma unroll
    for (int b = 0; b < NB; ++b) bc_col[b] = NW * (16 * b + (lane & 15)) + q;

    for (int j = 0; j < n_steps; ++j) {
        wg_sync();  // B1: sx, ssig of column j
        const int jn = j + 1;
        const d2 xme = sx[j & 1][row];
        const d2 al = sx[j & 1][jn];
        const double sigma = ssig[j & 1][0] + ssig[j & 1][1];
        d2 xb[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) xb[b] = sx[j & 1][min(bc_col[b], NR - 1)];
        const bool own_next = (jn & (NW - 1)) == q && jn != n_steps;  // (column n_steps is the next kernel's)
        const int tn = own_next ? jn / NW : -1;
        const double alr = al[0], ali = al[1];
        const bool ident = sigma == 0.0 && ali == 0.0;  // already reduced: H = I, this step changes nothing
        double root, rroot;
        fast_sqrt_rsqrt(ident ? 1.0 : alr * alr + ali * ali + sigma, root, rroot);
        const double beta = -copysign(root, alr);
        const double rbeta = -copysign(rroot, alr);
        const double qr = alr - beta, qi = ali;
        const double qn = fast_rcp(ident ? 1.0 : qr * qr + qi * qi);
        const double sr = ident ? 0.0 : qr * qn, si = ident ? 0.0 : -qi * qn;  // 1 / (alpha - beta)
        double tr = ident ? 0.0 : (beta - alr) * rbeta, ti = ident ? 0.0 : -ali * rbeta;
        if (NT * 4 > 176) {
            // tau is the same in every lane and lives across the partial products and the second barrier: with 192 matrix
            // registers per lane (<96, 2>) it moves to scalar registers (the last 8 bytes of scratch memory of this kernel)
            tr = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(tr)), __builtin_amdgcn_readfirstlane(__double2loint(tr)));
            ti = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(ti)), __builtin_amdgcn_readfirstlane(__double2loint(ti)));
        }
        const double one = ident ? 0.0 : 1.0;
        if (wv == 0 && lane == 0) Em[j] = ident ? alr : beta;
        double vr = 0.0, vi = 0.0;
        if (row > jn && row < n) {
            vr = xme[0] * sr - xme[1] * si;
            vi = xme[0] * si + xme[1] * sr;
        } else if (row == jn) {
            vr = one;
        }
        d2 vb[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int c = bc_col[b];
            vb[b] = (d2){0.0, 0.0};
            if (c > jn && c < n)
                vb[b] = (d2){xb[b][0] * sr - xb[b][1] * si, xb[b][0] * si + xb[b][1] * sr};
            else if (c == jn)
                vb[b] = (d2){one, 0.0};
        }

        // partial p = A v over this wave's rows and columns
        double par[2] = {0.0, 0.0}, pai[2] = {0.0, 0.0};
        static_for<0, NT / TB>([&](auto tbc) {
            constexpr int tb = decltype(tbc)::value * TB;
            if (NW * (tb + TB) - 1 > j) {
                static_for<0, TB>([&](auto uc) {
                    constexpr int t = tb + decltype(uc)::value;
                    fmac_bc<t & 15>(par[t & 1], vb[t >> 4][0], ar[t]);
                    fmac_bc<t & 15>(pai[t & 1], vb[t >> 4][1], ar[t]);
                    fnmac_bc<t & 15>(par[t & 1], vb[t >> 4][1], ai[t]);
                    fmac_bc<t & 15>(pai[t & 1], vb[t >> 4][0], ai[t]);
                });
            }
        });
        {
            const double p_r = par[0] + par[1], p_i = pai[0] + pai[1];
            sp[q][row] = (d2){p_r, p_i};
            const double share = wave_sum((row > j && row < n) ? (p_r * vr + p_i * vi) : 0.0);
            if (lane == 0) srho[wv] = share;
        }
        wg_sync();  // B2: partial products and shares of rho of all eight waves
        double ur = 0.0, ui = 0.0, rho = 0.0;
        double ubr[NB], ubi[NB];
        {
            // (all reads of a batch in flight together, then the sums.  With 48 columns per lane -- <96, 2>: 192 matrix
            // registers -- the reads go out in two batches: in one, their 36 landing registers pushed 13 dwords of the
            // matrix into scratch memory)
            constexpr int B0 = (NT * 4 > 176 && NB > 1) ? 1 : NB;  // broadcast registers summed in the first batch
            d2 t[NW], tb2[NB][NW];
            double rs[2 * NW];
#pragma unroll
            for (int w2 = 0; w2 < NW; ++w2) t[w2] = sp[w2][row];
#pragma unroll
            for (int b = 0; b < B0; ++b)
#pragma unroll
                for (int w2 = 0; w2 < NW; ++w2) tb2[b][w2] = sp[w2][min(bc_col[b], NR - 1)];
#pragma unroll
            for (int w2 = 0; w2 < 2 * NW; ++w2) rs[w2] = srho[w2];
#pragma unroll
            for (int w2 = 0; w2 < NW; ++w2) {
                ur += t[w2][0];
                ui += t[w2][1];
            }
#pragma unroll
            for (int w2 = 0; w2 < 2 * NW; ++w2) rho += rs[w2];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (b >= B0) {  // a later batch of one: requested behind the sums of the batch before it
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int w2 = 0; w2 < NW; ++w2) tb2[b][w2] = sp[w2][min(bc_col[b], NR - 1)];
                }
                ubr[b] = ubi[b] = 0.0;
#pragma unroll
                for (int w2 = 0; w2 < NW; ++w2) {
                    ubr[b] += tb2[b][w2][0];
                    ubi[b] += tb2[b][w2][1];
                }
                if (!(bc_col[b] > j && bc_col[b] < n)) ubr[b] = ubi[b] = 0.0;
            }
            if (!(row > j && row < n)) ur = ui = 0.0;
        }
        // w = tau u - (|tau|^2 rho / 2) v  (kernel 1b)
        const double a2 = -0.5 * (tr * tr + ti * ti) * rho;
        const double wr = fma(a2, vr, ur * tr - ui * ti);
        const double wi = fma(a2, vi, ur * ti + ui * tr);
        d2 wb[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b)
            wb[b] = (d2){fma(a2, vb[b][0], ubr[b] * tr - ubi[b] * ti), fma(a2, vb[b][1], ubr[b] * ti + ubi[b] * tr)};

        // A -= v w^H + w v^H on this wave's rows and columns; the owners of column j + 1 catch it on the way
        double nxr = 0.0, nxi = 0.0;
        static_for<0, NT / TB>([&](auto tbc) {
            constexpr int tb = decltype(tbc)::value * TB;
            if (NW * (tb + TB) - 1 > j) {
                static_for<0, TB>([&](auto uc) {
                    constexpr int t = tb + decltype(uc)::value;
                    double r = ar[t], m = ai[t];
                    fnmac_bc<t & 15>(r, wb[t >> 4][0], vr);
                    fnmac_bc<t & 15>(m, wb[t >> 4][0], vi);
                    fnmac_bc<t & 15>(r, wb[t >> 4][1], vi);
                    fmac_bc<t & 15>(m, wb[t >> 4][1], vr);
                    fnmac_bc<t & 15>(r, vb[t >> 4][0], wr);
                    fnmac_bc<t & 15>(m, vb[t >> 4][0], wi);
                    fnmac_bc<t & 15>(r, vb[t >> 4][1], wi);
                    fmac_bc<t & 15>(m, vb[t >> 4][1], wr);
                    ar[t] = r;
                    ai[t] = m;
                    if (t == tn) {  // uniform -- and kept a branch (the asm statement): as selects this was 4 VALU issues
                        nxr = r;    // per column and step, a sixth of the loop
                        nxi = m;
                        asm volatile("" : "+v"(nxr), "+v"(nxi));
                    }
                });
            }
        });
        if (own_next) publish(jn, nxr, nxi);
    }
    // hand-over: entry (i, c), i >= c, of the trailing block goes to the upper-triangle slot (c, i) of a 64 x 64 row-major
    // matrix at the head of this matrix' storage, conjugated (both triangles are up to date here; the loads of H were
    // consumed before the first barrier)
    {
        const int s0 = n_steps;
        const int ldt = n - n_steps;
        static_for<0, NT>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            const int c = NW * t + q;
            if (c >= s0 && c < n && row >= c && row < n)
                *reinterpret_cast<d2*>(Hm + ((size_t)(c - s0) * ldt + (row - s0)) * 2) = (d2){ar[t], -ai[t]};
        });
    }
}

// ------------------------------------------------------------------------------------------------
// kernel 2: implicit QL with Wilkinson shift on 64 tridiagonals per wave (one per lane)
// ------------------------------------------------------------------------------------------------

template <int QL_LD>  // matrices per block = row length of the [index][lane] LDS arrays
__global__ void __launch_bounds__(64)
tridiag_ql_kernel(const double* __restrict__ D, const double* __restrict__ E, int n, int64_t nk,
                  double* __restrict__ out, int* __restrict__ fail_count) {
    extern __shared__ __attribute__((aligned(16))) double ql_smem[];  // 2 * n * 64 doubles (<= 64 KiB)
    double* sd = ql_smem;
    double* se = ql_smem + (size_t)n * QL_LD;
    const int lane = threadIdx.x;
    const int64_t m0 = (int64_t)blockIdx.x * QL_LD;
    const int nmat = (int)min((int64_t)QL_LD, nk - m0);

    // coalesced fill: the 64 x n block of (d, e) is contiguous in memory
    for (int idx = lane; idx < nmat * n; idx += 64) {
        const int mm = idx / n, c = idx % n;
        sd[c * QL_LD + mm] = D[m0 * n + idx];
        se[c * QL_LD + mm] = E[m0 * n + idx];
    }
    __syncthreads();

    if (lane < nmat) {
        double* d = sd + lane;
        double* e = se + lane;
        bool failed = false;
        // NaN / Inf anywhere in H(k) reaches (d, e): hand out NaN eigenvalues (the caller raises the ValueError of
        // scipy's check_finite) instead of iterating to the non-convergence limit
        bool finite = true;
        for (int i = 0; i < n; ++i) finite = finite && isfinite(d[i * QL_LD]) && isfinite(e[i * QL_LD]);
        if (!finite) {
            for (int i = 0; i < n; ++i) d[i * QL_LD] = __builtin_nan("");
            atomicAdd(fail_count + 1, 1);  // flags[1]: non-finite input (tbk_eigenval_check -> TBK_ERR_NOT_FINITE)
        }
        // Every lane runs its own (l, m, iteration) state in ONE flat loop: with `for l { while ... }` the lanes met
        // again after every eigenvalue, so a wave paid max-over-lanes sweeps for each l (233 sweeps for 64 random
        // 64 x 64 matrices against 145 +- 3 per matrix).  The negligible off-diagonal that ends the next sweep (m) is
        // found DURING the sweep, on the values it writes -- the separate scan over (d, e) after every sweep was a
        // chain of dependent LDS reads, a fifth of the kernel; a scan is only left where a block ends (l reaches m).
        const double EPS = 2.220446049250313e-16;
        int l = 0, m = 0, iter = 0;
        bool done = !finite;
        // first l that has not converged, and the end m of its unreduced block (scan from l; d, e as stored)
        auto settle = [&]() {
            while (l < n - 1) {
                m = l;
                double ad = fabs(d[l * QL_LD]);
                for (; m < n - 1; ++m) {
                    const double ad1 = fabs(d[(m + 1) * QL_LD]);
                    if (fabs(e[m * QL_LD]) <= EPS * (ad + ad1)) break;
                    ad = ad1;
                }
                if (m != l) break;
                ++l;
            }
            if (l >= n - 1) done = true;
            iter = 0;
        };
        if (!done) settle();
        while (!done) {
            if (++iter > 60) {
                failed = true;
                break;
            }
            const double el = e[l * QL_LD];
            const double dl = d[l * QL_LD];
            double g = (d[(l + 1) * QL_LD] - dl) / (2.0 * el);
            double r = sqrt(fma(g, g, 1.0));
            g = d[m * QL_LD] - dl + el / (g + copysign(r, g));
            double s = 1.0, c = 1.0, p = 0.0;
            // The sweep is one long dependent chain per lane (this kernel is latency-bound), so it is
            // kept short: d[i+1] stays in a register from the previous step, (d[i-1], e[i-1]) are
            // fetched from LDS one step ahead, and the rotation uses one rsqrt instead of sqrt + divide
            // (|T| = O(1..10): no overflow guard needed).
            double d_up = d[m * QL_LD];                                      // d[i + 1]
            double d_i = d[(m - 1) * QL_LD], e_i = e[(m - 1) * QL_LD];       // step i = m - 1
            bool underflow = false;
            int m_next = m;        // lowest j in (l, m] whose new e[j] is negligible (e[m] becomes 0)
            double ad_prev = 0.0;  // |d[i + 2]| as written by the previous step
            for (int i = m - 1; i >= l; --i) {
                const int ip = (i > l) ? i - 1 : l;
                const double d_nx = d[ip * QL_LD], e_nx = e[ip * QL_LD];     // for step i - 1
                const double f = s * e_i;
                const double b = c * e_i;
                const double h2 = fma(f, f, g * g);
                if (h2 == 0.0) {
                    e[(i + 1) * QL_LD] = 0.0;
                    d[(i + 1) * QL_LD] = d_up - p;
                    e[m * QL_LD] = 0.0;
                    underflow = true;
                    break;
                }
                const double rinv = rsqrt(h2);
                const double e_new = h2 * rinv;
                e[(i + 1) * QL_LD] = e_new;
                s = f * rinv;
                c = g * rinv;
                const double gg = d_up - p;
                r = (d_i - gg) * s + 2.0 * c * b;
                p = s * r;
                const double d_new = gg + p;
                d[(i + 1) * QL_LD] = d_new;
                g = c * r - b;
                const double ad_cur = fabs(d_new);
                if (i + 1 < m && e_new <= EPS * (ad_cur + ad_prev)) m_next = i + 1;
                ad_prev = ad_cur;
                d_up = d_i;
                d_i = d_nx;
                e_i = e_nx;
            }
            if (underflow) {  // (never seen on H(k) data) same l again, block end from the stored values
                const int it = iter;
                settle();
                iter = it;
                continue;
            }
            const double dl_new = d_up - p;
            d[l * QL_LD] = dl_new;
            e[l * QL_LD] = g;
            e[m * QL_LD] = 0.0;
            m = m_next;
            if (fabs(g) <= EPS * (fabs(dl_new) + ad_prev)) {  // e[l] negligible: d[l] is an eigenvalue
                ++l;
                iter = 0;
                if (l >= m) settle();  // the block ended here (or e[l + 1] went as well): look for the next one
            }
        }
        if (failed) atomicAdd(fail_count, 1);
        // ascending order, like eigvalsh: insertion sort of this lane's column
        for (int a = 1; a < n && finite; ++a) {
            const double key = d[a * QL_LD];
            int b = a - 1;
            while (b >= 0 && d[b * QL_LD] > key) {
                d[(b + 1) * QL_LD] = d[b * QL_LD];
                --b;
            }
            d[(b + 1) * QL_LD] = key;
        }
    }
    __syncthreads();
    for (int idx = lane; idx < nmat * n; idx += 64) {
        const int mm = idx / n, c = idx % n;
        out[m0 * n + idx] = sd[c * QL_LD + mm];
    }
}

}  // namespace

bool tbk_eig_small_supported(int n) { return n >= 1 && n <= 64; }

// d_de holds the tridiagonal of every matrix: d[nk][n] followed by e[nk][n]
// The register-resident reduction of nk n x n matrices (32 < n <= 64) that sit h_stride doubles apart in d_H; (d, e) of
// matrix m to d_D / d_E + m * ldd + off.  Two launches: the first n - 32 steps (four or two waves per matrix), then the
// trailing 32 x 32 blocks two per wave.
static int launch_tridiag_33_64(hipStream_t s, double* d_H, int n, int64_t nk, double* d_D, double* d_Eo, int64_t h_stride, int ldd, int off,
                                int64_t call_nk) {
    const dim3 grid((unsigned)nk), block(64);
    // columns per lane = padded size / 4: a 40-orbital matrix in the 64-row instantiation does 16 column
    // updates per lane and step where 10 are enough (n = 48: 8.0 -> 7.2 ms per 65536 matrices)
    // Round 3: the four-wave kernel only does the first n - 32 steps; the trailing 32 x 32 block goes through the
    // head of the matrix' own storage to the packed kernel (two matrices per wave).  TBK_SMALL_SPLIT=0: one kernel.
    // Calls of a few matrices (all of them resident at once: what counts is one matrix' latency, not issue slots) keep the
    // whole reduction in ONE launch of the four-wave kernel: a single 64 x 64 matrix 99 -> 78 us.  By the size of the CALL,
    // not of this chunk: TBK_OPT_K_CHUNK must not change results, and the forms differ in the last bit.
    static const bool split_env = !(tbk_exp_env("TBK_SMALL_SPLIT") && atoi(tbk_exp_env("TBK_SMALL_SPLIT")) == 0);
    const bool split_on = split_env && std::max(call_nk, nk) > 512;
    const int n_steps = split_on ? n - 32 : n - 1;
    // TWO waves per matrix at every size when the kernel only does the first n - 32 steps (round 3; TBK_SMALL_NW2=0:
    // four): those are the steps with the most FMAs per reduction / barrier / scalar chain, and halving the copies of
    // that overhead buys more than the lower occupancy costs (178 registers at 64 rows: two waves per SIMD) -- cfg2
    // 951 -> 963 k, cfg4 8.84 -> 9.26 M k-points/s.  For the WHOLE reduction it was a wash (4.07 vs 4.14 ms, round 2).
    static const bool two_env = !(tbk_exp_env("TBK_SMALL_NW2") && atoi(tbk_exp_env("TBK_SMALL_NW2")) == 0);
    const bool two_waves = split_on && two_env;
#define TBK_T4(NRV, NWV) \
    hipLaunchKernelGGL((herm_tridiag4_kernel<NRV, NWV>), grid, dim3(NWV * 64), 0, s, d_H, n, d_D, d_Eo, n_steps, h_stride, ldd, off)
    if (n <= 40 && split_on)
        TBK_T4(40, 2);
    else if (n <= 48 && two_waves)
        TBK_T4(48, 2);
    else if (n <= 48)
        TBK_T4(48, 4);
    else if (n <= 56 && two_waves)
        TBK_T4(56, 2);
    else if (n <= 56)
        TBK_T4(56, 4);
    else if (two_waves)
        TBK_T4(64, 2);
    else
        TBK_T4(64, 4);
#undef TBK_T4
    TBK_HIP(hipGetLastError());
    if (split_on) {
        hipLaunchKernelGGL(herm_tridiag_packed_kernel<32>, dim3((unsigned)((nk + 1) / 2)), block, 0, s, d_H, 32, nk, d_D, d_Eo,
                           h_stride, ldd, off + n - 32);
        TBK_HIP(hipGetLastError());
    }
    return TBK_OK;
}

// The tail of the streaming reduction (tbk_eig_stream.hip): the trailing 64 x 64 blocks it left at the head of the
// n_full x n_full matrices, (d, e)[n_full - 64 ...] of every matrix.  No stage timer: the caller holds one.
int tbk_launch_tridiag_tail64(hipStream_t s, double* d_H, int64_t nk, double* d_D, double* d_E, int n_full, int64_t call_nk) {
    return launch_tridiag_33_64(s, d_H, 64, nk, d_D, d_E, (int64_t)n_full * n_full * 2, n_full, n_full - 64, call_nk);
}

// The first n - 64 Householder steps of nk n x n matrices, 64 < n <= 128, in the eight-wave register kernel; the trailing
// 64 x 64 blocks are left at the head of every matrix' storage for tbk_launch_tridiag_tail64.  (d, e)[0 .. n - 65] of
// matrix m to d_D / d_E + m * ldd + off.
bool tbk_eig_reg128_supported(int n) {
    static const bool on = !(getenv("TBK_REG128") && atoi(getenv("TBK_REG128")) == 0);  // measurements: 0 = streaming kernel
    return on && n > 64 && n <= 128;
}
int tbk_launch_tridiag_reg128(hipStream_t s, double* d_H, int n, int64_t nk, double* d_D, double* d_E, int64_t h_stride, int ldd, int off) {
    const dim3 grid((unsigned)nk);
#define TBK_T8(NCV, NWV) \
    hipLaunchKernelGGL((herm_tridiag8_kernel<NCV, NWV>), grid, dim3(NWV * 128), 0, s, d_H, n, d_D, d_E, n - 64, h_stride, ldd, off)
    // up to 96 orbitals FOUR waves per matrix (two column residues: 48 complex per lane, ~250 registers): two matrices
    // per CU, so that one's barriers and scalar chains run under the other's FMAs (TBK_REG128_NW2=0: eight waves)
    static const bool nw2 = !(getenv("TBK_REG128_NW2") && atoi(getenv("TBK_REG128_NW2")) == 0);
    if (n <= 80 && nw2)
        TBK_T8(80, 2);
    else if (n <= 80)
        TBK_T8(80, 4);
    else if (n <= 96 && nw2)
        TBK_T8(96, 2);
    else if (n <= 96)
        TBK_T8(96, 4);
    else if (n <= 112)
        TBK_T8(112, 4);
    else
        TBK_T8(128, 4);
#undef TBK_T8
    TBK_HIP(hipGetLastError());
    return TBK_OK;
}

// d_de holds the tridiagonal of every matrix: d[nk][n] followed by e[nk][n]
int tbk_launch_tridiag(tbk_model* m, hipStream_t s, double* d_H, int64_t nk, double* d_de) {
    const int n = m->n_orb;
    if (nk == 0) return TBK_OK;
    double* d_D = d_de;
    double* d_Eo = d_de + (size_t)nk * n;
    StageTimer t(m, TBK_T_EIG, s);
    const dim3 block(64);
    const int64_t packed = (int64_t)n * n * 2;
    if (n > 32) return launch_tridiag_33_64(s, d_H, n, nk, d_D, d_Eo, packed, n, 0, m->call_nk);
    if (n <= 8)
        hipLaunchKernelGGL(herm_tridiag_packed_kernel<8>, dim3((unsigned)((nk + 7) / 8)), block, 0, s, d_H, n, nk, d_D, d_Eo, packed, n, 0);
    else if (n <= 16)
        hipLaunchKernelGGL(herm_tridiag_packed_kernel<16>, dim3((unsigned)((nk + 3) / 4)), block, 0, s, d_H, n, nk, d_D, d_Eo, packed, n, 0);
    else
        hipLaunchKernelGGL(herm_tridiag_packed_kernel<32>, dim3((unsigned)((nk + 1) / 2)), block, 0, s, d_H, n, nk, d_D, d_Eo, packed, n, 0);
    TBK_HIP(hipGetLastError());
    return TBK_OK;
}

int tbk_launch_ql(tbk_model* m, hipStream_t s, const double* d_de, int64_t nk, double* d_E, bool beside_ql) {
    const int n = m->n_orb;
    if (nk == 0) return TBK_OK;
    StageTimer t(m, TBK_T_QL, s);
    // matrices per block: as many as fit 64 KiB of LDS (2 n doubles per matrix), at most one per lane
#define TBK_QL_LAUNCH(MPB)                                                                                  \
    hipLaunchKernelGGL(tridiag_ql_kernel<MPB>, dim3((unsigned)((nk + MPB - 1) / MPB)), dim3(64),             \
                       (size_t)2 * n * MPB * sizeof(double), s, d_de, d_de + (size_t)nk * n, n, nk, d_E,     \
                       m->ws_flag.as<int>())
    if (n <= 64 && beside_ql && nk <= 8192) {
        TBK_QL_LAUNCH(32);
    } else if (n <= 64) {
        TBK_QL_LAUNCH(64);
    } else if (n <= 128) {
        TBK_QL_LAUNCH(32);
    } else if (n <= 256) {
        TBK_QL_LAUNCH(16);
    } else if (n <= 512) {
        TBK_QL_LAUNCH(8);
    } else {
        tbk_set_error("tridiagonal QL kernel handles n <= 512 (n = %d)", n);
        return TBK_ERR_ARGUMENT;
    }
#undef TBK_QL_LAUNCH
    TBK_HIP(hipGetLastError());
    return TBK_OK;
}
