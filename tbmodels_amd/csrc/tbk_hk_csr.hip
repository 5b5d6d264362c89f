// tbk_hk_csr.hip -- H(k) for sparse hoppings (model._sparse == True).
//
// The reference has no sparse arithmetic: it densifies every CSR block on every call
// (`_array_cast`, /root/reference/src/tbmodels/_tb_model.py:1326-1331 with
// src/tbmodels/_sparse_matrix.py:20-21) and then runs the dense loop of :1111-1122.  Results must
// equal the dense path (tests/test_sparse_dense.py:15-25 of the reference).
//
// Here the sparse hoppings are transposed ONCE at staging into "per matrix element, which lattice
// vectors touch it":  for packed element e = (i <= j)
//
//     H[k][e] = sum_{rec in list(e)}  p_{r(rec)}(k) * v          direct      (entry (i, j) of hop[r])
//                                   + conj(p_r(k)) * conj(v)     transposed  (entry (j, i) of hop[r])
//                                   + 2 Re(p_r(k) v)             diagonal    (i == j)
//
// so every output element is produced by exactly one thread: no atomics, no zero-fill pass, and
// the 16 B/element output stream (the binding resource: N^2 * 16 B per k-point against ~20 short
// records per element) is written once, coalesced along j.  One thread walks its record list for
// KT = 8 consecutive k-points, so each record (20 B) is amortised over 8 outputs and the phase
// gathers are 64 B contiguous.
//
// Roofline: HBM write bound (8 * N(N+1) B per k in TRI mode, 16 N^2 B in FULL mode).

#include <algorithm>
#include <cstdlib>
#include <utility>
#include <vector>

#include "tbk_internal.h"

namespace {

typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int KT = 8;

template <int MODE, int CONV>
__global__ void __launch_bounds__(256)
hk_csr_kernel(const double* __restrict__ A, const int64_t* __restrict__ cptr,
              const int32_t* __restrict__ rec_r, const double* __restrict__ rec_v,
              const int32_t* __restrict__ colmap, const double* __restrict__ kpts,
              const double* __restrict__ pos, int dim, int ncol, int n_orb, int64_t nk,
              int64_t nk_pad, double* __restrict__ H) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t k0 = (int64_t)blockIdx.y * KT;
    if (e >= ncol) return;
    double re[KT], im[KT];
#pragma unroll
    for (int q = 0; q < KT; ++q) re[q] = im[q] = 0.0;

    const int64_t beg = cptr[e], end = cptr[e + 1];
    for (int64_t t = beg; t < end; ++t) {
        const int32_t packed = rec_r[t];
        const int kind = packed >> 28;
        const int64_t r = packed & 0x0fffffff;
        const double vr = rec_v[2 * t], vi = rec_v[2 * t + 1];
        const double* pc = A + (2 * r) * nk_pad + k0;
        const double* ps = pc + nk_pad;
        double c[KT], s[KT];
#pragma unroll
        for (int q = 0; q < KT; q += 2) {
            const d2 cc = *reinterpret_cast<const d2*>(pc + q);
            const d2 ss = *reinterpret_cast<const d2*>(ps + q);
            c[q] = cc[0];
            c[q + 1] = cc[1];
            s[q] = ss[0];
            s[q + 1] = ss[1];
        }
        // direct: (c + i s)(vr + i vi); transposed: conj of that; diagonal: twice the real part
        const double sign_im = (kind == 1) ? -1.0 : ((kind == 2) ? 0.0 : 1.0);
        const double scale_re = (kind == 2) ? 2.0 : 1.0;
#pragma unroll
        for (int q = 0; q < KT; ++q) {
            re[q] += scale_re * (c[q] * vr - s[q] * vi);
            im[q] += sign_im * (c[q] * vi + s[q] * vr);
        }
    }

    const int32_t ij = colmap[e];
    const int oi = ij >> 16, oj = ij & 0xffff;
    const size_t nn = (size_t)n_orb * n_orb;
#pragma unroll
    for (int q = 0; q < KT; ++q) {
        const int64_t kq = k0 + q;
        if (kq >= nk) break;
        double vr = re[q], vi = im[q];
        if (CONV == 1 && oi != oj) {  // (diagonal: conj(e_i) e_i = 1, Im stays exactly 0)
            const d2 ei = *reinterpret_cast<const d2*>(pos + ((size_t)kq * n_orb + oi) * 2);  // e[k][p] table
            const d2 ej = *reinterpret_cast<const d2*>(pos + ((size_t)kq * n_orb + oj) * 2);
            const double cs = ei[0] * ej[0] + ei[1] * ej[1], sn = ei[0] * ej[1] - ei[1] * ej[0];
            const double t = vr * cs - vi * sn;
            vi = vr * sn + vi * cs;
            vr = t;
        }
        double* hk = H + (size_t)kq * nn * 2;
        *reinterpret_cast<d2*>(hk + ((size_t)oi * n_orb + oj) * 2) = (d2){vr, vi};
        if (MODE == HK_FULL && oi != oj)
            *reinterpret_cast<d2*>(hk + ((size_t)oj * n_orb + oi) * 2) = (d2){vr, -vi};
    }
}


// ------------------------------------------------------------------------------------------------
// LDS variant (used whenever the phase tile fits): a workgroup takes KT consecutive k-points, copies
// their phases for ALL lattice vectors into LDS once ((cos, sin) pairs, [r][q] with a 16-byte pad per r so
// that lanes gathering different r spread over the banks) and then sweeps every packed element, 1024 at a
// time.  The phase gathers -- 16 B per (record, k-point), the dominant traffic: ~20 records per element --
// come out of LDS instead of L2; records (20 B each, L2 resident) are read once per k tile.
// ------------------------------------------------------------------------------------------------
constexpr int LDS_THREADS = 1024;  // 16 waves share one phase tile: the record walk is latency-bound

// SCHED: the records come in the conflict-free walk order of tbk_csr_schedule -- per wave round (64 consecutive packed
// elements) a common number of steps, srec[step][lane] = the lane's record of that step or none -- instead of every lane
// following its own list.  The phase reads of a step then hit 16 different 16-byte slots of the LDS row in each of the
// four 16-lane groups a ds_read_b128 is served in: one LDS cycle per group instead of ~2.2 with random rows (54 % of the
// LDS cycles were bank conflicts, and the LDS pipe was this kernel's bound).
template <int MODE, int CONV, int KT, bool SCHED>
__global__ void __launch_bounds__(LDS_THREADS)
hk_csr_lds_kernel(const double* __restrict__ A, const int64_t* __restrict__ cptr,
                  const int32_t* __restrict__ rec_r, const double* __restrict__ rec_v,
                  const int32_t* __restrict__ colmap, const double* __restrict__ kpts,
                  const double* __restrict__ pos, int dim, int ncol, int n_orb, int64_t n_r, int64_t nk,
                  int64_t nk_pad, double* __restrict__ H, int tiles_per_xcd, int slice_elems) {
    extern __shared__ __attribute__((aligned(16))) double sph[];  // [n_r][KT * 2 + 2]
    constexpr int LD = KT * 2 + 2;
    // Workgroup b runs on XCD b % 8 (round-robin dispatch): XCD x owns the k tiles [x T, (x + 1) T) and walks them SLICE by
    // slice of `slice_elems` packed elements -- its ~64 resident workgroups then read the same ~1.4 MB of records, which
    // stay in that XCD's 4 MB L2.  With every workgroup sweeping ALL elements for its tile the 22 MB record stream was
    // re-read from beyond L2 by each of them (L2 hit rate 42 %, waves waiting 70 % of their time).
    // (tiles_per_xcd == 0: a handful of k tiles -- one-k calls: fewer tiles than XCDs would leave the others idle; workgroup b
    // takes tile b % n_tiles, slice b / n_tiles, and the slices are single rounds so that the element sweep spreads over the chip)
    const int n_tiles_flat = (int)((nk + KT - 1) / KT);
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int64_t tile = tiles_per_xcd > 0 ? (int64_t)xcd * tiles_per_xcd + seq % tiles_per_xcd : (int64_t)(blockIdx.x % n_tiles_flat);
    const int slice = tiles_per_xcd > 0 ? seq / tiles_per_xcd : (int)(blockIdx.x / n_tiles_flat);
    const int64_t k0 = tile * KT;
    if (k0 >= nk) return;
    for (int64_t idx = threadIdx.x; idx < n_r * KT; idx += LDS_THREADS) {
        const int64_t r = idx / KT;
        const int q = (int)(idx % KT);
        sph[r * LD + 2 * q] = A[(2 * r) * nk_pad + k0 + q];
        sph[r * LD + 2 * q + 1] = A[(2 * r + 1) * nk_pad + k0 + q];
    }
    __syncthreads();
    const size_t nn = (size_t)n_orb * n_orb;
    const int lane = threadIdx.x & 63;
    const int ncol_round = min((ncol + 63) & ~63, (slice + 1) * slice_elems);
    for (int e = slice * slice_elems + threadIdx.x; e < ncol_round; e += LDS_THREADS) {
        if (!SCHED && e >= ncol) break;
        double re[KT], im[KT];
#pragma unroll
        for (int q = 0; q < KT; ++q) re[q] = im[q] = 0.0;
        auto accumulate = [&](int32_t packed, d2 val) {
            const double vr = val[0], vi = val[1];
            double ar, ai, br, bi;
            const double* ph;
            if (SCHED) {
                // scheduled records: the byte offset of the phase row in the low 30 bits, the sign of the imaginary part
                // (+1 direct, -1 transposed, 0 diagonal) as a two-bit signed code above it, diagonal values doubled at
                // staging -- 4 VALU instructions of decoding per record instead of 12 (the kernel is VALU-bound once
                // the LDS conflicts and the record misses are gone: 78 % busy)
                const double sign_im = (double)(packed >> 30);
                ar = vr;
                ai = vi;
                br = sign_im * vi;
                bi = sign_im * vr;
                ph = reinterpret_cast<const double*>(reinterpret_cast<const char*>(sph) + (packed & 0x3fffffff));
            } else {
                const int kind = packed >> 28;
                const int64_t r = packed & 0x0fffffff;
                const double sign_im = (kind == 1) ? -1.0 : ((kind == 2) ? 0.0 : 1.0);
                const double scale_re = (kind == 2) ? 2.0 : 1.0;
                ar = scale_re * vr;
                ai = scale_re * vi;
                br = sign_im * vi;
                bi = sign_im * vr;
                ph = sph + r * LD;
            }
#pragma unroll
            for (int q = 0; q < KT; ++q) {
                const d2 cs = *reinterpret_cast<const d2*>(ph + 2 * q);
                re[q] = fma(cs[0], ar, re[q]);
                re[q] = fma(-cs[1], ai, re[q]);
                im[q] = fma(cs[0], br, im[q]);
                im[q] = fma(cs[1], bi, im[q]);
            }
        };
        if (SCHED) {
            // cptr = first step of every wave round; rec_r / rec_v = [step][lane] (-1: no record for the lane in the step):
            // coalesced, the next step's records in flight.  (Storing only the active lanes of a step -- a 64-bit mask and
            // a record offset per step, the lane's index by mbcnt -- was measured: 25.2 instead of 22.4 ms per 50 000
            // k-points; the extra dependent loads cost more than the 38 % of padding bytes they save.)
            const int wr = __builtin_amdgcn_readfirstlane(e >> 6);
            const int64_t beg = cptr[wr], end = cptr[wr + 1];
            constexpr int32_t NONE = (int32_t)0x80000000;  // code -2: no record for this lane in this step
            // (two record slots in turn with scalar step pointers -- 9 instead of 13 non-FMA instructions per step -- were
            // measured: 16.9 instead of 15.3 ms per 50 000 k-points; the simple form below schedules its loads better)
            int32_t packed_n = NONE;
            d2 val_n = {0.0, 0.0};
            if (beg < end) {
                packed_n = rec_r[beg * 64 + lane];
                val_n = *reinterpret_cast<const d2*>(rec_v + 2 * (beg * 64 + lane));
            }
            for (int64_t t = beg; t < end; ++t) {
                const int32_t packed = packed_n;
                const d2 val = val_n;
                if (t + 1 < end) {
                    packed_n = rec_r[(t + 1) * 64 + lane];
                    val_n = *reinterpret_cast<const d2*>(rec_v + 2 * ((t + 1) * 64 + lane));
                }
                if (packed != NONE) accumulate(packed, val);
            }
            if (e >= ncol) continue;
        } else {
            const int64_t beg = cptr[e], end = cptr[e + 1];
            // the record walk is a chain of dependent L2 loads: keep the next record in flight
            int32_t packed_n = 0;
            d2 val_n = {0.0, 0.0};
            if (beg < end) {
                packed_n = rec_r[beg];
                val_n = *reinterpret_cast<const d2*>(rec_v + 2 * beg);
            }
            for (int64_t t = beg; t < end; ++t) {
                const int32_t packed = packed_n;
                const d2 val = val_n;
                if (t + 1 < end) {
                    packed_n = rec_r[t + 1];
                    val_n = *reinterpret_cast<const d2*>(rec_v + 2 * (t + 1));
                }
                accumulate(packed, val);
            }
        }
        const int32_t ij = colmap[e];
        const int oi = ij >> 16, oj = ij & 0xffff;
#pragma unroll
        for (int q = 0; q < KT; ++q) {
            const int64_t kq = k0 + q;
            if (kq >= nk) break;
            double vr = re[q], vi = im[q];
            if (CONV == 1 && oi != oj) {  // (diagonal: conj(e_i) e_i = 1, Im stays exactly 0)
                const d2 ei = *reinterpret_cast<const d2*>(pos + ((size_t)kq * n_orb + oi) * 2);  // e[k][p] table
                const d2 ej = *reinterpret_cast<const d2*>(pos + ((size_t)kq * n_orb + oj) * 2);
                const double cs = ei[0] * ej[0] + ei[1] * ej[1], sn = ei[0] * ej[1] - ei[1] * ej[0];
                const double t = vr * cs - vi * sn;
                vi = vr * sn + vi * cs;
                vr = t;
            }
            double* hk = H + (size_t)kq * nn * 2;
            *reinterpret_cast<d2*>(hk + ((size_t)oi * n_orb + oj) * 2) = (d2){vr, vi};
            if (MODE == HK_FULL && oi != oj)
                *reinterpret_cast<d2*>(hk + ((size_t)oj * n_orb + oi) * 2) = (d2){vr, -vi};
        }
    }
}

template <int MODE, int CONV, int KT>
hipError_t launch_lds(tbk_model* m, const double* d_A, int64_t nk, int64_t nk_pad, const double* d_k,
                      const double* d_pos, double* d_H) {
    const size_t lds = (size_t)std::max<int64_t>(m->n_r, 1) * (KT * 2 + 2) * sizeof(double);
    static const bool sched_on = !(tbk_exp_env("TBK_CSR_SCHED") && atoi(tbk_exp_env("TBK_CSR_SCHED")) == 0);  // 0: measurements
    const bool sched = sched_on && m->sched_kt == KT && m->d_sptr != nullptr;
    static std::atomic<bool> raised[2][TBK_MAX_DEVICES] = {};
    // slices of 4 x 1024 packed elements (measured: 1 -> 18.5, 2 -> 17.0, 3 -> 16.6, 4 -> 16.1, 6 -> 18.3, 8 -> 19.8 ms per 50 000 k-points at cfg3) (TBK_CSR_SLICE_ROUNDS: measurements); one slice = the whole triangle for small models
    static const int slice_rounds = tbk_exp_env("TBK_CSR_SLICE_ROUNDS") ? std::max(1, atoi(tbk_exp_env("TBK_CSR_SLICE_ROUNDS"))) : 4;
    const int64_t n_tiles = (nk + KT - 1) / KT;
    const bool flat = n_tiles < 8;  // one-k calls and short lines: see the kernel
    const int tiles_per_xcd = flat ? 0 : (int)((n_tiles + 7) / 8);
    const int slice_elems = (flat ? 1 : slice_rounds) * LDS_THREADS;
    const int n_slices = std::max(1, (((m->ncol + 63) & ~63) + slice_elems - 1) / slice_elems);
    const dim3 grid((unsigned)(flat ? n_tiles * n_slices : 8 * (int64_t)tiles_per_xcd * n_slices)), block(LDS_THREADS);
    if (sched) {
        hipError_t e = tbk_raise_lds_limit(reinterpret_cast<const void*>(&hk_csr_lds_kernel<MODE, CONV, KT, true>), (int)(160 * 1024), raised[1]);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((hk_csr_lds_kernel<MODE, CONV, KT, true>), grid, block, lds, m->stream, d_A, m->d_sptr, m->d_srec_r,
                           m->d_srec_v, m->d_colmap, d_k, d_pos, m->dim, m->ncol, m->n_orb, m->n_r, nk, nk_pad, d_H, tiles_per_xcd,
                           slice_elems);
    } else {
        hipError_t e = tbk_raise_lds_limit(reinterpret_cast<const void*>(&hk_csr_lds_kernel<MODE, CONV, KT, false>), (int)(160 * 1024), raised[0]);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((hk_csr_lds_kernel<MODE, CONV, KT, false>), grid, block, lds, m->stream, d_A, m->d_cptr, m->d_rec_r,
                           m->d_rec_v, m->d_colmap, d_k, d_pos, m->dim, m->ncol, m->n_orb, m->n_r, nk, nk_pad, d_H, tiles_per_xcd,
                           slice_elems);
    }
    return hipGetLastError();
}

template <int KT>
hipError_t launch_lds_mode(tbk_model* m, const double* d_A, int64_t nk, int64_t nk_pad, int mode, int convention,
                           const double* d_k, const double* d_pos, double* d_H) {
    if (mode == HK_TRI) return launch_lds<HK_TRI, 2, KT>(m, d_A, nk, nk_pad, d_k, d_pos, d_H);
    if (convention == 1) return launch_lds<HK_FULL, 1, KT>(m, d_A, nk, nk_pad, d_k, d_pos, d_H);
    return launch_lds<HK_FULL, 2, KT>(m, d_A, nk, nk_pad, d_k, d_pos, d_H);
}

}  // namespace

// phase tile of KT k-points x all lattice vectors in <= 80 KiB of LDS (two workgroups per CU): 8, 4, 2, 1, or 0 = no fit
int tbk_csr_tile_kpoints(int64_t n_r) {
    const int64_t budget = 80 * 1024 / (int64_t)sizeof(double);
    // TBK_CSR_KT16=1 (measurement, round 4): tiles of 16 k-points -- twice the FMAs per record decode, ONE workgroup per CU
    static const bool kt16 = tbk_exp_env("TBK_CSR_KT16") && atoi(tbk_exp_env("TBK_CSR_KT16")) != 0;
    if (kt16 && n_r > 0 && n_r * 34 <= 2 * budget - 1024) return 16;
    if (n_r <= 0 || n_r * 4 > budget) return 0;
    if (n_r * 18 <= budget) return 8;
    if (n_r * 10 <= budget) return 4;
    if (n_r * 6 <= budget) return 2;
    return 1;
}

// The walk order of the LDS kernel.  A ds_read_b128 of a wave is served in four groups of 16 lanes, one LDS cycle per group
// when its 16 lanes read 16 different 16-byte slots of the 256-byte LDS row (MI355X_MICROARCH.md, LDS).  A lane reads the
// phase row of ITS record's lattice vector r: slot (r * (kt + 1) + q) mod 16 for k-point q of the tile (rows of kt + 1
// slots) -- the same shift q for every lane, so what decides is the class c(r) = r * (kt + 1) mod 16.  Per wave round (64
// consecutive packed elements) and lane group this is an edge colouring of the bipartite multigraph lanes x classes, one
// edge per record: colours = steps, and a colouring with max-degree colours always exists (Koenig); it is built here by
// the alternating-path algorithm.  A round takes the largest of its four groups' step counts; lanes without a record in a
// step get the code 0x80000000.  A scheduled record carries the byte offset of its phase row and a two-bit sign code instead
// of (kind, r), diagonal values doubled.  The order of a lane's records changes (so does the last bit of the sums),
// deterministically.
void tbk_csr_schedule(int ncol, int kt, const std::vector<int64_t>& cptr, const std::vector<int32_t>& rec_r, const std::vector<double>& rec_v,
                      std::vector<int64_t>& sptr, std::vector<int32_t>& srec_r, std::vector<double>& srec_v) {
    static const int group_of_lane[32] = {0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1};
    const int n_rounds = (ncol + 63) / 64;
    const int spr = kt + 1;  // 16-byte slots per phase row
    sptr.assign((size_t)n_rounds + 1, 0);
    srec_r.clear();
    srec_v.clear();
    struct Edge {
        int u, v;       // lane slot 0..15 inside the group, class 0..15
        int64_t rec;    // index into rec_r / rec_v
        int lane;       // lane 0..63 of the wave
        int colour;
    };
    std::vector<Edge> edges;
    std::vector<std::vector<int>> at_u, at_v;  // [vertex][colour] -> edge index or -1
    for (int wr = 0; wr < n_rounds; ++wr) {
        int round_steps = 0;
        std::vector<Edge> round_edges;
        for (int g = 0; g < 4; ++g) {
            edges.clear();
            int deg_u[16] = {0}, deg_v[16] = {0}, slot = 0;
            for (int lane = 0; lane < 64; ++lane) {
                if ((lane >> 5) * 2 + group_of_lane[lane & 31] != g) continue;
                const int e = wr * 64 + lane;
                const int u = slot++;
                if (e >= ncol) continue;
                for (int64_t t = cptr[(size_t)e]; t < cptr[(size_t)e + 1]; ++t) {
                    const int64_t r = rec_r[(size_t)t] & 0x0fffffff;
                    const int v = (int)((r * spr) & 15);
                    edges.push_back(Edge{u, v, t, lane, -1});
                    deg_u[u]++;
                    deg_v[v]++;
                }
            }
            int delta = 0;
            for (int i = 0; i < 16; ++i) delta = std::max(delta, std::max(deg_u[i], deg_v[i]));
            at_u.assign(16, std::vector<int>((size_t)delta, -1));
            at_v.assign(16, std::vector<int>((size_t)delta, -1));
            for (int ei = 0; ei < (int)edges.size(); ++ei) {
                Edge& ed = edges[(size_t)ei];
                int a = 0, b = 0;
                while (at_u[(size_t)ed.u][(size_t)a] >= 0) ++a;  // free at the lane
                while (at_v[(size_t)ed.v][(size_t)b] >= 0) ++b;  // free at the class
                if (a != b) {
                    // make colour a free at the class vertex: flip the a / b path that starts there with its a-edge
                    // (in a bipartite graph it cannot come back to this edge's lane vertex, where a is free)
                    std::vector<int> path;
                    bool at_class = true;
                    int vertex = ed.v, want = a;
                    for (;;) {
                        const int next = at_class ? at_v[(size_t)vertex][(size_t)want] : at_u[(size_t)vertex][(size_t)want];
                        if (next < 0) break;
                        path.push_back(next);
                        vertex = at_class ? edges[(size_t)next].u : edges[(size_t)next].v;
                        at_class = !at_class;
                        want = (want == a) ? b : a;
                    }
                    for (int pe : path) {  // take the path's edges out ...
                        Edge& q = edges[(size_t)pe];
                        at_u[(size_t)q.u][(size_t)q.colour] = -1;
                        at_v[(size_t)q.v][(size_t)q.colour] = -1;
                    }
                    for (int pe : path) {  // ... and put them back with the two colours exchanged
                        Edge& q = edges[(size_t)pe];
                        q.colour = (q.colour == a) ? b : a;
                        at_u[(size_t)q.u][(size_t)q.colour] = pe;
                        at_v[(size_t)q.v][(size_t)q.colour] = pe;
                    }
                }
                ed.colour = a;
                at_u[(size_t)ed.u][(size_t)a] = ei;
                at_v[(size_t)ed.v][(size_t)a] = ei;
            }
            round_steps = std::max(round_steps, delta);
            round_edges.insert(round_edges.end(), edges.begin(), edges.end());
        }
        const size_t base = srec_r.size();
        srec_r.resize(base + (size_t)round_steps * 64, (int32_t)0x80000000);
        srec_v.resize((base + (size_t)round_steps * 64) * 2, 0.0);
        for (const Edge& ed : round_edges) {
            const size_t at = base + (size_t)ed.colour * 64 + (size_t)ed.lane;
            const int32_t old = rec_r[(size_t)ed.rec];
            const int kind = old >> 28;  // 0 direct, 1 transposed, 2 diagonal
            const uint32_t code = kind == 0 ? 1u : (kind == 1 ? 3u : 0u);  // +1, -1, 0 as a two-bit signed field
            const uint32_t row_bytes = (uint32_t)(old & 0x0fffffff) * (uint32_t)(kt * 2 + 2) * 8u;
            srec_r[at] = (int32_t)((code << 30) | row_bytes);
            const double scale = kind == 2 ? 2.0 : 1.0;
            srec_v[2 * at] = scale * rec_v[2 * (size_t)ed.rec];
            srec_v[2 * at + 1] = scale * rec_v[2 * (size_t)ed.rec + 1];
        }
        sptr[(size_t)wr + 1] = sptr[(size_t)wr] + round_steps;
    }
}

int tbk_launch_hk_csr(tbk_model* m, const double* d_A, int64_t nk, int64_t nk_pad, int mode,
                      int convention, const double* d_k, const double* d_pos, double* d_H) {
    if (nk == 0) return TBK_OK;
    const int kt = tbk_csr_tile_kpoints(m->n_r);
    if (kt > 0) {
        StageTimer t(m, TBK_T_HK);
        if (kt == 16)
            TBK_HIP((launch_lds_mode<16>(m, d_A, nk, nk_pad, mode, convention, d_k, d_pos, d_H)));
        else if (kt == 8)
            TBK_HIP((launch_lds_mode<8>(m, d_A, nk, nk_pad, mode, convention, d_k, d_pos, d_H)));
        else if (kt == 4)
            TBK_HIP((launch_lds_mode<4>(m, d_A, nk, nk_pad, mode, convention, d_k, d_pos, d_H)));
        else if (kt == 2)
            TBK_HIP((launch_lds_mode<2>(m, d_A, nk, nk_pad, mode, convention, d_k, d_pos, d_H)));
        else
            TBK_HIP((launch_lds_mode<1>(m, d_A, nk, nk_pad, mode, convention, d_k, d_pos, d_H)));
        return TBK_OK;
    }
    dim3 grid((m->ncol + 255) / 256, (unsigned)((nk + KT - 1) / KT));
    TBK_ARG(grid.y <= 65535, "k chunk too large for the sparse kernel grid");
    StageTimer t(m, TBK_T_HK);
#define TBK_CSR_LAUNCH(MODE, CONV)                                                                 \
    hipLaunchKernelGGL((hk_csr_kernel<MODE, CONV>), grid, dim3(256), 0, m->stream, d_A, m->d_cptr, \
                       m->d_rec_r, m->d_rec_v, m->d_colmap, d_k, d_pos, m->dim, m->ncol, m->n_orb,  \
                       nk, nk_pad, d_H)
    if (mode == HK_TRI) {
        TBK_CSR_LAUNCH(HK_TRI, 2);
    } else if (convention == 1) {
        TBK_CSR_LAUNCH(HK_FULL, 1);
    } else {
        TBK_CSR_LAUNCH(HK_FULL, 2);
    }
#undef TBK_CSR_LAUNCH
    TBK_HIP(hipGetLastError());
    return TBK_OK;
}
