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
        if (CONV == 1) {
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

template <int MODE, int CONV, int KT>
__global__ void __launch_bounds__(LDS_THREADS)
hk_csr_lds_kernel(const double* __restrict__ A, const int64_t* __restrict__ cptr,
                  const int32_t* __restrict__ rec_r, const double* __restrict__ rec_v,
                  const int32_t* __restrict__ colmap, const double* __restrict__ kpts,
                  const double* __restrict__ pos, int dim, int ncol, int n_orb, int64_t n_r, int64_t nk,
                  int64_t nk_pad, double* __restrict__ H) {
    extern __shared__ __attribute__((aligned(16))) double sph[];  // [n_r][KT * 2 + 2]
    constexpr int LD = KT * 2 + 2;
    const int64_t k0 = (int64_t)blockIdx.x * KT;
    for (int64_t idx = threadIdx.x; idx < n_r * KT; idx += LDS_THREADS) {
        const int64_t r = idx / KT;
        const int q = (int)(idx % KT);
        sph[r * LD + 2 * q] = A[(2 * r) * nk_pad + k0 + q];
        sph[r * LD + 2 * q + 1] = A[(2 * r + 1) * nk_pad + k0 + q];
    }
    __syncthreads();
    const size_t nn = (size_t)n_orb * n_orb;
    for (int e = threadIdx.x; e < ncol; e += LDS_THREADS) {
        double re[KT], im[KT];
#pragma unroll
        for (int q = 0; q < KT; ++q) re[q] = im[q] = 0.0;
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
            const double vr = val_n[0], vi = val_n[1];
            if (t + 1 < end) {
                packed_n = rec_r[t + 1];
                val_n = *reinterpret_cast<const d2*>(rec_v + 2 * (t + 1));
            }
            const int kind = packed >> 28;
            const int64_t r = packed & 0x0fffffff;
            const double sign_im = (kind == 1) ? -1.0 : ((kind == 2) ? 0.0 : 1.0);
            const double scale_re = (kind == 2) ? 2.0 : 1.0;
            const double ar = scale_re * vr, ai = scale_re * vi, br = sign_im * vi, bi = sign_im * vr;
            const double* ph = sph + r * LD;
#pragma unroll
            for (int q = 0; q < KT; ++q) {
                const d2 cs = *reinterpret_cast<const d2*>(ph + 2 * q);
                re[q] = fma(cs[0], ar, re[q]);
                re[q] = fma(-cs[1], ai, re[q]);
                im[q] = fma(cs[0], br, im[q]);
                im[q] = fma(cs[1], bi, im[q]);
            }
        }
        const int32_t ij = colmap[e];
        const int oi = ij >> 16, oj = ij & 0xffff;
#pragma unroll
        for (int q = 0; q < KT; ++q) {
            const int64_t kq = k0 + q;
            if (kq >= nk) break;
            double vr = re[q], vi = im[q];
            if (CONV == 1) {
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
    static bool raised[TBK_MAX_DEVICES] = {};
    {
        hipError_t e = tbk_raise_lds_limit(reinterpret_cast<const void*>(&hk_csr_lds_kernel<MODE, CONV, KT>), (int)(80 * 1024), raised);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((hk_csr_lds_kernel<MODE, CONV, KT>), dim3((unsigned)((nk + KT - 1) / KT)), dim3(LDS_THREADS), lds,
                       m->stream, d_A, m->d_cptr, m->d_rec_r, m->d_rec_v, m->d_colmap, d_k, d_pos, m->dim, m->ncol,
                       m->n_orb, m->n_r, nk, nk_pad, d_H);
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

int tbk_launch_hk_csr(tbk_model* m, const double* d_A, int64_t nk, int64_t nk_pad, int mode,
                      int convention, const double* d_k, const double* d_pos, double* d_H) {
    if (nk == 0) return TBK_OK;
    // phase tile of KT k-points x all lattice vectors in <= 80 KiB of LDS (two workgroups per CU)
    const int64_t budget = 80 * 1024 / (int64_t)sizeof(double);
    if (m->n_r > 0 && m->n_r * 4 <= budget) {
        StageTimer t(m, TBK_T_HK);
        if (m->n_r * 18 <= budget)
            TBK_HIP((launch_lds_mode<8>(m, d_A, nk, nk_pad, mode, convention, d_k, d_pos, d_H)));
        else if (m->n_r * 10 <= budget)
            TBK_HIP((launch_lds_mode<4>(m, d_A, nk, nk_pad, mode, convention, d_k, d_pos, d_H)));
        else if (m->n_r * 6 <= budget)
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
