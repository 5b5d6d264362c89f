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
            double dot = 0.0;
            for (int d = 0; d < dim; ++d)
                dot = fma(kpts[kq * dim + d], pos[oj * dim + d] - pos[oi * dim + d], dot);
            double sn, cs;
            sincospi(2.0 * dot, &sn, &cs);
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

}  // namespace

int tbk_launch_hk_csr(tbk_model* m, const double* d_A, int64_t nk, int64_t nk_pad, int mode,
                      int convention, const double* d_k, const double* d_pos, double* d_H) {
    if (nk == 0) return TBK_OK;
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
