// tbk_phase.hip -- phase-factor rows for one k-chunk.
//
// Reference step: the `np.exp(2j * np.pi * np.dot(k_array, R))` of
// /root/reference/src/tbmodels/_tb_model.py:1118, evaluated there once per stored R inside the
// Python loop.  Here every phase of a chunk is produced once, as two real rows per lattice vector:
//
//     A[2r    ][k] = cos(2 pi k.R_r)
//     A[2r + 1][k] = sin(2 pi k.R_r)          A is [k2][nk_pad], k contiguous
//
// which is the left operand of the real contraction in tbk_hk_dense.hip (its K index) and the
// gather table of tbk_hk_csr.hip.  k.R is accumulated in f64 with FMAs and handed to sincospi(2 k.R),
// whose argument reduction is exact, so |k| >> 1 costs no accuracy beyond the rounding of k.R itself
// (the reference multiplies by a rounded 2 pi first, which is worse).
//
// HBM-write bound: 16 B written per (k, R) against ~50 flops; one thread per (k, R), k fastest so a
// wave writes two 512 B row segments.

#include "tbk_internal.h"

namespace {

__global__ void __launch_bounds__(256)
phase_rows_kernel(const double* __restrict__ k, const int32_t* __restrict__ R, int dim, int64_t nk,
                  int64_t nk_pad, int64_t n_r, double* __restrict__ A) {
    const int64_t kidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t r = blockIdx.y;
    if (kidx >= nk_pad) return;
    double c = 0.0, s = 0.0;
    if (kidx < nk && r < n_r) {
        double dot = 0.0;
        for (int d = 0; d < dim; ++d)
            dot = fma(k[kidx * dim + d], (double)R[r * dim + d], dot);
        sincospi(2.0 * dot, &s, &c);
    }
    A[(2 * r) * nk_pad + kidx] = c;
    A[(2 * r + 1) * nk_pad + kidx] = s;
}

// k.p monomials (kdotp.py:71-78): A[p][k] = prod_d k_d^powers[p][d]
__global__ void __launch_bounds__(256)
monomial_rows_kernel(const double* __restrict__ k, const int32_t* __restrict__ powers, int dim,
                     int64_t nk, int64_t nk_pad, int64_t n_p, double* __restrict__ A) {
    const int64_t kidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = blockIdx.y;
    if (kidx >= nk_pad) return;
    double v = 0.0;
    if (kidx < nk && p < n_p) {
        v = 1.0;
        for (int d = 0; d < dim; ++d) {
            const double x = k[kidx * dim + d];
            const int e = powers[p * dim + d];
            // repeated multiplication in increasing order, like numpy's integer power for small e
            double acc = 1.0;
            for (int t = 0; t < e; ++t) acc *= x;
            v *= acc;
        }
    }
    A[p * nk_pad + kidx] = v;
}

// convention 1 (_tb_model.py:1124-1128): e[k][p] = exp(2 pi i k.pos_p), one (cos, sin) pair per (k, orbital)
__global__ void __launch_bounds__(256)
orbital_phase_kernel(const double* __restrict__ k, const double* __restrict__ pos, int dim, int64_t nk,
                     int n_orb, double* __restrict__ orb) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nk * n_orb) return;
    const int64_t kq = idx / n_orb;
    const int p = (int)(idx % n_orb);
    double dot = 0.0;
    for (int d = 0; d < dim; ++d) dot = fma(k[kq * dim + d], pos[p * dim + d], dot);
    double s, c;
    sincospi(2.0 * dot, &s, &c);
    orb[2 * idx] = c;
    orb[2 * idx + 1] = s;
}

}  // namespace

int tbk_launch_orbital_phases(tbk_model* m, const double* d_k, const double* d_pos, int64_t nk, double* d_orb) {
    if (nk == 0) return TBK_OK;
    const int64_t total = nk * m->n_orb;
    hipLaunchKernelGGL(orbital_phase_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, m->stream, d_k,
                       d_pos, m->dim, nk, m->n_orb, d_orb);
    TBK_HIP(hipGetLastError());
    return TBK_OK;
}

int tbk_launch_phase(tbk_model* m, const double* d_k, int64_t nk, int64_t nk_pad, double* d_A) {
    if (m->n_r_pad == 0 || nk_pad == 0) return TBK_OK;
    StageTimer t(m, TBK_T_PHASE);
    dim3 grid((unsigned)((nk_pad + 255) / 256), (unsigned)m->n_r_pad);
    hipLaunchKernelGGL(phase_rows_kernel, grid, dim3(256), 0, m->stream, d_k, m->d_R, m->dim, nk,
                       nk_pad, m->n_r, d_A);
    TBK_HIP(hipGetLastError());
    return TBK_OK;
}

int tbk_launch_monomials(hipStream_t s, const int32_t* d_powers, int dim, int64_t n_p,
                         int64_t n_p_pad, const double* d_k, int64_t nk, int64_t nk_pad,
                         double* d_A) {
    if (n_p_pad == 0 || nk_pad == 0) return TBK_OK;
    dim3 grid((unsigned)((nk_pad + 255) / 256), (unsigned)n_p_pad);
    hipLaunchKernelGGL(monomial_rows_kernel, grid, dim3(256), 0, s, d_k, d_powers, dim, nk, nk_pad,
                       n_p, d_A);
    TBK_HIP(hipGetLastError());
    return TBK_OK;
}
