// tbk_stage.hip -- one-time staging of the hoppings into the layout the H(k) kernel contracts.
//
// The reference walks `self.hop` (dict R -> N x N complex) on every call
// (/root/reference/src/tbmodels/_tb_model.py:1111) and adds the Hermitian conjugate afterwards
// (:1123).  Because H = A + A^H is real-linear in (cos, sin) of each phase, the conjugate part can
// be folded into the staged operand ONCE, and only the upper triangle i <= j is ever contracted:
//
//     H[i][j] = sum_R  p_R h_R[i][j] + conj(p_R) conj(h_R[j][i]),      p_R = c_R + i s_R
//     Re H[i][j] = sum_R  c_R (hr_ij + hr_ji)  -  s_R (hi_ij + hi_ji)
//     Im H[i][j] = sum_R  c_R (hi_ij - hi_ji)  +  s_R (hr_ij - hr_ji)
//
// i.e. a REAL contraction  H[k][e] = sum_kk A[kk][k] * B[kk][e]  with two K rows per lattice vector
// (kk = 2r: cos, kk = 2r+1: sin) and two real output columns (Re, Im) per packed element e = (i <= j).
// That halves the flops of the straightforward complex contraction (N(N+1)/2 instead of N^2 complex
// columns) and removes the transpose-add pass.  Diagonal elements get Im == 0 exactly, as in the
// reference.
//
// Layout written here ("tile-interleaved", so one workgroup row segment is contiguous):
//
//     Bt[kk][e / 16][plane][e % 16]      plane 0 = Re column, 1 = Im column; f64
//
// Padding rows (r >= n_r) and padding elements (e >= ncol) are zero.

#include "tbk_internal.h"

namespace {

__global__ void __launch_bounds__(256)
stage_dense_kernel(const double* __restrict__ hop, const int32_t* __restrict__ colmap, int n_orb,
                   int64_t n_r, int ncol_pad, double* __restrict__ Bt) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t r = blockIdx.y;
    if (e >= ncol_pad || r >= n_r) return;
    const int32_t ij = colmap[e];
    if (ij < 0) return;
    const int i = ij >> 16, j = ij & 0xffff;
    const double* blk = hop + (size_t)r * n_orb * n_orb * 2;
    const double hr = blk[((size_t)i * n_orb + j) * 2], hi = blk[((size_t)i * n_orb + j) * 2 + 1];
    const double gr = blk[((size_t)j * n_orb + i) * 2], gi = blk[((size_t)j * n_orb + i) * 2 + 1];
    const size_t tiles = ncol_pad / TBK_CT;
    const size_t base0 = (((size_t)(2 * r) * tiles + e / TBK_CT) * 2) * TBK_CT + e % TBK_CT;
    const size_t base1 = (((size_t)(2 * r + 1) * tiles + e / TBK_CT) * 2) * TBK_CT + e % TBK_CT;
    Bt[base0] = hr + gr;              // cos row, Re column
    Bt[base0 + TBK_CT] = hi - gi;     // cos row, Im column
    Bt[base1] = -(hi + gi);           // sin row, Re column
    Bt[base1 + TBK_CT] = hr - gr;     // sin row, Im column
}

// k.p coefficients: one real K row per monomial, H[e] = sum_p mono_p * C_p[i][j]
__global__ void __launch_bounds__(256)
stage_kdotp_kernel(const double* __restrict__ coeff, const int32_t* __restrict__ colmap, int n_orb,
                   int64_t n_p, int ncol_pad, double* __restrict__ Bt) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = blockIdx.y;
    if (e >= ncol_pad || p >= n_p) return;
    const int32_t ij = colmap[e];
    if (ij < 0) return;
    const int i = ij >> 16, j = ij & 0xffff;
    const double* blk = coeff + (size_t)p * n_orb * n_orb * 2;
    const size_t tiles = ncol_pad / TBK_CT;
    const size_t base = (((size_t)p * tiles + e / TBK_CT) * 2) * TBK_CT + e % TBK_CT;
    Bt[base] = blk[((size_t)i * n_orb + j) * 2];
    Bt[base + TBK_CT] = blk[((size_t)i * n_orb + j) * 2 + 1];
}

}  // namespace

int tbk_stage_dense(tbk_model* m, const double* d_hop_raw) {
    const size_t bytes = (size_t)m->k2 * m->ncol_pad * 2 * sizeof(double);
    if (bytes == 0) return TBK_OK;
    TBK_HIP(hipMalloc((void**)&m->d_B, bytes));
    m->staged_bytes += (int64_t)bytes;
    TBK_HIP(hipMemsetAsync(m->d_B, 0, bytes, m->stream));
    if (m->n_r > 0) {
        TBK_ARG(m->n_r <= 65535, "more than 65535 lattice vectors");
        dim3 grid((m->ncol_pad + 255) / 256, (unsigned)m->n_r);
        hipLaunchKernelGGL(stage_dense_kernel, grid, dim3(256), 0, m->stream, d_hop_raw, m->d_colmap,
                           m->n_orb, m->n_r, m->ncol_pad, m->d_B);
        TBK_HIP(hipGetLastError());
    }
    return TBK_OK;
}

int tbk_stage_kdotp(tbk_model* m, const double* d_coeff_raw) {
    const size_t bytes = (size_t)m->k2 * m->ncol_pad * 2 * sizeof(double);
    if (bytes == 0) return TBK_OK;
    TBK_HIP(hipMalloc((void**)&m->d_B, bytes));
    m->staged_bytes += (int64_t)bytes;
    TBK_HIP(hipMemsetAsync(m->d_B, 0, bytes, m->stream));
    if (m->n_r > 0) {
        TBK_ARG(m->n_r <= 65535, "more than 65535 Taylor coefficients");
        dim3 grid((m->ncol_pad + 255) / 256, (unsigned)m->n_r);
        hipLaunchKernelGGL(stage_kdotp_kernel, grid, dim3(256), 0, m->stream, d_coeff_raw,
                           m->d_colmap, m->n_orb, m->n_r, m->ncol_pad, m->d_B);
        TBK_HIP(hipGetLastError());
    }
    return TBK_OK;
}
