// tbk_kdotp.hip -- Taylor coefficients of H(k) around one k-point: Model.construct_kdotp
// (/root/reference/src/tbmodels/_tb_model.py:942-982), SURVEY.md section 8(f) rank 2.
//
//     C_p = (2 pi i)^{|p|} / prod_d p_d!  *  sum_R [ R^p e^{+2 pi i k.R} hop_R  +  (-R)^p e^{-2 pi i k.R} hop_R^H ]
//
// the derivative flavour of the Fourier sum: the same staged operand Bt (tbk_stage.hip), the phase of ONE
// k-point, and a monomial weight R^p per lattice vector.  C_p is Hermitian, so only the packed upper
// triangle is accumulated and the lower one is written as its conjugate.  The work is n_p * N(N+1)/2 * N_R
// complex multiply-adds (20 * 2080 * 4096 at order 3 of the headline model): one thread per (element, p).

#include <algorithm>

#include "tbk_internal.h"

namespace {

typedef double d2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256)
kdotp_phase_kernel(const double* __restrict__ k0, const int32_t* __restrict__ R, int dim, int64_t n_r,
                   double* __restrict__ ph) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_r) return;
    double dot = 0.0;
    for (int d = 0; d < dim; ++d) dot = fma(k0[d], (double)R[r * dim + d], dot);
    double s, c;
    sincospi(2.0 * dot, &s, &c);
    ph[2 * r] = c;
    ph[2 * r + 1] = s;
}

__global__ void __launch_bounds__(256)
kdotp_coeff_kernel(const double* __restrict__ Bt, const int32_t* __restrict__ colmap,
                   const int32_t* __restrict__ R, const double* __restrict__ ph,
                   const int32_t* __restrict__ powers, const double* __restrict__ prefactor, int dim,
                   int64_t n_r, int ncol, int ncol_pad, int n_orb, double* __restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int p = blockIdx.y;
    if (e >= ncol) return;
    const int32_t ij = colmap[e];
    const int oi = ij >> 16, oj = ij & 0xffff;
    int pw[TBK_MAX_DIM];
    int deg = 0;
    for (int d = 0; d < dim; ++d) {
        pw[d] = powers[p * dim + d];
        deg += pw[d];
    }
    const double sgn = (deg & 1) ? -1.0 : 1.0;
    const size_t tiles = ncol_pad / TBK_CT;
    const size_t col = ((size_t)(e / TBK_CT) * 2) * TBK_CT + e % TBK_CT;
    double xr = 0.0, xi = 0.0;
    for (int64_t r = 0; r < n_r; ++r) {
        double mono = 1.0;
        for (int d = 0; d < dim; ++d) {
            const double base = (double)R[r * dim + d];
            for (int t = 0; t < pw[d]; ++t) mono *= base;
        }
        if (mono == 0.0) continue;
        const double* row_c = Bt + (size_t)(2 * r) * tiles * 2 * TBK_CT + col;      // cos row
        const double* row_s = Bt + (size_t)(2 * r + 1) * tiles * 2 * TBK_CT + col;  // sin row
        // staged planes: cos row = (hr + gr, hi - gi), sin row = (-(hi + gi), hr - gr);  h = hop[i][j], g = hop[j][i]
        const double s_re = row_c[0], d_im = row_c[TBK_CT], ms_im = row_s[0], d_re = row_s[TBK_CT];
        const double hr = 0.5 * (s_re + d_re), gr = 0.5 * (s_re - d_re);
        const double hi = 0.5 * (d_im - ms_im), gi = 0.5 * (-ms_im - d_im);
        const double c = ph[2 * r], s = ph[2 * r + 1];
        // ph h + sgn conj(ph) conj(g)
        const double tr = (c * hr - s * hi) + sgn * (c * gr - s * gi);
        const double ti = (c * hi + s * hr) - sgn * (c * gi + s * gr);
        xr = fma(mono, tr, xr);
        xi = fma(mono, ti, xi);
    }
    const double fr = prefactor[2 * p], fi = prefactor[2 * p + 1];
    const double cr = fr * xr - fi * xi, ci = fr * xi + fi * xr;
    double* cp = out + (size_t)p * n_orb * n_orb * 2;
    *reinterpret_cast<d2*>(cp + ((size_t)oi * n_orb + oj) * 2) = (d2){cr, ci};
    if (oi != oj) *reinterpret_cast<d2*>(cp + ((size_t)oj * n_orb + oi) * 2) = (d2){cr, -ci};
}

}  // namespace

// powers: host [n_p][dim]; prefactor: host [n_p][2]; coeffs_out: host [n_p][n_orb][n_orb][2]
extern "C" int tbk_kdotp_coefficients(tbk_model* m, const double* k0, int64_t n_p, const int32_t* powers,
                                      const double* prefactor, double* coeffs_out) {
    TBK_ARG(m != nullptr, "model is NULL");
    TBK_LOCK(m);
    TBK_ARG(!m->sparse && !m->kdotp, "construct_kdotp needs a dense tight-binding model handle");
    TBK_ARG(n_p >= 0 && n_p <= 65535, "n_p out of range");
    if (n_p == 0) return TBK_OK;
    TBK_ARG(k0 && powers && prefactor && coeffs_out, "NULL argument");
    for (int64_t t = 0; t < n_p * m->dim; ++t) TBK_ARG(powers[t] >= 0, "negative power");
    TBK_HIP(hipSetDevice(m->device));
    const size_t nn2 = (size_t)m->n_orb * m->n_orb * 2;
    const size_t out_bytes = (size_t)n_p * nn2 * sizeof(double);
    const size_t aux_bytes = (size_t)m->dim * sizeof(double) + (size_t)n_p * m->dim * sizeof(int32_t) +
                             (size_t)n_p * 2 * sizeof(double) + (size_t)std::max<int64_t>(m->n_r, 1) * 2 * sizeof(double);
    TBK_CHECK(m->ws_out.reserve(out_bytes));
    TBK_CHECK(m->ws_k.reserve(aux_bytes + 64));
    char* aux = m->ws_k.as<char>();
    double* d_k0 = reinterpret_cast<double*>(aux);
    double* d_pref = d_k0 + m->dim;
    double* d_ph = d_pref + n_p * 2;
    int32_t* d_pw = reinterpret_cast<int32_t*>(d_ph + std::max<int64_t>(m->n_r, 1) * 2);
    TBK_HIP(hipMemcpyAsync(d_k0, k0, m->dim * sizeof(double), hipMemcpyHostToDevice, m->stream));
    TBK_HIP(hipMemcpyAsync(d_pref, prefactor, (size_t)n_p * 2 * sizeof(double), hipMemcpyHostToDevice, m->stream));
    TBK_HIP(hipMemcpyAsync(d_pw, powers, (size_t)n_p * m->dim * sizeof(int32_t), hipMemcpyHostToDevice, m->stream));
    TBK_HIP(hipMemsetAsync(m->ws_out.ptr, 0, out_bytes, m->stream));
    if (m->n_r > 0) {
        hipLaunchKernelGGL(kdotp_phase_kernel, dim3((unsigned)((m->n_r + 255) / 256)), dim3(256), 0, m->stream, d_k0,
                           m->d_R, m->dim, m->n_r, d_ph);
        TBK_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(kdotp_coeff_kernel, dim3((m->ncol + 255) / 256, (unsigned)n_p), dim3(256), 0, m->stream,
                       m->d_B, m->d_colmap, m->d_R, d_ph, d_pw, d_pref, m->dim, m->n_r, m->ncol, m->ncol_pad,
                       m->n_orb, m->ws_out.as<double>());
    TBK_HIP(hipGetLastError());
    TBK_HIP(hipMemcpyAsync(coeffs_out, m->ws_out.ptr, out_bytes, hipMemcpyDeviceToHost, m->stream));
    TBK_HIP(hipStreamSynchronize(m->stream));
    return TBK_OK;
}
