// tbk_eig.hip -- batched Hermitian eigenvalues of a chunk of H(k).
//
// Reference step: `[la.eigvalsh(ham) for ham in hamiltonians]`
// (/root/reference/src/tbmodels/_tb_model.py:1147-1150): one LAPACK zheevr call per k-point from a
// Python loop, eigenvalues ascending.  Here one batched call per chunk.
//
// H is stored row-major (H[k][i][j]); read as column-major it is H^T = conj(H), which has the same
// (real) spectrum, and the upper triangle i <= j that the H(k) kernels write in TRI mode is the
// LOWER triangle of that column-major matrix -- hence rocblas_fill_lower.

#include <rocsolver/rocsolver.h>

#include "tbk_internal.h"

namespace {

__global__ void count_info_kernel(const int* __restrict__ info, int64_t n, int* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && info[i] != 0) atomicAdd(flag, 1);
}

// LAPACK's zheev* scale the matrix into the safe range before reducing it (zlanhe / zlascl); rocSOLVER's zheevd
// deflates against absolute thresholds instead: on a 1e-30-scaled matrix its eigenvalues came out 17 % off
// (tests/test_gpu_parity.py, structured matrices).  So: one workgroup per matrix brings max |h_ij| of the stored
// triangle to [1, 2) with an exact power of two, and the eigenvalues are scaled back afterwards.
__global__ void __launch_bounds__(256) scale_to_unit_kernel(double* __restrict__ H, int n, double* __restrict__ scale) {
    __shared__ double smax[4];
    double* A = H + (size_t)blockIdx.x * n * n * 2;
    // only the stored triangle (row-major upper, c >= r): in TRI mode the rest of the buffer was never written
    const size_t count = (size_t)n * n;
    double mx = 0.0;
    for (size_t i = threadIdx.x; i < count; i += 256) {
        const size_t r = i / n, c = i % n;
        if (c < r) continue;
        const double v = fmax(fabs(A[2 * i]), fabs(A[2 * i + 1]));
        const double w = (A[2 * i] != A[2 * i] || A[2 * i + 1] != A[2 * i + 1]) ? A[2 * i] + A[2 * i + 1] : v;
        mx = (w > mx || w != w) ? w : mx;  // NaN sticks
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(mx, off, 64);
        mx = (o > mx || o != o) ? o : mx;
    }
    if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = smax[0];
    for (int w = 1; w < 4; ++w) mx = (smax[w] > mx || smax[w] != smax[w]) ? smax[w] : mx;
    double s = 1.0;
    if (mx > 0.0 && isfinite(mx)) s = ldexp(1.0, -ilogb(mx));
    if (s != 1.0)
        for (size_t i = threadIdx.x; i < count; i += 256) {
            const size_t r = i / n, c = i % n;
            if (c < r) continue;
            A[2 * i] *= s;
            A[2 * i + 1] *= s;
        }
    if (threadIdx.x == 0) scale[blockIdx.x] = 1.0 / s;
}

__global__ void unscale_eigenvalues_kernel(double* __restrict__ E, int n, int64_t total, const double* __restrict__ scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) E[i] *= scale[i / n];
}

}  // namespace

size_t tbk_eig_scratch_per_k(const tbk_model* m) {
    return (size_t)m->n_orb * 4 * sizeof(double) + sizeof(int)  // (d, e) + complex tau
           + (tbk_eig_band_supported(m->n_orb) ? tbk_band_scratch_per_matrix(m->n_orb) + 2 * tbk_band_bytes_per_matrix(m->n_orb) +
                                                     tbk_band_xl_buffer_per_matrix(m->n_orb)
                                                   : 0);
}

int tbk_eig_batched(tbk_model* m, double* d_H, int64_t nk, double* d_E) {
    if (nk == 0 || m->n_orb == 0) return TBK_OK;
    const int n = m->n_orb;
    TBK_CHECK(m->ws_E.reserve((size_t)nk * n * sizeof(double)));
    TBK_CHECK(m->ws_info.reserve((size_t)nk * sizeof(int)));
    TBK_CHECK(m->ws_E2.reserve((size_t)nk * sizeof(double)));
    StageTimer t(m, TBK_T_EIG);
    hipLaunchKernelGGL(scale_to_unit_kernel, dim3((unsigned)nk), dim3(256), 0, m->stream, d_H, n, m->ws_E2.as<double>());
    TBK_HIP(hipGetLastError());
    TBK_ROCBLAS(rocsolver_zheevd_strided_batched(
        m->blas, rocblas_evect_none, rocblas_fill_lower, n,
        reinterpret_cast<rocblas_double_complex*>(d_H), n, (rocblas_stride)n * n, d_E,
        (rocblas_stride)n, m->ws_E.as<double>(), (rocblas_stride)n, m->ws_info.as<int>(),
        (rocblas_int)nk));
    hipLaunchKernelGGL(unscale_eigenvalues_kernel, dim3((unsigned)((nk * n + 255) / 256)), dim3(256), 0, m->stream, d_E, n,
                       nk * n, m->ws_E2.as<double>());
    TBK_HIP(hipGetLastError());
    hipLaunchKernelGGL(count_info_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, m->stream,
                       m->ws_info.as<int>(), nk, m->ws_flag.as<int>());
    TBK_HIP(hipGetLastError());
    return TBK_OK;
}
