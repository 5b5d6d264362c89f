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

}  // namespace

size_t tbk_eig_scratch_per_k(const tbk_model* m) {
    return (size_t)m->n_orb * 4 * sizeof(double) + sizeof(int);  // (d, e) + complex tau
}

int tbk_eig_batched(tbk_model* m, double* d_H, int64_t nk, double* d_E) {
    if (nk == 0 || m->n_orb == 0) return TBK_OK;
    const int n = m->n_orb;
    TBK_CHECK(m->ws_E.reserve((size_t)nk * n * sizeof(double)));
    TBK_CHECK(m->ws_info.reserve((size_t)nk * sizeof(int)));
    StageTimer t(m, TBK_T_EIG);
    TBK_ROCBLAS(rocsolver_zheevd_strided_batched(
        m->blas, rocblas_evect_none, rocblas_fill_lower, n,
        reinterpret_cast<rocblas_double_complex*>(d_H), n, (rocblas_stride)n * n, d_E,
        (rocblas_stride)n, m->ws_E.as<double>(), (rocblas_stride)n, m->ws_info.as<int>(),
        (rocblas_int)nk));
    hipLaunchKernelGGL(count_info_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, m->stream,
                       m->ws_info.as<int>(), nk, m->ws_flag.as<int>());
    TBK_HIP(hipGetLastError());
    return TBK_OK;
}
