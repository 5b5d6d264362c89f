// tbk_peak.hip -- microbenchmark of the sustained v_mfma_f64_16x16x4_f64 rate.
//
// The local hardware guide lists no FP64 matrix peak; AMD's public figure for MI355X is
// 78.6 TFLOP/s.  bench.py reports both the spec and what this loop sustains on the device it runs
// on, so roofline fractions can be read against either.

#include <cstdlib>

#include "tbk_internal.h"

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) mfma_f64_loop(double* out, int iters, double seed) {
    d4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
    double a = seed == 0.0 ? 0.0 : seed + threadIdx.x * 1e-3;
    double b = seed == 0.0 ? 0.0 : seed - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[blockIdx.x * blockDim.x + threadIdx.x] = s;  // keep the loop alive
}

}  // namespace

int tbk_run_mfma_f64_peak(double* tflops) {
    hipDeviceProp_t prop;
    int dev = 0;
    TBK_HIP(hipGetDevice(&dev));
    TBK_HIP(hipGetDeviceProperties(&prop, dev));
    const int iters = 4000;
    const bool verbose = getenv("TBK_VERBOSE") != nullptr;
    double* d_out = nullptr;
    TBK_HIP(hipMalloc((void**)&d_out, (size_t)prop.multiProcessorCount * 4 * 256 * sizeof(double)));
    hipEvent_t e0, e1;
    TBK_HIP(hipEventCreate(&e0));
    TBK_HIP(hipEventCreate(&e1));
    double best = 0.0;
    // waves per SIMD x {non-trivial operands, all-zero operands}: the chip clocks to its power budget, so
    // the sustained rate depends on both (MI355X_MICROARCH.md, DVFS give-back)
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int grid = prop.multiProcessorCount * wps;
        for (int zero = 0; zero < 2; ++zero) {
            double best_cfg = 0.0;
            for (int rep = 0; rep < 3; ++rep) {
                TBK_HIP(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(mfma_f64_loop, dim3(grid), dim3(256), 0, 0, d_out, iters,
                                   zero ? 0.0 : 1.0 + rep);
                TBK_HIP(hipEventRecord(e1, 0));
                TBK_HIP(hipEventSynchronize(e1));
                float ms = 0.f;
                TBK_HIP(hipEventElapsedTime(&ms, e0, e1));
                const double flops = (double)grid * 4 /*waves*/ * iters * 8.0 * (2.0 * 16 * 16 * 4);
                const double tf = flops / (ms * 1e-3) / 1e12;
                if (rep > 0 && tf > best_cfg) best_cfg = tf;
            }
            if (verbose)
                fprintf(stderr, "[tbk] mfma_f64 loop: %d waves/SIMD, %s operands: %.2f TFLOP/s\n", wps,
                        zero ? "zero" : "non-zero", best_cfg);
            if (!zero && best_cfg > best) best = best_cfg;
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(d_out);
    *tflops = best;
    return TBK_OK;
}
