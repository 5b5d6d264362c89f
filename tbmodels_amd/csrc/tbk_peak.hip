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
    // The whole loop is ONE asm block with the eight accumulators in AGPRs: written with the builtin, hipcc moved all
    // 64 accumulator registers between VGPRs and AGPRs on every trip (128 v_accvgpr moves per 8 MFMAs), and with the
    // accumulators in VGPRs the loop reported 47 TFLOP/s while the H(k) kernel (AGPR accumulators) sustained 68.
    asm volatile(
        "s_mov_b32 s20, %10\n"
        "1:\n"
        "v_mfma_f64_16x16x4_f64 %0, %8, %9, %0\n"
        "v_mfma_f64_16x16x4_f64 %1, %8, %9, %1\n"
        "v_mfma_f64_16x16x4_f64 %2, %8, %9, %2\n"
        "v_mfma_f64_16x16x4_f64 %3, %8, %9, %3\n"
        "v_mfma_f64_16x16x4_f64 %4, %8, %9, %4\n"
        "v_mfma_f64_16x16x4_f64 %5, %8, %9, %5\n"
        "v_mfma_f64_16x16x4_f64 %6, %8, %9, %6\n"
        "v_mfma_f64_16x16x4_f64 %7, %8, %9, %7\n"
        "s_sub_u32 s20, s20, 1\n"
        "s_cmp_lg_u32 s20, 0\n"
        "s_cbranch_scc1 1b\n"
        : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]), "+a"(acc[7])
        : "v"(a), "v"(b), "s"(iters)
        : "s20", "scc");
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[blockIdx.x * blockDim.x + threadIdx.x] = s;  // keep the loop alive
}

__global__ void __launch_bounds__(256) fma_f64_loop(double* out, int iters, double seed) {
    double acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = seed * (i + 1);
    const double a = 1.0 + 1e-9 * threadIdx.x, b = 1e-7;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = fma(acc[i], a, b);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
    if (s == 12345.678) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

}  // namespace

// TBK_VERBOSE probe: does a VALU-f64 kernel run in the shadow of an MFMA-f64 kernel on the same SIMDs?
static int overlap_probe(int cus, double* d_out) {
    hipStream_t sa, sb;
    TBK_HIP(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    TBK_HIP(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t e0, e1, f0, f1;
    TBK_HIP(hipEventCreate(&e0));
    TBK_HIP(hipEventCreate(&e1));
    TBK_HIP(hipEventCreate(&f0));
    TBK_HIP(hipEventCreate(&f1));
    const int it_m = 100000, it_f = 400000;
    float ms_m = 0, ms_f = 0, ms_m2 = 0, ms_f2 = 0;
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            TBK_HIP(hipDeviceSynchronize());
            if (mode == 0) {  // each alone
                TBK_HIP(hipEventRecord(e0, sa));
                hipLaunchKernelGGL(mfma_f64_loop, dim3(cus * 2), dim3(256), 0, sa, d_out, it_m, 1.5);
                TBK_HIP(hipEventRecord(e1, sa));
                TBK_HIP(hipEventSynchronize(e1));
                TBK_HIP(hipEventRecord(f0, sb));
                hipLaunchKernelGGL(fma_f64_loop, dim3(cus), dim3(256), 0, sb, d_out, it_f, 1.5);
                TBK_HIP(hipEventRecord(f1, sb));
                TBK_HIP(hipEventSynchronize(f1));
                TBK_HIP(hipEventElapsedTime(&ms_m, e0, e1));
                TBK_HIP(hipEventElapsedTime(&ms_f, f0, f1));
            } else {  // together
                TBK_HIP(hipEventRecord(e0, sa));
                hipLaunchKernelGGL(mfma_f64_loop, dim3(cus * 2), dim3(256), 0, sa, d_out, it_m, 1.5);
                TBK_HIP(hipEventRecord(e1, sa));
                TBK_HIP(hipEventRecord(f0, sb));
                hipLaunchKernelGGL(fma_f64_loop, dim3(cus), dim3(256), 0, sb, d_out, it_f, 1.5);
                TBK_HIP(hipEventRecord(f1, sb));
                TBK_HIP(hipEventSynchronize(e1));
                TBK_HIP(hipEventSynchronize(f1));
                TBK_HIP(hipEventElapsedTime(&ms_m2, e0, e1));
                TBK_HIP(hipEventElapsedTime(&ms_f2, f0, f1));
            }
        }
    }
    const double fma_tf = (double)cus * 256 * it_f * 8.0 * 2.0 / (ms_f * 1e-3) / 1e12;
    fprintf(stderr,
            "[tbk] overlap probe: mfma alone %.2f ms, fma alone %.2f ms (%.1f TFLOP/s VALU f64); together: mfma %.2f ms, "
            "fma %.2f ms\n",
            ms_m, ms_f, fma_tf, ms_m2, ms_f2);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipEventDestroy(f0);
    (void)hipEventDestroy(f1);
    (void)hipStreamDestroy(sa);
    (void)hipStreamDestroy(sb);
    return TBK_OK;
}

int tbk_run_mfma_f64_peak(double* tflops) {
    hipDeviceProp_t prop;
    int dev = 0;
    TBK_HIP(hipGetDevice(&dev));
    TBK_HIP(hipGetDeviceProperties(&prop, dev));
    const int iters = 100000;  // ~20 ms per launch: long enough for the clock to settle
    const bool verbose = tbk_exp_env("TBK_VERBOSE") != nullptr;
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
    if (verbose) (void)overlap_probe(prop.multiProcessorCount, d_out);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(d_out);
    *tflops = best;
    return TBK_OK;
}
