// probe: semantics and issue rate of v_fmac_f64_dpp row_newbcast (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* o, const double* a) {
    double acc = o[threadIdx.x], v = a[threadIdx.x], x = a[64 + threadIdx.x];
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(x));
    asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:7 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(x));
    o[threadIdx.x] = acc;
}
template <bool DPP>
__global__ void rate(double* o, const double* a, int iters) {
    double v = a[threadIdx.x], x = a[64 + threadIdx.x];
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
    for (int i = 0; i < iters; ++i) {
        if (DPP) {
            asm volatile(
                "v_fmac_f64_dpp %0, %8, %9 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f64_dpp %1, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f64_dpp %2, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f64_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f64_dpp %4, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f64_dpp %5, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f64_dpp %6, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f64_dpp %7, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(v), "v"(x));
        } else {
            asm volatile(
                "v_fmac_f64 %0, %8, %9\nv_fmac_f64 %1, %8, %9\nv_fmac_f64 %2, %8, %9\nv_fmac_f64 %3, %8, %9\n"
                "v_fmac_f64 %4, %8, %9\nv_fmac_f64 %5, %8, %9\nv_fmac_f64 %6, %8, %9\nv_fmac_f64 %7, %8, %9\n"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(v), "v"(x));
        }
    }
    o[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}
int main() {
    double *o, *a; (void)hipMalloc(&o, 1 << 26); (void)hipMalloc(&a, 128*8);
    double ho[64], ha[128];
    for (int i = 0; i < 64; ++i) { ho[i] = 0; ha[i] = i + 1; ha[64+i] = 1000 + i; }
    (void)hipMemcpy(o, ho, sizeof ho, hipMemcpyHostToDevice); (void)hipMemcpy(a, ha, sizeof ha, hipMemcpyHostToDevice);
    k<<<1,64>>>(o, a); (void)hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; i += 9) printf("lane %d: %g (expect %g)\n", i, ho[i], (double)((i/16)*16+5+1)*(1000+i) - (double)((i/16)*16+7+1)*(1000+i));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000, blocks = 256 * 8, threads = 256;
    for (int dpp = 0; dpp < 2; ++dpp) for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        if (dpp) rate<true><<<blocks, threads>>>(o, a, iters); else rate<false><<<blocks, threads>>>(o, a, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.3f ms, %.1f TFLOP/s\n", dpp ? "v_fmac_f64_dpp row_newbcast" : "v_fmac_f64", ms, 2.0 * 8 * iters * (double)blocks * threads / ms / 1e9);
    }
}
