// probe: do f64 MFMA and VALU instructions of DIFFERENT waves on one SIMD run concurrently on gfx950?
//
// One workgroup of 8 waves per CU (LDS-limited): waves 0..3 run the "matrix" loop, waves 4..7 the "vector" loop
// (wave w and wave w + 4 share a SIMD: waves of a workgroup are dealt round-robin over the four SIMDs).
// Either side can be switched off; the vector side has flavours: f64 FMA, f64 FMA with a DPP operand, f64 add,
// 32-bit integer adds, v_mov_b32 DPP (the pieces the n <= 64 Householder kernel is made of).
// Printed: time of each side alone and of both together.  together ~ max  =>  separate pipes;  ~ sum  =>  shared.
//   hipcc --offload-arch=gfx950 -O3 tools/pipe_probe.hip -o tools/pipe_probe
#include <hip/hip_runtime.h>

#include <cstdio>

typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void matrix_loop(int iters, double a, double b, double& sink) {
    d4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
    asm volatile(
        "s_mov_b32 s20, %6\n"
        "1:\n"
        "v_mfma_f64_16x16x4_f64 %0, %4, %5, %0\n"
        "v_mfma_f64_16x16x4_f64 %1, %4, %5, %1\n"
        "v_mfma_f64_16x16x4_f64 %2, %4, %5, %2\n"
        "v_mfma_f64_16x16x4_f64 %3, %4, %5, %3\n"
        "s_sub_u32 s20, s20, 1\n"
        "s_cmp_lg_u32 s20, 0\n"
        "s_cbranch_scc1 1b\n"
        : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3])
        : "v"(a), "v"(b), "s"(iters)
        : "s20", "scc");
    sink = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

template <int FLAVOUR>
__device__ __forceinline__ void vector_loop(int iters, double a, double b, double& sink) {
    double c0 = a, c1 = b, c2 = a + b, c3 = a - b, c4 = 1.0, c5 = 2.0, c6 = 3.0, c7 = 4.0;
    if (FLAVOUR == 0) {
        asm volatile(
            "s_mov_b32 s21, %10\n"
            "2:\n"
            "v_fmac_f64 %0, %8, %9\nv_fmac_f64 %1, %8, %9\nv_fmac_f64 %2, %8, %9\nv_fmac_f64 %3, %8, %9\n"
            "v_fmac_f64 %4, %8, %9\nv_fmac_f64 %5, %8, %9\nv_fmac_f64 %6, %8, %9\nv_fmac_f64 %7, %8, %9\n"
            "s_sub_u32 s21, s21, 1\ns_cmp_lg_u32 s21, 0\ns_cbranch_scc1 2b\n"
            : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
            : "v"(a), "v"(b), "s"(iters)
            : "s21", "scc");
    } else if (FLAVOUR == 1) {
        asm volatile(
            "s_mov_b32 s21, %10\n"
            "2:\n"
            "v_fmac_f64_dpp %0, %8, %9 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %1, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %2, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %4, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %5, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %6, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %7, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
            "s_sub_u32 s21, s21, 1\ns_cmp_lg_u32 s21, 0\ns_cbranch_scc1 2b\n"
            : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
            : "v"(a), "v"(b), "s"(iters)
            : "s21", "scc");
    } else if (FLAVOUR == 2) {
        asm volatile(
            "s_mov_b32 s21, %10\n"
            "2:\n"
            "v_add_f64 %0, %0, %8\nv_add_f64 %1, %1, %8\nv_add_f64 %2, %2, %8\nv_add_f64 %3, %3, %8\n"
            "v_add_f64 %4, %4, %9\nv_add_f64 %5, %5, %9\nv_add_f64 %6, %6, %9\nv_add_f64 %7, %7, %9\n"
            "s_sub_u32 s21, s21, 1\ns_cmp_lg_u32 s21, 0\ns_cbranch_scc1 2b\n"
            : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
            : "v"(a), "v"(b), "s"(iters)
            : "s21", "scc");
    } else {
        int i0 = (int)a, i1 = (int)b, i2 = 3, i3 = 4, i4 = 5, i5 = 6, i6 = 7, i7 = 8;
        const int x = threadIdx.x;
        if (FLAVOUR == 3) {
            asm volatile(
                "s_mov_b32 s21, %9\n"
                "2:\n"
                "v_add_u32 %0, %0, %8\nv_add_u32 %1, %1, %8\nv_add_u32 %2, %2, %8\nv_add_u32 %3, %3, %8\n"
                "v_add_u32 %4, %4, %8\nv_add_u32 %5, %5, %8\nv_add_u32 %6, %6, %8\nv_add_u32 %7, %7, %8\n"
                "s_sub_u32 s21, s21, 1\ns_cmp_lg_u32 s21, 0\ns_cbranch_scc1 2b\n"
                : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7)
                : "v"(x), "s"(iters)
                : "s21", "scc");
        } else {
            asm volatile(
                "s_mov_b32 s21, %9\n"
                "2:\n"
                "v_mov_b32_dpp %0, %8 row_ror:1 row_mask:0xf bank_mask:0xf\n"
                "v_mov_b32_dpp %1, %8 row_ror:2 row_mask:0xf bank_mask:0xf\n"
                "v_mov_b32_dpp %2, %8 row_ror:4 row_mask:0xf bank_mask:0xf\n"
                "v_mov_b32_dpp %3, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                "v_mov_b32_dpp %4, %8 row_ror:1 row_mask:0xf bank_mask:0xf\n"
                "v_mov_b32_dpp %5, %8 row_ror:2 row_mask:0xf bank_mask:0xf\n"
                "v_mov_b32_dpp %6, %8 row_ror:4 row_mask:0xf bank_mask:0xf\n"
                "v_mov_b32_dpp %7, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                "s_sub_u32 s21, s21, 1\ns_cmp_lg_u32 s21, 0\ns_cbranch_scc1 2b\n"
                : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7)
                : "v"(x), "s"(iters)
                : "s21", "scc");
        }
        c0 = i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7;
    }
    sink = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}

template <int FLAVOUR>
__global__ void __launch_bounds__(512) probe(double* out, int it_matrix, int it_vector, int vector_waves) {
    extern __shared__ double lds[];  // sized so that ONE workgroup fits a CU
    const int wave = threadIdx.x >> 6;
    double sink = 0.0;
    const double a = 1.0 + 1e-9 * threadIdx.x, b = 1e-7;
    if (wave < 4) {
        if (it_matrix > 0) matrix_loop(it_matrix, a, b, sink);
    } else if (wave - 4 < vector_waves) {
        if (it_vector > 0) vector_loop<FLAVOUR>(it_vector, a, b, sink);
    }
    if (sink == 12345.678) out[blockIdx.x * blockDim.x + threadIdx.x] = sink + lds[threadIdx.x];
}

template <int FLAVOUR>
float run(double* out, int cus, int it_matrix, int it_vector, int vector_waves = 4) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e30f;
    (void)hipFuncSetAttribute((const void*)probe<FLAVOUR>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        probe<FLAVOUR><<<cus, 512, 100 * 1024>>>(out, it_matrix, it_vector, vector_waves);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    double* out;
    (void)hipMalloc(&out, (size_t)cus * 512 * 8);
    const int it_m = 40000;  // x 4 MFMA x 64 cycles  = 10.2 M cycles
    const int it_v = 320000;  // x 8 VALU x 4 cycles   = 10.2 M cycles
    const char* names[5] = {"v_fmac_f64", "v_fmac_f64_dpp row_newbcast", "v_add_f64", "v_add_u32", "v_mov_b32_dpp row_ror"};
    const float m_alone = run<0>(out, cus, it_m, 0);
    printf("matrix side alone (4 waves x %d x 4 v_mfma_f64_16x16x4): %.3f ms = %.1f TFLOP/s\n", it_m, m_alone,
           (double)cus * 4 * it_m * 4 * 2048.0 / m_alone / 1e9);
    float v_alone[5], both[5];
    v_alone[0] = run<0>(out, cus, 0, it_v); both[0] = run<0>(out, cus, it_m, it_v);
    v_alone[1] = run<1>(out, cus, 0, it_v); both[1] = run<1>(out, cus, it_m, it_v);
    v_alone[2] = run<2>(out, cus, 0, it_v); both[2] = run<2>(out, cus, it_m, it_v);
    v_alone[3] = run<3>(out, cus, 0, it_v); both[3] = run<3>(out, cus, it_m, it_v);
    v_alone[4] = run<4>(out, cus, 0, it_v); both[4] = run<4>(out, cus, it_m, it_v);
    for (int f = 0; f < 5; ++f)
        printf("%-30s alone %.3f ms (%.2f cycles/instr at 2.4 GHz); with the matrix side: %.3f ms  (max %.3f, sum %.3f)\n", names[f],
               v_alone[f], v_alone[f] * 1e-3 * 2.4e9 / ((double)it_v * 8), both[f], m_alone > v_alone[f] ? m_alone : v_alone[f],
               m_alone + v_alone[f]);
    // two vector waves per SIMD cannot be had in this layout; one vector wave on ONE SIMD only: does the matrix side notice?
    printf("one vector wave per CU beside four matrix waves: %.3f ms\n", run<0>(out, cus, it_m, it_v, 1));
    return 0;
}
