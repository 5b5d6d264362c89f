// Cycles per phase of band_reduce_kernel (workgroup 0, while the whole grid runs) and the time of both kernels of
// the two-stage reduction without PCIe in the way.  Build here, run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DTBK_PHASE_CLOCK -Iinclude -Itbmodels_amd/csrc \
//         tools/band_phase_clock.hip -o tools/band_phase_clock -lrocblas && tools/band_phase_clock 256 4096
#include <cstdio>
#include <cstdlib>
#include <vector>

#define TBK_EXPERIMENTS 1  // (this driver flips the measurement switches: tbk_exp_env reads the environment)
#include "../tbmodels_amd/csrc/tbk_eig_band.hip"
#include "../tbmodels_amd/csrc/tbk_eig_band_chase.hip"
#include "../tbmodels_amd/csrc/tbk_eig_band_xl.hip"

void tbk_set_error(const char*, ...) {}
int DevBuf::reserve(size_t) { return 0; }
void DevBuf::release() {}
StageTimer::StageTimer(tbk_model* m_, int, hipStream_t s) : m(m_), on(false), stream(s) {}
StageTimer::~StageTimer() {}

int main(int argc, char** argv) {
    setenv("TBK_BAND_SPLIT", "0", 1);  // (this driver has no workspace allocator: the one-launch kernels only)
    const int n = argc > 1 ? atoi(argv[1]) : 256;
    const int nk = argc > 2 ? atoi(argv[2]) : 4096;
    std::vector<double> h((size_t)n * n * 2);
    srand(1);
    for (auto& x : h) x = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < n; ++i) h[((size_t)i * n + i) * 2 + 1] = 0.0;
    double *d_H0, *d_H, *d_de;
    void *d_vw, *d_band;
    hipMalloc(&d_H0, (size_t)nk * n * n * 16);
    hipMalloc(&d_H, (size_t)nk * n * n * 16);
    hipMalloc(&d_de, (size_t)nk * n * 16);
    hipMalloc(&d_vw, (size_t)nk * tbk_band_scratch_per_matrix(n));
    hipMalloc(&d_band, (size_t)nk * tbk_band_bytes_per_matrix(n));
    for (int k = 0; k < nk; ++k) hipMemcpy(d_H0 + (size_t)k * n * n * 2, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    tbk_model m;
    m.n_orb = n;
    unsigned long long zero[32] = {0};
    for (int rep = 0; rep < 3; ++rep) {
        hipMemcpy(d_H, d_H0, (size_t)nk * n * n * 16, hipMemcpyDeviceToDevice);
#ifdef TBK_PHASE_CLOCK
        hipMemcpyToSymbol(HIP_SYMBOL(tbk_band_clock), zero, sizeof(zero));
#endif
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipDeviceSynchronize();
        hipEventRecord(a, nullptr);
        if (tbk_band_fused(n)) {
            tbk_launch_band_reduce(&m, nullptr, d_H, nk, d_vw, nullptr, d_de);
        } else {
            tbk_launch_band_reduce(&m, nullptr, d_H, nk, d_vw, d_band);
            tbk_launch_band_chase(&m, nullptr, d_band, nk, d_de);
        }
        hipEventRecord(b, nullptr);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        unsigned long long clk[32] = {0};
#ifdef TBK_PHASE_CLOCK
        hipMemcpyFromSymbol(clk, HIP_SYMBOL(tbk_band_clock), sizeof(clk));
#endif
        printf("n=%d nk=%d  both kernels %.2f ms  (%.3f us per matrix)\n", n, nk, ms, ms * 1e3 / nk);
#ifndef TBK_PHASE_CLOCK
        (void)zero;
        continue;
#endif
        if (rep == 2) {
            const char* name[16] = {"0 look-ahead", "1 panel QR", "2 T factor", "3 hand-over", "4 big pass (rest)", "5 W", "6 loop top",
                                    "7 pass: request + wait", "8 pass: update + store", "9 pass: transposition",
                                    "10 pass: products + X", "11 pass: barrier", "12 QR: products", "13 QR: wave sums", "14 QR: barrier",
                                    "15 QR: reflector + update"};
            double total = 0;
            for (int k = 0; k <= 15; ++k) total += (double)clk[k];
            for (int k = 0; k <= 15; ++k)
                if (name[k][0]) printf("   %-28s %12.0f cycles  %5.1f %%\n", name[k], (double)clk[k], 100.0 * clk[k] / total);
            printf("   total %.0f\n", total);
        }
    }
    return 0;
}
