// Cycles per phase of one Householder step of herm_tridiag_stream_kernel, measured on workgroup 0 while the whole
// grid runs (so the memory system is loaded as in production).  Build + run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DTBK_PHASE_CLOCK -Iinclude -Itbmodels_amd/csrc \
//         tools/stream_phase_clock.hip -o /tmp/spc -lrocblas && /tmp/spc 80 36864
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../tbmodels_amd/csrc/tbk_eig_stream.hip"

void tbk_set_error(const char*, ...) {}
int DevBuf::reserve(size_t) { return 0; }
void DevBuf::release() {}
StageTimer::StageTimer(tbk_model* m_, int, hipStream_t s) : m(m_), on(false), stream(s) {}
StageTimer::~StageTimer() {}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 80;
    const int nk = argc > 2 ? atoi(argv[2]) : 8192;
    std::vector<double> h((size_t)n * n * 2);
    srand(1);
    for (auto& x : h) x = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < n; ++i) h[((size_t)i * n + i) * 2 + 1] = 0.0;
    double *d_H, *d_de, *d_E;
    hipMalloc(&d_H, (size_t)nk * n * n * 16);
    hipMalloc(&d_de, (size_t)nk * n * 16);
    hipMalloc(&d_E, (size_t)nk * n * 8);
    for (int k = 0; k < nk; ++k) hipMemcpy(d_H + (size_t)k * n * n * 2, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    tbk_model m;
    m.n_orb = n;
    unsigned long long zero[16] = {0};
    for (int rep = 0; rep < 2; ++rep) {
        hipMemcpyToSymbol(HIP_SYMBOL(tbk_phase_clock), zero, sizeof(zero));
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipEventRecord(a, nullptr);
        tbk_launch_tridiag_stream(&m, nullptr, d_H, nk, d_de);
        hipEventRecord(b, nullptr);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        unsigned long long clk[16];
        hipMemcpyFromSymbol(clk, HIP_SYMBOL(tbk_phase_clock), sizeof(clk));
        const char* name[8] = {"", "1 column build", "2 reflector", "3 pass (plain steps)", "4 column sums", "5 panel corrections", "6 w'", "7 pass (flush steps)"};
        unsigned long long total = 0;
        for (int k = 1; k <= 7; ++k) total += clk[k];
        printf("n=%d nk=%d  kernel %.2f ms  (%.3f us per matrix)  workgroup 0: %.0f cycles per step\n", n, nk, ms, ms * 1e3 / nk,
               total / (double)(n - 1));
        for (int k = 1; k <= 7; ++k) printf("   %-22s %8.0f cycles per step  %5.1f %%\n", name[k], clk[k] / (double)(n - 1), 100.0 * clk[k] / total);
    }
    return 0;
}
