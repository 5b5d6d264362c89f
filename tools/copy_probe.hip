#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
int main() {
    char* d; (void)hipMalloc(&d, 1 << 22);
    std::vector<char> h(1 << 22);
    char* p; (void)hipHostMalloc(&p, 1 << 22, hipHostMallocDefault);
    hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (size_t bytes : {512, 4096, 16384, 20000, 32768, 49152, 65535, 65536, 65537, 131072, 1 << 20}) {
        double best[3] = {1e9, 1e9, 1e9};
        for (int rep = 0; rep < 20; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            (void)hipMemcpy(h.data(), d, bytes, hipMemcpyDeviceToHost);
            auto t1 = std::chrono::steady_clock::now();
            (void)hipMemcpyAsync(p, d, bytes, hipMemcpyDeviceToHost, s);
            (void)hipStreamSynchronize(s);
            auto t2 = std::chrono::steady_clock::now();
            (void)hipMemcpyAsync(h.data(), d, bytes, hipMemcpyDeviceToHost, s);
            (void)hipStreamSynchronize(s);
            auto t3 = std::chrono::steady_clock::now();
            double a = std::chrono::duration<double, std::micro>(t1 - t0).count(), b = std::chrono::duration<double, std::micro>(t2 - t1).count(), c = std::chrono::duration<double, std::micro>(t3 - t2).count();
            if (a < best[0]) best[0] = a; if (b < best[1]) best[1] = b; if (c < best[2]) best[2] = c;
        }
        printf("%8zu B: hipMemcpy pageable %7.1f us | async pinned + sync %7.1f us | async pageable + sync %7.1f us\n", bytes, best[0], best[1], best[2]);
    }
}
