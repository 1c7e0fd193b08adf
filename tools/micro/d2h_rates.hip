// What bounds rtc_render's download of a 4K canvas (199 MB): the DMA into pinned memory, or the copy out of the staging
// buffer into the caller's pageable buffer?  Build: hipcc -O2 --offload-arch=gfx950 -o d2h_rates d2h_rates.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t bytes = (size_t)3840 * 2160 * 24, chunk = (size_t)16 << 20;
    char *d = nullptr, *pinned = nullptr, *stage = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess || hipHostMalloc(&pinned, bytes) != hipSuccess ||
        hipHostMalloc(&stage, chunk) != hipSuccess)
        return 1;
    (void)hipMemset(d, 1, bytes);
    std::vector<char> pageable(bytes);
    for (int rep = 0; rep < 2; rep++) {
        double t = now();
        (void)hipMemcpy(pinned, d, bytes, hipMemcpyDeviceToHost);
        printf("D2H into pinned memory            %6.1f GB/s\n", bytes / (now() - t) / 1e9);
        t = now();
        (void)hipMemcpy(pageable.data(), d, bytes, hipMemcpyDeviceToHost);
        printf("D2H into pageable memory (driver) %6.1f GB/s\n", bytes / (now() - t) / 1e9);
        for (int threads : {1, 2, 4, 8}) {
            t = now();
            for (size_t off = 0; off < bytes; off += chunk) {
                const size_t n = std::min(chunk, bytes - off);
                std::vector<std::thread> pool;
                for (int k = 1; k < threads; k++)
                    pool.emplace_back([&, k] {
                        const size_t a = n * k / threads, b = n * (k + 1) / threads;
                        std::memcpy(pageable.data() + off + a, stage + a, b - a);
                    });
                std::memcpy(pageable.data() + off, stage, n / threads);
                for (auto &th : pool) th.join();
            }
            printf("staging -> pageable, %d thread(s)   %6.1f GB/s\n", threads, bytes / (now() - t) / 1e9);
        }
    }
    return 0;
}
