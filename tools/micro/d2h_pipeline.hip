// The download pipeline of rtc_render in isolation: device canvas -> pinned staging chunks -> pageable destination,
// DMA of chunk k+1 overlapped with the host copy of chunk k.  Sweeps chunk size, number of staging buffers and copy threads.
// Build: hipcc -O2 --offload-arch=gfx950 -o d2h_pipeline d2h_pipeline.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void copy_mt(char *dst, const char *src, size_t n, int threads) {
    std::vector<std::thread> pool;
    for (int k = 1; k < threads; k++)
        pool.emplace_back([=] {
            const size_t a = n * k / threads, b = n * (k + 1) / threads;
            std::memcpy(dst + a, src + a, b - a);
        });
    std::memcpy(dst, src, n / threads);
    for (auto &th : pool) th.join();
}

int main() {
    const size_t bytes = (size_t)3840 * 2160 * 24;
    char *d = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess) return 1;
    (void)hipMemset(d, 1, bytes);
    std::vector<char> dst(bytes);
    std::memset(dst.data(), 0, bytes);
    hipStream_t st;
    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (size_t chunk_mb : {4, 8, 16, 32}) {
        for (int nbuf : {2, 3, 4}) {
            for (int threads : {1, 2}) {
                const size_t chunk = chunk_mb << 20;
                std::vector<char *> stage(nbuf);
                std::vector<hipEvent_t> ev(nbuf);
                for (int i = 0; i < nbuf; i++) {
                    (void)hipHostMalloc(&stage[i], chunk);
                    (void)hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
                }
                const size_t n_chunks = (bytes + chunk - 1) / chunk;
                double best = 1e9;
                for (int rep = 0; rep < 3; rep++) {
                    const double t = now();
                    auto issue = [&](size_t k) {
                        const size_t off = k * chunk, n = std::min(chunk, bytes - off);
                        (void)hipMemcpyAsync(stage[k % nbuf], d + off, n, hipMemcpyDeviceToHost, st);
                        (void)hipEventRecord(ev[k % nbuf], st);
                    };
                    for (size_t k = 0; k < std::min<size_t>(nbuf - 1, n_chunks); k++) issue(k);
                    for (size_t k = 0; k < n_chunks; k++) {
                        if (k + nbuf - 1 < n_chunks) issue(k + nbuf - 1);
                        (void)hipEventSynchronize(ev[k % nbuf]);
                        const size_t off = k * chunk;
                        copy_mt(dst.data() + off, stage[k % nbuf], std::min(chunk, bytes - off), threads);
                    }
                    best = std::min(best, now() - t);
                }
                printf("chunk %2zu MB, %d buffers, %d copy thread(s): %6.2f ms  %5.1f GB/s\n", chunk_mb, nbuf, threads,
                       best * 1e3, bytes / best / 1e9);
                for (int i = 0; i < nbuf; i++) (void)hipHostFree(stage[i]), (void)hipEventDestroy(ev[i]);
            }
        }
    }
    return 0;
}
