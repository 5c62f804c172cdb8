// xfer_probe.hip -- what the host-pointer boundary can reach on this box: PCIe rates of pageable / pinned / registered copies and the host's own
// memcpy rate with 1 .. 16 threads (the numbers behind csrc/host_xfer.hip's choices).   hipcc -O2 tools/xfer_probe.hip -o tools/xfer_probe -lpthread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <chrono>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x)                                                                 \
    do {                                                                      \
        hipError_t e = (x);                                                   \
        if (e != hipSuccess) {                                                \
            printf("%s failed: %s\n", #x, hipGetErrorString(e));              \
            return 1;                                                         \
        }                                                                     \
    } while (0)

static void par_memcpy(char* dst, const char* src, size_t n, int threads) {
    std::vector<std::thread> th;
    const size_t per = (n / threads + 4095) & ~size_t(4095);
    for (int t = 0; t < threads; ++t) {
        const size_t a = std::min(n, per * t), b = std::min(n, per * (t + 1));
        if (b > a) th.emplace_back([=] { memcpy(dst + a, src + a, b - a); });
    }
    for (auto& x : th) x.join();
}

int main() {
    const size_t n = 44ull << 20;
    char *pageable = (char*)malloc(n), *pageable2 = (char*)malloc(n), *pinned = nullptr, *dev = nullptr;
    memset(pageable, 1, n), memset(pageable2, 2, n);
    CK(hipHostMalloc((void**)&pinned, n, hipHostMallocDefault));
    CK(hipMalloc((void**)&dev, n));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    printf("host threads available: %u\n", std::thread::hardware_concurrency());
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        CK(hipMemcpyAsync(dev, pageable, n, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        double t1 = now();
        CK(hipMemcpyAsync(pageable2, dev, n, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        double t2 = now();
        CK(hipMemcpyAsync(dev, pinned, n, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        double t3 = now();
        CK(hipMemcpyAsync(pinned, dev, n, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        double t4 = now();
        CK(hipHostRegister(pageable, n, hipHostRegisterDefault));
        double t5 = now();
        CK(hipMemcpyAsync(dev, pageable, n, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        double t6 = now();
        CK(hipHostUnregister(pageable));
        double t7 = now();
        printf("44 MiB: pageable H2D %.1f GB/s, D2H %.1f GB/s | pinned H2D %.1f, D2H %.1f | register %.2f ms + copy %.1f GB/s + unregister %.2f ms\n", n / (t1 - t0) / 1e9,
               n / (t2 - t1) / 1e9, n / (t3 - t2) / 1e9, n / (t4 - t3) / 1e9, (t5 - t4) * 1e3, n / (t6 - t5) / 1e9, (t7 - t6) * 1e3);
    }
    for (int threads : {1, 2, 4, 8, 16}) {
        par_memcpy(pinned, pageable, n, threads);
        double t0 = now();
        for (int r = 0; r < 4; ++r) par_memcpy(pinned, pageable, n, threads);
        double t1 = now();
        for (int r = 0; r < 4; ++r) par_memcpy(pageable2, pinned, n, threads);
        double t2 = now();
        printf("memcpy with %2d threads (thread start included): pageable -> pinned %.1f GB/s, pinned -> pageable %.1f GB/s\n", threads, 4 * n / (t1 - t0) / 1e9, 4 * n / (t2 - t1) / 1e9);
    }
    // chunked pinned copies, enqueued back to back: what one command costs beyond its bytes (chunk size; an event recorded behind every chunk;
    // chunks alternating over two streams)
    hipStream_t s2;
    CK(hipStreamCreate(&s2));
    hipEvent_t ev[64];
    for (auto& evk : ev) CK(hipEventCreateWithFlags(&evk, hipEventDisableTiming));
    for (int dir = 0; dir < 2; ++dir)
        for (size_t chunk : {size_t(1) << 20, size_t(2) << 20, size_t(4) << 20, size_t(8) << 20, size_t(16) << 20})
            for (int variant = 0; variant < 4; ++variant) {  // 0: one stream; 1: + event per chunk; 2: two streams; 3: two streams + events
                const bool events = variant & 1, two = variant & 2;
                double best = 1e9;
                for (int rep = 0; rep < 4; ++rep) {
                    double t0 = now();
                    int k = 0;
                    for (size_t off = 0; off < n; off += chunk, ++k) {
                        const size_t len = std::min(chunk, n - off);
                        hipStream_t st = two && (k & 1) ? s2 : s;
                        if (dir == 0)
                            CK(hipMemcpyAsync(dev + off, pinned + off, len, hipMemcpyHostToDevice, st));
                        else
                            CK(hipMemcpyAsync(pinned + off, dev + off, len, hipMemcpyDeviceToHost, st));
                        if (events) CK(hipEventRecord(ev[k & 63], st));
                    }
                    CK(hipStreamSynchronize(s));
                    CK(hipStreamSynchronize(s2));
                    best = std::min(best, now() - t0);
                }
                printf("%s 44 MiB in %2zu MiB chunks, %s%s: %.1f GB/s (%.0f us)\n", dir ? "D2H" : "H2D", chunk >> 20, two ? "two streams" : "one stream", events ? " + event per chunk" : "",
                       n / best / 1e9, best * 1e6);
            }
    // small-copy latency
    double t0 = now();
    for (int r = 0; r < 100; ++r) {
        CK(hipMemcpyAsync(dev, pinned, 4096, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
    }
    printf("4 KiB pinned H2D + sync: %.1f us\n", (now() - t0) / 100 * 1e6);
    return 0;
}
