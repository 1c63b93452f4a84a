// How fast do 4 GB of pageable host memory reach the device?  (engine creation at config 4 uploads 500 dispersion matrices of 8 MB)
//   hipcc --offload-arch=gfx950 -O2 -o h2d_probe h2d_probe.hip -lpthread && ./h2d_probe
// Prints GB/s for: plain hipMemcpy from pageable memory; hipHostRegister + one async copy (+ the registration's own time);
// threaded staging through two pinned buffers (the copy of chunk c + 1 into the staging buffer overlaps the DMA of chunk c).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void par_memcpy(char *dst, const char *src, size_t n, int threads) {
    std::vector<std::thread> th;
    const size_t per = (n + threads - 1) / threads;
    for (int t = 0; t < threads; t++) {
        const size_t o = (size_t)t * per;
        if (o >= n) break;
        const size_t len = std::min(per, n - o);
        th.emplace_back([=] { memcpy(dst + o, src + o, len); });
    }
    for (auto &x : th) x.join();
}

int main() {
    const size_t total = (size_t)4 << 30, chunk = (size_t)256 << 20;
    char *h = (char *)malloc(total);
    for (size_t i = 0; i < total; i += 4096) h[i] = (char)i;       // touch
    char *d = nullptr;
    if (hipMalloc(&d, total) != hipSuccess) return 1;
    hipStream_t s;
    hipStreamCreate(&s);
    for (int rep = 0; rep < 2; rep++) {
        double t = now();
        hipMemcpy(d, h, total, hipMemcpyHostToDevice);
        printf("pageable hipMemcpy: %.2f GB/s (%.3f s)\n", total / (now() - t) / 1e9, now() - t);
    }
    {
        double t = now();
        hipError_t e = hipHostRegister(h, total, hipHostRegisterDefault);
        double tr = now() - t;
        if (e == hipSuccess) {
            t = now();
            hipMemcpyAsync(d, h, total, hipMemcpyHostToDevice, s);
            hipStreamSynchronize(s);
            double tc = now() - t;
            t = now();
            hipHostUnregister(h);
            printf("hipHostRegister %.3f s + copy %.3f s (%.2f GB/s) + unregister %.3f s -> %.2f GB/s overall\n", tr, tc, total / tc / 1e9, now() - t,
                   total / (tr + tc) / 1e9);
        } else
            printf("hipHostRegister failed: %s\n", hipGetErrorString(e));
    }
    for (int threads : {1, 2, 4, 8}) {
        char *p[2];
        hipHostMalloc((void **)&p[0], chunk, hipHostMallocDefault);
        hipHostMalloc((void **)&p[1], chunk, hipHostMallocDefault);
        hipEvent_t ev[2];
        hipEventCreate(&ev[0]); hipEventCreate(&ev[1]);
        double t = now();
        int c = 0;
        for (size_t o = 0; o < total; o += chunk, c++) {
            const int b = c & 1;
            if (c >= 2) hipEventSynchronize(ev[b]);
            par_memcpy(p[b], h + o, chunk, threads);
            hipMemcpyAsync(d + o, p[b], chunk, hipMemcpyHostToDevice, s);
            hipEventRecord(ev[b], s);
        }
        hipStreamSynchronize(s);
        printf("staged through 2 x %zu MB pinned, %d copy thread(s): %.2f GB/s (%.3f s)\n", chunk >> 20, threads, total / (now() - t) / 1e9, now() - t);
        hipHostFree(p[0]); hipHostFree(p[1]);
    }
    return 0;
}
