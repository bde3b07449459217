// alloc_probe -- what the set-up path pays per HIP memory call on this box (round 4):
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/alloc_probe tools/alloc_probe.hip && /tmp/alloc_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now_ms() { return 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipSetDevice(0);
    hipFree(nullptr);
    void* warm = nullptr;
    hipMalloc(&warm, 1 << 20);
    hipFree(warm);
    const size_t sizes[] = {4096, 65536, 1 << 20, 8 << 20, 32 << 20, 128 << 20, 512 << 20};
    for (size_t sz : sizes) {
        double t_alloc = 0, t_free = 0, t_set = 0;
        const int n = 8;
        std::vector<void*> p(n);
        for (int i = 0; i < n; ++i) {
            double t0 = now_ms();
            hipMalloc(&p[i], sz);
            t_alloc += now_ms() - t0;
        }
        {
            double t0 = now_ms();
            hipMemset(p[0], 0, sz);
            hipDeviceSynchronize();
            t_set = now_ms() - t0;
        }
        for (int i = 0; i < n; ++i) {
            double t0 = now_ms();
            hipFree(p[i]);
            t_free += now_ms() - t0;
        }
        printf("%10zu B: hipMalloc %.3f ms  hipFree %.3f ms  first memset+sync %.3f ms\n", sz, t_alloc / n, t_free / n, t_set);
    }
    // small synchronous copies from pageable memory
    void* d = nullptr;
    hipMalloc(&d, 8 << 20);
    std::vector<char> h(8 << 20, 1);
    for (size_t sz : {(size_t)256, (size_t)65536, (size_t)(1 << 20), (size_t)(8 << 20)}) {
        double t0 = now_ms();
        for (int i = 0; i < 10; ++i) hipMemcpy(d, h.data(), sz, hipMemcpyHostToDevice);
        printf("hipMemcpy H2D pageable %8zu B: %.3f ms\n", sz, (now_ms() - t0) / 10);
    }
    double t0 = now_ms();
    void* hp = nullptr;
    hipHostMalloc(&hp, 1 << 20, hipHostMallocMapped | hipHostMallocCoherent);
    printf("hipHostMalloc 1 MiB: %.3f ms\n", now_ms() - t0);
    t0 = now_ms();
    hipHostFree(hp);
    printf("hipHostFree: %.3f ms\n", now_ms() - t0);
    hipStream_t s;
    t0 = now_ms();
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    printf("hipStreamCreate: %.3f ms\n", now_ms() - t0);
    hipEvent_t e;
    t0 = now_ms();
    hipEventCreate(&e);
    printf("hipEventCreate: %.3f ms\n", now_ms() - t0);
    return 0;
}
