// lds_dma_probe.hip -- semantics of global_load_lds_dwordx4 on gfx950 as k2c_panel (cafe_amd/csrc/k2c_panel.hpp) relies on them:
// lane l's 16 bytes land at LDS byte address M0 + 16 l (M0 = wave-uniform byte address, any 16-byte aligned value up to 160 KB),
// the source address is per lane, completion is counted by vmcnt in issue order with ordinary loads.
//   hipcc --offload-arch=gfx950 -O2 -o tools/lds_dma_probe tools/lds_dma_probe.hip && tools/lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__global__ void probe(const double* src, double* out, int lds_off_bytes)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // every wave copies 64 pieces of 16 bytes, source piece index permuted (lane -> 63 - lane), to its own 1 KB of LDS
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + lds_off_bytes + __builtin_amdgcn_readfirstlane(wave) * 1024;
    dma16(src + 2 * ((size_t)wave * 64 + (63 - lane)), base);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const double* p = lds + lds_off_bytes / 8 + wave * 128;
    out[(size_t)threadIdx.x * 2] = p[2 * lane];
    out[(size_t)threadIdx.x * 2 + 1] = p[2 * lane + 1];
}

int main()
{
    const int T = 256;
    std::vector<double> h(2 * T);
    for (int i = 0; i < 2 * T; ++i) h[i] = 1000.0 + i;
    double *d_src, *d_out;
    hipMalloc(&d_src, h.size() * 8);
    hipMalloc(&d_out, h.size() * 8);
    hipMemcpy(d_src, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    int bad_total = 0;
    for (int off : {0, 1024, 65536, 140 * 1024}) {
        hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(probe, dim3(1), dim3(T), off + 4096, 0, d_src, d_out, off);
        std::vector<double> o(2 * T);
        hipMemcpy(o.data(), d_out, o.size() * 8, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int t = 0; t < T; ++t) {
            const int w = t / 64, l = t % 64;
            const double want0 = 1000.0 + 2 * (w * 64 + (63 - l));
            if (o[2 * t] != want0 || o[2 * t + 1] != want0 + 1) ++bad;
        }
        printf("LDS offset %6d: %d of %d lanes wrong%s\n", off, bad, T, bad ? "" : "  (lane l's 16 bytes at M0 + 16 l, source per lane: as assumed)");
        bad_total += bad;
    }
    return bad_total != 0;
}
