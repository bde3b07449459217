// simd_map_probe.hip -- which SIMD of its CU does wave w of a workgroup run on?  (HW_REG_HW_ID: wave slot [3:0], SIMD [5:4],
// CU [11:8], ...).  The family walk deals its row tiles 3,3,2,2 to the four wave rows and assumes that waves w and w + 4 of an
// 8-wave workgroup share a SIMD (3 + 2 tiles each); this prints the placement the hardware really makes.
//   hipcc --offload-arch=gfx950 -O2 -o tools/simd_map_probe tools/simd_map_probe.hip && tools/simd_map_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <string>

__global__ void probe(unsigned* out, int lds_dummy)
{
    extern __shared__ double smem[];
    if (lds_dummy < 0) smem[threadIdx.x] = 0;
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = hw;
    // stay resident a little so that the workgroups of one launch do not reuse each other's slots
    const long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 20000) {}
}

int main()
{
    for (int threads : {512, 640, 256}) {
        for (size_t lds : {(size_t)54544, (size_t)8704}) {
            const int grid = 250, waves = threads / 64;
            unsigned* d;
            hipMalloc(&d, grid * 16 * sizeof(unsigned));
            hipMemset(d, 0, grid * 16 * sizeof(unsigned));
            hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            hipLaunchKernelGGL(probe, dim3(grid), dim3(threads), lds, 0, d, 0);
            hipDeviceSynchronize();
            std::vector<unsigned> h(grid * 16);
            hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
            std::map<std::string, int> patterns;
            for (int b = 0; b < grid; ++b) {
                std::string s;
                for (int w = 0; w < waves; ++w) s += char('0' + ((h[b * 16 + w] >> 4) & 3));
                patterns[s]++;
            }
            printf("threads %d lds %zu: SIMD of wave 0..%d -> count of workgroups\n", threads, lds, waves - 1);
            for (auto& p : patterns) printf("   %s  x %d\n", p.first.c_str(), p.second);
            hipFree(d);
        }
    }
    return 0;
}
