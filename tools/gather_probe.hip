// gather_probe.hip -- throwaway measurement (not part of the product): throughput of K2's leaf-column gathers at
// the cfg 2 shape under full-chip load, for a few access patterns.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_variants/probe/gather_probe tools/gather_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int LD = 176, KP = 152, NF = 40, G = 5, NLEAF = 16;

template <int MODE>
__global__ __launch_bounds__(512) void gather(const double* __restrict__ PT, const int* __restrict__ counts, double* out,
                                              unsigned long long* cyc, int reps)
{
    __shared__ int s_cnt[NF * NLEAF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int wf = wave & 1, wr = wave >> 1;
    const int rt0 = wr * 2 + min(wr, 2), ntile = 2 + (wr < 2 ? 1 : 0);   // 10 row tiles over 4 wave rows: 3,3,2,2
    for (int i = tid; i < NF * NLEAF; i += 512) s_cnt[i] = counts[(size_t)blockIdx.x * NF * NLEAF + i];
    __syncthreads();
    double acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        const int leaf = r % NLEAF;
        const double* M = PT + (size_t)leaf * KP * LD;
        if (MODE == 0 || MODE == 2) {
            double v[G][3];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int f = wf * 20 + 4 * g + lk;
                const int cnt = (MODE == 2) ? s_cnt[leaf] : s_cnt[f * NLEAF + leaf];
                const double* col = M + (size_t)cnt * LD + rt0 * 16 + li;
#pragma unroll
                for (int j = 0; j < 3; ++j) v[g][j] = (j < ntile) ? col[j * 16] : 0.0;
            }
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc += v[g][j];
        } else if (MODE == 1) {
            d2 v2[G];
            double v1[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int f = wf * 20 + 4 * g + lk;
                const int cnt = s_cnt[f * NLEAF + leaf];
                const double* col = M + (size_t)cnt * LD + rt0 * 16;
                v2[g] = *reinterpret_cast<const d2*>(col + 2 * li);
                v1[g] = (ntile > 2) ? col[32 + li] : 0.0;
            }
#pragma unroll
            for (int g = 0; g < G; ++g) acc += v2[g].x + v2[g].y + v1[g];
        } else if (MODE == 3) {   // one family per 16 lanes replaced by: one family per wave-load, 64 contiguous rows
            // lane l loads row l of family (g, k): 512 contiguous bytes per instruction; same bytes per wave overall
            double v[15];
#pragma unroll
            for (int q = 0; q < 15; ++q) {
                const int f = wf * 20 + (q * 4) / 3;   // ~20 families x 48 rows = 15 x 64
                const int cnt = s_cnt[(f % NF) * NLEAF + leaf];
                v[q] = M[(size_t)cnt * LD + (rt0 * 16 + lane + 64 * (q % 1)) % 150];
            }
#pragma unroll
            for (int q = 0; q < 15; ++q) acc += v[q];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[(size_t)blockIdx.x * 512 + tid] = acc;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 64;
    const int grid = argc > 2 ? atoi(argv[2]) : 250;
    double *PT, *out;
    int* cnt;
    unsigned long long* cyc;
    hipMalloc(&PT, sizeof(double) * NLEAF * KP * LD);
    hipMalloc(&out, sizeof(double) * grid * 512);
    hipMalloc(&cnt, sizeof(int) * grid * NF * NLEAF);
    hipMalloc(&cyc, sizeof(unsigned long long) * grid);
    std::vector<double> h(NLEAF * KP * LD, 1.0);
    hipMemcpy(PT, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    std::vector<int> hc((size_t)grid * NF * NLEAF);
    srand(7);
    for (auto& c : hc) {   // geometric-ish counts, mean ~20, as simulated family sizes are
        int k = 0;
        while (k < 100 && (rand() % 100) < 95) ++k;
        c = k;
    }
    hipMemcpy(cnt, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](int mode, const char* name) {
        for (int it = 0; it < 3; ++it) {
            hipEventRecord(e0);
            switch (mode) {
                case 0: gather<0><<<grid, 512>>>(PT, cnt, out, cyc, reps); break;
                case 1: gather<1><<<grid, 512>>>(PT, cnt, out, cyc, reps); break;
                case 2: gather<2><<<grid, 512>>>(PT, cnt, out, cyc, reps); break;
                case 3: gather<3><<<grid, 512>>>(PT, cnt, out, cyc, reps); break;
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> hcy(grid);
        hipMemcpy(hcy.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
        double m = 0;
        for (auto c : hcy) m += (double)c;
        m /= grid;
        const double bytes = 40.0 * 160 * 8;   // per workgroup per gather
        printf("%-34s %8.1f cycles per leaf gather  (%.1f B/clk/CU)  kernel %.3f ms\n", name, m / reps, bytes * reps / m, ms);
    };
    run(0, "dwordx2, 4 families x 16 rows");
    run(1, "dwordx4 row pairs + dwordx2");
    run(2, "dwordx2, one column for all");
    run(3, "dwordx2, 64 contiguous rows");
    return 0;
}
