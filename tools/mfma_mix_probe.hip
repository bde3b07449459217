// probe: how much of the v_mfma_f64_4x4x4 rate survives the instruction mix of K2's inner loop
// (per 15 MFMAs: 5 LDS operand reads, 3 global operand reads, a dozen address ops), at 1 and 2 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>

constexpr int G = 5, NRT = 3;

template <bool LDS_A, bool GLB_B, int VALU>
__global__ __launch_bounds__(256) void mix(const double* __restrict__ B, double* out, int ksteps, int LD, int LDv, int reps)
{
    extern __shared__ double Lbuf[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 20 * LDv; i += 256) Lbuf[i] = 1e-3 * (i % 97);
    __syncthreads();
    const int li = lane & 15, lk = lane >> 4;
    double acc[G][NRT];
    for (int g = 0; g < G; ++g) for (int j = 0; j < NRT; ++j) acc[g][j] = 0;
    const double* ap4 = Lbuf + (size_t)(lane & 3) * LDv + lk;
    const double* bp = B + (size_t)lk * LD + li + wave * 48;
    unsigned junk = tid;
    for (int r = 0; r < reps; ++r) {
        double a0[G], a1[G], b0[NRT], b1[NRT];
        for (int g = 0; g < G; ++g) a0[g] = LDS_A ? ap4[(size_t)(4 * g) * LDv] : 1.0 + g;
        for (int j = 0; j < NRT; ++j) b0[j] = GLB_B ? bp[j * 16] : 2.0 + j;
        for (int ks = 0; ks + 2 <= ksteps; ks += 2) {
            const double* bp1 = bp + (size_t)(ks + 1) * 4 * LD;
            const double* ap1 = ap4 + (ks + 1) * 4;
#pragma unroll
            for (int g = 0; g < G; ++g) a1[g] = LDS_A ? ap1[(size_t)(4 * g) * LDv] : a0[g];
#pragma unroll
            for (int j = 0; j < NRT; ++j) b1[j] = GLB_B ? bp1[j * 16] : b0[j];
#pragma unroll
            for (int v = 0; v < VALU; ++v) junk = junk * 1664525u + 1013904223u;
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int j = 0; j < NRT; ++j) acc[g][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a0[g], b0[j], acc[g][j], 0, 0, 0);
            const int kn = (ks + 2 < ksteps) ? ks + 2 : ksteps - 1;
            const double* bp2 = bp + (size_t)kn * 4 * LD;
            const double* ap2 = ap4 + kn * 4;
#pragma unroll
            for (int g = 0; g < G; ++g) a0[g] = LDS_A ? ap2[(size_t)(4 * g) * LDv] : a1[g];
#pragma unroll
            for (int j = 0; j < NRT; ++j) b0[j] = GLB_B ? bp2[j * 16] : b1[j];
#pragma unroll
            for (int v = 0; v < VALU; ++v) junk = junk * 1664525u + 1013904223u;
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int j = 0; j < NRT; ++j) acc[g][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1[g], b1[j], acc[g][j], 0, 0, 0);
        }
    }
    double s = junk;
    for (int g = 0; g < G; ++g) for (int j = 0; j < NRT; ++j) s += acc[g][j];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

int main(int argc, char** argv)
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const int ksteps = 38, LD = 176, reps = 400;
    const int LDv = argc > 1 ? atoi(argv[1]) : 162;
    printf("LDv = %d\n", LDv);
    double* B; hipMalloc(&B, (size_t)160 * LD * 8 * 2); hipMemset(B, 0, (size_t)160 * LD * 8 * 2);
    double* out; hipMalloc(&out, (size_t)cus * 4 * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = 20 * LDv * 8;
    auto run = [&](const char* name, auto kern, int bpc) {
        dim3 g(cus * bpc), b(256);
        hipLaunchKernelGGL(kern, g, b, lds, 0, B, out, ksteps, LD, LDv, 4); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(kern, g, b, lds, 0, B, out, ksteps, LD, LDv, reps); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = 512.0 * 15 * ksteps * reps * (double)g.x * 4;  // per wave: 15 MFMAs x 512 flop per k-step
        printf("%-34s %d waves/SIMD: %8.3f ms  %6.2f TFLOP/s\n", name, bpc, ms, fl / ms / 1e9);
    };
    for (int bpc : {1, 2, 3}) {
        run("registers only", mix<false, false, 0>, bpc);
        run("+ 12 int VALU / 15 MFMA", mix<false, false, 12>, bpc);
        run("A from LDS (5 ds_read_b64)", mix<true, false, 0>, bpc);
        run("B from global (3 dwordx2)", mix<false, true, 0>, bpc);
        run("A LDS + B global", mix<true, true, 0>, bpc);
        run("A LDS + B global + 12 VALU", mix<true, true, 12>, bpc);
    }
    return 0;
}
