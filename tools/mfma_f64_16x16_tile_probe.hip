// probe: v_mfma_f64_16x16x4 rate with the operand structure of K2's 16x16 kernel (NFT x NRT independent
// accumulator tiles fed from NFT + NRT distinct operand registers), registers only
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NFT, int NRT>
__global__ __launch_bounds__(256) void rate(double* out, int iters)
{
    d4 acc[NFT][NRT];
    for (int i = 0; i < NFT; ++i) for (int j = 0; j < NRT; ++j) acc[i][j] = d4{0, 0, 0, 0};
    double a[NFT], b[NRT];
    for (int i = 0; i < NFT; ++i) a[i] = threadIdx.x * 1e-3 + i;
    for (int j = 0; j < NRT; ++j) b[j] = 1.0 + blockIdx.x * 1e-6 + j;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NFT; ++i)
#pragma unroll
            for (int j = 0; j < NRT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NFT; ++i) a[i] += 1e-9;   // operands change every k-step, as in the kernel
#pragma unroll
        for (int j = 0; j < NRT; ++j) b[j] += 1e-9;
    }
    double s = 0;
    for (int i = 0; i < NFT; ++i) for (int j = 0; j < NRT; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    double* out; hipMalloc(&out, (size_t)cus * 4 * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    auto run = [&](const char* name, auto kern, int bpc, int tiles) {
        dim3 g(cus * bpc), b(256);
        hipLaunchKernelGGL(kern, g, b, 0, 0, out, 100); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(kern, g, b, 0, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = 2048.0 * tiles * iters * (double)g.x * 4;
        printf("16x16x4 %-10s %d waves/SIMD: %8.3f ms  %6.2f TFLOP/s\n", name, bpc, ms, fl / ms / 1e9);
    };
    for (int bpc : {1, 2}) {
        run("1x4 tiles", rate<1, 4>, bpc, 4);
        run("2x4 tiles", rate<2, 4>, bpc, 8);
        run("1x8 tiles", rate<1, 8>, bpc, 8);
    }
    return 0;
}
