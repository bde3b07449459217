// mfma_f64_probe.hip -- one-off hardware probe for the K2 design: layout and issue rate of
// v_mfma_f64_16x16x4_f64 on gfx950, next to the v_fma_f64 vector rate.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_probe mfma_f64_probe.hip && ./mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void layout_probe(const double* A /*16x4 row-major*/, const double* B /*4x16 row-major*/,
                             double* D /*16x16 row-major*/)
{
    const int l = threadIdx.x;
    // A operand: lane l holds A[l & 15][l >> 4]; B operand: lane l holds B[l >> 4][l & 15]
    const double a = A[(l & 15) * 4 + (l >> 4)];
    const double b = B[(l >> 4) * 16 + (l & 15)];
    double4_t c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    // C/D: lane l, reg r: row = (l >> 4) + 4 r, col = l & 15
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma_rate(double* out, int iters)
{
    double4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = double4_t{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + blockIdx.x * 1e-6;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void fma_rate(double* out, int iters)
{
    double acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = i;
    double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9 * blockIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = fma(acc[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main()
{
    // layout check with asymmetric data
    std::vector<double> A(64), B(64), D(256), R(256, 0.0);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = 1 + i * 0.5 + k * 7;
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = 3 + k * 1.25 - j * 0.0625 * (k + 1);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dD;
    CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dD, 256 * 8));
    CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(layout_probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    CK(hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int i = 0; i < 256; ++i) worst = fmax(worst, fabs(D[i] - R[i]));
    printf("layout check: max |D - A*B| = %g  (%s)\n", worst, worst < 1e-9 ? "OK" : "MISMATCH");

    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", p.gcnArchName, cus, p.clockRate);
    double* out; CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    auto run = [&](auto kern, int blocks_per_cu, double flops_per_thread_iter, const char* name) {
        dim3 g(cus * blocks_per_cu), b(256);
        hipLaunchKernelGGL(kern, g, b, 0, 0, out, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, g, b, 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double fl = flops_per_thread_iter * iters * (double)g.x * 256;
        printf("%-28s blocks/CU %d: %.3f ms  %.2f TFLOP/s\n", name, blocks_per_cu, ms, fl / ms / 1e9);
    };
    // one MFMA 16x16x4 = 2048 flops per wave = 32 per lane
    for (int bpc : {1, 2, 4}) {
        run(mfma_rate<1>, bpc, 32.0 * 1, "mfma_f64 16x16x4, 1 acc");
        run(mfma_rate<2>, bpc, 32.0 * 2, "mfma_f64 16x16x4, 2 acc");
        run(mfma_rate<4>, bpc, 32.0 * 4, "mfma_f64 16x16x4, 4 acc");
        run(mfma_rate<8>, bpc, 32.0 * 8, "mfma_f64 16x16x4, 8 acc");
        run(fma_rate, bpc, 2.0 * 16, "v_fma_f64, 16 chains");
    }
    return 0;
}
