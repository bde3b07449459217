// probe: v_mfma_f64_4x4x4_4b_f64 lane layout and issue rate on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
__global__ void layout(const double* a_in, const double* b_in, double* d_out) {
    int l = threadIdx.x;
    double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a_in[l], b_in[l], 0.0, 0, 0, 0);
    d_out[l] = d;
}
template <int NACC> __global__ __launch_bounds__(256) void rate(double* out, int iters) {
    double acc[NACC]; for (int i = 0; i < NACC; ++i) acc[i] = 0;
    double a = threadIdx.x * 1e-3, b = 1.0 + blockIdx.x * 1e-6;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    double s = 0; for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    // layout discovery: set A = one-hot at lane la, B = one-hot at lane lb; see which output lanes light up
    double *dA, *dB, *dD; hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 512);
    std::vector<double> A(64), B(64), D(64);
    // give every A lane value (1000 + l) and B one-hot to decode which A lanes pair with a B lane, etc.
    printf("B one-hot at lane lb -> nonzero D lanes (value = A value that multiplied it):\n");
    for (int lb : {0, 1, 4, 5, 16, 17, 20, 37}) {
        for (int l = 0; l < 64; ++l) { A[l] = 1000 + l; B[l] = (l == lb) ? 1.0 : 0.0; }
        hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        hipMemcpy(D.data(), dD, 512, hipMemcpyDeviceToHost);
        printf(" lb=%2d:", lb);
        for (int l = 0; l < 64; ++l) if (D[l] != 0) printf(" D[%d]=A[%d]", l, (int)D[l] - 1000);
        printf("\n");
    }
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    double* out; hipMalloc(&out, (size_t)cus * 8 * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 40000;
    auto run = [&](auto kern, int bpc, int nacc) {
        dim3 g(cus * bpc), b(256);
        hipLaunchKernelGGL(kern, g, b, 0, 0, out, 100); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(kern, g, b, 0, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double fl = 512.0 / 64 * nacc * iters * (double)g.x * 256;  // 4 blocks x 4x4x4 x 2 flops per wave instruction
        printf("4x4x4_4b %d acc, %d blocks/CU: %.3f ms %.2f TFLOP/s\n", nacc, bpc, ms, fl / ms / 1e9);
    };
    for (int bpc : {1, 2, 4}) { run(rate<1>, bpc, 1); run(rate<4>, bpc, 4); run(rate<8>, bpc, 8); }
    return 0;
}
