// probe.hip -- measured ceilings of THIS chip for bench.py's roofline (measurement infrastructure, not
// part of the likelihood path; built into its own library, cafe_amd/lib/libcafeprobe.so):
//   cafeprobe_hbm_triad       a[i] = b[i] + s * c[i] over arrays far larger than the 256 MiB Infinity Cache
//   cafeprobe_hbm_copy        a[i] = b[i]  (the guide's 6.29 TB/s figure is a float4 copy)
//   cafeprobe_mfma_f64        register-only issue rate of v_mfma_f64_4x4x4_4b (shape 4) / v_mfma_f64_16x16x4 (shape 16)
// All times are HIP-event times of one launch after a warm-up launch.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

namespace {

typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void triad(d2* __restrict__ a, const d2* __restrict__ b, const d2* __restrict__ c,
                                             double s, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) a[i] = b[i] + s * c[i];
}

__global__ __launch_bounds__(256) void copy16(d2* __restrict__ a, const d2* __restrict__ b, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) a[i] = b[i];
}

// NACC independent accumulators per wave, distinct A/B operand registers per accumulator
template <int NACC>
__global__ __launch_bounds__(512) void mfma4_rate(double* out, int iters)
{
    double acc[NACC], a[NACC], b[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        acc[i] = 0.0;
        a[i] = 1e-3 * (threadIdx.x + i);
        b[i] = 1.0 + 1e-6 * (blockIdx.x + i);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b[i], acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(512) void mfma16_rate(double* out, int iters)
{
    d4 acc[NACC];
    double a[NACC], b[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        acc[i] = d4{0, 0, 0, 0};
        a[i] = 1e-3 * (threadIdx.x + i);
        b[i] = 1.0 + 1e-6 * (blockIdx.x + i);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[i], acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define PCK(x)                                                                         \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "cafeprobe: %s failed: %s\n", #x, hipGetErrorString(e_));  \
            return -1;                                                                 \
        }                                                                              \
    } while (0)

template <class F>
int timed(F&& launch, float* ms)
{
    hipEvent_t e0, e1;
    PCK(hipEventCreate(&e0));
    PCK(hipEventCreate(&e1));
    launch();  // warm-up
    PCK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        PCK(hipEventRecord(e0, 0));
        launch();
        PCK(hipEventRecord(e1, 0));
        PCK(hipEventSynchronize(e1));
        float t = 0;
        PCK(hipEventElapsedTime(&t, e0, e1));
        if (t < best) best = t;
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    *ms = best;
    return 0;
}

}  // namespace

extern "C" {

// bytes_per_array: size of each of the three arrays (>= 1 GiB recommended).  *gbs = 3 * bytes / time.
int cafeprobe_hbm_triad(int device, size_t bytes_per_array, double* gbs)
{
    PCK(hipSetDevice(device));
    const size_t n = bytes_per_array / sizeof(d2);
    d2 *a = nullptr, *b = nullptr, *c = nullptr;
    PCK(hipMalloc(&a, n * sizeof(d2)));
    PCK(hipMalloc(&b, n * sizeof(d2)));
    PCK(hipMalloc(&c, n * sizeof(d2)));
    PCK(hipMemset(b, 0, n * sizeof(d2)));
    PCK(hipMemset(c, 0, n * sizeof(d2)));
    float ms = 0;
    const int rc = timed([&] { hipLaunchKernelGGL(triad, dim3(256 * 16), dim3(256), 0, 0, a, b, c, 3.0, n); }, &ms);
    hipFree(a);
    hipFree(b);
    hipFree(c);
    if (rc) return rc;
    *gbs = 3.0 * (double)(n * sizeof(d2)) / (ms * 1e-3) / 1e9;
    return 0;
}

int cafeprobe_hbm_copy(int device, size_t bytes_per_array, double* gbs)
{
    PCK(hipSetDevice(device));
    const size_t n = bytes_per_array / sizeof(d2);
    d2 *a = nullptr, *b = nullptr;
    PCK(hipMalloc(&a, n * sizeof(d2)));
    PCK(hipMalloc(&b, n * sizeof(d2)));
    PCK(hipMemset(b, 0, n * sizeof(d2)));
    float ms = 0;
    const int rc = timed([&] { hipLaunchKernelGGL(copy16, dim3(256 * 16), dim3(256), 0, 0, a, b, n); }, &ms);
    hipFree(a);
    hipFree(b);
    if (rc) return rc;
    *gbs = 2.0 * (double)(n * sizeof(d2)) / (ms * 1e-3) / 1e9;
    return 0;
}

// shape = 4: v_mfma_f64_4x4x4_4b (512 flop per wave-instruction); shape = 16: v_mfma_f64_16x16x4 (2048).
// 512-thread workgroups (2 waves per SIMD), `wg_per_cu` of them per CU, 12 (shape 4) / 8 (shape 16)
// independent accumulators per wave.
int cafeprobe_mfma_f64(int device, int shape, int wg_per_cu, double* tflops)
{
    PCK(hipSetDevice(device));
    hipDeviceProp_t p;
    PCK(hipGetDeviceProperties(&p, device));
    const int grid = p.multiProcessorCount * wg_per_cu;
    double* out = nullptr;
    PCK(hipMalloc(&out, (size_t)grid * 512 * sizeof(double)));
    const int iters = 20000;
    float ms = 0;
    int rc;
    double flop_per_wave_iter;
    if (shape == 4) {
        rc = timed([&] { hipLaunchKernelGGL(mfma4_rate<12>, dim3(grid), dim3(512), 0, 0, out, iters); }, &ms);
        flop_per_wave_iter = 12 * 512.0;
    } else {
        rc = timed([&] { hipLaunchKernelGGL(mfma16_rate<8>, dim3(grid), dim3(512), 0, 0, out, iters); }, &ms);
        flop_per_wave_iter = 8 * 2048.0;
    }
    hipFree(out);
    if (rc) return rc;
    *tflops = flop_per_wave_iter * iters * (double)grid * 8 / (ms * 1e-3) / 1e12;
    return 0;
}

}  // extern "C"
