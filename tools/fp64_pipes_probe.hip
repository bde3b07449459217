// fp64_pipes_probe.hip -- do FP64 vector FMAs run BESIDE FP64 matrix instructions on gfx950?  (round 5)
//
// MI355X quotes the same 78.6 TFLOP/s for FP64 vector and FP64 matrix work.  The walk's products run on the matrix pipe at
// ~77 % of it; if v_fma_f64 from the same or another wave of the SIMD executed concurrently, a product could put part of its
// row tiles on the vector pipe and exceed the matrix peak.  This probe times, per SIMD:
//   M   only v_mfma_f64_16x16x4 (8 per iteration, 4 independent accumulators)           -> 8 x 64 cycles if the pipe is full
//   V   only v_fma_f64          (128 per iteration, 16 independent chains)              -> 128 x 4 cycles if the pipe is full
//   MV  both in one wave, interleaved (16 FMAs behind every MFMA)
//   M|V two waves per SIMD, one all-matrix, one all-vector (wave specialisation)
//   M4  only v_mfma_f64_4x4x4 (32 per iteration: the walk's 4-family shape)  and  M4V the same with the FMAs
// and prints cycles per iteration: overlap shows as MV ~ max(M, V), a shared unit as MV ~ M + V.
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/fp64_pipes_probe tools/fp64_pipes_probe.hip && tools/fp64_pipes_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));

// MODE bit 0: matrix work, bit 1: vector work; SPLIT: odd waves vector-only, even waves matrix-only; SHAPE4: 4x4x4 instructions
template <int MODE, bool SPLIT, bool SHAPE4>
__global__ __launch_bounds__(512) void pipes(double* out, long long* cycles, int reps, double seed)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool do_m = MODE & 1, do_v = MODE & 2;
    if (SPLIT) {
        do_m = (wave & 1) == 0;
        do_v = (wave & 1) == 1;
    }
    d4 acc[4];
    double accs[4];
    for (int i = 0; i < 4; ++i) {
        acc[i] = d4{0.0, 0.0, 0.0, 0.0};
        accs[i] = 0.0;
    }
    double f[16];
    for (int i = 0; i < 16; ++i) f[i] = seed * (i + 1) + lane * 1e-9;
    const double a = 1.0 + seed * lane, b = 1.0 - seed * lane, m = 1.0 - 1e-12, c = 1e-13;
    __syncthreads();
    const long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        if (do_m && do_v) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (SHAPE4) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) accs[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, accs[u], 0, 0, 0);
                } else {
                    acc[q & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q & 3], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) f[i] = __builtin_fma(f[i], m, c);
            }
        } else if (do_m) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (SHAPE4) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) accs[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, accs[u], 0, 0, 0);
                } else {
                    acc[q & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q & 3], 0, 0, 0);
                }
            }
        } else if (do_v) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int i = 0; i < 16; ++i) f[i] = __builtin_fma(f[i], m, c);
        }
    }
    const long long t1 = clock64();
    double s = 0.0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + accs[i];
    for (int i = 0; i < 16; ++i) s += f[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cycles[(size_t)blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

template <int MODE, bool SPLIT, bool SHAPE4>
static void run(const char* what, int waves_per_simd, double* d_out, long long* d_cyc)
{
    const int threads = 256 * waves_per_simd, blocks = 256, reps = 2000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((pipes<MODE, SPLIT, SHAPE4>), dim3(blocks), dim3(threads), 0, 0, d_out, d_cyc, 10, 1e-7);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((pipes<MODE, SPLIT, SHAPE4>), dim3(blocks), dim3(threads), 0, 0, d_out, d_cyc, reps, 1e-7);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h((size_t)blocks * threads / 64);
    CK(hipMemcpy(h.data(), d_cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    double mean = 0;
    for (long long v : h) mean += (double)v;
    mean /= (double)h.size();
    // per iteration and wave: 8 x 2048 flop of matrix work (16x16x4; 4x4x4: 32 x 512 = the same), 128 x 64 x 2 of vector work
    const double waves = (double)blocks * threads / 64;
    const double mflop = (SPLIT ? 0.5 : ((MODE & 1) ? 1.0 : 0.0)) * waves * reps * 8.0 * 2048.0;
    const double vflop = (SPLIT ? 0.5 : ((MODE & 2) ? 1.0 : 0.0)) * waves * reps * 128.0 * 128.0;
    printf("%-58s %d wave(s)/SIMD: %8.1f clock64 ticks/iteration/wave   %.3f ms   matrix %6.1f + vector %6.1f = %6.1f TFLOP/s\n", what,
           waves_per_simd, mean / reps, ms, mflop / ms * 1e-9, vflop / ms * 1e-9, (mflop + vflop) / ms * 1e-9);
}

int main()
{
    CK(hipSetDevice(0));
    double* d_out;
    long long* d_cyc;
    CK(hipMalloc((void**)&d_out, sizeof(double) * 256 * 512));
    CK(hipMalloc((void**)&d_cyc, sizeof(long long) * 256 * 8));
    for (int w = 1; w <= 2; ++w) {
        run<1, false, false>("M   8 x v_mfma_f64_16x16x4", w, d_out, d_cyc);
        run<2, false, false>("V   128 x v_fma_f64", w, d_out, d_cyc);
        run<3, false, false>("MV  both, one wave (16 FMAs behind every MFMA)", w, d_out, d_cyc);
        run<1, false, true>("M4  32 x v_mfma_f64_4x4x4", w, d_out, d_cyc);
        run<3, false, true>("M4V both, one wave (16 FMAs behind every 4 MFMAs)", w, d_out, d_cyc);
    }
    run<3, true, false>("M|V even waves matrix only, odd waves vector only", 2, d_out, d_cyc);
    run<3, true, true>("M4|V the same with the 4x4x4 shape", 2, d_out, d_cyc);
    return 0;
}
