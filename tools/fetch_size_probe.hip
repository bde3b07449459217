// fetch_size_probe.hip -- measurement only (not part of the product): kernels that move a KNOWN number of bytes from
// memory with the access patterns of the pruning kernels, for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
// (/opt/skills/guides/MI355X_MICROARCH.md: "FETCH_SIZE reports 1/2 of the bytes of a 16 B/lane streaming read; other
// access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
// Every kernel touches each byte of a 2 GiB array (8x the 256 MiB Infinity Cache) exactly once:
//   stream16      16 B per lane, contiguous (the guide's reference pattern)
//   stream8       8 B per lane, contiguous (512 B per wave load)
//   b_operand     the walk's matrix operand: lane l loads 8 B at row (l >> 4), column tile + (l & 15) -- four 128-byte
//                 segments per wave load, k-steps walking down a [KP][LD] matrix (global_load_dwordx2, scalar base)
//   col_gather    the walk's leaf gathers: every 16 lanes read one 128-byte run of a pseudo-random matrix column
//   store8        8 B per lane stores in the accumulator layout (16-row tiles), for WRITE_SIZE
//   hipcc --offload-arch=gfx950 -O3 -o tools/_variants/fetch_size_probe tools/fetch_size_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr size_t BYTES = 2ull << 30;
constexpr int LD = 272, KP = 252;   // the configs[2] matrix shape (251 wide)

__global__ __launch_bounds__(256) void stream16(const d2* __restrict__ p, double* out, size_t n)
{
    d2 acc = {0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += p[i];
    if (acc.x + acc.y == 12345.678) out[0] = acc.x;
}
__global__ __launch_bounds__(256) void stream8(const double* __restrict__ p, double* out, size_t n)
{
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += p[i];
    if (acc == 12345.678) out[0] = acc;
}
// one workgroup (4 waves) per matrix; wave w owns column tiles w, w + 4, ...; per tile it walks all k-steps
__global__ __launch_bounds__(256) void b_operand(const double* __restrict__ p, double* out, int n_mat)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lk = lane >> 4;
    double acc = 0;
    for (int m = blockIdx.x; m < n_mat; m += gridDim.x) {
        const double* M = p + (size_t)m * KP * LD;
        for (int tile = wave; tile < LD / 16; tile += 4)
            for (int k = 0; k < KP; k += 4) acc += M[(size_t)(k + lk) * LD + tile * 16 + li];
    }
    if (acc == 12345.678) out[0] = acc;
}
__global__ __launch_bounds__(256) void col_gather(const double* __restrict__ p, double* out, size_t n_runs)
{
    // every 128-byte run of the array read once, in a scrambled order (runs = 16 doubles)
    const int li = threadIdx.x & 15;
    double acc = 0;
    const size_t groups = (size_t)gridDim.x * 16;   // 16-lane groups in flight
    for (size_t r = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4); r < n_runs; r += groups) {
        const size_t s = (r * 2654435761ull) % n_runs;   // odd multiplier: a permutation when n_runs is a power of two
        acc += p[s * 16 + li];
    }
    if (acc == 12345.678) out[0] = acc;
}
__global__ __launch_bounds__(256) void store8(double* __restrict__ p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = (double)i;
}

int main()
{
    double *buf, *out;
    if (hipMalloc(&buf, BYTES) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    hipMemset(buf, 0, BYTES);
    hipDeviceSynchronize();
    const size_t n8 = BYTES / 8;
    const int n_mat = (int)(n8 / ((size_t)KP * LD));
    hipLaunchKernelGGL(stream16, dim3(4096), dim3(256), 0, 0, (const d2*)buf, out, n8 / 2);
    hipLaunchKernelGGL(stream8, dim3(4096), dim3(256), 0, 0, buf, out, n8);
    hipLaunchKernelGGL(b_operand, dim3(4096), dim3(256), 0, 0, buf, out, n_mat);
    hipLaunchKernelGGL(col_gather, dim3(4096), dim3(256), 0, 0, buf, out, n8 / 16);
    hipLaunchKernelGGL(store8, dim3(4096), dim3(256), 0, 0, buf, n8);
    hipDeviceSynchronize();
    printf("bytes stream16 %zu stream8 %zu b_operand %zu col_gather %zu store8 %zu\n", BYTES, BYTES,
           (size_t)n_mat * KP * LD * 8, BYTES, BYTES);
    return 0;
}
