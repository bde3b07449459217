// probe 2: same MFMA mix as mfma_mix_probe, operands fetched for TWO k-steps per instruction
// (ds_read_b128 / global_load_dwordx4) -- is the cost of operand traffic per instruction or per byte?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int G = 5, NRT = 3;
typedef double d2 __attribute__((ext_vector_type(2)));

template <int LDS_A, int GLB_B>   // 0: registers, 1: 8-byte loads per k-step, 2: 16-byte loads per two k-steps
__global__ __launch_bounds__(256) void mix(const double* __restrict__ B, double* out, int kpairs, int LD, int LDv, int reps)
{
    extern __shared__ double Lbuf[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 20 * LDv; i += 256) Lbuf[i] = 1e-3 * (i % 97);
    __syncthreads();
    const int li = lane & 15, lk = lane >> 4;
    double acc[G][NRT];
    for (int g = 0; g < G; ++g) for (int j = 0; j < NRT; ++j) acc[g][j] = 0;
    // paired layouts: element (k-pair kp, lk) holds the operands of k-steps 2kp and 2kp+1 side by side
    const double* ap = Lbuf + (size_t)(lane & 3) * LDv + lk * 2;
    const double* bp = B + ((size_t)lk * LD + li + wave * 48) * 2;
    for (int r = 0; r < reps; ++r) {
        d2 a0[G], a1[G], b0[NRT], b1[NRT];
        auto load = [&](int kp, d2 (&a)[G], d2 (&b)[NRT]) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const double* q = ap + (size_t)(4 * g) * LDv + kp * 8;
                if (LDS_A == 2) a[g] = *reinterpret_cast<const d2*>(q);
                else if (LDS_A == 1) { a[g].x = q[0]; a[g].y = q[4 + 1]; }   // two separate 8-byte reads
                else { a[g].x = 1.0 + g; a[g].y = 2.0 + g; }
            }
#pragma unroll
            for (int j = 0; j < NRT; ++j) {
                const double* q = bp + (size_t)kp * 8 * LD + j * 32;
                if (GLB_B == 2) b[j] = *reinterpret_cast<const d2*>(q);
                else if (GLB_B == 1) { b[j].x = q[0]; b[j].y = q[4 * LD]; }
                else { b[j].x = 2.0 + j; b[j].y = 3.0 + j; }
            }
        };
        auto compute = [&](const d2 (&a)[G], const d2 (&b)[NRT]) {
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int j = 0; j < NRT; ++j) acc[g][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[g].x, b[j].x, acc[g][j], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int j = 0; j < NRT; ++j) acc[g][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[g].y, b[j].y, acc[g][j], 0, 0, 0);
        };
        load(0, a0, b0);
        for (int kp = 0; kp + 2 <= kpairs; kp += 2) {
            load(kp + 1, a1, b1);
            compute(a0, b0);
            load((kp + 2 < kpairs) ? kp + 2 : kpairs - 1, a0, b0);
            compute(a1, b1);
        }
    }
    double s = 0;
    for (int g = 0; g < G; ++g) for (int j = 0; j < NRT; ++j) s += acc[g][j];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

int main(int argc, char** argv)
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const int kpairs = 20, LD = 176, reps = 400, LDv = 164;
    double* B; hipMalloc(&B, (size_t)200 * LD * 8 * 4); hipMemset(B, 0, (size_t)200 * LD * 8 * 4);
    double* out; hipMalloc(&out, (size_t)cus * 4 * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = 20 * LDv * 8 + 1024;
    auto run = [&](const char* name, auto kern, int bpc) {
        dim3 g(cus * bpc), b(256);
        hipLaunchKernelGGL(kern, g, b, lds, 0, B, out, kpairs, LD, LDv, 4); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(kern, g, b, lds, 0, B, out, kpairs, LD, LDv, reps); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = 512.0 * 30 * kpairs * reps * (double)g.x * 4;
        printf("%-44s %d waves/SIMD: %8.3f ms  %6.2f TFLOP/s\n", name, bpc, ms, fl / ms / 1e9);
    };
    for (int bpc : {1, 2, 3}) {
        run("registers only", mix<0, 0>, bpc);
        run("A: 8-byte LDS reads", mix<1, 0>, bpc);
        run("A: 16-byte LDS reads (two k-steps each)", mix<2, 0>, bpc);
        run("B: 8-byte global loads", mix<0, 1>, bpc);
        run("B: 16-byte global loads (two k-steps each)", mix<0, 2>, bpc);
        run("A 8-byte + B 8-byte", mix<1, 1>, bpc);
        run("A 16-byte + B 16-byte", mix<2, 2>, bpc);
    }
    return 0;
}
