// lds_probe.hip -- how much dynamic LDS can one workgroup get on this device?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void touch(double* out, int n) { extern __shared__ double s[]; for (int i = threadIdx.x; i < n; i += blockDim.x) s[i] = i; __syncthreads(); if (threadIdx.x == 0) out[0] = s[n - 1]; }
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int v = 0; hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, 0);
    printf("sharedMemPerBlock=%zu attrMaxSharedPerBlock=%d maxSharedMemoryPerMultiProcessor=%zu\n", p.sharedMemPerBlock, v, p.maxSharedMemoryPerMultiProcessor);
    double* out; hipMalloc(&out, 8);
    for (int kb : {32, 64, 65, 96, 128, 144, 156, 159, 160}) {
        size_t bytes = (size_t)kb * 1024;
        hipError_t e1 = hipFuncSetAttribute((const void*)touch, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        hipLaunchKernelGGL(touch, dim3(1), dim3(256), bytes, 0, out, (int)(bytes / 8));
        hipError_t e2 = hipGetLastError(); hipError_t e3 = hipDeviceSynchronize();
        double h = 0; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
        printf("%3d KiB: setattr=%s launch=%s sync=%s val=%g\n", kb, hipGetErrorName(e1), hipGetErrorName(e2), hipGetErrorName(e3), h);
    }
    return 0;
}
