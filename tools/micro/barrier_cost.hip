#include <hip/hip_runtime.h>
#include <stdio.h>
// cost of a workgroup barrier for a 2-wave workgroup (main wave works, helper mostly waits), one WG per 2 SIMDs
template <int MODE> __global__ void k(long long* out, float* sink, int iters) {
    __shared__ float lds[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a = lane * 0.5f, b = 1.0001f;
    lds[threadIdx.x] = a;
    __syncthreads();
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (wave == 0) {
#pragma unroll
            for (int j = 0; j < 64; ++j) a = a * b + 0.25f;   // ~64 dependent fmas
            lds[lane] = a;
        } else if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 32; ++j) a = a * b + 0.25f;   // helper does half as much
            lds[64 + lane] = a;
        }
        if (MODE >= 1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        if (wave == 0) a += lds[64 + (lane ^ 1)];
    }
    long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[MODE] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
int main() {
    long long* out; float* sink;
    hipMalloc(&out, 64); hipMalloc(&sink, 1024 * 128 * 4);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<0>, dim3(1024), dim3(128), 0, 0, out, sink, iters);
        hipLaunchKernelGGL(k<1>, dim3(1024), dim3(128), 0, 0, out, sink, iters);
        hipLaunchKernelGGL(k<2>, dim3(1024), dim3(128), 0, 0, out, sink, iters);
        hipDeviceSynchronize();
    }
    long long h[3]; hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
    printf("cycles/iter: no barrier %.1f, barrier (idle helper) %.1f, barrier (busy helper) %.1f\n", h[0] / (double)iters, h[1] / (double)iters, h[2] / (double)iters);
    return 0;
}
