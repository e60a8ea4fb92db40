// Developer micro-benchmark: latency of dependent LDS round trips and of dependent VALU chains for ONE wavefront per SIMD
// (the regime of the dsim kernels at 1024 environments).  hipcc --offload-arch=gfx950 -O3 lds_latency.hip -o lds_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE> __global__ void k(long long* out, float* sink, int iters) {
    __shared__ float lds[4096];
    const int lane = threadIdx.x;
    float a = lane * 0.5f + 1.f;
    lds[lane] = a;
    __syncthreads();
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {          // write -> read another lane's word -> use (one phase boundary of the kernels)
            lds[lane] = a;
            asm volatile("" ::: "memory");
            a += lds[(lane + 1) & 63];
        } else if (MODE == 1) {   // read -> use -> read at an address that depends on nothing (pure load latency chain)
            a += lds[(lane * 7 + i) & 1023];
        } else if (MODE == 2) {   // 8 independent loads, one wait
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += lds[(lane + 64 * j + i) & 1023];
            a += s;
        } else if (MODE == 3) {   // dependent fma chain of 16
#pragma unroll
            for (int j = 0; j < 16; ++j) a = a * 1.0001f + 0.25f;
        } else if (MODE == 4) {   // 16 independent fmas (4 chains of 4)
            float b = a + 1.f, c = a + 2.f, d = a + 3.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { a = a * 1.0001f + 0.25f; b = b * 1.0001f + 0.25f; c = c * 1.0001f + 0.25f; d = d * 1.0001f + 0.25f; }
            a += b + c + d;
        } else if (MODE == 5) {   // sqrt + division (IEEE)
            a = sqrtf(a) + 1.0f / a;
        } else if (MODE == 6) {   // 128 dependent fmas per iteration: the loop overhead (a taken branch) is amortised away
#pragma unroll
            for (int j = 0; j < 128; ++j) a = a * 1.0001f + 0.25f;
        } else if (MODE == 7) {   // 128 fmas in 4 independent chains
            float b = a + 1.f, c = a + 2.f, d = a + 3.f;
#pragma unroll
            for (int j = 0; j < 32; ++j) { a = a * 1.0001f + 0.25f; b = b * 1.0001f + 0.25f; c = c * 1.0001f + 0.25f; d = d * 1.0001f + 0.25f; }
            a += b + c + d;
        } else if (MODE == 9 || MODE == 10) {   // MODE 6 with 2 / 4 wavefronts per SIMD
#pragma unroll
            for (int j = 0; j < 128; ++j) a = a * 1.0001f + 0.25f;
        } else if (MODE == 8) {   // an empty iteration: what the loop itself costs
            asm volatile("" : "+v"(a));
        }
    }
    long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[MODE] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
int main() {
    long long* out; float* sink;
    (void)hipMalloc(&out, 128); (void)hipMalloc(&sink, 1024 * 64 * 4);
    const int iters = 4000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<0>, dim3(1024), dim3(64), 0, 0, out, sink, iters);
        hipLaunchKernelGGL(k<1>, dim3(1024), dim3(64), 0, 0, out, sink, iters);
        hipLaunchKernelGGL(k<2>, dim3(1024), dim3(64), 0, 0, out, sink, iters);
        hipLaunchKernelGGL(k<3>, dim3(1024), dim3(64), 0, 0, out, sink, iters);
        hipLaunchKernelGGL(k<4>, dim3(1024), dim3(64), 0, 0, out, sink, iters);
        hipLaunchKernelGGL(k<5>, dim3(1024), dim3(64), 0, 0, out, sink, iters);
        hipLaunchKernelGGL(k<6>, dim3(1024), dim3(64), 0, 0, out, sink, iters);
        hipLaunchKernelGGL(k<7>, dim3(1024), dim3(64), 0, 0, out, sink, iters);
        hipLaunchKernelGGL(k<8>, dim3(1024), dim3(64), 0, 0, out, sink, iters);
        // the same two with TWO and FOUR resident wavefronts per SIMD (2048 / 4096 workgroups of one wave on 1024 SIMDs)
        hipLaunchKernelGGL(k<9>, dim3(2048), dim3(64), 0, 0, out, sink, iters);
        hipLaunchKernelGGL(k<10>, dim3(4096), dim3(64), 0, 0, out, sink, iters);
        (void)hipDeviceSynchronize();
    }
    long long h[11]; (void)hipMemcpy(h, out, 88, hipMemcpyDeviceToHost);
    const char* names[11] = {"store -> load other lane -> use", "load -> use (dependent loads)", "8 independent loads -> use", "16 dependent fma", "16 fma in 4 chains (+3 adds)", "sqrtf + 1/x (IEEE)",
                             "128 dependent fma", "128 fma in 4 chains (+3 adds)", "empty iteration", "128 dependent fma, 2 waves / SIMD", "128 dependent fma, 4 waves / SIMD"};
    for (int m = 0; m < 11; ++m) printf("%-36s %.1f cycles/iter\n", names[m], h[m] / (double)iters);
    return 0;
}
