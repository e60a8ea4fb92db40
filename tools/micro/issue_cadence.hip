// Developer micro-benchmark: issue cadence of wave64 fp32 VALU instructions on gfx950 -- dependent vs independent, plain vs packed,
// 1 / 2 / 4 / 8 wavefronts per SIMD.  Every block records its own duration; min / mean / max over blocks are printed.
// hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize issue_cadence.hip -o issue_cadence
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE> __global__ void k(long long* out, float* sink, int iters) {
    float a = threadIdx.x * 0.5f + 1.f, b = a + 1.f, c = a + 2.f, d = a + 3.f;
    f2 p = {a, b}, q = {c, d};
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {          // 128 dependent plain FMAs
#pragma unroll
            for (int j = 0; j < 128; ++j) a = a * 1.0001f + 0.25f;
        } else if (MODE == 1) {   // 128 plain FMAs in 4 independent chains (no SLP: plain v_fma_f32)
#pragma unroll
            for (int j = 0; j < 32; ++j) { a = a * 1.0001f + 0.25f; b = b * 1.0001f + 0.25f; c = c * 1.0001f + 0.25f; d = d * 1.0001f + 0.25f; }
        } else if (MODE == 2) {   // 128 packed FMAs in 2 independent chains
#pragma unroll
            for (int j = 0; j < 64; ++j) { p = p * 1.0001f + 0.25f; q = q * 1.0001f + 0.25f; }
        } else if (MODE == 3) {   // 128 dependent packed FMAs
#pragma unroll
            for (int j = 0; j < 128; ++j) p = p * 1.0001f + 0.25f;
        }
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + p.x + p.y + q.x + q.y;
}
template <int MODE> void run(const char* name, int blocks, long long* out, float* sink, int iters) {
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, sink, iters);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(blocks);
    (void)hipMemcpy(h.data(), out, 8 * blocks, hipMemcpyDeviceToHost);
    double mn = 1e30, mx = 0, s = 0;
    for (auto v : h) { double x = v / (double)iters / 128.0; mn = x < mn ? x : mn; mx = x > mx ? x : mx; s += x; }
    printf("%-44s blocks %5d (%.1f waves/SIMD): cycles per instruction min %.2f mean %.2f max %.2f\n", name, blocks, blocks / 1024.0, mn, s / blocks, mx);
}
int main() {
    long long* out; float* sink;
    (void)hipMalloc(&out, 8 * 8192); (void)hipMalloc(&sink, 8192 * 64 * 4);
    const int iters = 2000;
    for (int blocks : {1024, 2048, 4096, 8192}) {
        run<0>("plain v_fma_f32, one dependent chain", blocks, out, sink, iters);
        run<1>("plain v_fma_f32, four independent chains", blocks, out, sink, iters);
        run<2>("v_pk_fma_f32, two independent chains", blocks, out, sink, iters);
        run<3>("v_pk_fma_f32, one dependent chain", blocks, out, sink, iters);
    }
    return 0;
}
