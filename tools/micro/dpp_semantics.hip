// Developer micro-test (GPU box): pins the semantics of the DPP row shifts the row-tree sums of dsim_core.hpp rely on
// (dsim_hip.hip: DevExec::row_from_above / row_from_below).  hipcc --offload-arch=gfx950 -O2 dpp_semantics.hip -o dpp_semantics
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CTRL, bool BC> __device__ int dpp(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, 0xf, BC); }

__global__ void k(int* out) {
    const int lane = threadIdx.x;
    const int v = 100 + lane;
    out[0 * 64 + lane] = dpp<0x101, false>(-1, v);   // row_shl:1, old = -1, bound_ctrl off
    out[1 * 64 + lane] = dpp<0x111, false>(-1, v);   // row_shr:1
    out[2 * 64 + lane] = dpp<0x108, false>(-1, v);   // row_shl:8
    out[3 * 64 + lane] = dpp<0x101, true>(-1, v);    // row_shl:1, bound_ctrl on
    int r = -7;
    if (lane != 5) r = dpp<0x101, false>(-1, v);     // lane 5 disabled: what does lane 4 read?
    out[4 * 64 + lane] = r;
    out[5 * 64 + lane] = dpp<0x130, false>(-1, v);   // wave_shl:1
    out[6 * 64 + lane] = dpp<0x138, false>(-1, v);   // wave_shr:1
}

int main() {
    int* d;
    hipMalloc(&d, 7 * 64 * 4);
    hipLaunchKernelGGL(k, 1, 64, 0, 0, d);
    int h[7 * 64];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[7] = {"row_shl:1 old=-1 bc=0", "row_shr:1 old=-1 bc=0", "row_shl:8 old=-1 bc=0", "row_shl:1 old=-1 bc=1",
                            "row_shl:1, lane 5 disabled", "wave_shl:1", "wave_shr:1"};
    for (int t = 0; t < 7; ++t) {
        printf("%-28s:", names[t]);
        for (int l = 0; l < 20; ++l) printf(" %d", h[t * 64 + l]);
        printf(" ... l63=%d\n", h[t * 64 + 63]);
    }
    return 0;
}
