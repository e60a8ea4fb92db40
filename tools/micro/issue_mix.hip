// Developer micro-benchmark: what ONE wavefront per SIMD pays per instruction of each KIND the step kernels are made of -- not
// only plain VALU (issue_cadence.hip) but the control flow around predicated blocks, scalar ALU, cross-lane reads, LDS issue and
// waits.  Each pattern is a fixed asm block repeated 32 x per loop iteration; cycles per BLOCK are reported (clock64 around the
// loop, 1024 blocks of one wave = one wave per SIMD, and 2048 = two).
// hipcc --offload-arch=gfx950 -O3 issue_mix.hip -o issue_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define REP4(x) x x x x
#define REP32(x) REP4(REP4(x)) REP4(REP4(x))

template <int MODE> __global__ void k(long long* out, float* sink, int iters, int kpred) {
    __shared__ float lds[256];
    float a = threadIdx.x * 0.5f + 1.f, b = a + 1.f, c = a + 2.f, d = a + 3.f;
    const int lane = threadIdx.x;
    lds[lane] = a; lds[lane + 64] = b;
    __syncthreads();
    unsigned addr = (unsigned)(lane * 4);
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {   // 4 dependent plain FMAs
            REP32(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
        } else if (MODE == 1) {   // the same 4 FMAs inside an exec-masked block: v_cmp, s_and_saveexec, s_cbranch_execz, ..., s_or
            REP32(asm volatile("v_cmp_gt_i32 vcc, %3, %4\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 1f\n"
                               "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                               "1: s_or_b64 exec, exec, s[20:21]" : "+v"(a) : "v"(b), "v"(c), "s"(kpred), "v"(lane) : "vcc", "s20", "s21");)
        } else if (MODE == 2) {   // the same without the branch instruction (exec masking only)
            REP32(asm volatile("v_cmp_gt_i32 vcc, %3, %4\n s_and_saveexec_b64 s[20:21], vcc\n"
                               "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                               "s_or_b64 exec, exec, s[20:21]" : "+v"(a) : "v"(b), "v"(c), "s"(kpred), "v"(lane) : "vcc", "s20", "s21");)
        } else if (MODE == 3) {   // predication by select: 4 FMAs + v_cmp + 1 v_cndmask
            REP32(asm volatile("v_mov_b32 %5, %0\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                               "v_cmp_gt_i32 vcc, %3, %4\n v_cndmask_b32 %0, %5, %0, vcc" : "+v"(a) : "v"(b), "v"(c), "s"(kpred), "v"(lane), "v"(d) : "vcc");)
        } else if (MODE == 4) {   // 4 scalar ALU instructions + 4 FMAs
            REP32(asm volatile("s_add_u32 s20, s20, 1\n v_fma_f32 %0, %0, %1, %2\n s_add_u32 s21, s21, 1\n v_fma_f32 %0, %0, %1, %2\n"
                               "s_add_u32 s20, s20, 1\n v_fma_f32 %0, %0, %1, %2\n s_add_u32 s21, s21, 1\n v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c) : "s20", "s21", "scc");)
        } else if (MODE == 5) {   // v_readlane -> SGPR -> dependent FMA, 4 x
            REP32(asm volatile("v_readlane_b32 s20, %0, 3\n s_nop 0\n v_fma_f32 %0, %0, s20, %2\n v_readlane_b32 s20, %0, 5\n s_nop 0\n v_fma_f32 %0, %0, s20, %2\n"
                               "v_readlane_b32 s20, %0, 7\n s_nop 0\n v_fma_f32 %0, %0, s20, %2\n v_readlane_b32 s20, %0, 9\n s_nop 0\n v_fma_f32 %0, %0, s20, %2"
                               : "+v"(a) : "v"(b), "v"(c) : "s20");)
        } else if (MODE == 6) {   // DPP row shift on the operand of a dependent FMA, 4 x
            REP32(asm volatile("s_nop 1\n v_fmac_f32_dpp %0, %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "s_nop 1\n v_fmac_f32_dpp %0, %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "s_nop 1\n v_fmac_f32_dpp %0, %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "s_nop 1\n v_fmac_f32_dpp %0, %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a) : "v"(b));)
        } else if (MODE == 7) {   // ONE dependent LDS round trip: store, load, wait, use
            REP32(asm volatile("ds_write_b32 %3, %0\n ds_read_b32 %0, %3 offset:256\n s_waitcnt lgkmcnt(0)\n v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c), "v"(addr) : "memory");)
        } else if (MODE == 8) {   // 8 independent LDS loads issued together, one wait, 8 uses (a range sum's shape)
            REP32(asm volatile("ds_read_b32 v40, %3\n ds_read_b32 v41, %3 offset:4\n ds_read_b32 v42, %3 offset:8\n ds_read_b32 v43, %3 offset:12\n"
                               "ds_read_b32 v44, %3 offset:256\n ds_read_b32 v45, %3 offset:260\n ds_read_b32 v46, %3 offset:264\n ds_read_b32 v47, %3 offset:268\n"
                               "s_waitcnt lgkmcnt(0)\n v_add_f32 %0, %0, v40\n v_add_f32 %0, %0, v41\n v_add_f32 %0, %0, v42\n v_add_f32 %0, %0, v43\n"
                               "v_add_f32 %0, %0, v44\n v_add_f32 %0, %0, v45\n v_add_f32 %0, %0, v46\n v_add_f32 %0, %0, v47"
                               : "+v"(a) : "v"(b), "v"(c), "v"(addr) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "memory");)
        } else if (MODE == 9) {   // ds_bpermute round trip + use
            REP32(asm volatile("ds_bpermute_b32 %0, %3, %0\n s_waitcnt lgkmcnt(0)\n v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c), "v"(addr) : "memory");)
        } else if (MODE == 10) {  // s_waitcnt with nothing outstanding + FMA, 4 x
            REP32(asm volatile("s_waitcnt lgkmcnt(0)\n v_fma_f32 %0, %0, %1, %2\n s_waitcnt lgkmcnt(0)\n v_fma_f32 %0, %0, %1, %2\n"
                               "s_waitcnt lgkmcnt(0)\n v_fma_f32 %0, %0, %1, %2\n s_waitcnt lgkmcnt(0)\n v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
        } else if (MODE == 11) {  // correctly rounded 1/x as hipcc emits it for 1.0f / x (division by an FMA result)
            REP32(a = 1.0f / (a * 1.0001f + 2.0f);)
        } else if (MODE == 12) {  // correctly rounded sqrtf
            REP32(a = sqrtf(a * 1.0001f + 2.0f);)
        } else if (MODE == 13) {  // 4 INDEPENDENT plain FMAs (four accumulators)
            REP32(asm volatile("v_fma_f32 v40, v40, %1, %2\n v_fma_f32 v41, v41, %1, %2\n v_fma_f32 v42, v42, %1, %2\n v_fma_f32 v43, v43, %1, %2"
                               : "+v"(a) : "v"(b), "v"(c) : "v40", "v41", "v42", "v43");)
        } else if (MODE == 14) {  // 4 dependent PACKED FMAs: v_pk_fma_f32 on even-aligned register pairs (8 fp32 FMAs)
            REP32(asm volatile("v_pk_fma_f32 v[40:41], v[40:41], v[44:45], v[46:47]\n v_pk_fma_f32 v[40:41], v[40:41], v[44:45], v[46:47]\n"
                               "v_pk_fma_f32 v[40:41], v[40:41], v[44:45], v[46:47]\n v_pk_fma_f32 v[40:41], v[40:41], v[44:45], v[46:47]"
                               : "+v"(a) : "v"(b), "v"(c) : "v40", "v41", "v44", "v45", "v46", "v47");)
        } else if (MODE == 15) {  // 4 INDEPENDENT packed FMAs (four accumulator pairs: 8 fp32 FMAs)
            REP32(asm volatile("v_pk_fma_f32 v[40:41], v[40:41], v[48:49], v[50:51]\n v_pk_fma_f32 v[42:43], v[42:43], v[48:49], v[50:51]\n"
                               "v_pk_fma_f32 v[44:45], v[44:45], v[48:49], v[50:51]\n v_pk_fma_f32 v[46:47], v[46:47], v[48:49], v[50:51]"
                               : "+v"(a) : "v"(b), "v"(c) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");)
        } else if (MODE == 16) {  // packed multiply / add, dependent: 2 x (v_pk_mul_f32 + v_pk_add_f32)
            REP32(asm volatile("v_pk_mul_f32 v[40:41], v[40:41], v[44:45]\n v_pk_add_f32 v[40:41], v[40:41], v[46:47]\n"
                               "v_pk_mul_f32 v[40:41], v[40:41], v[44:45]\n v_pk_add_f32 v[40:41], v[40:41], v[46:47]"
                               : "+v"(a) : "v"(b), "v"(c) : "v40", "v41", "v44", "v45", "v46", "v47");)
        } else if (MODE == 17) {  // packed FMA with a BROADCAST scalar operand (op_sel_hi: both halves take the low register): a * s + b on pairs
            REP32(asm volatile("v_pk_fma_f32 v[40:41], v[40:41], v[44:45], v[46:47] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[40:41], v[40:41], v[44:45], v[46:47] op_sel_hi:[1,0,1]\n"
                               "v_pk_fma_f32 v[40:41], v[40:41], v[44:45], v[46:47] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[40:41], v[40:41], v[44:45], v[46:47] op_sel_hi:[1,0,1]"
                               : "+v"(a) : "v"(b), "v"(c) : "v40", "v41", "v44", "v45", "v46", "v47");)
        } else if (MODE == 18) {  // 2 moves + 1 packed FMA: what packing costs when the operands are NOT already an aligned pair
            REP32(asm volatile("v_mov_b32 v44, %1\n v_mov_b32 v45, %2\n v_pk_fma_f32 v[40:41], v[40:41], v[44:45], v[46:47]\n"
                               "v_mov_b32 v44, %1\n v_mov_b32 v45, %2\n v_pk_fma_f32 v[40:41], v[40:41], v[44:45], v[46:47]"
                               : "+v"(a) : "v"(b), "v"(c) : "v40", "v41", "v44", "v45", "v46", "v47");)
        }
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + lds[(lane + 1) & 63];
}
template <int MODE> void run(const char* name, int blocks, long long* out, float* sink, int iters, int kpred) {
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, sink, iters, kpred);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(blocks);
    (void)hipMemcpy(h.data(), out, 8 * blocks, hipMemcpyDeviceToHost);
    double mn = 1e30, s = 0;
    for (auto v : h) { double x = v / (double)iters / 32.0; mn = x < mn ? x : mn; s += x; }
    printf("%-78s %4.1f waves/SIMD: cycles per block min %6.1f mean %6.1f\n", name, blocks / 1024.0, mn, s / blocks);
}
int main() {
    long long* out; float* sink;
    (void)hipMalloc(&out, 8 * 8192); (void)hipMalloc(&sink, 8192 * 64 * 4);
    const int iters = 200;
    for (int blocks : {1024, 2048}) {
        run<0>("4 dependent v_fma", blocks, out, sink, iters, 9);
        run<1>("v_cmp + s_and_saveexec + s_cbranch_execz + 4 v_fma + s_or (9 lanes on)", blocks, out, sink, iters, 9);
        run<1>("  ... no lane on (branch taken)", blocks, out, sink, iters, 0);
        run<2>("v_cmp + s_and_saveexec + 4 v_fma + s_or (no branch instruction)", blocks, out, sink, iters, 9);
        run<3>("v_mov + 4 v_fma + v_cmp + v_cndmask (select instead of exec mask)", blocks, out, sink, iters, 9);
        run<4>("4 x (s_add_u32 + v_fma)", blocks, out, sink, iters, 9);
        run<5>("4 x (v_readlane -> SGPR + s_nop + dependent v_fma)", blocks, out, sink, iters, 9);
        run<6>("4 x (s_nop 1 + dependent v_fmac_dpp row_shl:1)", blocks, out, sink, iters, 9);
        run<7>("ds_write + dependent ds_read + wait + v_fma (LDS round trip)", blocks, out, sink, iters, 9);
        run<8>("8 ds_read issued together + wait + 8 v_add", blocks, out, sink, iters, 9);
        run<9>("ds_bpermute + wait + v_fma", blocks, out, sink, iters, 9);
        run<10>("4 x (s_waitcnt lgkmcnt(0), nothing outstanding + v_fma)", blocks, out, sink, iters, 9);
        run<11>("1.0f / x, correctly rounded (+ the v_fma feeding it)", blocks, out, sink, iters, 9);
        run<12>("sqrtf(x), correctly rounded (+ the v_fma feeding it)", blocks, out, sink, iters, 9);
        run<13>("4 INDEPENDENT v_fma (four accumulators)", blocks, out, sink, iters, 9);
        run<14>("4 dependent v_pk_fma_f32 (8 fp32 FMAs on aligned pairs)", blocks, out, sink, iters, 9);
        run<15>("4 INDEPENDENT v_pk_fma_f32 (8 fp32 FMAs)", blocks, out, sink, iters, 9);
        run<16>("2 x (v_pk_mul_f32 + v_pk_add_f32), dependent", blocks, out, sink, iters, 9);
        run<17>("4 dependent v_pk_fma_f32 with a broadcast operand (op_sel_hi)", blocks, out, sink, iters, 9);
        run<18>("2 x (2 v_mov into a pair + v_pk_fma_f32): packing from scattered registers", blocks, out, sink, iters, 9);
    }
    return 0;
}
