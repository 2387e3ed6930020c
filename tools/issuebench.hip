// issuebench.hip — VALU issue cost per encoding on gfx950 (k_extend<false> is VALU-issue-bound, so which encodings the
// compiler picks matters).  Each test issues 8 independent instructions per iteration from 8 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/issuebench.hip -o gpurun_out/issuebench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N_ITER 4096
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
template <int OP> __global__ void k(float *out, float fs) {
    float a[8], b = fs * 0.5f + 1.f, c = fs * 0.25f + 0.5f;
    unsigned long long d[8]; for (int i = 0; i < 8; i++) d[i] = threadIdx.x + i;
    for (int i = 0; i < 8; i++) a[i] = fs + threadIdx.x + i;
#pragma unroll 1
    for (int it = 0; it < N_ITER; it++) {
#define FMAC(i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define FMA3(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
#define FMA3S(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "s"(fs), "v"(c));
#define MUL2(i) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
#define MUL3(i) asm volatile("v_mul_f32_e64 %0, %1, %0" : "+v"(a[i]) : "v"(b));
#define MUL3N(i) asm volatile("v_mul_f32_e64 %0, -%1, |%0|" : "+v"(a[i]) : "v"(b));
#define ADD2(i) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
#define SUB2S(i) asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(a[i]) : "s"(fs));
#define MIN3(i) asm volatile("v_min3_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
#define MIN2(i) asm volatile("v_min_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
#define CND3(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a[i]) : "v"(b));
#define CND2(i) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
#define CMP3(i) asm volatile("v_cmp_gt_f32_e64 s[10:11], %0, %1" : : "v"(a[i]), "v"(b) : "s10", "s11");
#define CMP2(i) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
#define MOV(i) asm volatile("v_mov_b32_e32 %0, %1" : "+v"(a[i]) : "v"(b));
#define FIX(i) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define RCP(i) asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(a[i]));
#define XOR(i) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
#define MULLO(i) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
#define MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d[i]) : "v"(b), "v"(c) : "vcc");
#define XSDWA(i) asm volatile("v_xor_b32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(a[i]));
#define CVTU(i) asm volatile("v_cvt_f32_u32_e32 %0, %0" : "+v"(a[i]));
#define SMOV(i) asm volatile("s_mov_b32 s12, s13" : : : "s12");
#define SAND(i) asm volatile("s_and_b64 s[14:15], s[10:11], exec" : : : "s14", "s15", "scc");
#define MIXVS(i) asm volatile("v_mul_f32_e32 %0, %1, %0\n s_and_b64 s[14:15], s[10:11], exec" : "+v"(a[i]) : "v"(b) : "s14", "s15", "scc");
        if (OP == 0) { REP8(FMAC) } if (OP == 1) { REP8(FMA3) } if (OP == 2) { REP8(FMA3S) } if (OP == 3) { REP8(MUL2) }
        if (OP == 4) { REP8(MUL3) } if (OP == 5) { REP8(MUL3N) } if (OP == 6) { REP8(ADD2) } if (OP == 7) { REP8(SUB2S) }
        if (OP == 8) { REP8(MIN3) } if (OP == 9) { REP8(MIN2) } if (OP == 10) { REP8(CND3) } if (OP == 11) { REP8(CND2) }
        if (OP == 12) { REP8(CMP3) } if (OP == 13) { REP8(CMP2) } if (OP == 14) { REP8(MOV) } if (OP == 15) { REP8(FIX) }
        if (OP == 16) { REP8(RCP) } if (OP == 17) { REP8(XOR) } if (OP == 18) { REP8(MULLO) } if (OP == 19) { REP8(SMOV) }
        if (OP == 20) { REP8(SAND) } if (OP == 21) { REP8(MIXVS) }
        if (OP == 22) { REP8(MAD64) } if (OP == 23) { REP8(XSDWA) } if (OP == 24) { REP8(CVTU) }
    }
    float s = 0; for (int i = 0; i < 8; i++) s += a[i] + (float)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char *name, int waves_per_simd) {
    float *out; hipMalloc(&out, 256 * 16 * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(256 * waves_per_simd), block(256);   // n blocks/CU x 4 waves = n waves per SIMD
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, out, 1.5f);
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, out, 1.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double per_simd = (double)waves_per_simd * N_ITER * 8;            // instructions issued per SIMD
    printf("%-36s waves/SIMD %d: %6.2f cycles per wave-instruction (2.4 GHz nominal)\n", name, waves_per_simd, ms * 1e-3 * 2.4e9 / per_simd);
    hipFree(out);
}
#define RUN(OP, NAME) run<OP>(NAME, 8); run<OP>(NAME, 4); run<OP>(NAME, 1);
int main() {
    RUN(0, "v_fmac_f32_e32 (VOP2)") RUN(1, "v_fma_f32 (VOP3)") RUN(2, "v_fma_f32 sgpr src") RUN(3, "v_mul_f32_e32") RUN(4, "v_mul_f32_e64")
    RUN(5, "v_mul_f32_e64 neg/abs") RUN(6, "v_add_f32_e32") RUN(7, "v_sub_f32_e32 sgpr") RUN(8, "v_min3_f32") RUN(9, "v_min_f32_e32")
    RUN(10, "v_cndmask_b32_e64 (sgpr mask)") RUN(11, "v_cndmask_b32_e32 (vcc)") RUN(12, "v_cmp_gt_f32_e64 -> sgpr") RUN(13, "v_cmp_gt_f32_e32 -> vcc")
    RUN(14, "v_mov_b32") RUN(15, "v_div_fixup_f32") RUN(16, "v_rcp_f32") RUN(17, "v_xor_b32") RUN(18, "v_mul_lo_u32") RUN(19, "s_mov_b32") RUN(20, "s_and_b64")
    RUN(21, "v_mul_f32 + s_and_b64 pair (per pair)")
    RUN(22, "v_mad_u64_u32") RUN(23, "v_xor_b32_sdwa") RUN(24, "v_cvt_f32_u32")
    return 0;
}
