// microbench.hip — issue cost of the instruction classes the path tracer leans on (gfx950).
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench.hip -o gpurun_out/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define N_ITER 2048
typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP> __global__ void k(uint32_t *out, uint32_t seed, float fs) {
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u, c = b + 77u, d = c * 3u;
    float x = fs + threadIdx.x, y = x * 1.0001f, z = y + 0.5f, w = z * 0.999f;
#pragma unroll 1
    for (int i = 0; i < N_ITER; i++) {
        if (OP == 0) { x = fmaf(x, 1.0001f, 0.5f); y = fmaf(y, 1.0001f, 0.5f); z = fmaf(z, 1.0001f, 0.5f); w = fmaf(w, 1.0001f, 0.5f); }
        if (OP == 1) { a = a * 1664525u; b = b * 1664525u; c = c * 1664525u; d = d * 1664525u; }              // v_mul_lo_u32
        if (OP == 2) { a = a * b + c; b = b * c + d; c = c * d + a; d = d * a + b; }                          // v_mad_u64_u32 or mul+add
        if (OP == 3) { a = __umul24(a, b) + c; b = __umul24(b, c) + d; c = __umul24(c, d) + a; d = __umul24(d, a) + b; }  // v_mad_u32_u24
        if (OP == 4) { x = y / x; y = z / y; z = w / z; w = x / w; }                                          // IEEE division
        if (OP == 5) { x = y * __frcp_rn(x); y = z * __frcp_rn(y); z = w * __frcp_rn(z); w = x * __frcp_rn(w); }
        if (OP == 6) { x = sqrtf(x + 1.f); y = sqrtf(y + 1.f); z = sqrtf(z + 1.f); w = sqrtf(w + 1.f); }
        if (OP == 7) { a = __umulhi(a, 0xD2511F53u) ^ b; b = __umulhi(b, 0xCD9E8D57u) ^ c; c = __umulhi(c, 0xD2511F53u) ^ d; d = __umulhi(d, 0xCD9E8D57u) ^ a; }
        if (OP == 8) { a ^= a >> 16; b ^= b >> 15; c ^= c >> 13; d ^= d >> 16; a += b; b += c; c += d; d += a; }  // cheap int ops (2 per lane-op)
        if (OP == 10) { f2 p = {x, y}, q = {z, w}; p = __builtin_elementwise_fma(p, f2{1.0001f, 1.0001f}, f2{0.5f, 0.5f}); q = __builtin_elementwise_fma(q, f2{1.0001f, 1.0001f}, f2{0.5f, 0.5f}); x = p.x; y = p.y; z = q.x; w = q.y; }   // v_pk_fma_f32 x2 = 4 fma
        if (OP == 11) { f2 p = {x, y}, q = {z, w}; p = p * f2{1.0001f, 1.0001f}; q = q * f2{0.9999f, 0.9999f}; p = p + f2{0.5f, 0.5f}; q = q + f2{0.5f, 0.5f}; x = p.x; y = p.y; z = q.x; w = q.y; }   // v_pk_mul + v_pk_add x2
        if (OP == 12) { x = x * 1.0001f; y = y * 1.0001f; z = z * 0.9999f; w = w * 0.9999f; x = x + 0.5f; y = y + 0.5f; z = z + 0.5f; w = w + 0.5f; }   // scalar mul + add x4
        if (OP == 9) { x = __builtin_amdgcn_rcpf(x) + 1.f; y = __builtin_amdgcn_rcpf(y) + 1.f; z = __builtin_amdgcn_rcpf(z) + 1.f; w = __builtin_amdgcn_rcpf(w) + 1.f; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ __float_as_uint(x + y + z + w);
}

template <int OP> void run(const char *name, int ops_per_iter) {
    uint32_t *out; hipMalloc(&out, 256 * 8 * 256 * 4 * sizeof(uint32_t));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(256 * 8), block(256);   // 8 blocks/CU x 4 waves = 8 waves per SIMD
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, out, 1u, 1.5f);
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, out, 1u + r, 1.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double wave_insts = (double)256 * 8 * 4 * N_ITER * ops_per_iter;            // per kernel, all waves
    double simd_cycles = ms * 1e-3 * 2.4e9 * 1024;                               // 1024 SIMDs at nominal 2.4 GHz
    printf("%-28s %8.3f ms  -> %6.2f SIMD-cycles per wave-instruction-group (of %d source ops)\n", name, ms, simd_cycles / (wave_insts / ops_per_iter) / 1.0, ops_per_iter);
    hipFree(out);
}
int main() {
    run<0>("fma_f32 x4", 4); run<1>("mul_lo_u32 x4", 4); run<2>("mad u32 (a*b+c) x4", 4); run<3>("mad_u32_u24 x4", 4);
    run<4>("IEEE div x4", 4); run<5>("mul * frcp_rn x4", 4); run<6>("sqrtf x4", 4); run<7>("mul_hi_u32 ^ x4", 4); run<8>("xorshift+add x8", 8);
    run<9>("v_rcp_f32 + add x4", 4);
    run<10>("v_pk_fma_f32 x2 (4 fma)", 2); run<11>("v_pk_mul+v_pk_add x2 (8 flop-ops)", 4); run<12>("mul + add x4 unpacked (8 ops)", 8);
    return 0;
}
