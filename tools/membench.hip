// membench.hip — HBM efficiency of the wavefront state streams: separate SoA arrays vs chunk-interleaved records.
// Each single-wave workgroup w streams its private region [w*cap,(w+1)*cap): reads 16+8+16+16 B/lane, writes 16+8+16 B/lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(64) void k_soa(const float4 *a, const float2 *b, const float4 *c, const float4 *d,
                                            float4 *oa, float2 *ob, float4 *oc, uint32_t cap) {
    uint32_t base = blockIdx.x * cap, lane = threadIdx.x;
    float4 A = a[base + lane]; float2 B = b[base + lane]; float4 C = c[base + lane]; float4 D = d[base + lane];
    for (uint32_t j = 0; j < cap; j += 64) {
        uint32_t i = base + j + lane;
        float4 x = A, z = C, h = D; float2 y = B;
        if (j + 64 < cap) { A = a[i + 64]; B = b[i + 64]; C = c[i + 64]; D = d[i + 64]; }
        x.x += h.x + z.y; y.x += h.y; z.z += x.w;
        oa[i] = x; ob[i] = y; oc[i] = z;
    }
}
// chunk record: [ray_a 64x16][ray_b 64x8][state 64x16][hit 64x16] = 3584 B ; out record [ray_a][ray_b][state] in the other pool (same stride)
__global__ __launch_bounds__(64) void k_aos(const char *in, char *out, uint32_t cap) {
    const uint32_t REC = 3584;
    size_t base = (size_t)blockIdx.x * (cap / 64) * REC; uint32_t lane = threadIdx.x;
    auto ld = [&](size_t r, float4 &A, float2 &B, float4 &C, float4 &D) {
        const char *p = in + base + r * REC;
        A = ((const float4 *)p)[lane]; B = ((const float2 *)(p + 1024))[lane]; C = ((const float4 *)(p + 1536))[lane]; D = ((const float4 *)(p + 2560))[lane]; };
    float4 A, C, D; float2 B; ld(0, A, B, C, D);
    for (uint32_t r = 0; r < cap / 64; r++) {
        float4 x = A, z = C, h = D; float2 y = B;
        if (r + 1 < cap / 64) ld(r + 1, A, B, C, D);
        x.x += h.x + z.y; y.x += h.y; z.z += x.w;
        char *p = out + base + (size_t)r * REC;
        ((float4 *)p)[lane] = x; ((float2 *)(p + 1024))[lane] = y; ((float4 *)(p + 1536))[lane] = z;
    }
}
int main() {
    const uint32_t n_waves = 131072, cap = 2048;                 // 268M slots like the cornell frame
    const size_t n = (size_t)n_waves * cap;
    float4 *a, *c, *d, *oa, *oc; float2 *b, *ob; char *rin, *rout;
    CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&c, n * 16)); CK(hipMalloc(&d, n * 16));
    CK(hipMalloc(&oa, n * 16)); CK(hipMalloc(&ob, n * 8)); CK(hipMalloc(&oc, n * 16));
    CK(hipMalloc(&rin, n / 64 * 3584)); CK(hipMalloc(&rout, n / 64 * 3584));
    CK(hipMemset(a, 0, n * 16)); CK(hipMemset(b, 0, n * 8)); CK(hipMemset(c, 0, n * 16)); CK(hipMemset(d, 0, n * 16)); CK(hipMemset(rin, 0, n / 64 * 3584));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double bytes = (double)n * (56 + 40);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0); hipLaunchKernelGGL(k_soa, dim3(n_waves), dim3(64), 0, 0, a, b, c, d, oa, ob, oc, cap); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); printf("SoA arrays        %7.2f ms  %6.2f TB/s\n", ms, bytes / ms / 1e9);
        hipEventRecord(e0); hipLaunchKernelGGL(k_aos, dim3(n_waves), dim3(64), 0, 0, rin, rout, cap); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("chunk-interleaved %7.2f ms  %6.2f TB/s\n", ms, bytes / ms / 1e9);
    }
    return 0;
}
