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
// same bytes, but wave w streams rows w, w + n_waves, w + 2 n_waves, ... (concurrent waves touch neighbouring KBs) -> the plain streaming rate
__global__ __launch_bounds__(64) void k_lin(const float4 *a, const float2 *b, const float4 *c, const float4 *d,
                                            float4 *oa, float2 *ob, float4 *oc, uint32_t cap, uint32_t n_waves) {
    uint32_t lane = threadIdx.x;
    for (uint32_t j = 0; j < cap / 64; j++) {
        size_t i = ((size_t)j * n_waves + blockIdx.x) * 64 + lane;
        float4 x = a[i], z = c[i], h = d[i]; float2 y = b[i];
        x.x += h.x + z.y; y.x += h.y; z.z += x.w;
        oa[i] = x; ob[i] = y; oc[i] = z;
    }
}
// wave-private regions whose bases are skewed: region w starts at w * (cap + skew)
__global__ __launch_bounds__(64) void k_skew(const float4 *a, const float2 *b, const float4 *c, const float4 *d,
                                             float4 *oa, float2 *ob, float4 *oc, uint32_t cap, uint32_t stride) {
    size_t base = (size_t)blockIdx.x * stride; uint32_t lane = threadIdx.x;
    float4 A = a[base + lane]; float2 B = b[base + lane]; float4 C = c[base + lane]; float4 D = d[base + lane];
    for (uint32_t j = 0; j < cap; j += 64) {
        size_t i = base + j + lane;
        float4 x = A, z = C, h = D; float2 y = B;
        if (j + 64 < cap) { A = a[i + 64]; B = b[i + 64]; C = c[i + 64]; D = d[i + 64]; }
        x.x += h.x + z.y; y.x += h.y; z.z += x.w;
        oa[i] = x; ob[i] = y; oc[i] = z;
    }
}
__global__ __launch_bounds__(256) void k_copy(const float4 *a, float4 *o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) o[i] = a[i];
}
__global__ __launch_bounds__(256) void k_read(const float4 *a, float4 *o, size_t n) {
    float4 acc = make_float4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { float4 v = a[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    if (acc.x == 12345.f) o[0] = acc;
}
int main() {
    const uint32_t n_waves = 131072, cap = 2048;                 // 268M slots like the cornell frame
    const size_t n = (size_t)n_waves * cap;
    float4 *a, *c, *d, *oa, *oc; float2 *b, *ob; char *rin, *rout;
    const size_t np = n + (size_t)n_waves * 192;                 // room for skewed region bases
    CK(hipMalloc(&a, np * 16)); CK(hipMalloc(&b, np * 8)); CK(hipMalloc(&c, np * 16)); CK(hipMalloc(&d, np * 16));
    CK(hipMalloc(&oa, np * 16)); CK(hipMalloc(&ob, np * 8)); CK(hipMalloc(&oc, np * 16));
    CK(hipMalloc(&rin, n / 64 * 3584)); CK(hipMalloc(&rout, n / 64 * 3584));
    CK(hipMemset(a, 0, n * 16)); CK(hipMemset(b, 0, n * 8)); CK(hipMemset(c, 0, n * 16)); CK(hipMemset(d, 0, n * 16)); CK(hipMemset(rin, 0, n / 64 * 3584));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double bytes = (double)n * (56 + 40);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0); hipLaunchKernelGGL(k_soa, dim3(n_waves), dim3(64), 0, 0, a, b, c, d, oa, ob, oc, cap); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); printf("SoA arrays        %7.2f ms  %6.2f TB/s\n", ms, bytes / ms / 1e9);
        hipEventRecord(e0); hipLaunchKernelGGL(k_aos, dim3(n_waves), dim3(64), 0, 0, rin, rout, cap); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("chunk-interleaved %7.2f ms  %6.2f TB/s\n", ms, bytes / ms / 1e9);
        hipEventRecord(e0); hipLaunchKernelGGL(k_lin, dim3(n_waves), dim3(64), 0, 0, a, b, c, d, oa, ob, oc, cap, n_waves); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("row-interleaved waves (plain streaming) %7.2f ms  %6.2f TB/s\n", ms, bytes / ms / 1e9);
        for (uint32_t skew : {0u, 16u, 64u, 80u, 192u}) {
            hipEventRecord(e0); hipLaunchKernelGGL(k_skew, dim3(n_waves), dim3(64), 0, 0, a, b, c, d, oa, ob, oc, cap, cap + skew); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1); printf("wave-private regions, base skew %3u slots %7.2f ms  %6.2f TB/s\n", skew, ms, bytes / ms / 1e9);
        }
        hipEventRecord(e0); hipLaunchKernelGGL(k_copy, dim3(256 * 32), dim3(256), 0, 0, a, oa, n); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("float4 copy       %7.2f ms  %6.2f TB/s (read+write)\n", ms, (double)n * 32 / ms / 1e9);
        hipEventRecord(e0); hipLaunchKernelGGL(k_read, dim3(256 * 32), dim3(256), 0, 0, a, oa, n); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("float4 read       %7.2f ms  %6.2f TB/s\n", ms, (double)n * 16 / ms / 1e9);
    }
    return 0;
}
