// fw_libm.h — the libm-class functions of the path, restated so that the device computes the BITS the reference computes.
//
// The reference (Rust) lowers f32::log10 / sin / powf / atan2 / asin / acos to the platform's libm: on the Linux hosts it runs
// on that is glibc (this image: 2.35, x86_64, the FMA ifunc variants).  ROCm's ocml versions of the same functions differ from
// glibc's by an ulp in a few inputs per million, and an ulp in a free-flight distance, a Fresnel weight or a texel coordinate
// is enough to send a path elsewhere (tools/diverge.py: every diverging path of part2_all started with the fog medium's
// log10f).  Each function below follows glibc's algorithm operation for operation — same tables, same polynomial order, same
// intermediate precision (double where glibc uses double) — and is compiled with -ffp-contract=off, with an explicit fma()
// exactly where glibc's FMA build contracts; tests/test_libm_cpu.py compiles this header with g++ and compares every function
// with the libm of the machine it runs on (exhaustively over the inputs the renderer can produce where that is feasible),
// tests/test_gpu_libm.py compares the device with the host compilation of the same header.
//
// Compiles as plain C++ (host, for the tests) and as HIP device code (the product).  Nothing here calls libm.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define FW_LM __device__ __forceinline__
#define FW_LM_TABLE static __device__ __constant__ const
#else
#define FW_LM static inline
#define FW_LM_TABLE static const
#endif

namespace fwlm {

FW_LM uint32_t asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
FW_LM float asfloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
FW_LM uint64_t asuint64(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
FW_LM double asdouble(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }

// ---------------------------------------------------------------------------------------------------------------------
// logf — glibc sysdeps/ieee754/flt-32/e_logf.c (Szabolcs Nagy's table-driven version, in glibc since 2.28), data = __logf_data
// (read back from this image's libm.so.6: identical).  Every float in (0, 2^32) gives the bits of glibc's logf
// (tools/libm_sweep.cpp); the plain and the FMA-contracted evaluation agree on all of them, so the plain one is written.
// ---------------------------------------------------------------------------------------------------------------------
struct LogfEntry { double invc, logc; };
#define FW_LOGF_TAB                                                                                                              \
    {{0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},                             \
     {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},                             \
     {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},                                \
     {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},                             \
     {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},                                                          \
     {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},                               \
     {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},                               \
     {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}}
FW_LM_TABLE LogfEntry LOGF_TAB[16] = FW_LOGF_TAB;

FW_LM float logf_glibc(float x) {
    const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2, Ln2 = 0x1.62e42fefa39efp-1;
    uint32_t ix = asuint(x);
    if (ix == 0x3f800000u) return 0.f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix * 2u == 0u) return -__builtin_inff();                       // log(+-0) = -inf
        if (ix == 0x7f800000u) return x;                                  // log(inf) = inf
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return __builtin_nanf("");   // x < 0 or NaN
        ix = asuint(x * 0x1p23f);                                         // subnormal: normalise
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> (23 - 4)) % 16u);
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    const double invc = LOGF_TAB[i].invc, logc = LOGF_TAB[i].logc;
    const double z = (double)asfloat(iz);
    const double r = z * invc - 1.0;
    const double y0 = logc + (double)k * Ln2;
    const double r2 = r * r;
    double y = A1 * r + A2;
    y = A0 * r2 + y;
    y = y * r2 + (y0 + r);
    return (float)y;
}

// ---------------------------------------------------------------------------------------------------------------------
// log10f — glibc 2.35 sysdeps/ieee754/flt-32/e_log10f.c (the fdlibm form, float arithmetic around logf):
//   log10(x) = (n * log10_2lo + ivln10 * log(m)) + n * log10_2hi,  x = 2^n * m with the sign trick of the original.
// volume.rs:67 draws the free-flight distance with it (base 10, as written there).
// ---------------------------------------------------------------------------------------------------------------------
FW_LM float log10f_glibc(float x) {
    const float two25 = 3.3554432000e+07f, ivln10 = 4.3429449201e-01f, log10_2hi = 3.0102920532e-01f, log10_2lo = 7.9034151668e-07f;
    int32_t hx = (int32_t)asuint(x), k = 0;
    if (hx < 0x00800000) {                                               // x < 2^-126
        if ((hx & 0x7fffffff) == 0) return -__builtin_inff();            // log(+-0) = -inf  (-two25 / |x|)
        if (hx < 0) return __builtin_nanf("");                           // log(-#) = NaN
        k -= 25; x *= two25;
        hx = (int32_t)asuint(x);
    }
    if (hx >= 0x7f800000) return x + x;
    k += (hx >> 23) - 127;
    const int32_t i = (int32_t)(((uint32_t)k & 0x80000000u) >> 31);
    hx = (hx & 0x007fffff) | ((0x7f - i) << 23);
    const float y = (float)(k + i);
    x = asfloat((uint32_t)hx);
    const float z = y * log10_2lo + ivln10 * logf_glibc(x);
    return z + y * log10_2hi;
}

}  // namespace fwlm
