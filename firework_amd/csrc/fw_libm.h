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

// ---------------------------------------------------------------------------------------------------------------------
// sinf — glibc sysdeps/ieee754/flt-32/s_sinf.c + sincosf.h (Wilco Dijkstra's double-precision version, glibc >= 2.28), the
// FMA build (sysdeps/x86_64/fpu/multiarch/s_sinf-fma.c: same source, a*b+c contracted): all 2^32 inputs give libm's bits; the
// uncontracted evaluation differs in 12 of them.  Tables = __sincosf_table / __inv_pio4 (read back from libm.so.6).
// texture.rs:59-72 (Checker), :239-249 (Marble).
// ---------------------------------------------------------------------------------------------------------------------
struct SinCosT { double sign[4]; double hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3; };
FW_LM_TABLE SinCosT SINCOS_TAB[2] = {
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5,
     -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5,
     0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}};
FW_LM_TABLE uint32_t INV_PIO4[24] = {0xa2, 0xa2f9, 0xa2f983, 0xa2f9836e, 0xf9836e4e, 0x836e4e44, 0x6e4e4415, 0x4e441529,
                                     0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1, 0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0,
                                     0x34ddc0db, 0xddc0db62, 0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041};
FW_LM uint32_t abstop12(float x) { return (asuint(x) >> 20) & 0x7ffu; }
FW_LM float sinf_poly(double x, double x2, int tab, int n) {
    const SinCosT &p = SINCOS_TAB[tab];
    if ((n & 1) == 0) {
        const double x3 = x * x2, s1 = __builtin_fma(x2, p.s3, p.s2), x7 = x3 * x2, s = __builtin_fma(x3, p.s1, x);
        return (float)__builtin_fma(x7, s1, s);
    }
    const double x4 = x2 * x2, c2 = __builtin_fma(x2, p.c4, p.c3), c1 = __builtin_fma(x2, p.c1, p.c0), x6 = x4 * x2,
                 c = __builtin_fma(x4, p.c2, c1);
    return (float)__builtin_fma(x6, c2, c);
}
FW_LM float sinf_glibc(float y) {
    double x = (double)y, s;
    int n;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {                          // |y| < pi/4
        s = x * x;
        if (abstop12(y) < abstop12(0x1p-12f)) return y;
        return sinf_poly(x, s, 0, 0);
    }
    if (abstop12(y) < abstop12(120.0f)) {                                  // reduce_fast (x86_64: the scaled float-to-int form)
        const double r = x * SINCOS_TAB[0].hpi_inv;
        n = ((int32_t)r + 0x800000) >> 24;
        x = __builtin_fma(-(double)n, SINCOS_TAB[0].hpi, x);
        s = SINCOS_TAB[0].sign[n & 3];
        return sinf_poly(x * s, x * x, (n & 2) ? 1 : 0, n);
    }
    if (abstop12(y) < abstop12(__builtin_inff())) {                       // reduce_large
        uint32_t xi = asuint(y);
        const int sign = (int)(xi >> 31);
        const uint32_t a0 = INV_PIO4[(xi >> 26) & 15u], a4 = INV_PIO4[((xi >> 26) & 15u) + 4], a8 = INV_PIO4[((xi >> 26) & 15u) + 8];
        const int shift = (int)((xi >> 23) & 7u);
        xi = (xi & 0xffffffu) | 0x800000u;
        xi <<= shift;
        uint64_t res0 = (uint64_t)(uint32_t)(xi * a0);                      // 32-bit product, as in the source
        const uint64_t res1 = (uint64_t)xi * a4, res2 = (uint64_t)xi * a8;
        res0 = (res2 >> 32) | (res0 << 32);
        res0 += res1;
        const uint64_t nn = (res0 + (1ULL << 61)) >> 62;
        res0 -= nn << 62;
        x = (double)(int64_t)res0;
        n = (int)nn;
        x = x * 0x1.921FB54442D18p-62;
        s = SINCOS_TAB[0].sign[(n + sign) & 3];
        return sinf_poly(x * s, x * x, ((n + sign) & 2) ? 1 : 0, n);
    }
    return __builtin_nanf("");
}

// ---------------------------------------------------------------------------------------------------------------------
// powf — glibc sysdeps/ieee754/flt-32/e_powf.c (Szabolcs Nagy; log2 by table + polynomial in double, then exp2 by table), the FMA
// build.  x^5, x^(1/2.2), x^0.5, x^2.4 over all 2^32 x and 4*10^8 random (x, y) pairs give libm's bits; the uncontracted form
// differs in a few inputs per exponent.  Tables = __powf_log2_data / __exp2f_data (read back from libm.so.6).
// util.rs:69-73 (schlick: powf(5.)), render.rs:186 (gamma).
// ---------------------------------------------------------------------------------------------------------------------
FW_LM_TABLE LogfEntry POWF_LOG2_TAB[16] = {
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2}, {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},
    {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2}, {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4}, {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5},
    {0x1p+0, 0x0p+0},                              {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},  {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},
    {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}};
FW_LM_TABLE uint64_t EXP2F_TAB[32] = {
    0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51, 0x3fef72b83c7d517b, 0x3fef54873168b9aa,
    0x3fef387a6e756238, 0x3fef1e9df51fdee1, 0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d,
    0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429, 0x3feea47eb03a5585, 0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74,
    0x3feea11473eb0187, 0x3feea589994cce13, 0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d,
    0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069, 0x3fef5818dcfba487, 0x3fef7c97337b9b5f,
    0x3fefa4afa2a490da, 0x3fefd0765b6e4540};
FW_LM bool zeroinfnan(uint32_t ix) { return 2u * ix - 1u >= 2u * 0x7f800000u - 1u; }
// 0: y is not an integer, 1: odd integer, 2: even integer
FW_LM int checkint(uint32_t iy) {
    const int e = (int)(iy >> 23 & 0xffu);
    if (e < 0x7f) return 0;
    if (e > 0x7f + 23) return 2;
    if (iy & ((1u << (0x7f + 23 - e)) - 1u)) return 0;
    if (iy & (1u << (0x7f + 23 - e))) return 1;
    return 2;
}
FW_LM float powf_glibc(float x, float y) {
    const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2, A3 = -0x1.7154748bef6c8p-1,
                 A4 = 0x1.71547652ab82bp0, C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1,
                 SHIFT = 0x1.8p+47;
    uint32_t sign_bias = 0, ix = asuint(x);
    const uint32_t iy = asuint(y);
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || zeroinfnan(iy)) {
        if (zeroinfnan(iy)) {                                                // y is 0, inf or NaN
            if (2u * iy == 0u) return 1.0f;
            if (ix == 0x3f800000u) return 1.0f;
            if (2u * ix > 2u * 0x7f800000u || 2u * iy > 2u * 0x7f800000u) return x + y;
            if (2u * ix == 2u * 0x3f800000u) return 1.0f;
            if ((2u * ix < 2u * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;
            return y * y;
        }
        if (zeroinfnan(ix)) {                                                // x is 0, inf or NaN
            float x2 = x * x;
            if ((ix & 0x80000000u) && checkint(iy) == 1) x2 = -x2;
            return (iy & 0x80000000u) ? 1.f / x2 : x2;
        }
        if (ix & 0x80000000u) {                                              // finite x < 0
            const int yint = checkint(iy);
            if (yint == 0) return __builtin_nanf("");
            if (yint == 1) sign_bias = 1u << (5 + 11);
            ix &= 0x7fffffffu;
        }
        if (ix < 0x00800000u) {                                              // subnormal x
            ix = asuint(x * 0x1p23f);
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        }
    }
    // log2_inline
    const uint32_t tmp = ix - 0x3f330000u, top = tmp & 0xff800000u, iz = ix - top;
    const int i = (int)((tmp >> (23 - 4)) % 16u), k = (int32_t)top >> 23;
    const double invc = POWF_LOG2_TAB[i].invc, logc = POWF_LOG2_TAB[i].logc, z = (double)asfloat(iz);
    const double r = __builtin_fma(z, invc, -1.0), y0 = logc + (double)k, r2 = r * r;
    double yy = __builtin_fma(A0, r, A1);
    const double p = __builtin_fma(A2, r, A3), r4 = r2 * r2;
    double q = __builtin_fma(A4, r, y0);
    q = __builtin_fma(p, r2, q);
    yy = __builtin_fma(yy, r4, q);
    const double ylogx = (double)y * yy;
    if ((asuint64(ylogx) >> 47 & 0xffffu) >= asuint64(126.0) >> 47) {        // |y * log2(x)| >= 126
        if (ylogx > 0x1.fffffffd1d571p+6) return sign_bias ? -__builtin_inff() : __builtin_inff();
        if (ylogx <= -150.0) return sign_bias ? -0.0f : 0.0f;
    }
    // exp2_inline
    double kd = ylogx + SHIFT;
    const uint64_t ki = asuint64(kd);
    kd -= SHIFT;
    const double rr = ylogx - kd;
    uint64_t t = EXP2F_TAB[ki % 32u];
    t += (ki + sign_bias) << (52 - 5);
    const double sc = asdouble(t), zz = __builtin_fma(C0, rr, C1), rr2 = rr * rr;
    double y2 = __builtin_fma(C2, rr, 1.0);
    y2 = __builtin_fma(zz, rr2, y2);
    return (float)(y2 * sc);
}

// ---------------------------------------------------------------------------------------------------------------------
// asinf / acosf / atanf / atan2f — glibc 2.35 e_asinf.c (the 2011 Chebyshev variant), e_acosf.c, s_atanf.c, e_atan2f.c: the
// fdlibm float code, plain float arithmetic, no FMA build exists.  asinf, acosf, atanf: all 2^32 inputs give libm's bits;
// atan2f: 10^9 random pairs (half of them in [-1,1]^2).  sphere.rs:22-29 (atan2, asin), cone.rs (acos), cylinder.rs / disk.rs (atan2).
// ---------------------------------------------------------------------------------------------------------------------
FW_LM float asinf_glibc(float x) {
    const float one = 1.0f, huge = 1.000e+30f, pio2_hi = 1.57079637050628662109375f, pio2_lo = -4.37113900018624283e-8f,
                pio4_hi = 0.785398185253143310546875f, p0 = 1.666675248e-1f, p1 = 7.495297643e-2f, p2 = 4.547037598e-2f,
                p3 = 2.417951451e-2f, p4 = 4.216630880e-2f;
    float t, w, p, q, c, r, s;
    const int32_t hx = (int32_t)asuint(x), ix = hx & 0x7fffffff;
    if (ix == 0x3f800000) return x * pio2_hi + x * pio2_lo;
    if (ix > 0x3f800000) return (x - x) / (x - x);
    if (ix < 0x3f000000) {
        if (ix < 0x32000000) { if (huge + x > one) return x; }
        else { t = x * x; w = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4)))); return x + x * w; }
    }
    w = one - __builtin_fabsf(x);
    t = w * 0.5f;
    p = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4))));
    s = __builtin_sqrtf(t);
    if (ix >= 0x3F79999A) t = pio2_hi - (2.0f * (s + s * p) - pio2_lo);
    else {
        w = asfloat(asuint(s) & 0xfffff000u);
        c = (t - w * w) / (s + w);
        r = p;
        p = 2.0f * s * r - (pio2_lo - 2.0f * c);
        q = pio4_hi - 2.0f * w;
        t = pio4_hi - (p - q);
    }
    return hx > 0 ? t : -t;
}
FW_LM float acosf_glibc(float x) {
    const float one = 1.0f, pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f, pS0 = 1.6666667163e-01f,
                pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f, pS4 = 7.9153501429e-04f,
                pS5 = 3.4793309169e-05f, qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f, qS3 = -6.8828397989e-01f,
                qS4 = 7.7038154006e-02f;
    float z, p, q, r, w, s, c, df;
    const int32_t hx = (int32_t)asuint(x), ix = hx & 0x7fffffff;
    if (ix == 0x3f800000) return hx > 0 ? 0.0f : pi + 2.0f * pio2_lo;
    if (ix > 0x3f800000) return (x - x) / (x - x);
    if (ix < 0x3f000000) {
        if (ix <= 0x23000000) return pio2_hi + pio2_lo;
        z = x * x;
        p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    if (hx < 0) {
        z = (one + x) * 0.5f;
        p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        s = __builtin_sqrtf(z);
        r = p / q;
        w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    }
    z = (one - x) * 0.5f;
    s = __builtin_sqrtf(z);
    df = asfloat(asuint(s) & 0xfffff000u);
    c = (z - df * df) / (s + df);
    p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    r = p / q;
    w = r * s + c;
    return 2.0f * (df + w);
}
FW_LM float atanf_glibc(float x) {
    const float one = 1.0f, huge = 1.0e30f;
    const float hi0 = 4.6364760399e-01f, hi1 = 7.8539812565e-01f, hi2 = 9.8279368877e-01f, hi3 = 1.5707962513e+00f;
    const float lo0 = 5.0121582440e-09f, lo1 = 3.7748947079e-08f, lo2 = 3.4473217170e-08f, lo3 = 7.5497894159e-08f;
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f,
                aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f,
                aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    float w, s1, s2, z, hi = 0.f, lo = 0.f;
    const int32_t hx = (int32_t)asuint(x), ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) {                                                 // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;
        return hx > 0 ? hi3 + lo3 : -hi3 - lo3;
    }
    if (ix < 0x3ee00000) {                                                  // |x| < 0.4375
        if (ix < 0x31000000) { if (huge + x > one) return x; }
        id = -1;
    } else {
        x = __builtin_fabsf(x);
        if (ix < 0x3f980000) {
            if (ix < 0x3f300000) { id = 0; hi = hi0; lo = lo0; x = (2.0f * x - one) / (2.0f + x); }
            else { id = 1; hi = hi1; lo = lo1; x = (x - one) / (x + one); }
        } else {
            if (ix < 0x401c0000) { id = 2; hi = hi2; lo = lo2; x = (x - 1.5f) / (one + 1.5f * x); }
            else { id = 3; hi = hi3; lo = lo3; x = -1.0f / x; }
        }
    }
    z = x * x;
    w = z * z;
    s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    z = hi - ((x * (s1 + s2) - lo) - x);
    return hx < 0 ? -z : z;
}
FW_LM float atan2f_glibc(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    float z;
    const int32_t hx = (int32_t)asuint(x), ix = hx & 0x7fffffff, hy = (int32_t)asuint(y), iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return atanf_glibc(y);
    const int32_t m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) { if (m < 2) return y; return m == 2 ? pi + tiny : -pi - tiny; }
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) return m == 0 ? pi_o_4 + tiny : (m == 1 ? -pi_o_4 - tiny : (m == 2 ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny));
        return m == 0 ? 0.0f : (m == 1 ? -0.0f : (m == 2 ? pi + tiny : -pi - tiny));
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int32_t k = (iy - ix) >> 23;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = atanf_glibc(__builtin_fabsf(y / x));
    if (m == 0) return z;
    if (m == 1) return asfloat(asuint(z) ^ 0x80000000u);
    if (m == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
}

}  // namespace fwlm
