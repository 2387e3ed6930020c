// tools/libm_sweep.cpp — compares every function of firework_amd/csrc/fw_libm.h (compiled for the host) with the libm of the
// machine it runs on, bit for bit.  `libm_sweep quick` = the renderer's own input sets (seconds; run by tests/test_libm_cpu.py);
// `libm_sweep full` = all 2^32 floats per one-argument function, x^y for the exponents the path uses over all x, 10^9 random
// pairs for the two-argument functions (minutes per function; results quoted in fw_libm.h).
//   g++ -O2 -ffp-contract=off -mfma -o libm_sweep tools/libm_sweep.cpp -lm
#include "../firework_amd/csrc/fw_libm.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
using namespace fwlm;
static unsigned long total_bad = 0;
static bool same(float a, float b) { return asuint(a) == asuint(b) || (a != a && b != b); }
template <class F, class G> static void sweep1(const char *name, F ref, G mine, uint64_t step) {
    unsigned long bad = 0; uint32_t first = 0;
    for (uint64_t u = 0; u <= 0xffffffffull; u += step) { const float x = asfloat((uint32_t)u); if (!same(ref(x), mine(x))) { if (!bad) first = (uint32_t)u; bad++; } }
    printf("%-10s %10llu inputs  mismatches %lu%s\n", name, (unsigned long long)(0x100000000ull / step), bad, bad ? " (first shown below)" : "");
    if (bad) printf("    first: %08x\n", first);
    total_bad += bad;
}
int main(int argc, char **argv) {
    const bool full = argc > 1 && !strcmp(argv[1], "full");
    const uint64_t step = full ? 1 : 4099;            // quick: every 4099th bit pattern (a prime: all exponents, varied mantissas)
    sweep1("logf", [](float x) { return logf(x); }, [](float x) { return logf_glibc(x); }, step);
    sweep1("log10f", [](float x) { return log10f(x); }, [](float x) { return log10f_glibc(x); }, step);
    sweep1("sinf", [](float x) { return sinf(x); }, [](float x) { return sinf_glibc(x); }, step);
    sweep1("asinf", [](float x) { return asinf(x); }, [](float x) { return asinf_glibc(x); }, step);
    sweep1("acosf", [](float x) { return acosf(x); }, [](float x) { return acosf_glibc(x); }, step);
    sweep1("atanf", [](float x) { return atanf(x); }, [](float x) { return atanf_glibc(x); }, step);
    const float ys[5] = {5.0f, 1.0f / 2.2f, 0.5f, 1.0f / 2.0f, 2.4f};
    for (float y : ys) { char nm[32]; snprintf(nm, sizeof nm, "powf(x,%.3g)", y);
        sweep1(nm, [y](float x) { return powf(x, y); }, [y](float x) { return powf_glibc(x, y); }, step); }
    {   // the counter RNG's whole output set: xi = k * 2^-24, k = 0 .. 2^24-1 (the only arguments log10f ever sees on the path)
        unsigned long bad = 0;
        for (uint32_t k = 0; k < (1u << 24); k++) { const float xi = (float)k * (1.0f / 16777216.0f); if (!same(log10f(xi), log10f_glibc(xi))) bad++; }
        printf("%-10s %10u inputs  mismatches %lu   (every xi the RNG can draw)\n", "log10f(xi)", 1u << 24, bad); total_bad += bad;
    }
    uint64_t st = 88172645463325252ull; unsigned long bad2 = 0, badp = 0; const long n = full ? 1000000000L : 20000000L;
    for (long i = 0; i < n; i++) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        float y = asfloat((uint32_t)st), x = asfloat((uint32_t)(st >> 32));
        if (!same(powf(x, y), powf_glibc(x, y))) badp++;
        if (i & 1) { x = (float)((int32_t)(st >> 32)) * (1.0f / 2147483648.0f); y = (float)((int32_t)st) * (1.0f / 2147483648.0f); }
        if (!same(atan2f(y, x), atan2f_glibc(y, x))) bad2++;
    }
    printf("%-10s %10ld pairs   mismatches %lu\n%-10s %10ld pairs   mismatches %lu\n", "atan2f", n, bad2, "powf", n, badp);
    total_bad += bad2 + badp;
    printf("TOTAL mismatches %lu\n", total_bad);
    return total_bad ? 1 : 0;
}
