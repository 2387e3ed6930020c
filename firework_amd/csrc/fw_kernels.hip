// fw_kernels.hip — the wavefront path tracer's kernels for gfx950 (MI355X, CDNA4).
//
// Replaces the body of the reference's `Renderer::render()` pixel loop (src/render.rs:127-196):
//
//   k_raygen      render_pixel's loop head + Camera::ray          render.rs:172-180, camera.rs:109-116
//   k_extend_linear / k_extend_tlas / k_extend_tlas_park + k_blas (/ k_extend_bvh)
//                 root.hit(): linear scene / TLAS + per-object     scene.rs:137-149,235-266, bvh.rs:115-151,
//                 transform + primitive intersectors + mesh BLAS   objects/*.rs
//                 (one body, two entry points with their own register budgets)
//   k_shade       Material::emit/scatter, Texture::sample,         render.rs:19-31, material.rs, texture.rs,
//                 Environment::sample, + stream compaction         environment.rs
//   k_accumulate  `total_color += color(..)` in sample order      render.rs:181
//   k_resolve     /spp, powf(1/gamma), clamp, Color::from          render.rs:184-190, util.rs:14-23
//   k_bounce      k_extend + k_shade of one segment in one launch (FIREWORK_FUSED=1; measured slower, kept for A/B)
//
// One lane = one path.  Path state lives in HBM as SoA arrays.  The path pool is split into WAVE-PRIVATE
// queues: wavefront w owns slots [w*cap, (w+1)*cap) of every array and compacts its
// survivors there with a wave ballot + mbcnt prefix — no atomics, no barriers, no cross-wave traffic, and
// a deterministic slot order.  All paths of a launch are at the same depth, so the depth is a kernel
// argument, not state.  Every random draw is a pure function of
// (seed, pixel, sample, dimension), so the image does not depend on scheduling.
//
// Numerics: compiled with -ffp-contract=off; +,-,*,/ and sqrt are IEEE-exact and written in the same
// association as the reference's expressions, so hit/miss decisions follow the CPU oracle bit for
// bit; libm-class functions (sin, atan2, asin, acos, log10, pow) are glibc's algorithms restated (fw_libm.h): the same bits.
#include "fw_device.h"
#include "fw_libm.h"
#include <atomic>
#include <type_traits>

namespace fw {

// ------------------------------------------------------------------------------------------------
// small vector type (ultraviolet::Vec3 semantics: dot/cross/mag_sq via fma, `/` component-wise)
// ------------------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 mk(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return {fmaf(a.y, b.z, -a.z * b.y), fmaf(a.z, b.x, -a.x * b.z), fmaf(a.x, b.y, -a.y * b.x)};
}
__device__ __forceinline__ float mag_sq(V3 a) { return dot(a, a); }
__device__ __forceinline__ float comp(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
__device__ __forceinline__ V3 ld3(const float *p) { return {p[0], p[1], p[2]}; }

// ------------------------------------------------------------------------------------------------
// Division and square root.  hipcc lowers `a / b` to v_div_scale x2, v_rcp, 6 fma/mul, v_div_fmas,
// v_div_fixup (~41 SIMD-cycles per wave64 on MI355X, tools/microbench.hip) and sqrtf to ~53.  fdiv/fsqrt run
// the same Newton/residual chain without the scaling steps: identical, correctly rounded bits whenever
// v_div_scale would be the identity (every operand magnitude a renderer produces; exponents within ~2^+-96),
// v_div_fixup still supplies the IEEE results for zero / infinite / NaN operands (x/0 = +-inf is what the
// reference's range tests rely on), and a denominator shared by several quotients is inverted once.
// Build with -DFW_FAST_DIV=0 to use the compiler's expansion everywhere (A/B: tests pass bit-identically).
// ------------------------------------------------------------------------------------------------
#ifndef FW_FAST_DIV
#define FW_FAST_DIV 1
#endif
#ifndef FW_WB
#define FW_WB 64   // threads per workgroup of the queue kernels: single-wave workgroups retire independently
#endif
struct Rcp { float b, r; };
__device__ __forceinline__ Rcp make_rcp(float b) {
    float r = __builtin_amdgcn_rcpf(b);
    r = fmaf(fmaf(-b, r, 1.0f), r, r);
    return Rcp{b, r};
}
__device__ __forceinline__ float fdiv(float a, const Rcp &c) {
#if FW_FAST_DIV
    float q = a * c.r;
    q = fmaf(fmaf(-c.b, q, a), c.r, q);
    q = fmaf(fmaf(-c.b, q, a), c.r, q);
    return __builtin_amdgcn_div_fixupf(q, c.b, a);
#else
    return a / c.b;
#endif
}
__device__ __forceinline__ float fdiv(float a, float b) { return fdiv(a, make_rcp(b)); }
// v_sqrt_f32 is within 1 ulp; pick among s-1ulp, s, s+1ulp by the sign of the exact residuals (the test the
// compiler's own expansion uses).  0, inf and NaN pass through unchanged (their residual tests are false).
__device__ __forceinline__ float fsqrt(float x) {
#if FW_FAST_DIV
    float s = __builtin_amdgcn_sqrtf(x);
    float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
    float vd = fmaf(-sd, s, x), vu = fmaf(-su, s, x);
    s = (vd <= 0.f) ? sd : s;
    s = (vu > 0.f) ? su : s;
    return s;
#else
    return sqrtf(x);
#endif
}
__device__ __forceinline__ V3 operator/(V3 a, float s) { Rcp c = make_rcp(s); return {fdiv(a.x, c), fdiv(a.y, c), fdiv(a.z, c)}; }
__device__ __forceinline__ float mag(V3 a) { return fsqrt(mag_sq(a)); }
__device__ __forceinline__ V3 normalized(V3 a) { return a / mag(a); }

struct Ray { V3 o, d; };
__device__ __forceinline__ V3 ray_point(const Ray &r, float t) { return r.o + t * r.d; }

// Phase statistics (debug builds only: tools/build_variant.sh phase -DFW_PHASE_STATS; tools/phase_stats.py): for up to PH_N code
// sections of a kernel the cycles the WAVE spent in them (clock64 deltas, each active lane adding delta / active lanes), the
// LANE-cycles (every active lane adding delta: lane-cycles / (64 x wave cycles) is the section's lane utilisation) and how often
// the wave entered them.  Where the walks and k_shade lose their lanes, section by section (round 5).
#ifdef FW_PHASE_STATS
constexpr int PH_N = 20;
__device__ unsigned long long g_phase[2 * 3 * PH_N];     // [kernel class: 0 the wide walks, 1 k_shade][section][lane-cycles, wave cycles, entries]
struct Phase {
    float lane[PH_N], wave[PH_N], cnt[PH_N];
    __device__ __forceinline__ void init() { for (int k = 0; k < PH_N; k++) { lane[k] = 0.f; wave[k] = 0.f; cnt[k] = 0.f; } }
    __device__ __forceinline__ void add(int k, long long t0) {      // called by the section's active lanes, at its end
        const float d = (float)(clock64() - t0), pc = (float)__popcll(__ballot(1));
        lane[k] += d; wave[k] += d / pc; cnt[k] += 1.f / pc;
    }
    __device__ __forceinline__ void count(int k) { const float pc = (float)__popcll(__ballot(1)); lane[k] += 1.f; cnt[k] += 1.f / pc; }   // lanes / entries only
    __device__ __forceinline__ void flush(int cls) {
        unsigned long long *g_phase_ = g_phase + cls * 3 * PH_N;
        for (int k = 0; k < PH_N; k++) {
            float a = lane[k], b = wave[k], c = cnt[k];
            for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); c += __shfl_xor(c, o); }
            if ((threadIdx.x & 63u) == 0) {
                if (a > 0.f) atomicAdd(&g_phase_[3 * k], (unsigned long long)(a + 0.5f));
                if (b > 0.f) atomicAdd(&g_phase_[3 * k + 1], (unsigned long long)(b + 0.5f));
                if (c > 0.f) atomicAdd(&g_phase_[3 * k + 2], (unsigned long long)(c + 0.5f));
            }
        }
    }
};
#define PH_DECL Phase ph_; ph_.init(); const long long ph_k0_ = clock64()      // the kernel's own span goes to section PH_N - 2
#define PH_T0 const long long ph_t0_ = clock64()
#define PH_ADD(k) ph_.add((k), ph_t0_)
#define PH_COUNT(k) ph_.count(k)
#define PH_FLUSH(cls) do { ph_.add(PH_N - 2, ph_k0_); ph_.flush(cls); } while (0)
#define PH_ARG , Phase &ph_
#define PH_PASS , ph_
#else
#define PH_DECL do { } while (0)
#define PH_T0 do { } while (0)
#define PH_ADD(k) do { } while (0)
#define PH_COUNT(k) do { } while (0)
#define PH_FLUSH(cls) do { } while (0)
#define PH_ARG
#define PH_PASS
#endif

// ------------------------------------------------------------------------------------------------
// counter RNG: pcg4d(pixel, sample, dimension, seed32) -> 4 x u32; float = (u >> 8) * 2^-24
// ------------------------------------------------------------------------------------------------
enum : uint32_t { P_JITTER = 0, P_LENS = 1, P_SCATTER = 2, P_FRESNEL = 3, P_VOLUME = 4 };

struct RngKey { uint32_t pixel, sample, seed32; };

__device__ __forceinline__ uint4 pcg4d(uint4 v) {
    v.x = v.x * 1664525u + 1013904223u; v.y = v.y * 1664525u + 1013904223u;
    v.z = v.z * 1664525u + 1013904223u; v.w = v.w * 1664525u + 1013904223u;
    v.x += v.y * v.w; v.y += v.z * v.x; v.z += v.x * v.y; v.w += v.y * v.z;
    v.x ^= v.x >> 16; v.y ^= v.y >> 16; v.z ^= v.z >> 16; v.w ^= v.w >> 16;
    v.x += v.y * v.w; v.y += v.z * v.x; v.z += v.x * v.y; v.w += v.y * v.z;
    return v;
}
__device__ __forceinline__ float u2f(uint32_t u) { return (float)(u >> 8) * (1.0f / 16777216.0f); }
__device__ __forceinline__ uint4 draw(const RngKey &k, uint32_t purpose, uint32_t segment, uint32_t index) {
    return pcg4d(make_uint4(k.pixel, k.sample, purpose | (segment << 3) | (index << 7), k.seed32));
}

#ifndef FW_MAX_REJECT
#define FW_MAX_REJECT 1024
#endif
constexpr uint32_t MAX_REJECT = FW_MAX_REJECT;   // exit condition for the rejection loops (P(miss 1024x) ~ 1e-330)

// util.rs:36-43
__device__ __forceinline__ V3 random_in_unit_sphere(const RngKey &k, uint32_t segment PH_ARG) {
    V3 p = mk(0.f, 0.f, 0.f);
    for (uint32_t attempt = 0; attempt < MAX_REJECT; attempt++) {
        PH_COUNT(19);
        uint4 u = draw(k, P_SCATTER, segment, attempt);
        p = 2.0f * mk(u2f(u.x), u2f(u.y), u2f(u.z)) - mk(1.f, 1.f, 1.f);
        if (mag_sq(p) < 1.0f) break;
    }
    return p;
}
// util.rs:45-52
__device__ __forceinline__ V3 random_in_unit_disk(const RngKey &k) {
    V3 p = mk(0.f, 0.f, 0.f);
    for (uint32_t attempt = 0; attempt < MAX_REJECT; attempt++) {
        uint4 u = draw(k, P_LENS, 0, attempt);
        p = 2.0f * mk(u2f(u.x), u2f(u.y), 0.f) - mk(1.f, 1.f, 0.f);
        if (dot(p, p) < 1.0f) break;
    }
    return p;
}

// A path is named by its HOME SLOT: the slot k_raygen put it in, home = wave * cap + chunk * 64 + lane.  Its linear id
// (chunk * n_waves + wave) * 64 + lane = s_local * n_pixels + p_local gives the pixel (render.rs:127) and the sample.
// Radiance is deposited at sample_rad[home]: a wave's deposits stay inside its own 16 B * cap window instead of being
// strewn over the whole [sample][pixel] array (k_shade 26.9 -> see DESIGN §5), and k_accumulate walks the same mapping.
// s_local < 2^12, so a float quotient is off by at most one: fix it up instead of a 32-bit integer division.
__device__ __forceinline__ RngKey key_of_linear(const DFrame &f, uint32_t linear) {
    uint32_t s_local = (uint32_t)((float)linear * f.inv_n_pixels);
    int32_t p_local = (int32_t)(linear - s_local * f.n_pixels);
    if (p_local < 0) { s_local--; p_local += (int32_t)f.n_pixels; }
    else if ((uint32_t)p_local >= f.n_pixels) { s_local++; p_local -= (int32_t)f.n_pixels; }
    RngKey k;
#ifdef FW_ABL_NO_PIXGATHER      // timing ablation (wrong frames): no lookup of the pixel id
    k.pixel = (uint32_t)p_local;
#else
    k.pixel = f.pixel_ids ? f.pixel_ids[p_local] : (uint32_t)p_local;
#endif
    k.sample = f.sample0 + s_local;
    k.seed32 = f.seed32;
    return k;
}
// Bit of a path in DFrame.dep_bits ("this path wrote a radiance record") in the PIXEL-major layout, p_local * spp_batch +
// s_local: one pixel's samples are consecutive bits and k_accumulate reads them as a few words (the records themselves stay
// at the path's home slot).  Used for small pixel sets (DFrame.dep_pixel_major: a rank's share of a tiled frame), where
// k_accumulate has too few threads to hide a chain of one bit-word load per sample (1/8 of cornell: 0.47 -> 0.13 ms, the
// rank's frame 6.5 -> 6.1 ms); a whole frame keeps the slot-major bits, bit = home slot, which cost k_shade nothing to address
// (the decode below is 12 instructions in a kernel that is bound by them: +0.3 ms per cornell frame) and leave k_accumulate
// bound by its 16-byte record gathers out of a multi-GB array either way (0.5-0.6 ms from 1/4 of the frame upwards).
__device__ __forceinline__ uint32_t dep_bit_of(const DFrame &f, uint32_t home) {
    const uint32_t lane = home & 63u, c = (home >> 6) & ((1u << f.q_shift) - 1u), w = home >> (f.q_shift + 6u);
    const uint32_t linear = (c * f.q_n_waves + w) * 64u + lane;
    uint32_t s_local = (uint32_t)((float)linear * f.inv_n_pixels);
    int32_t p_local = (int32_t)(linear - s_local * f.n_pixels);
    if (p_local < 0) { s_local--; p_local += (int32_t)f.n_pixels; }
    else if ((uint32_t)p_local >= f.n_pixels) { s_local++; p_local -= (int32_t)f.n_pixels; }
    return (uint32_t)p_local * f.spp_batch + s_local;
}
__device__ __forceinline__ RngKey key_of(const DFrame &f, uint32_t home) {
    const uint32_t lane = home & 63u, c = (home >> 6) & ((1u << f.q_shift) - 1u), w = home >> (f.q_shift + 6u);
    return key_of_linear(f, (c * f.q_n_waves + w) * 64u + lane);
}

// Streaming cache policy.  Every queue element is written once and read once, ~30 GB apart, and the radiance deposits
// are scattered 16-byte pieces: with the `nt` bit these do not linger in L2 as partially filled lines
// (k_shade 29.8 -> 25.5 ms for the deposits alone).  FW_NT_* switch the three classes for A/B runs.
#ifndef FW_NT_RAD
#define FW_NT_RAD 1
#endif
#ifndef FW_NT_ST
#define FW_NT_ST 0
#endif
#ifndef FW_NT_LD
#define FW_NT_LD 0
#endif
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_nt(float4 *p, float4 v) { f4v x = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(x, reinterpret_cast<f4v *>(p)); }
__device__ __forceinline__ void st_nt(float2 *p, float2 v) { f2v x = {v.x, v.y}; __builtin_nontemporal_store(x, reinterpret_cast<f2v *>(p)); }
__device__ __forceinline__ float4 ld_nt(const float4 *p) { f4v x = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p)); return make_float4(x.x, x.y, x.z, x.w); }
__device__ __forceinline__ float2 ld_nt(const float2 *p) { f2v x = __builtin_nontemporal_load(reinterpret_cast<const f2v *>(p)); return make_float2(x.x, x.y); }
template <class T> __device__ __forceinline__ void qst(T *p, T v) { if (FW_NT_ST) st_nt(p, v); else *p = v; }
template <class T> __device__ __forceinline__ T qld(const T *p) { if (FW_NT_LD) return ld_nt(p); return *p; }

// Camera rays of a pinhole camera (aperture 0) all start at the camera position (camera.rs:109-116 adds an offset of
// exactly 0), so segment 0 stores only the direction, in ray_a, and ray_b is neither written nor read: 8 B less per
// access, three accesses per sample.  The host sets pinhole0 only when no coordinate of the position is a negative zero (-0 + 0 = +0).
__device__ __forceinline__ bool short_rays(const DFrame &f, int segment) { return segment == 0 && f.pinhole0 != 0u; }
__device__ __forceinline__ float2 load_ray_b(const DPaths &in, uint32_t i, const DFrame &f, int segment) {
    if (short_rays(f, segment)) return make_float2(0.f, 0.f);
    return qld(&in.ray_b[i]);
}
__device__ __forceinline__ Ray make_ray(float4 ra, float2 rb, const DFrame &f, int segment) {
    if (short_rays(f, segment)) return Ray{mk(f.cam_pos[0], f.cam_pos[1], f.cam_pos[2]), mk(ra.x, ra.y, ra.z)};
    return Ray{mk(ra.x, ra.y, ra.z), mk(ra.w, rb.x, rb.y)};
}
// Path state of slot i in segment `segment`.  A camera path starts with throughput (1,1,1) and its home slot IS its slot,
// so k_raygen does not write segment-0 state and its readers do not load it (32 B per sample less HBM traffic).
__device__ __forceinline__ float4 load_state(const DPaths &in, uint32_t i, int segment) {
    if (segment == 0) return make_float4(1.f, 1.f, 1.f, __uint_as_float(i));
    return qld(&in.state[i]);
}
// CHAIN state (DFrame.chain_bits != 0), 8 bytes per path instead of 16: (the materials the path has scattered on so far, chain_bits
// bits per segment, segment 0 lowest | home slot).  Where every attenuation is a constant of its material (no texture that varies:
// cornell, suzanne, hdri, volume) the running product beta.rgb is replaced by the material ids, and the ONE path in 30 that ends in
// light multiplies them out when it deposits — back to front, a0 * (a1 * (... * emit)), which is the association of the reference's
// recursion (render.rs:23-28): the pre-gamma sample sums are then the oracle's bit for bit, and k_shade, which is bound by its queue
// streams, moves 8 bytes less per ray in each direction.  Returned as (chain bits, -, -, home bits).
__device__ __forceinline__ float4 load_state_chain(const DPaths &in, uint32_t i, int segment) {
    if (segment == 0) return make_float4(0.f, 0.f, 0.f, __uint_as_float(i));
    const float2 v = qld(&reinterpret_cast<const float2 *>(in.state)[i]);
    return make_float4(v.x, 0.f, 0.f, v.y);
}
// the home slot alone (bits, as a float), for the kernels that need the path's RNG key (a medium's draw) and nothing else of the state
__device__ __forceinline__ float load_home(const DPaths &in, uint32_t i, const DFrame &f, int segment) {
    if (segment == 0) return __uint_as_float(i);
    if (f.chain_bits) return reinterpret_cast<const float2 *>(in.state)[i].y;
    return in.state[i].w;
}

// ------------------------------------------------------------------------------------------------
// The exact walk's flag rule (fw_device.h DExact): does this ray's result depend on how the trees are walked?
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool ill_direction(float dx, float dy, float dz, float shear) {
    // util.rs:104-118 picks the SIGNED largest component as the shear axis of mesh.rs:147-162
    const float dk = dx > dy ? (dz > dx ? dz : dx) : (dz > dy ? dz : dy);
    const float am = fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz));
    return fabsf(dk) < am * shear;        // DExact.shear: 2^-10 (option EXACT_SHEAR_LOG2)
}
// The SOFT class (round 4).  Between "ill-conditioned enough for the exact list" (2^-10) and well-conditioned lies a band of rays
// whose triangle hits carry errors of ~(max|d| / |d_kz|) ulps: a t that can precede the entry of its own triangle's box by more
// than cull_bound's slack.  The reference never culls, so it finds such a hit; a walk that culls against the best t so far may not
// (two paths of suzanne's 1.1e9 at 512 spp, shear ratios 2^-7.2 and 2^-9.97: tools/full_parity.py, gpurun_out/r04b).  A ray in this
// band — 2 % of uniformly distributed directions — is walked WITHOUT culling: with the item boxes being the reference's leaf-node
// boxes (fw_runtime.cpp) the walk then tests exactly the reference's set of items, and the smallest t with the rank rule on a tie
// is the reference's winner.  Costs such a ray a few times its culled walk, inside the ordinary kernels.
constexpr float NO_CULL = 3.40282347e+38f;
// shear = DScene.soft_shear (2^-5 unless option SOFT_SHEAR_LOG2 says otherwise; 0: the class is off)
__device__ __forceinline__ bool soft_direction(float dx, float dy, float dz, float shear) {
    const float dk = dx > dy ? (dz > dx ? dz : dx) : (dz > dy ? dz : dy);
    const float am = fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz));
    return fabsf(dk) < am * shear;
}
// at the level of the TLAS: in the world frame or in the frame of any rotated mesh (DExact.frames)
__device__ __forceinline__ bool soft_ray(const DExact &ex, V3 d, float shear) {
    if (soft_direction(d.x, d.y, d.z, shear)) return true;
    for (uint32_t k = 0; k < ex.n_frames; k++) {
        const float *m = ex.frames[k];
        if (soft_direction(m[0] * d.x + m[3] * d.y + m[6] * d.z, m[1] * d.x + m[4] * d.y + m[7] * d.z, m[2] * d.x + m[5] * d.y + m[8] * d.z, shear)) return true;
    }
    return false;
}
// A ray with a NaN in it passes every box (the slab test's min / max drop NaNs) and "hits" triangles with a NaN t that no comparison
// rejects (mesh.rs:164-177) and that `!(best < t)` lets replace and be replaced: its result depends on the order of the tests like
// nothing else, and an ordinary walk spends a millisecond on it, one lane through a whole tree while its launch waits (suzanne:
// one such path made four k_blas_lds launches of ~150 us last ~1 ms each).  With the exact walk on it is flagged (needs_exact)
// and the ordinary walks leave it out (skip_ray).  (inf - inf counts: no harm.)
__device__ __forceinline__ bool nan_ray(float ox, float oy, float oz, float dx, float dy, float dz) {
    const float s = ((ox + oy) + oz) + ((dx + dy) + dz);
    return s != s;
}
__device__ __forceinline__ bool needs_exact(const DExact &ex, float ox, float oy, float oz, float dx, float dy, float dz) {
    if (ex.mode & 4u) return true;
    if (nan_ray(ox, oy, oz, dx, dy, dz)) return true;
    if (ex.mode & 1u) {
        if (ill_direction(dx, dy, dz, ex.shear)) return true;
        for (uint32_t k = 0; k < ex.n_frames; k++) {            // inv_rotation_mat * d = rows of rotation_mat as columns (rot_inv)
            const float *m = ex.frames[k];
            if (ill_direction(m[0] * dx + m[3] * dy + m[6] * dz, m[1] * dx + m[4] * dy + m[7] * dz, m[2] * dx + m[5] * dy + m[8] * dz, ex.shear)) return true;
        }
    }
    if (ex.mode & 2u) {
        const float far = fmaxf(fmaxf(fabsf(ox - ex.far_c[0]), fabsf(oy - ex.far_c[1])), fabsf(oz - ex.far_c[2]));
        if (far > ex.far_r) {       // a far origin: does the ray come near the small objects at all?  (conservative slab test)
            const float ix = __builtin_amdgcn_rcpf(dx), iy = __builtin_amdgcn_rcpf(dy), iz = __builtin_amdgcn_rcpf(dz);
            float t0 = (ex.box_lo[0] - ox) * ix, t1 = (ex.box_hi[0] - ox) * ix;
            float tn = fminf(t0, t1), tf = fmaxf(t0, t1);
            t0 = (ex.box_lo[1] - oy) * iy; t1 = (ex.box_hi[1] - oy) * iy;
            tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
            t0 = (ex.box_lo[2] - oz) * iz; t1 = (ex.box_hi[2] - oz) * iz;
            tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
            if (!(tf < tn * (1.f - 1e-3f)) && !(tf < 0.f)) return true;
        }
    }
    return false;
}
template <class R>
__device__ __forceinline__ bool skip_ray(const DExact &ex, const R &r) { return ex.mode != 0u && nan_ray(r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z); }
// The flagged rays of one wave-chunk go on the list of their segment (DExact): k_extend_exact walks them again, literally, after
// the ordinary k_extend and before k_shade.  Called by every lane of the wave (fl = false where there is nothing to flag): one
// device-wide atomic per wave and chunk that flags anything.  slot: the ray's slot in the queue of `segment`.
__device__ __forceinline__ void flag_exact(const DExact &ex, bool fl, uint32_t slot, int segment) {
    const unsigned long long m = __ballot(fl);
    if (!m) return;
    const uint32_t lane = threadIdx.x & 63u, leader = (uint32_t)__ffsll((long long)m) - 1u;
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    uint32_t at = 0;
    if (lane == leader) at = atomicAdd(&ex.count[segment], (uint32_t)__popcll(m));
    at = (uint32_t)__shfl((int)at, (int)leader) + rank;
    if (fl && at < ex.cap) ex.slots[segment & 1][at] = slot;
}

// ------------------------------------------------------------------------------------------------
// K1  ray generation
// ------------------------------------------------------------------------------------------------
constexpr int WB = FW_WB;
// Traversal statistics (debug builds only: make CXXFLAGS_EXTRA=-DFW_TRAV_STATS): per phase, useful lane-iterations and the
// lane-slots the wave spent on them (64 per executed iteration) -> lane utilisation of the node walks and leaf tests.
#ifdef FW_TRAV_STATS
__device__ unsigned long long g_trav[8];
__shared__ float g_ts[8 * 64];
#define TS_TICK(k) do { float pc_ = (float)__popcll(__ballot(1)); g_ts[(k) * 64 + (threadIdx.x & 63u)] += 1.f; g_ts[((k) + 1) * 64 + (threadIdx.x & 63u)] += 64.f / pc_; } while (0)
#define TS_COUNT(k, pred) do { g_ts[(k) * 64 + (threadIdx.x & 63u)] += (pred) ? 1.f : 0.f; g_ts[((k) + 1) * 64 + (threadIdx.x & 63u)] += 1.f; } while (0)
#define TS_BEGIN() do { for (int k_ = 0; k_ < 8; k_++) g_ts[k_ * 64 + (threadIdx.x & 63u)] = 0.f; } while (0)
#define TS_END() do { for (int k_ = 0; k_ < 8; k_++) { float v_ = g_ts[k_ * 64 + (threadIdx.x & 63u)]; for (int o_ = 32; o_ > 0; o_ >>= 1) v_ += __shfl_xor(v_, o_); \
                      if ((threadIdx.x & 63u) == 0) atomicAdd(&g_trav[k_], (unsigned long long)(v_ + 0.5f)); } } while (0)
#else
#define TS_TICK(k) do { } while (0)
#define TS_COUNT(k, pred) do { } while (0)
#define TS_BEGIN() do { } while (0)
#define TS_END() do { } while (0)
#endif

#ifndef FW_XCD_SWIZZLE
#define FW_XCD_SWIZZLE 1
#endif
#ifndef FW_XCD_MIN
#define FW_XCD_MIN 8192u        // queues from which a launch takes the XCD-contiguous assignment
#endif
// Which queue a single-wave workgroup takes.  Workgroups are dealt round-robin to the 8 XCDs (block b -> XCD b mod 8), so with w = b the
// waves resident on one XCD own every 8th queue — eight XCDs' address translation and L2s all working on the same stretch of every path
// array.  Round 5: XCD x takes the contiguous eighth [x n/8, (x+1) n/8) of the queues instead, so the windows its waves stream are
// neighbours.  Interleaved on one box (profiles/r05d_shade_ablations.txt): cornell 34.0 -> 32.3 ms (k_extend 14.3 -> 13.75, k_shade 19.3 ->
// 19.05 exclusive, and the two batches in flight overlap better), hdri 33.9 -> 33.2; on a second box (profiles/r05e_xcd_ab.txt) volume, suzanne and teapot
// gain 1-1.5 % and cornell's runs scatter by more than the difference.  Any bijection is correct: queues are private.
__device__ __forceinline__ uint32_t wave_index() {
#if FW_XCD_SWIZZLE
    // (not for small launches: with a few hundred queues per XCD a contiguous eighth is a contiguous piece of the IMAGE, and one XCD gets the
    // expensive pixels — random_spheres, 5 600 queues: 1.65 -> 1.72-1.77 ms, profiles/r05e_xcd_ab.txt, r05g_share_xcd.txt.  From 8 192 queues
    // on it pays most where batches are short: rank 0's share of a 4-rank cornell frame, 12 288 queues, 10.05 -> 8.82 ms = 79 -> 95 % of
    // ideal; of an 8-rank frame 4.73 -> 4.50 ms = 84 -> 94 %: its k_extend_linear_defer 8.2 -> 6.5 ms)
    const uint32_t per = gridDim.x >> 3;
    if ((gridDim.x & 7u) == 0u && gridDim.x >= FW_XCD_MIN) return (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
#endif
    return blockIdx.x * (WB / 64) + (threadIdx.x >> 6);
}

// Hit record: 8 bytes (t, code) with code = object << prim_bits | primitive (rect3d face / mesh triangle), MISS = all
// ones.  k_shade is HBM-bound (88 % of what its streams can reach), so bytes are what it pays for.
__device__ __forceinline__ float2 pack_hit(float t, uint32_t obj, uint32_t prim, uint32_t prim_bits) {
    return make_float2(t, __uint_as_float(obj == MISS ? MISS : ((obj << prim_bits) | prim)));
}

// Wave w generates the paths  id = chunk * (n_waves*64) + w*64 + lane  (chunks of 64 consecutive pixels of
// one sample index, dealt round-robin to the waves: coherent inside a wave, balanced across waves).
// render.rs:172-180 + camera.rs:109-116 for one path: the camera ray of (pixel, sample).
// (Round 2 tried to RECOMPUTE this ray in k_shade at segment 0 instead of loading it — 16 B per sample, 4.3 GB per cornell
// frame less HBM traffic, bit-identical by construction.  It lost: k_shade 21.9 -> 23.1 ms, hdri 5.9 -> 6.4 ms, at 125
// instead of 115 registers: the kernel is not bound by bytes alone.  Removed.)
__device__ __forceinline__ Ray camera_ray(const DCamera &cam, const DFrame &f, const RngKey &k) {
    const Rcp rw = make_rcp((float)f.width), rh = make_rcp((float)f.height);
    uint32_t py_row = (uint32_t)((float)k.pixel * f.inv_width);      // k.pixel / width without an integer division
    int32_t px = (int32_t)(k.pixel - py_row * f.width);
    if (px < 0) { py_row--; px += (int32_t)f.width; } else if ((uint32_t)px >= f.width) { py_row++; px -= (int32_t)f.width; }
    uint32_t py = f.height - py_row;                                  // util.rs:31-33 Coord::from_index
    uint4 j = draw(k, P_JITTER, 0, 0);
    float u = fdiv((float)px + u2f(j.x), rw);                         // render.rs:178
    float v = fdiv((float)py + u2f(j.y), rh);                         // render.rs:179
    V3 pos = ld3(cam.position);
    if (cam.lens_radius == 0.f) {
        // camera.rs:109-116 draws the disk sample even when the aperture is 0; then rd = 0 * sample = (+-0, +-0) and so is
        // `offset`.  x + (+-0) = x and x - (+-0) = x for every x except -0, so unless a component of the position or of the
        // direction-before-offset is a negative zero, the sample's signs cannot reach the ray: skip its rejection loop (half of
        // k_raygen's instructions for a pinhole camera).  A lane that does meet a -0 takes the full expression below.
        const V3 dd = ld3(cam.lower_left) + u * ld3(cam.horizontal) + v * ld3(cam.vertical) - pos;
        const uint32_t NZ = 0x80000000u;
        const bool neg0 = __float_as_uint(dd.x) == NZ || __float_as_uint(dd.y) == NZ || __float_as_uint(dd.z) == NZ ||
                          __float_as_uint(pos.x) == NZ || __float_as_uint(pos.y) == NZ || __float_as_uint(pos.z) == NZ;
        if (!neg0) return Ray{pos, dd};
    }
    V3 rd = cam.lens_radius * random_in_unit_disk(k);
    V3 cu = ld3(cam.u), cv = ld3(cam.v);
    V3 offset = cu * rd.x + cv * rd.y;
    V3 o = pos + offset;
    V3 d = ld3(cam.lower_left) + u * ld3(cam.horizontal) + v * ld3(cam.vertical) - pos - offset;
    return Ray{o, d};
}

__global__ __launch_bounds__(WB) void k_raygen(DCamera cam, DFrame f, DPaths out, float4 *__restrict__ sample_rad, DQueue q, uint32_t n_paths) {
    const uint32_t w = wave_index(), lane = threadIdx.x & 63u;
    if (w >= q.n_waves) return;
    uint32_t produced = 0;
    for (uint32_t chunk = 0;; chunk++) {
        uint32_t id0 = chunk * (q.n_waves * 64u) + w * 64u;
        if (id0 >= n_paths) break;
        uint32_t i = id0 + lane;
        const uint32_t slot = w * q.cap + chunk * 64u + lane;
        bool fl = false; V3 o = mk(0, 0, 0), d = o;
        if (i < n_paths) {
            const Ray r = camera_ray(cam, f, key_of_linear(f, i));
            o = r.o; d = r.d;
            if (f.pinhole0) qst(&out.ray_a[slot], make_float4(d.x, d.y, d.z, 0.f));
            else {
                qst(&out.ray_a[slot], make_float4(o.x, o.y, o.z, d.x));
                qst(&out.ray_b[slot], make_float2(d.y, d.z));
            }
            fl = f.ex.mode && needs_exact(f.ex, o.x, o.y, o.z, d.x, d.y, d.z);
        }
        if (f.ex.mode) flag_exact(f.ex, fl, slot, 0);
        produced += min(64u, n_paths - id0);
    }
    if (lane == 0) q.wcount[w] = produced;                                    // segment 0 queue length of this wave
}

// ------------------------------------------------------------------------------------------------
// object record access + world<->object transform (scene.rs:235-266)
// ------------------------------------------------------------------------------------------------
struct Obj { float4 q0, q1, q2, q3, q4; uint32_t kf, material, aux0, aux1; };

// record layout (fw_device.h): r0 = (pos.xyz, kind|flags) r1 = q3 r2 = q4 r3..r5 = (R row i, material | aux0 | aux1)
__device__ __forceinline__ Obj load_obj(const float4 *__restrict__ objs, uint32_t j) {
    const float4 *p = objs + (size_t)j * OBJ_Q;
    float4 r0 = p[0], r3 = p[3], r4 = p[4], r5 = p[5];
    Obj o;
    o.q0 = make_float4(r3.x, r3.y, r3.z, r0.x); o.q1 = make_float4(r4.x, r4.y, r4.z, r0.y); o.q2 = make_float4(r5.x, r5.y, r5.z, r0.z);
    o.q3 = p[1]; o.q4 = p[2];
    o.kf = __float_as_uint(r0.w); o.material = __float_as_uint(r3.w);
    o.aux0 = __float_as_uint(r4.w); o.aux1 = __float_as_uint(r5.w);
    return o;
}
// Per-lane (TLAS leaf) fetch for k_extend: only what the intersection needs.  An unrotated sphere costs two
// 16-byte loads instead of six — these divergent gathers, not arithmetic, bound the BVH scenes (64 B/clk/CU).
__device__ __forceinline__ Obj load_obj_for_hit(const float4 *__restrict__ objs, uint32_t j) {
    const float4 *p = objs + (size_t)j * OBJ_Q;
    float4 r0 = p[0];
    Obj o;
    o.kf = __float_as_uint(r0.w); o.material = 0; o.aux0 = 0; o.aux1 = 0;
    o.q0 = make_float4(1.f, 0.f, 0.f, r0.x); o.q1 = make_float4(0.f, 1.f, 0.f, r0.y); o.q2 = make_float4(0.f, 0.f, 1.f, r0.z);
    o.q3 = p[1]; o.q4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t kind = o.kf & 0xffu, flags = (o.kf >> 8) & 0xffffu, inner = o.kf >> 24;
    if (kind == 1u || kind == 2u || kind == 3u || kind == 4u || kind == 6u) o.q4 = p[2];
    const bool mesh = kind == 5u || (kind == 6u && inner == 5u);
    if ((flags & OF_ROTATED) || mesh) {
        float4 r3 = p[3], r4 = p[4], r5 = p[5];
        o.q0 = make_float4(r3.x, r3.y, r3.z, r0.x); o.q1 = make_float4(r4.x, r4.y, r4.z, r0.y); o.q2 = make_float4(r5.x, r5.y, r5.z, r0.z);
        o.aux0 = __float_as_uint(r4.w); o.aux1 = __float_as_uint(r5.w);
    }
    return o;
}
__device__ __forceinline__ uint32_t obj_kind(const Obj &o) { return o.kf & 0xffu; }
__device__ __forceinline__ uint32_t obj_flags(const Obj &o) { return (o.kf >> 8) & 0xffffu; }
__device__ __forceinline__ uint32_t obj_inner(const Obj &o) { return o.kf >> 24; }

// inv_rotation_mat * v  (columns of R^T are the rows of R; Mat3*Vec3 = c0*x + c1*y + c2*z, plain mul/add)
__device__ __forceinline__ V3 rot_inv(const Obj &o, V3 v) {
    return {o.q0.x * v.x + o.q1.x * v.y + o.q2.x * v.z,
            o.q0.y * v.x + o.q1.y * v.y + o.q2.y * v.z,
            o.q0.z * v.x + o.q1.z * v.y + o.q2.z * v.z};
}
// rotation_mat * v
__device__ __forceinline__ V3 rot_fwd(const Obj &o, V3 v) {
    return {o.q0.x * v.x + o.q0.y * v.y + o.q0.z * v.z,
            o.q1.x * v.x + o.q1.y * v.y + o.q1.z * v.z,
            o.q2.x * v.x + o.q2.y * v.y + o.q2.z * v.z};
}
__device__ __forceinline__ Ray to_object_space(const Obj &o, const Ray &r) {
    V3 pos = mk(o.q0.w, o.q1.w, o.q2.w);
    if (obj_flags(o) & OF_ROTATED) return Ray{rot_inv(o, r.o - pos), rot_inv(o, r.d)};
    return Ray{r.o - pos, r.d};
}

// ------------------------------------------------------------------------------------------------
// K3  primitive intersectors (object space).  Each returns t (and a primitive id where one exists).
// ------------------------------------------------------------------------------------------------
// objects/mod.rs:19-31 + objects/sphere.rs:31-60
__device__ __forceinline__ bool hit_sphere(float radius, const Ray &r, float tmin, float tmax, float &t_out) {
    float a = dot(r.d, r.d);
    float b = 2.f * dot(r.o, r.d);
    float c = dot(r.o, r.o) - radius * radius;
    float disc = b * b - 4.f * a * c;
    if (disc < 0.f) return false;
    float t1, t2; bool two;
    Rcp two_a = make_rcp(2.f * a);
    if (disc == 0.f) { t1 = fdiv(-b, two_a); t2 = 0.f; two = false; }
    else { float sq = fsqrt(disc); t1 = fdiv(-b - sq, two_a); t2 = fdiv(-b + sq, two_a); two = true; }
    if (t1 < tmax && t1 > tmin) { t_out = t1; return true; }
    if (two && t2 < tmax && t2 > tmin) { t_out = t2; return true; }
    return false;
}

// objects/rect.rs:47-73.  AX: 0 = XY (plane axis z), 1 = XZ (plane axis y), 2 = YZ (plane axis x)
template <int AX, bool DEGENERATE>
__device__ __forceinline__ bool hit_rect(float a_min, float a_max, float b_min, float b_max, float k, const Ray &r,
                                         float tmin, float tmax, float &t_out) {
    constexpr int A1 = (AX == 2) ? 1 : 0, A2 = (AX == 0) ? 1 : 2, OT = (AX == 0) ? 2 : (AX == 1 ? 1 : 0);
    // rect.rs:47-62: three interval tests, each rejecting on `x < lo || x > hi` (so a NaN passes, as in the reference).
    // 18 of these run per cornell ray and the kernel is VALU-issue-bound, so the form matters:
    //   x in [lo, hi]  <=>  x == med3(x, lo, hi)   when lo <= hi:  v_med3 + one compare per interval, "not (less or
    //   greater)" so that a NaN x still passes (med3 returns a non-NaN bound then) — 18.8 vs 19.7 ms against the previous form,
    //   min of the six differences (x - lo, hi - x, ...) >= 0 with NaN-dropping fminf, which is kept for DEGENERATE rects
    //   (lo > hi on some interval, flagged by the host: nothing passes there, but a median would).
    // Six v_cmp + five s_or_b64 per rect (the literal form) kept the scalar unit ~80 % busy in the first version.
    float t = fdiv(k - comp(r.o, OT), comp(r.d, OT));
    float pa = comp(r.o, A1) + t * comp(r.d, A1), pb = comp(r.o, A2) + t * comp(r.d, A2);
    t_out = t;
    if (DEGENERATE) {
        float slack = fminf(fminf(fminf(t - tmin, tmax - t), fminf(pa - a_min, a_max - pa)), fminf(pb - b_min, b_max - pb));
        return !(slack < 0.f);
    }
    // The t interval keeps its two literal comparisons: its upper bound is the closest hit so far, and that can be a NaN.  A ray
    // that lies exactly in a rectangle's plane gives t = 0/0, which rect.rs:47-62 ACCEPTS (`t < t_min || t > t_max` is false for
    // a NaN) — after which `closest` is NaN and every later, ordinary hit is accepted too, because `t > NaN` is false as well.
    // v_med3 with a NaN operand returns min3 instead, i.e. it rejects them (and lets t < t_min through).  This was the one
    // diverging path of cornell 512x512 @1024 (tools/diverge.py: pixel 112819, sample 170, segment 8: a bounce ray with
    // d.y = 0 starting on the ceiling; the reference goes on to hit the back wall).  Position intervals have constant bounds.
    const float ma = __builtin_amdgcn_fmed3f(pa, a_min, a_max), mb = __builtin_amdgcn_fmed3f(pb, b_min, b_max);
    return !(t < tmin || t > tmax) && !(pa < ma || pa > ma) && !(pb < mb || pb > mb);
}
// The non-degenerate form above with the reciprocal of the plane axis' direction component handed in: the walls of a room share three of
// them (k_extend_linear_defer).  rc == make_rcp(comp(r.d, OT)), so fdiv gives the bits of the form above.
template <int AX>
__device__ __forceinline__ bool hit_rect_rcp(float a_min, float a_max, float b_min, float b_max, float k, const Ray &r, const Rcp &rc,
                                             float tmin, float tmax, float &t_out) {
    constexpr int A1 = (AX == 2) ? 1 : 0, A2 = (AX == 0) ? 1 : 2, OT = (AX == 0) ? 2 : (AX == 1 ? 1 : 0);
    const float t = fdiv(k - comp(r.o, OT), rc);
    const float pa = comp(r.o, A1) + t * comp(r.d, A1), pb = comp(r.o, A2) + t * comp(r.d, A2);
    t_out = t;
    const float ma = __builtin_amdgcn_fmed3f(pa, a_min, a_max), mb = __builtin_amdgcn_fmed3f(pb, b_min, b_max);
    return !(t < tmin || t > tmax) && !(pa < ma || pa > ma) && !(pb < mb || pb > mb);
}
template <bool DEGENERATE>
__device__ __forceinline__ bool hit_rect_kind_t(uint32_t kind, float4 q3, float k, const Ray &r, float tmin, float tmax, float &t) {
    if (kind == 1) return hit_rect<0, DEGENERATE>(q3.x, q3.y, q3.z, q3.w, k, r, tmin, tmax, t);
    if (kind == 2) return hit_rect<1, DEGENERATE>(q3.x, q3.y, q3.z, q3.w, k, r, tmin, tmax, t);
    return hit_rect<2, DEGENERATE>(q3.x, q3.y, q3.z, q3.w, k, r, tmin, tmax, t);
}
// q4 = (k, degenerate flag, -, -)
__device__ __forceinline__ bool hit_rect_kind(uint32_t kind, float4 q3, float4 q4, const Ray &r, float tmin, float tmax, float &t) {
    if (q4.y != 0.f) return hit_rect_kind_t<true>(kind, q3, q4.x, r, tmin, tmax, t);
    return hit_rect_kind_t<false>(kind, q3, q4.x, r, tmin, tmax, t);
}

// objects/rect3d.rs:18-100 — faces +z, -z, +y, -y, +x, -x; linear closest with narrowing, later wins ties.
// (Measured and rejected: the two parallel faces of an axis as one packed-f32 computation — v_pk_fma/mul/add issue at half
// the rate of their scalar forms on gfx950, tools/microbench.hip, so nothing is gained and the repacking costs: 28.0 vs 24.3 ms.)
template <bool DEGENERATE>
__device__ __forceinline__ bool hit_rect3d_t(float4 q3, float4 q4, const Ray &r, float tmin, float tmax, float &t_out, uint32_t &face) {
    float px = q3.x, py = q3.y, pz = q3.z, sx = q3.w, sy = q4.x, sz = q4.y;
    float closest = tmax, t; face = 7u;          // 7 = no face yet: "any hit" is one vector compare at the end, not five scalar ORs
    bool h;   // predicated: no exec-mask bookkeeping between the six faces
    h = hit_rect<0, DEGENERATE>(px, px + sx, py, py + sy, pz + sz, r, tmin, closest, t); closest = h ? t : closest; face = h ? 0u : face;
    h = hit_rect<0, DEGENERATE>(px, px + sx, py, py + sy, pz, r, tmin, closest, t);      closest = h ? t : closest; face = h ? 1u : face;
    h = hit_rect<1, DEGENERATE>(px, px + sx, pz, pz + sz, py + sy, r, tmin, closest, t); closest = h ? t : closest; face = h ? 2u : face;
    h = hit_rect<1, DEGENERATE>(px, px + sx, pz, pz + sz, py, r, tmin, closest, t);      closest = h ? t : closest; face = h ? 3u : face;
    h = hit_rect<2, DEGENERATE>(py, py + sy, pz, pz + sz, px + sx, r, tmin, closest, t); closest = h ? t : closest; face = h ? 4u : face;
    h = hit_rect<2, DEGENERATE>(py, py + sy, pz, pz + sz, px, r, tmin, closest, t);      closest = h ? t : closest; face = h ? 5u : face;
    t_out = closest;
    const bool any = face != 7u;
    face &= 7u * (uint32_t)any;                 // callers read `face` only after a hit, but keep it a valid face id
    return any;
}
// q4 = (size.y, size.z, degenerate flag: some size component is negative, -)
__device__ __forceinline__ bool hit_rect3d(float4 q3, float4 q4, const Ray &r, float tmin, float tmax, float &t_out, uint32_t &face) {
    if (q4.z != 0.f) return hit_rect3d_t<true>(q3, q4, r, tmin, tmax, t_out, face);
    return hit_rect3d_t<false>(q3, q4, r, tmin, tmax, t_out, face);
}

constexpr float PI_F = 3.14159265358979323846f;

// objects/mod.rs:19-31: number of roots (0/1/2) of a t^2 + b t + c
__device__ __forceinline__ int solve_quadratic(float a, float b, float c, float &t1, float &t2) {
    float disc = b * b - 4.f * a * c;
    if (disc < 0.f) return 0;
    Rcp two_a = make_rcp(2.f * a);
    if (disc == 0.f) { t1 = fdiv(-b, two_a); t2 = 0.f; return 1; }
    float sq = fsqrt(disc);
    t1 = fdiv(-b - sq, two_a); t2 = fdiv(-b + sq, two_a);
    return 2;
}
// objects/cone.rs:26-88 (the accepted root only; normal/uv are rebuilt in k_shade)
__device__ __forceinline__ bool cone_ok(float height, const Ray &r, float t, float tmin, float tmax) {
    if (t > tmax || t < tmin) return false;
    float py = r.o.y + t * r.d.y;
    return !(py < 0.f || py > height);
}
__device__ __forceinline__ bool hit_cone(float radius, float height, const Ray &r, float tmin, float tmax, float &t_out) {
    V3 o = r.o, d = r.d;
    float r2_div_h2 = fdiv(radius * radius, height * height);
    float a = d.x * d.x + d.z * d.z - r2_div_h2 * d.y * d.y;
    float b = 2.f * (d.x * o.x + d.z * o.z - r2_div_h2 * d.y * (o.y - height));
    float c = o.x * o.x + o.z * o.z - r2_div_h2 * (o.y - height) * (o.y - height);
    float t1, t2;
    int n = solve_quadratic(a, b, c, t1, t2);
    if (n == 0) return false;
    if (cone_ok(height, r, t1, tmin, tmax)) { t_out = t1; return true; }
    if (n == 2 && cone_ok(height, r, t2, tmin, tmax)) { t_out = t2; return true; }
    return false;
}
// objects/cylinder.rs:40-90
__device__ __forceinline__ bool cylinder_ok(float height, float max_phi, const Ray &r, float t, float tmin, float tmax) {
    if (t > tmax || t < tmin) return false;
    V3 p = ray_point(r, t);
    float phi = fwlm::atan2f_glibc(p.z, p.x);
    if (phi < 0.f) phi = phi + PI_F * 2.f;
    return p.y > 0.f && p.y < height && phi < max_phi;
}
__device__ __forceinline__ bool hit_cylinder(float radius, float height, float max_phi, const Ray &r, float tmin, float tmax, float &t_out) {
    V3 o = r.o, d = r.d;
    float a = d.x * d.x + d.z * d.z;
    float b = 2.f * (d.x * o.x + d.z * o.z);
    float c = o.x * o.x + o.z * o.z - radius * radius;
    float disc = b * b - 4.f * a * c;
    if (!(disc > 0.0f)) return false;
    float t1, t2;
    int n = solve_quadratic(a, b, c, t1, t2);
    if (n == 0) return false;
    if (cylinder_ok(height, max_phi, r, t1, tmin, tmax)) { t_out = t1; return true; }
    if (n == 2 && cylinder_ok(height, max_phi, r, t2, tmin, tmax)) { t_out = t2; return true; }
    return false;
}
// objects/disk.rs:39-83
__device__ __forceinline__ bool hit_disk(float radius, float phi_max, float inner_radius, const Ray &r, float tmin, float tmax, float &t_out) {
    if (r.d.y == 0.f) return false;
    float t = fdiv(-r.o.y, r.d.y);
    if (t < tmin || t > tmax) return false;
    V3 p = ray_point(r, t);
    float dist2 = p.x * p.x + p.z * p.z;
    if (dist2 > radius * radius || dist2 < inner_radius * inner_radius) return false;
    float phi = fwlm::atan2f_glibc(p.z, p.x);
    if (phi < 0.f) phi = phi + 2.f * PI_F;
    if (phi > phi_max) return false;
    t_out = t;
    return true;
}

// util.rs:104-118 — SIGNED comparison
__device__ __forceinline__ int max_component_idx(V3 v) {
    if (v.x > v.y) return (v.z > v.x) ? 2 : 0;
    return (v.z > v.y) ? 2 : 1;
}

// objects/mesh.rs:139-219.  The permutation (kx,ky,kz) and the shear (sx,sy,sz) depend on the ray only
// (mesh.rs:147-162), so a BLAS walk computes them once (TriRay) instead of once per triangle: three divisions and the
// signed max-component search leave the per-triangle path.  Returns t and the barycentrics.
struct TriRay { V3 o; int kx, ky, kz; float sx, sy, sz; };
__device__ __forceinline__ TriRay make_triray(const Ray &r) {
    TriRay q;
    q.o = r.o;
    q.kz = max_component_idx(r.d);
    q.kx = (q.kz + 1) % 3;
    q.ky = (q.kx + 1) % 3;
    V3 d = mk(comp(r.d, q.kx), comp(r.d, q.ky), comp(r.d, q.kz));
    Rcp dz = make_rcp(d.z);
    q.sx = fdiv(-d.x, dz); q.sy = fdiv(-d.y, dz); q.sz = fdiv(1.f, dz);
    return q;
}
__device__ __forceinline__ bool hit_triangle(V3 p0, V3 p1, V3 p2, const TriRay &q, float tmin, float tmax,
                                             float &t_out, float &b0, float &b1, float &b2) {
    V3 p0t = p0 - q.o, p1t = p1 - q.o, p2t = p2 - q.o;
    p0t = mk(comp(p0t, q.kx), comp(p0t, q.ky), comp(p0t, q.kz));
    p1t = mk(comp(p1t, q.kx), comp(p1t, q.ky), comp(p1t, q.kz));
    p2t = mk(comp(p2t, q.kx), comp(p2t, q.ky), comp(p2t, q.kz));
    p0t.x += q.sx * p0t.z; p0t.y += q.sy * p0t.z;
    p1t.x += q.sx * p1t.z; p1t.y += q.sy * p1t.z;
    p2t.x += q.sx * p2t.z; p2t.y += q.sy * p2t.z;
    float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
    float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
    float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
    // (e0<0 || e1<0 || e2<0) && (e0>0 || e1>0 || e2>0), as two comparisons (fminf/fmaxf drop NaNs like the ORs do)
    if (fminf(fminf(e0, e1), e2) < 0.f && fmaxf(fmaxf(e0, e1), e2) > 0.f) return false;
    float det = e0 + e1 + e2;
    if (det == 0.f) return false;
    p0t.z *= q.sz; p1t.z *= q.sz; p2t.z *= q.sz;
    float t_scaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    if (det < 0.f && (t_scaled >= tmin * det || t_scaled < tmax * det)) return false;
    else if (det > 0.f && (t_scaled <= tmin * det || t_scaled > tmax * det)) return false;
    float inv_det = fdiv(1.f, det);
    b0 = e0 * inv_det; b1 = e1 * inv_det; b2 = e2 * inv_det;
    t_out = t_scaled * inv_det;
    return true;
}

// aabb.rs:30-50 with the reciprocal directions hoisted (1/d is the same value every call) and ONE comparison: the
// reference leaves its per-axis `all()` at the first axis with tmax <= tmin, but tmin only grows and tmax only shrinks
// (f32::max/min ignore NaN, so neither ever becomes NaN), hence a failed axis implies the final tmax <= tmin as well.
__device__ __forceinline__ bool hit_aabb(float4 lo, float4 hi, V3 o, V3 inv, float tmin, float tmax) {
    float t0 = (lo.x - o.x) * inv.x, t1 = (hi.x - o.x) * inv.x;
    tmin = fmaxf(tmin, inv.x < 0.f ? t1 : t0); tmax = fminf(tmax, inv.x < 0.f ? t0 : t1);
    t0 = (lo.y - o.y) * inv.y; t1 = (hi.y - o.y) * inv.y;
    tmin = fmaxf(tmin, inv.y < 0.f ? t1 : t0); tmax = fminf(tmax, inv.y < 0.f ? t0 : t1);
    t0 = (lo.z - o.z) * inv.z; t1 = (hi.z - o.z) * inv.z;
    tmin = fmaxf(tmin, inv.z < 0.f ? t1 : t0); tmax = fminf(tmax, inv.z < 0.f ? t0 : t1);
    return tmax > tmin;
}
// The box test of the WALKS (round 4): aabb.rs:30-50 with the planes chosen by the ray's signs before the arithmetic, the entry
// distance returned (clamped to tmin: it orders the children and is what culling compares), and the EXIT planes multiplied by
// inv_hi = inv * (1 + 2^-12) instead of inv (relaxed(): three more registers per ray, no more instructions per box).
// Why relaxed.  Which items the reference tests is decided by its own leaf-node boxes (tri_gate_ok / obj_gate_ok below, exact);
// the walked trees only have to REACH every item that rule admits and that the item test accepts.  Such an item can lie a hair
// outside its own, tight box: a ray that touches a box exactly at an edge (entry == exit: `tmax > tmin` fails, the union behind a
// DoubleLeaf passes), a sphere whose discriminant is rounding noise at its silhouette (the ray may miss it by 2^-24 L^2 / r and still
// "hit": 1.2e-4 for r = 0.1 seen from L = 14), a triangle hit carrying (max|d| / |d_kz|) ulps.  Relative to the distance these are
// <= 2^-12 for every ray that is not on the exact list (far rule: L < 2048 r; shear rule: < 2^10), so a box admits every ray whose
// exit is within that of its entry.  Four such paths in part2's 1.8e9 at 256 spp, one in teapot's 2.7e8 (gpurun_out/r04a, r04b).
constexpr float GATE_RELAX = 1.0f + 1.0f / 4096.0f;
__device__ __forceinline__ V3 relaxed(V3 inv) { return mk(inv.x * GATE_RELAX, inv.y * GATE_RELAX, inv.z * GATE_RELAX); }
// ... and the lower end of the interval the BOXES are clipped to (the item tests keep the caller's t_min).  A path that leaves a
// surface starts ON a triangle's neighbour, whose plane it meets at t ~ 0; the computed t is that plus noise — (max|d| / |d_kz|)^2
// ulps of the distance: 6e-5 at a shear of 32, 1.4e-3 at 150 — and the reference accepts it when the noise lifts it over
// t_min = 0.001 (suzanne pixel 515109, sample 499: t = 0.00105 for a hit point 5e-6 from the origin).  The ray has left such a
// triangle's own, thin box long before t_min, so a box test clipped at t_min never reaches it; the reference's does, through the
// DoubleLeaf's union.  Boxes are therefore entered from 15/16 of t_min on, and from behind the origin for the SOFT class.
__device__ __forceinline__ float box_tmin(float tmin, bool soft) { return tmin <= 0.f ? tmin : (soft ? -0.0625f : tmin * 0.9375f); }
__device__ __forceinline__ bool hit_aabb_entry(float4 lo, float4 hi, V3 o, V3 inv, V3 inv_hi, float tmin, float tmax, float &entry) {
    const bool sx = inv.x < 0.f, sy = inv.y < 0.f, sz = inv.z < 0.f;
    tmin = fmaxf(tmin, ((sx ? hi.x : lo.x) - o.x) * inv.x); tmax = fminf(tmax, ((sx ? lo.x : hi.x) - o.x) * inv_hi.x);
    tmin = fmaxf(tmin, ((sy ? hi.y : lo.y) - o.y) * inv.y); tmax = fminf(tmax, ((sy ? lo.y : hi.y) - o.y) * inv_hi.y);
    tmin = fmaxf(tmin, ((sz ? hi.z : lo.z) - o.z) * inv.z); tmax = fminf(tmax, ((sz ? lo.z : hi.z) - o.z) * inv_hi.z);
    entry = tmin;
    return tmax > tmin;
}
// The reference's gating rule, exact (round 4).  bvh.rs:115-151 tests an item iff the ray passes the box of every node down to the
// item's LEAF NODE — i.e. iff it passes that node's box (a DoubleLeaf's is the union of its two items'; the ancestors' are supersets
// and the slab arithmetic is monotone).  The walked trees hold the items' own boxes, relaxed (hit_aabb_entry), so a hit that is about
// to become its ray's best is put to the rule itself, with the reference's own test (hit_aabb, the caller's [tmin, tmax]):
//   triangle: the box of its three vertices is inside its leaf node's, so passing it suffices (no fetch); otherwise the leaf node's
//             box decides (tri_gate, 32 B from HBM: flat axis-aligned triangles, and one other hit in ~1e7);
//   object:   its reference leaf-node box (obj_gate), fetched with the object record.
__device__ __forceinline__ bool tri_gate_ok(const DScene &sc, size_t tri, V3 p0, V3 p1, V3 p2, V3 o, V3 inv, float tmin, float tmax) {
#ifdef FW_AB_NO_VERIFY      // timing variant only (tools/build_variant.sh noverify -DFW_AB_NO_VERIFY): what the rule costs
    return true;
#endif
    // Sufficient, and all that nearly every hit needs: the box of the three vertices WITHOUT mesh.rs:230-241's padding of flat axes,
    // entry = the smaller and exit = the larger of an axis' two plane distances.  With lo <= hi the smaller IS the plane the
    // reference picks by the sign of 1/d (the slab arithmetic is monotone; a NaN distance — the origin in a plane it runs along —
    // makes this form fail where the reference ignores it, which only sends the hit to the exact test below); the padded box and the
    // leaf node's contain this one.  ~30 instructions, nothing fetched.
    const float lx = fminf(fminf(p0.x, p1.x), p2.x), ly = fminf(fminf(p0.y, p1.y), p2.y), lz = fminf(fminf(p0.z, p1.z), p2.z);
    const float hx = fmaxf(fmaxf(p0.x, p1.x), p2.x), hy = fmaxf(fmaxf(p0.y, p1.y), p2.y), hz = fmaxf(fmaxf(p0.z, p1.z), p2.z);
    const float ax = (lx - o.x) * inv.x, bx = (hx - o.x) * inv.x, ay = (ly - o.y) * inv.y, by = (hy - o.y) * inv.y, az = (lz - o.z) * inv.z, bz = (hz - o.z) * inv.z;
    const float tn = fmaxf(fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz)), tmin);
    const float tf = fminf(fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)), tmax);
    if (tf > tn) return true;
    return hit_aabb(sc.tri_gate[2 * tri], sc.tri_gate[2 * tri + 1], o, inv, tmin, tmax);     // the rule itself: the reference leaf node's box, aabb.rs:30-50
}
__device__ __forceinline__ bool obj_gate_ok(const DScene &sc, uint32_t k, V3 o, V3 inv) {
    return hit_aabb(sc.obj_gate[2 * (size_t)k], sc.obj_gate[2 * (size_t)k + 1], o, inv, 0.001f, 2e9f);      // render.rs:19: the walk's [t_min, t_max]
}
// Per-lane traversal stack in LDS, laid out [level][lane] so a push/pop by the whole wave touches
// 64 consecutive dwords (conflict-free).  `base` = first level this traversal may use.
struct LdsStack {
    uint32_t *s; int sp;
    __device__ __forceinline__ void push(uint32_t v) { s[sp * FW_WB] = v; sp++; }
    __device__ __forceinline__ uint32_t pop() { sp--; return s[sp * FW_WB]; }
};
// The same with 16-bit entries (trees of fewer than 2^15 nodes and items: bit 15 = leaf flag), [level][lane] over the 64 lanes
// of ONE wave: half the LDS, which is what lets a 1024-thread workgroup keep 16 stacks next to a whole tree (k_blas_lds).
struct LdsStack16 {
    uint16_t *s; int sp;
    __device__ __forceinline__ void push(uint32_t v) { s[sp * 64] = (uint16_t)((v & 0x7fffu) | ((v >> 16) & 0x8000u)); sp++; }
    __device__ __forceinline__ uint32_t pop() { sp--; const uint32_t e = s[sp * 64]; return (e & 0x7fffu) | ((e & 0x8000u) << 16); }
};

// One step of a walk over PAIR NODES (fw_device.h: a node holds the boxes of BOTH its children, so one 64-byte fetch
// decides two boxes; with one box per node every box test waited for its own dependent fetch and the walk was
// latency-bound).  Returns the next reference: the nearer hit child (the farther one is pushed), else a popped one.
// A child is entered when the reference's own test passes (aabb.rs:30-50 with the caller's [tmin, tmax]) and its entry
// is not clearly beyond the best hit so far: `cull` = cull_bound(best t) = best t + 2^-10 |best t|.  The slack covers what a
// computed slab entry can exceed the t of a hit lying on the box face, and keeps exact ties (which the in-order rank
// decides) reachable — so the result does not depend on the shape of the walked tree.
template <class Stack>
__device__ __forceinline__ uint32_t pair_step(const float4 *__restrict__ nodes, uint32_t node, V3 o, V3 inv, V3 inv_hi, float tmin, float tmax,
                                              float cull, Stack &st) {
    const float4 *nd = nodes + 4 * (size_t)node;
    const float4 q0 = nd[0], q1 = nd[1], q2 = nd[2], q3 = nd[3];
    float tl, tr;
    bool hl = hit_aabb_entry(q0, q1, o, inv, inv_hi, tmin, tmax, tl), hr = hit_aabb_entry(q2, q3, o, inv, inv_hi, tmin, tmax, tr);
    hl = hl && !(tl > cull); hr = hr && !(tr > cull);
    const uint32_t rl = __float_as_uint(q0.w), rr = __float_as_uint(q1.w);
    // Flat, not nested: one predicated push and one predicated pop around a select.  The nested form (both -> push and return,
    // else left, else right, else pop) cost the walk loops four extra scalar mask copies per step and made the compiler wait
    // for the node's first quad before issuing the loads of the other three (part2 @16 10.35 -> 10.15 ms, suzanne @64 10.3 -> 10.2).
    const bool left_first = tl <= tr;
    uint32_t next = (hl && (!hr || left_first)) ? rl : rr;
    if (hl && hr) st.push(left_first ? rr : rl);
    if (!hl && !hr) { next = REF_DONE; if (st.sp) next = st.pop(); }
    return next;
}
// (Round 3, also measured and removed — tools/experiments/r03_branchfree_stack.diff: the step without its three predicated regions
// (the stack's top read with the node, the farther child written to the free slot whether or not it is kept, the count moved by
// a select): k_blas_lds 448 -> 422 us per launch, and the frames within the boxes' noise or worse — suzanne 65.4 -> 66.2 ms, @64
// 10.1 -> 10.5, part2 @16 10.7 -> 11.0, teapot @32 15.5 -> 15.7.)
// (Round 3, measured and removed — tools/experiments/r03_lds_planes.diff: for the LDS-resident walks the nodes as seven PLANES of
// float2 (left, right) per number, each lane reading the near and far planes its ray's signs select (no selects: 27 vector
// instructions for the two box tests instead of 56, the subtractions and multiplications as v_pk_add_f32 / v_pk_mul_f32) and
// ds_read_b64 at 8-byte stride instead of whole nodes 64 bytes apart (whose quads fall on 4 of 16 slots of a bank row: 4-way
// conflicts in every lane group, ~96 LDS cycles per step).  Bit-identical, GPU suite green — and k_blas_lds 448 -> 430 us per
// launch, suzanne 65.4 -> 65.6 ms, part2 @16 10.6 -> 10.7: nothing.  The packed instructions take two passes each (the same
// cycles as the scalar pairs), and the step is not bound by its box arithmetic or by LDS cycles but by the serial issue of one
// wave's ~110 instructions with 2.4 waves per SIMD to hide it.  The same planes read as single dwords from whole nodes
// (ds_read2_b32, 16-dword stride: 16-way conflicts) cost +20 %: suzanne 65 -> 78 ms.)
// best t -> culling bound, a little beyond it whatever its sign (a medium's inner mesh is walked with t in (-MAX, MAX))
#ifdef FW_NO_CULL     // A/B build (tools/diverge.py): no culling against the best hit, every box test is the reference's alone
__device__ __forceinline__ float cull_bound(float) { return 3.40282347e+38f; }
#else
// Round 4: the slack is 2^-10 of t (it was 1e-6).  A quadratic's root at the silhouette of a sphere, cone or cylinder carries the
// square root of its discriminant's rounding: up to 2^-11.5 of the distance, whatever the radius — such a hit can precede the entry
// of its own box by that much, and the reference, which never culls, takes it (part2 pixel 845940, sample 176: two spheres of the
// cluster 1.2e-5 apart in t, gpurun_out/r04d).  Boxes entered up to 0.1 % beyond the best hit cost nothing measurable.
__device__ __forceinline__ float cull_bound(float t) { return t + fabsf(t) * (1.0f / 1024.0f); }
#endif

// K4  mesh BLAS (bvh.rs:100-151 over Triangle items).  The reference visits BOTH children with the caller's
// [tmin,tmax] and keeps the smaller t, the right/later item winning ties.  Here: front-to-back traversal
// (a Branch's children are split along axis depth%3, so the sign of the ray direction on that axis picks the
// near child), boxes culled against the best t so far, every triangle still tested against the caller's tmax,
// and ties resolved by the item's in-order rank in the REFERENCE tree (tri_rank / obj_rank, fetched only on a
// tie) — the reference's winner in any visiting order and for any tree topology (the walked tree is SAH-built).  Differs only if the winning hit lies exactly on a culled
// box's entry plane, or for NaN t.
__device__ __forceinline__ bool hit_mesh(const DScene &sc, uint32_t root, uint32_t tri_base, const Ray &r, float tmin,
                                         float tmax, float cull_t, uint32_t *stack_base, float &t_out, uint32_t &tri_out) {
    V3 inv = mk(fdiv(1.f, r.d.x), fdiv(1.f, r.d.y), fdiv(1.f, r.d.z));
    const TriRay tr = make_triray(r);
    LdsStack st{stack_base, 0};
    bool have = false; float best = tmax; uint32_t best_tri = 0;
    const bool soft = soft_direction(r.d.x, r.d.y, r.d.z, sc.soft_shear);               // the SOFT class: walked without culling
    uint32_t cur = root;                                                 // a reference: pair node, REF_LEAF | triangle, or REF_DONE
    while (cur != REF_DONE) {
        // while-while: every lane first walks pair nodes until it holds a leaf (or runs out of tree); only then do the
        // lanes test triangles, together — the long triangle test is not run for one or two lanes at a time
        while (!(cur & REF_LEAF)) {
            TS_TICK(4);
            // cull_t: a t the caller already holds from another object (hits beyond it cannot win; equal t still can)
            cur = pair_step(sc.blas, cur, r.o, inv, relaxed(inv), box_tmin(tmin, soft), tmax, soft ? NO_CULL : cull_bound(have ? fminf(best, cull_t) : cull_t), st);
        }
        if (cur == REF_DONE) break;
        const uint32_t item = cur & NODE_MASK;
        cur = st.sp ? st.pop() : REF_DONE;
        TS_TICK(6);
        const float4 *tp = sc.tri + 3 * (size_t)(tri_base + item);
        float4 a = tp[0], b = tp[1], c = tp[2];
        float t, b0, b1, b2;
        if (hit_triangle(mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), tr, tmin, tmax, t, b0, b1, b2)) {
            // tie -> the item that comes later in the reference tree's in-order (ranks fetched only then)
            if ((!have || t < best || (t == best && sc.tri_rank[tri_base + item] > sc.tri_rank[tri_base + best_tri])) &&
                tri_gate_ok(sc, tri_base + item, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), r.o, inv, tmin, tmax)) { have = true; best = t; best_tri = item; }
        }
    }
    t_out = best; tri_out = best_tri;
    return have;
}

// The literal mesh walk of the reference (bvh.rs:115-151 over mesh.rs's Triangle items) for k_extend_exact: the mesh's own
// median-split tree, a node's items tested whenever the ray passes the NODE's box (a DoubleLeaf holds two behind one box),
// both children always, nothing culled; the smaller t wins, a tie goes to the later item (`if lh.t < rh.t {lh} else {rh}`).
// ref_root = first node of the mesh's tree in sc.ref_blas (child indices are relative to it).
constexpr int EXACT_LEVELS = 40;    // a median-split tree over N items is ceil(log2 N) deep: < 2^32 items
constexpr int EXACT_STACK = 512;    // entries of the stack the 64 lanes of a wave share in hit_mesh_exact_wave
constexpr uint32_t EXACT_WAVE_RAYS = 8;   // k_extend_exact: one ray per wave while the list is at most this many rays per wave of the launch
constexpr int EXACT_WB = 64;        // threads per workgroup of k_extend_exact (one wave): its stacks are [level][EXACT_WB] columns in LDS, a few KB that fit
                                    // on a CU next to the LDS-resident walks' 137-150 KB of the other batch in flight
// The reference's walk over one mesh (bvh.rs:92-131), one ray per LANE: k_extend_exact's form for long lists.  Never narrows
// its interval, `!(best < t)` replaces.  stack: this lane's column of an LDS array [level][EXACT_WB].
__device__ __forceinline__ bool hit_mesh_exact_lane(const DScene &sc, uint32_t ref_root, uint32_t tri_base, const Ray &r, float tmin, float tmax,
                                                    uint32_t *stack, float &t_out, uint32_t &tri_out) {
    const V3 inv = mk(fdiv(1.f, r.d.x), fdiv(1.f, r.d.y), fdiv(1.f, r.d.z));      // aabb.rs:33: 1.0 / r.direction()[a]
    const TriRay tr = make_triray(r);
    const float4 *nodes = sc.ref_blas + 2 * (size_t)ref_root;
    int sp = 0;
    uint32_t cur = 0; bool have = false; float best = tmax; uint32_t best_tri = 0;
    // while-while: every lane first goes down to its next leaf node (or runs out of tree); only then do the lanes test their
    // triangles, together — with the leaf test inside the node loop every step of the wave paid for a triangle fetch from L2
    for (bool more = true; more;) {
        uint32_t A = 0, B = 0; bool leaf = false;
        for (;;) {
            const float4 lo = nodes[2 * (size_t)cur], hi = nodes[2 * (size_t)cur + 1];
            A = __float_as_uint(lo.w); B = __float_as_uint(hi.w);
            if (hit_aabb(lo, hi, r.o, inv, tmin, tmax)) {
                if ((A >> 30) == 0u) { if (sp < EXACT_LEVELS) { stack[sp * EXACT_WB] = A & NODE_MASK; sp++; } cur = cur + 1u; continue; }     // Branch: left = next node, right later
                leaf = true; break;
            }
            if (sp == 0) { more = false; break; }
            sp--; cur = stack[sp * EXACT_WB];
        }
        if (leaf) {
            const bool two = (A >> 30) == NODE_DOUBLE;
            const uint32_t i0 = A & NODE_MASK, i1 = two ? B : i0;
            const float4 *p0 = sc.tri + 3 * (size_t)(tri_base + i0), *p1 = sc.tri + 3 * (size_t)(tri_base + i1);
            const float4 a0 = p0[0], b0 = p0[1], c0 = p0[2], a1 = p1[0], b1 = p1[1], c1 = p1[2];     // both triangles of a DoubleLeaf requested at once
            float t, u0, u1, u2;
            if (hit_triangle(mk(a0.x, a0.y, a0.z), mk(b0.x, b0.y, b0.z), mk(c0.x, c0.y, c0.z), tr, tmin, tmax, t, u0, u1, u2))
                if (!have || !(best < t)) { have = true; best = t; best_tri = i0; }
            if (two && hit_triangle(mk(a1.x, a1.y, a1.z), mk(b1.x, b1.y, b1.z), mk(c1.x, c1.y, c1.z), tr, tmin, tmax, t, u0, u1, u2))
                if (!have || !(best < t)) { have = true; best = t; best_tri = i1; }
            if (sp == 0) more = false;
            else { sp--; cur = stack[sp * EXACT_WB]; }
        }
    }
    t_out = best; tri_out = best_tri;
    return have;
}

__device__ __forceinline__ bool hit_mesh_exact_wave(const DScene &sc, uint32_t ref_root, uint32_t tri_base, const Ray &r, float tmin, float tmax,
                               uint32_t *stack, float &t_out, uint32_t &tri_out) {
    // ONE RAY PER WAVE: the 64 lanes hold the same ray (k_extend_exact's form for short lists: every lane the same instructions
    // on the same values) and share this walk.  The reference's walk (bvh.rs:92-131) never narrows its interval: it tests every
    // node whose ancestors' boxes the ray hits with the caller's (tmin, tmax), and of the triangles hit it returns the smallest
    // t, the LATER one in its depth-first order on a tie (`!(best < t)` replaces).  Neither the set of nodes nor that choice
    // depends on the order of the tests, so the lanes pop up to 64 nodes of a shared stack per round, and the wave reduces
    // (t, depth-first position) at the end; the nodes lie in depth-first order (left child = next node), so the position is the
    // node's index.  A walk is then ~depth rounds of one L2 latency each instead of one latency per node visited (60-150 us),
    // and the lanes of a wave no longer wait for each other's different paths.
    // One case does depend on the order: mesh.rs:164-177 lets a NaN t through (no comparison rejects it), and `!(best < t)` then
    // replaces whatever came before and is replaced by whatever comes next — so the reference returns what its rule makes of
    // the hits BEHIND the last NaN one in depth-first order, or that NaN hit if none follows.  A wave that meets a NaN t takes
    // the largest position of one and walks once more, counting only hits behind it.
    const uint32_t lane = threadIdx.x & 63u;
    const V3 inv = mk(fdiv(1.f, r.d.x), fdiv(1.f, r.d.y), fdiv(1.f, r.d.z));      // aabb.rs:33: 1.0 / r.direction()[a]
    const TriRay tr = make_triray(r);
    const float4 *nodes = sc.ref_blas + 2 * (size_t)ref_root;
    uint32_t *ws = stack - lane;                                         // the wave's EXACT_STACK entries
    bool have = false; float best = tmax; uint32_t best_tri = 0, best_pos = 0;
    int after = -1;                                                      // only hits at positions > after count (second pass)
    int nan_pos = -1; uint32_t nan_tri = 0, nan_bits = 0;                // the last NaN hit
    for (bool second = false;; second = true) {
        have = false; best = tmax; best_tri = 0; best_pos = 0;
        uint32_t sp = 1;                                                 // wave-uniform
        if (lane == 0) ws[0] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        while (sp > 0u) {
            // as many nodes as fit: a round pops k and pushes at most 2k; with little room left the wave goes node by node, which
            // needs at most one more entry per level of the tree below the node (EXACT_LEVELS, kept free)
            const int room = EXACT_STACK - EXACT_LEVELS - 8 - (int)sp;
            uint32_t k = min(64u, sp);
            if ((int)k > room) k = (uint32_t)max(1, room);
            const bool active = lane < k;
            const uint32_t node = active ? ws[sp - 1u - lane] : 0u;
            sp -= k;
            float4 lo = make_float4(0, 0, 0, 0), hi = lo;
            if (active) { lo = nodes[2 * (size_t)node]; hi = nodes[2 * (size_t)node + 1]; }
            const uint32_t A = __float_as_uint(lo.w), B = __float_as_uint(hi.w);
            const bool hitb = active && hit_aabb(lo, hi, r.o, inv, tmin, tmax);
            const bool branch = hitb && (A >> 30) == 0u;
            const unsigned long long m = __ballot(branch);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");       // the pops above before the pushes below
            if (branch) {
                const uint32_t at = sp + 2u * (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                ws[at] = A & NODE_MASK; ws[at + 1u] = node + 1u;
            }
            sp += 2u * (uint32_t)__popcll(m);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            if (hitb && !branch) {
                const bool two = (A >> 30) == NODE_DOUBLE;
                const uint32_t i0 = A & NODE_MASK, i1 = two ? B : i0;
                const float4 *p0 = sc.tri + 3 * (size_t)(tri_base + i0), *p1 = sc.tri + 3 * (size_t)(tri_base + i1);
                const float4 a0 = p0[0], b0 = p0[1], c0 = p0[2], a1 = p1[0], b1 = p1[1], c1 = p1[2];     // both triangles of a DoubleLeaf requested at once
                auto take = [&](float t, uint32_t tri, uint32_t pos) {
                    if (t != t) { if ((int)pos > nan_pos) { nan_pos = (int)pos; nan_tri = tri; nan_bits = __float_as_uint(t); } }
                    else if ((int)pos > after && (!have || t < best || (t == best && pos > best_pos))) { have = true; best = t; best_tri = tri; best_pos = pos; }
                };
                float t, u0, u1, u2;
                if (hit_triangle(mk(a0.x, a0.y, a0.z), mk(b0.x, b0.y, b0.z), mk(c0.x, c0.y, c0.z), tr, tmin, tmax, t, u0, u1, u2)) take(t, i0, 2u * node);
                if (two && hit_triangle(mk(a1.x, a1.y, a1.z), mk(b1.x, b1.y, b1.z), mk(c1.x, c1.y, c1.z), tr, tmin, tmax, t, u0, u1, u2)) take(t, i1, 2u * node + 1u);
            }
        }
        if (second || __ballot(nan_pos >= 0) == 0ull) break;
        for (int d = 1; d < 64; d <<= 1) {        // the last NaN hit of the wave
            const int op = __shfl_xor(nan_pos, d); const uint32_t otri = (uint32_t)__shfl_xor((int)nan_tri, d), ob = (uint32_t)__shfl_xor((int)nan_bits, d);
            if (op > nan_pos) { nan_pos = op; nan_tri = otri; nan_bits = ob; }
        }
        after = nan_pos;
    }
    for (int d = 1; d < 64; d <<= 1) {            // butterfly: every lane ends with the wave's winner
        const bool oh = __shfl_xor((int)have, d) != 0;
        const float ot = __shfl_xor(best, d);
        const uint32_t otri = (uint32_t)__shfl_xor((int)best_tri, d), opos = (uint32_t)__shfl_xor((int)best_pos, d);
        if (oh && (!have || ot < best || (ot == best && opos > best_pos))) { have = true; best = ot; best_tri = otri; best_pos = opos; }
    }
    if (!have && after >= 0) { t_out = __uint_as_float(nan_bits); tri_out = nan_tri; return true; }
    t_out = best; tri_out = best_tri;
    return have;
}

// shape dispatch in object space.  prim: rect3d face / mesh triangle, else 0.  EXACT: a mesh takes the literal reference walk
// and aux0 is its reference tree's first node (k_extend_exact; 1: one ray per lane, 2: per wave).
// SIMPLE (round 5): the caller's scene holds spheres, axis-aligned rectangles and Rect3d boxes only, plus media whose inner shape is a sphere
// (LaunchCfg.simple_set: the host checks) — the other six shapes' code is compiled out of the caller (part2's k_extend_tlas_wide: 143 KB of
// code against a 64 KB instruction cache; a medium's two boundary tests each inlined the whole dispatch)
template <int EXACT = 0, bool MESH = true, bool SIMPLE = false>
__device__ __forceinline__ bool hit_shape(const DScene &sc, uint32_t kind, float4 q3, float4 q4, uint32_t aux0, uint32_t aux1,
                                          const Ray &r, float tmin, float tmax, uint32_t *stack_base, float &t, uint32_t &prim) {
    prim = 0;
    if (SIMPLE) {
        if (kind == 0u) return hit_sphere(q3.x, r, tmin, tmax, t);
        if (kind == 4u) return hit_rect3d(q3, q4, r, tmin, tmax, t, prim);
        if (kind >= 1u && kind <= 3u) return hit_rect_kind(kind, q3, q4, r, tmin, tmax, t);
        return false;
    }
    if (!MESH && kind == 5u) return false;     // the caller's scene holds no mesh (the host checks): the walk is compiled out
    switch (kind) {
    case 0: return hit_sphere(q3.x, r, tmin, tmax, t);
    case 1: case 2: case 3: return hit_rect_kind(kind, q3, q4, r, tmin, tmax, t);
    case 4: return hit_rect3d(q3, q4, r, tmin, tmax, t, prim);
    case 5: if (EXACT == 2) return hit_mesh_exact_wave(sc, aux0, aux1, r, tmin, tmax, stack_base, t, prim);
            if (EXACT == 1) return hit_mesh_exact_lane(sc, aux0, aux1, r, tmin, tmax, stack_base, t, prim);
            return hit_mesh(sc, aux0, aux1, r, tmin, tmax, tmax, stack_base, t, prim);
    case 7: return hit_cone(q3.x, q3.y, r, tmin, tmax, t);
    case 8: return hit_cylinder(q3.x, q3.y, q3.z, r, tmin, tmax, t);
    case 9: return hit_disk(q3.x, q3.z, q3.w, r, tmin, tmax, t);
    default: return false;
    }
}

// objects/volume.rs:56-82
template <int EXACT = 0, bool MESH = true, bool SIMPLE = false>
__device__ __forceinline__ bool hit_medium(const DScene &sc, const Obj &o, const Ray &r, float tmin, float tmax,
                                           uint32_t *stack_base, const RngKey &key, uint32_t segment, uint32_t obj_index, float &t_out) {
    const float FMAX = 3.40282347e+38f;
    uint32_t ik = obj_inner(o), prim;
    float t1, t2;
    const uint32_t root = EXACT ? sc.obj_ref_blas[obj_index] : o.aux0;
    if (SIMPLE) {       // the inner shape is a sphere (the host checks): the same two calls of volume.rs:58-61, without the dispatch
        if (!hit_sphere(o.q3.x, r, -FMAX, FMAX, t1)) return false;
        if (!hit_sphere(o.q3.x, r, t1 + 0.0001f, FMAX, t2)) return false;
    } else {
    if (!hit_shape<EXACT, MESH>(sc, ik, o.q3, o.q4, root, o.aux1, r, -FMAX, FMAX, stack_base, t1, prim)) return false;
    if (!hit_shape<EXACT, MESH>(sc, ik, o.q3, o.q4, root, o.aux1, r, t1 + 0.0001f, FMAX, stack_base, t2, prim)) return false;
    }
    t1 = fmaxf(t1, tmin);
    t2 = fminf(t2, tmax);
    if (t1 >= t2) return false;
    t1 = fmaxf(t1, 0.f);
    float dmag = mag(r.d);
    float dist_inside_boundary = (t2 - t1) * dmag;
    float xi = u2f(draw(key, P_VOLUME, segment, obj_index).x);
    float hit_distance = -fdiv(1.f, o.q4.w) * fwlm::log10f_glibc(xi);       // log10, as written (volume.rs:67); glibc's bits (fw_libm.h)
    if (hit_distance < dist_inside_boundary) { t_out = t1 + fdiv(hit_distance, dmag); return true; }
    return false;
}

// RenderObjectInternal::hit up to the object-space t (the world-space point/normal are rebuilt in k_shade)
// MEDIUM = false: the caller's scene holds no ConstantMedium (k_extend_linear_defer: the host checks), so the medium's code —
// its double-precision log10 costs registers even where it never runs — is compiled out; MESH = false likewise for the mesh walk
template <bool MEDIUM = true, int EXACT = 0, bool MESH = true, bool SIMPLE = false>
__device__ __forceinline__ bool hit_object(const DScene &sc, const Obj &o, uint32_t obj_index, const Ray &world, float tmin,
                                           float tmax, uint32_t *stack_base, const RngKey &key, uint32_t segment, float &t, uint32_t &prim) {
    Ray r = to_object_space(o, world);
    uint32_t kind = obj_kind(o);
    if (MEDIUM && kind == 6) { prim = 0; return hit_medium<EXACT, MESH, SIMPLE>(sc, o, r, tmin, tmax, stack_base, key, segment, obj_index, t); }
    const uint32_t root = (EXACT && kind == 5u) ? sc.obj_ref_blas[obj_index] : o.aux0;
    return hit_shape<EXACT, MESH, SIMPLE>(sc, kind, o.q3, o.q4, root, o.aux1, r, tmin, tmax, stack_base, t, prim);
}

// ------------------------------------------------------------------------------------------------
// K2  extend: closest hit of every queued ray
// ------------------------------------------------------------------------------------------------
extern __shared__ uint32_t lds_stack[];

// Deferred mesh work (k_extend_bvh): most rays of a chunk miss a mesh's box while a few walk its BLAS
// (rocprofv3 on suzanne: 14 % of lanes active).  A ray that reaches a mesh leaf of the TLAS does not enter the
// BLAS; it parks (slot, mesh object, best hit so far) in a wave-private LDS list, and whenever 64 entries have
// accumulated the whole wave walks BLASes together, one parked ray per lane, and writes the final hit records.
#ifndef FW_BLAS_RUN_MIN
#define FW_BLAS_RUN_MIN 128
#endif
#ifndef FW_BLAS_REFILL_MIN
#define FW_BLAS_REFILL_MIN 16
#endif
constexpr uint32_t DEFER_CAP = FW_BLAS_RUN_MIN + 64;        // entries; a chunk adds at most 64 and the list is emptied when it reaches BLAS_RUN_MIN
constexpr uint32_t BLAS_RUN_MIN = FW_BLAS_RUN_MIN;     // parked rays that start a BLAS run (the more, the smaller the share of its tail)
#ifndef FW_TLAS_REFILL_MIN
#define FW_TLAS_REFILL_MIN 16
#endif
#ifndef FW_TLAS_WAVES
#define FW_TLAS_WAVES 4      // round 4: the gate check (obj_gate_ok) and the relaxed exit planes took the refilling L2 walks past the 96 registers of 5 waves (24-36 spills)
#endif
#ifndef FW_BLAS_WAVES
#define FW_BLAS_WAVES 5
#endif

#ifndef FW_TLAS_SCAN_MAX
#define FW_TLAS_SCAN_MAX 8
#endif
constexpr uint32_t TLAS_SCAN_MAX = FW_TLAS_SCAN_MAX;       // up to this many objects the TLAS is scanned, not walked
constexpr uint32_t TLAS_REFILL_MIN = FW_TLAS_REFILL_MIN;   // idle lanes that trigger a refill of the TLAS walk
constexpr uint32_t BLAS_REFILL_MIN = FW_BLAS_REFILL_MIN;   // idle lanes that trigger a refill inside a run
#ifndef FW_BLAS_WALK_NUM
#define FW_BLAS_WALK_NUM 2
#define FW_BLAS_WALK_DEN 1
#endif
#ifndef FW_WALK_NUM
#define FW_WALK_NUM 4
#define FW_WALK_DEN 1
#endif
static_assert(FW_WALK_NUM > FW_WALK_DEN, "the walk must continue while every busy lane walks");
constexpr uint32_t WALK_NUM = FW_WALK_NUM, WALK_DEN = FW_WALK_DEN;   // same rule for the TLAS walk
static_assert(FW_BLAS_WALK_NUM > FW_BLAS_WALK_DEN, "the walk must continue while every busy lane walks");
constexpr uint32_t BLAS_WALK_NUM = FW_BLAS_WALK_NUM, BLAS_WALK_DEN = FW_BLAS_WALK_DEN;   // node walking stops when walkers * NUM <= busy lanes * DEN
static_assert(FW_WB == 64, "the parked-ray list and the wave-private queues assume single-wave workgroups");
static_assert(FW_TLAS_SCAN_MAX <= 24, "k_extend_scan parks a ray with a bit mask of its meshes");
// a parked entry's second word: the mesh object to walk, or PARK_MASK | a bit per mesh object (k_extend_scan: all the meshes the ray reaches)
constexpr uint32_t PARK_MASK = 0x80000000u;
__device__ __forceinline__ uint32_t park_next_mesh(uint32_t &word) {       // -> the next object to walk; word = what is left (0: nothing)
    if (!(word & PARK_MASK)) { const uint32_t obj = word; word = 0u; return obj; }
    const uint32_t m = word & ~PARK_MASK, obj = (uint32_t)__ffs((int)m) - 1u, rest = m & (m - 1u);
    word = rest ? (PARK_MASK | rest) : 0u;
    return obj;
}

// The LDS-resident walks (k_blas_lds, k_extend_tlas_lds) read the entries of SEVERAL wave queues as one stream of blocks of
// <= 64 (one register read-ahead buffer each): a wave that has handed out its queue takes the next one of its workgroup, so
// rays refill across queue boundaries and only a workgroup's very last rays walk in a thinning wave.  Which physical wave
// walks a ray does not matter: every result is written to the ray's own slot.
// (Tried on the L2-fetching walks too — several consecutive queues per wave, statically dealt or taken from a device-wide
// atomic: the busy lanes of suzanne's k_blas node loop rose from 55 to 80 %, wave-iterations fell 29 %, and the kernel got
// SLOWER, 10.2 -> 15.4 / 11.7 ms: those walks are bound by their node gathers, runs of neighbouring queues are unevenly
// loaded, and the extra registers cost a wave per SIMD.  They keep one queue per wave.
// Also tried here: ONE device-wide counter instead of one per workgroup, so that no CU idles while another still has
// queues (the LDS walks average 2.4-2.8 of 4 waves per SIMD): suzanne @64 11.1 -> 13.4 ms, part2 @16 10.4 -> 12.6 ms — a
// device-scope atomic per queue from 4 096 waves on one address is slow, and the early-finishing CUs were not idle: they
// run the other batch's k_shade / k_extend_scan.)
struct BlockStream {
    // the wave's queues are named by an index k: queue id = q_off + k * q_mul (k-ranges are dealt statically or taken from a counter)
    uint32_t q, q_end, q0;     // next index to open; end and start of this wave's static range
    uint32_t q_off, q_mul;
    uint32_t stride;           // slots per queue region
    uint32_t base, n, off;     // the open queue: first slot, entries, offset of its next block
    uint32_t cnt;              // lane j holds the entry count of the queue with index q0 + j (read once, at the start)
    // dynamic mode (dyn_ctr != nullptr): after its static range the wave takes further indices, one at a time, from a counter
    // (in LDS: the waves of a workgroup share the workgroup's queues)
    uint32_t *dyn_ctr; const uint32_t *dyn_counts; uint32_t dyn_first, dyn_end; bool dyn_done;
    // unit mode (unit > 0; the LDS walks since round 3): the counter hands out UNITS of `unit` rays — queue index u / bpq, rays
    // [u % bpq * unit, + unit) of it — instead of whole queues: a queue holds up to ~2 900 rays, 225 us of walking for the wave that
    // took the last one while the other fifteen of its workgroup had nothing left (the LDS walks averaged 2.7 of 4 waves per SIMD).
    // unit_counts: the workgroup's queue counts, staged in LDS (a unit beyond its queue's count is skipped for ~100 ns).
    uint32_t unit, bpq; const uint32_t *unit_counts;
    uint32_t *ptotal;          // statistics: per-queue totals of parked rays (dynamic mode of k_blas_lds), or nullptr
    __device__ __forceinline__ void init(const uint32_t *counts, uint32_t first, uint32_t last, uint32_t stride_, uint32_t lane,
                                         uint32_t off_ = 0u, uint32_t mul_ = 1u) {
        q = q0 = first; q_end = last; q_off = off_; q_mul = mul_; stride = stride_; base = 0; n = 0; off = 0;
        cnt = (first + lane < last) ? counts[off_ + (first + lane) * mul_] : 0u;
        dyn_ctr = nullptr; dyn_counts = counts; dyn_first = dyn_end = 0; dyn_done = true; ptotal = nullptr; unit = 0; bpq = 1; unit_counts = nullptr;
    }
    __device__ __forceinline__ void init_dynamic(uint32_t *ctr, uint32_t first, uint32_t end) { dyn_ctr = ctr; dyn_first = first; dyn_end = end; dyn_done = first >= end; }
    __device__ __forceinline__ void init_units(uint32_t *ctr, uint32_t n_queues, uint32_t unit_, const uint32_t *counts_in_lds) {
        dyn_ctr = ctr; dyn_first = 0; dyn_end = n_queues; dyn_done = n_queues == 0; unit = unit_; bpq = (stride + unit_ - 1u) / unit_; unit_counts = counts_in_lds;
    }
    // the next block of the stream: first slot and entry count (wave-uniform); false when the wave's queues are used up
    __device__ __forceinline__ bool next(uint32_t &b_base, uint32_t &b_n) {
        for (;;) {
            if (off < n) { b_base = base + off; b_n = min(64u, n - off); off += 64u; return true; }
            if (q >= q_end) {
                if (dyn_done) return false;
                uint32_t v = 0;
                if ((threadIdx.x & 63u) == 0) v = atomicAdd(dyn_ctr, 1u);
                const uint32_t u = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
                if (unit) {
                    const uint32_t ku = u / bpq, b = u - ku * bpq;
                    if (ku >= dyn_end) { dyn_done = true; return false; }
                    const uint32_t nqu = q_off + ku * q_mul, cnt_q = unit_counts[ku], start = b * unit;
                    if (ptotal && b == 0u && cnt_q && (threadIdx.x & 63u) == 0) ptotal[nqu] += cnt_q;
                    if (start >= cnt_q) continue;
                    base = nqu * stride + start; n = min(unit, cnt_q - start); off = 0;
                    continue;
                }
                const uint32_t k = dyn_first + u;
                if (k >= dyn_end) { dyn_done = true; return false; }
                const uint32_t nq = q_off + k * q_mul;
                n = dyn_counts[nq]; base = nq * stride; off = 0;
                if (ptotal && n && (threadIdx.x & 63u) == 0) ptotal[nq] += n;
                continue;
            }
            n = (uint32_t)__shfl((int)cnt, (int)(q - q0));
            base = (q_off + q * q_mul) * stride; off = 0; q++;
        }
    }
};
// The objects the host kept out of the walked TLAS (DScene.hoisted: boxes that cover most of the scene, met by nearly every
// ray): tested here for one ray with wave-uniform object indices — scalar loads, no divergence between lanes — exactly as
// their leaf would: the object's leaf box first (obj_leaf = its box in the walked tree: that of its reference leaf node), the gate box where it has one,
// then the object with the caller's [TMIN, TMAX], ties by reference rank.  A walk that starts from this result culls
// against its t like against any other hit.
template <bool SIMPLE = false>
__device__ __forceinline__ void hoisted_hits(const DScene &sc, const Ray &r, V3 inv, const RngKey &key, int segment, uint32_t *blas_stack,
                                             bool &have, float &best_t, uint32_t &best_obj, uint32_t &best_prim) {
    const float TMIN = 0.001f, TMAX = 2e9f;                          // render.rs:19
    for (uint32_t h = 0; h < sc.n_hoisted; h++) {
        const uint32_t k = sc.hoisted[h];
        float entry;
        if (!hit_aabb_entry(sc.obj_leaf[2 * (size_t)k], sc.obj_leaf[2 * (size_t)k + 1], r.o, inv, relaxed(inv), box_tmin(TMIN, false), TMAX, entry)) continue;
        Obj o = load_obj(sc.obj, k);
        float t; uint32_t prim;
        if (SIMPLE ? hit_object<true, 0, false, true>(sc, o, k, r, TMIN, TMAX, blas_stack, key, segment, t, prim) : hit_object(sc, o, k, r, TMIN, TMAX, blas_stack, key, segment, t, prim)) {
            if ((!have || t < best_t || (t == best_t && sc.obj_rank[k] > sc.obj_rank[best_obj])) && obj_gate_ok(sc, k, r.o, inv)) { have = true; best_t = t; best_obj = k; best_prim = prim; }
        }
    }
}

// Closest hit of one ray: the linear scan of scene.rs:137-149 or the TLAS walk of bvh.rs:115-151.  With DEFER a
// ray that reaches a mesh leaf of the TLAS reports (deferred, deferred_obj) instead of entering the BLAS.
// MESH = false: the caller's scene holds no mesh (launch_extend checks): the mesh walk is compiled out of the linear scan, which is
// bound by instruction issue and whose code then fits the instruction cache (k_extend_linear: 93 KB with it)
// MEDIUM = false likewise for a scene without a ConstantMedium (its two boundary tests and double-precision log10): hdri's scan is then
// a quarter of the code and keeps its registers
template <bool USE_BVH, bool DEFER, bool MESH = true, bool MEDIUM = true, bool SIMPLE = false>
__device__ __forceinline__ void closest_hit(const DScene &sc, const Ray &r, const RngKey &key, int segment,
                                            uint32_t *my_stack, uint32_t *blas_stack, float &best_t, uint32_t &best_obj,
                                            uint32_t &best_prim, bool &deferred, uint32_t &deferred_obj, bool soft = false) {
    const float TMIN = 0.001f, TMAX = 2e9f;                          // render.rs:19
    if (!USE_BVH) {
        // scene.rs:137-149: linear scan with narrowing; a later object replaces on t <= closest
        // Camera rays of one chunk (4 x 16 neighbouring pixels) mostly miss the same objects, so an expensive shape is
        // skipped for the whole wave when no lane's ray can reach its inflated world box (cornell: 2 boxes x ~190
        // instructions, skipped for ~85 % of the segment-0 chunks).  Later segments are incoherent: some lane always
        // hits, the pre-test would only cost.  The pre-test is conservative by construction: approximate reciprocals, a
        // box inflated by 1e-4 of the scene and ray-origin scale (the exact tests err by a few ulp of that), no clipping
        // against the best t, NaN-dropping min/max (a lane on a slab boundary with d = 0 counts as a possible hit).
        const bool cull0 = segment == 0;
        V3 ainv = mk(0.f, 0.f, 0.f); float eps = 0.f;
        if (cull0) {
            ainv = mk(__builtin_amdgcn_rcpf(r.d.x), __builtin_amdgcn_rcpf(r.d.y), __builtin_amdgcn_rcpf(r.d.z));
            eps = 1e-4f * (fabsf(r.o.x) + fabsf(r.o.y) + fabsf(r.o.z));
        }
        if (cull0) {
            for (uint32_t k = 0; k < sc.n_objects; k++) {
                Obj o = load_obj(sc.obj, k);     // wave-uniform index: scalar loads
                if (obj_flags(o) & OF_CULL0) {
                    const float4 lo = sc.obj_cull[2 * (size_t)k], hi = sc.obj_cull[2 * (size_t)k + 1];
                    const float m = eps + 1e-4f * (fmaxf(fmaxf(fabsf(lo.x), fabsf(lo.y)), fabsf(lo.z)) + fmaxf(fmaxf(fabsf(hi.x), fabsf(hi.y)), fabsf(hi.z))) + 1e-6f;
                    float t0 = (lo.x - m - r.o.x) * ainv.x, t1 = (hi.x + m - r.o.x) * ainv.x;
                    float tn = fminf(t0, t1), tf = fmaxf(t0, t1);
                    t0 = (lo.y - m - r.o.y) * ainv.y; t1 = (hi.y + m - r.o.y) * ainv.y;
                    tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
                    t0 = (lo.z - m - r.o.z) * ainv.z; t1 = (hi.z + m - r.o.z) * ainv.z;
                    tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
                    const bool maybe = !(tf < tn * (1.f - 1e-4f) - 1e-4f) && !(tf < 0.f);
                    if (__ballot(maybe) == 0ull) continue;
                }
                float t; uint32_t prim;
                if (hit_object<MEDIUM, 0, MESH, SIMPLE>(sc, o, k, r, TMIN, best_t, blas_stack, key, segment, t, prim)) { best_t = t; best_obj = k; best_prim = prim; }
            }
        } else {
            // the loop of all later segments: no pre-test, nothing but the scan (the scalar unit is as busy as the VALUs here)
            const float4 *op = sc.obj;
            for (uint32_t k = 0; k < sc.n_objects; k++, op += OBJ_Q) {
                Obj o = load_obj(op, 0);
                float t; uint32_t prim;
                if (hit_object<MEDIUM, 0, MESH, SIMPLE>(sc, o, k, r, TMIN, best_t, blas_stack, key, segment, t, prim)) { best_t = t; best_obj = k; best_prim = prim; }
            }
        }
    } else {
        // bvh.rs:88-98,115-151 over RenderObjectInternal items; same scheme as hit_mesh
        V3 inv = mk(fdiv(1.f, r.d.x), fdiv(1.f, r.d.y), fdiv(1.f, r.d.z));
        LdsStack st{my_stack, 0};
        bool have = false;
        uint32_t cur = sc.tlas_root;
        if (sc.n_hoisted) hoisted_hits(sc, r, inv, key, segment, blas_stack, have, best_t, best_obj, best_prim);
        // Rounds of (node steps, then object tests).  A lane walks pair nodes until it holds a leaf or is out of tree; the
        // WAVE stops walking once no more than a quarter of its busy lanes still walk, so that most lanes test their object now
        // instead of waiting for the stragglers, who go on in the next round (suzanne BLAS: 15.2 -> 12.2 ms with the same
        // rule; the plain while-while ran the node loop at 30 % lane utilisation).  All lanes that entered call this
        // together: the ballots see exactly them.
        for (;;) {
            const bool busy = cur != REF_DONE;
            const uint32_t n_busy = (uint32_t)__popcll(__ballot(busy));
            if (n_busy == 0u) break;
            for (;;) {
                const bool walking = busy && !(cur & REF_LEAF);
                const uint32_t n_walk = (uint32_t)__popcll(__ballot(walking));
                if (n_walk == 0u || (n_walk < n_busy && n_walk * WALK_NUM <= n_busy * WALK_DEN)) break;   // n_walk < n_busy: somebody holds a leaf or has finished
                if (walking) {
                    TS_TICK(0);
                    cur = pair_step(sc.tlas, cur, r.o, inv, relaxed(inv), box_tmin(TMIN, soft), TMAX, soft ? NO_CULL : cull_bound(have ? best_t : TMAX), st);
                }
            }
            if (!busy || !(cur & REF_LEAF) || cur == REF_DONE) continue;
            const uint32_t item = cur & NODE_MASK;
            cur = st.sp ? st.pop() : REF_DONE;
            Obj o = load_obj_for_hit(sc.obj, item);
            if (DEFER && sc.has_mesh && obj_kind(o) == 5u && !deferred) {                  // park the first mesh — if the reference's walk reaches it (obj_gate_ok)
                if (obj_gate_ok(sc, item, r.o, inv)) { deferred = true; deferred_obj = item; }
                continue;
            }
            TS_TICK(2);
            float t; uint32_t prim;
            if (hit_object(sc, o, item, r, TMIN, TMAX, blas_stack, key, segment, t, prim)) {
                if ((!have || t < best_t || (t == best_t && sc.obj_rank[item] > sc.obj_rank[best_obj])) && obj_gate_ok(sc, item, r.o, inv)) { have = true; best_t = t; best_obj = item; best_prim = prim; }
            }
        }
    }
}

template <bool USE_BVH, bool REFILL, bool PARK, bool MESH = true, bool MEDIUM = true, bool SIMPLE = false>
__device__ __forceinline__ void extend_body(const DScene &sc, const DFrame &f, const DPaths &in, float2 *__restrict__ hits,
                                            const DQueue &q, int segment, int tlas_levels, int stack_levels, const DPark &park) {
    const uint32_t w = wave_index(), lane = threadIdx.x & 63u;
    if (w >= q.n_waves) return;
    TS_BEGIN();
    const uint32_t n = q.wcount[(size_t)segment * q.n_waves + w];
    const uint32_t base = w * q.cap;
    uint32_t *my_stack = lds_stack + threadIdx.x;                    // [level][lane]
    uint32_t *blas_stack = my_stack + (size_t)tlas_levels * WB;      // BLAS levels sit above the TLAS levels
    // deferred-mesh list (SoA) behind the stacks; WB == 64 when USE_BVH defers (one wave per workgroup)
    uint32_t *e_slot = lds_stack + (size_t)stack_levels * WB, *e_obj = e_slot + DEFER_CAP, *e_bobj = e_obj + DEFER_CAP,
             *e_bprim = e_bobj + DEFER_CAP;
    float *e_t = reinterpret_cast<float *>(e_bprim + DEFER_CAP);
    uint32_t list_n = 0;
    const float TMIN = 0.001f, TMAX = 2e9f;                          // render.rs:19

    // Walks the BLASes of ALL parked rays.  64 rays start together; a lane that finishes its ray writes the hit record and,
    // once BLAS_REFILL_MIN lanes are idle, the idle lanes take further parked rays (from the end of the LDS list), so the
    // few deep walks of a batch overlap with many short ones instead of holding 63 idle lanes (suzanne: the node loop ran
    // at 14 % lane utilisation, tools/trav_stats.py).  Same arithmetic and tie rules as hit_mesh + the merge of the old
    // one-shot flush; the order in which parked rays are walked does not matter (each writes its own slot).
    auto blas_run = [&]() {
        bool act = false, have = false;
        uint32_t slot = 0, obj = 0, tri_base = 0, bobj = MISS, bprim = 0, mtri = 0, cur = REF_DONE;
        float bt = TMAX, mbest = TMAX;
        V3 ro = mk(0, 0, 0), inv = ro;
        bool soft = false;
        TriRay tr{mk(0, 0, 0), 0, 1, 2, 0.f, 0.f, 0.f};
        LdsStack st{blas_stack, 0};
        for (;;) {
            const unsigned long long idle_mask = __ballot(!act);
            const uint32_t n_idle = (uint32_t)__popcll(idle_mask);
            if (list_n > 0u && n_idle >= BLAS_REFILL_MIN) {
                const uint32_t take = min(n_idle, list_n);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_mask, 0u));
                if (!act && rank < take) {
                    const uint32_t e = list_n - 1u - rank;
                    slot = e_slot[e]; obj = e_obj[e]; bt = e_t[e]; bobj = e_bobj[e]; bprim = e_bprim[e];
                    float4 ra = qld(&in.ray_a[slot]); float2 rb = load_ray_b(in, slot, f, segment);
                    Obj o = load_obj(sc.obj, obj);
                    Ray r = to_object_space(o, make_ray(ra, rb, f, segment));
                    ro = r.o; inv = mk(fdiv(1.f, r.d.x), fdiv(1.f, r.d.y), fdiv(1.f, r.d.z));
                    soft = soft_direction(r.d.x, r.d.y, r.d.z, sc.soft_shear);
                    tr = make_triray(r);
                    tri_base = o.aux1; cur = o.aux0; st.sp = 0;
                    have = false; mbest = TMAX; mtri = 0; act = true;
                }
                list_n -= take;
            }
            const uint32_t n_idle_after = (uint32_t)__popcll(__ballot(!act));
            if (n_idle_after == 64u) break;                            // the list is empty too: 64 idle lanes would have refilled
            // walk pair nodes until this lane holds a triangle (or is out of tree) — but the WAVE stops walking as soon as
            // no more than a quarter of its busy lanes still walk: the rest test their triangles now instead of waiting
            // for the stragglers, who simply go on in the next round
            const uint32_t n_act = 64u - n_idle_after;
            for (;;) {
                const bool walking = act && !(cur & REF_LEAF);
                const uint32_t n_walk = (uint32_t)__popcll(__ballot(walking));
                if (n_walk == 0u || (n_walk < n_act && n_walk * BLAS_WALK_NUM <= n_act * BLAS_WALK_DEN)) break;   // n_walk < n_act: somebody holds a leaf, the round makes progress
                if (walking) {
                    TS_TICK(4);
                    cur = pair_step(sc.blas, cur, ro, inv, relaxed(inv), box_tmin(TMIN, soft), TMAX, soft ? NO_CULL : cull_bound(have ? fminf(mbest, bt) : bt), st);
                }
            }
            if (act && (cur & REF_LEAF)) {
                if (cur != REF_DONE) {
                    const uint32_t item = cur & NODE_MASK;
                    cur = st.sp ? st.pop() : REF_DONE;
                    TS_TICK(6);
                    const float4 *tp = sc.tri + 3 * (size_t)(tri_base + item);
                    float4 a = tp[0], b = tp[1], c = tp[2];
                    float t, b0, b1, b2;
                    if (hit_triangle(mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), tr, TMIN, TMAX, t, b0, b1, b2)) {
                        if ((!have || t < mbest || (t == mbest && sc.tri_rank[tri_base + item] > sc.tri_rank[tri_base + mtri])) &&
                        tri_gate_ok(sc, tri_base + item, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), ro, inv, TMIN, TMAX)) { have = true; mbest = t; mtri = item; }
                    }
                }
                if (cur == REF_DONE) {                                  // this ray is finished: merge with what the TLAS walk held
                    if (have && (bobj == MISS || mbest < bt || (mbest == bt && sc.obj_rank[obj] > sc.obj_rank[bobj]))) { bt = mbest; bobj = obj; bprim = mtri; }
                    qst(&hits[slot], pack_hit(bt, bobj, bprim, sc.prim_bits));
                    act = false;
                }
            }
        }
    };

    if (USE_BVH && REFILL) {
        // ---- TLAS walk with in-wave refill.  The chunked loop below gives every lane one ray of a 64-ray chunk and waits
        // for the slowest: on part2 only 40 % of the lanes are still busy in an average round (tools/trav_stats.py).  Here a
        // lane that has finished its ray writes the hit record (or parks the ray for its mesh) and, once TLAS_REFILL_MIN
        // lanes are idle, the idle lanes take the next rays of the wave's queue.  The queue is read 64 rays ahead into
        // registers (one buffer being handed out through ds_bpermute, one in flight), so a refill never waits for HBM.
        // Same tests, same tie rules, same culling as closest_hit<true>: the bits do not change.
        float4 ca = make_float4(0, 0, 0, 0), na = ca; float2 cb = make_float2(0, 0), nb = cb; float cs = 0.f, ns = 0.f;
        uint32_t cur_base = 0, q_next = 0;
        auto fetch = [&](uint32_t j, float4 &a, float2 &b, float &st) {
            if (j < n) { a = qld(&in.ray_a[base + j]); b = load_ray_b(in, base + j, f, segment); if (sc.has_medium) st = load_home(in, base + j, f, segment); }
        };
        fetch(lane, ca, cb, cs); fetch(64u + lane, na, nb, ns);
        const uint32_t IDLE = 0xffffffffu;
        uint32_t slot = IDLE, cur = REF_DONE, path_id = 0, best_obj = MISS, best_prim = 0, deferred_obj = 0;
        float best_t = TMAX; bool have = false, deferred = false;
        V3 wo = mk(0, 0, 0), wd = wo, inv = wo;
        bool soft = false;                                              // the SOFT class (scenes with meshes): no culling
        LdsStack st{my_stack, 0};
        uint32_t park_n = 0;                                            // rays handed over to k_blas so far (wave-uniform)
        for (;;) {
            const unsigned long long idle_mask = __ballot(slot == IDLE);
            const uint32_t n_idle = (uint32_t)__popcll(idle_mask);
            if (q_next < n) {
                if (n_idle >= TLAS_REFILL_MIN) {
                    const uint32_t take = min(n_idle, min(n, cur_base + 64u) - q_next);
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_mask, 0u));
                    const uint32_t src = (q_next - cur_base) + rank;
                    const int sel = (int)((src & 63u) << 2);
                    float4 ra; float2 rb;
                    ra.x = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(ca.x)));
                    ra.y = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(ca.y)));
                    ra.z = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(ca.z)));
                    ra.w = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(ca.w)));
                    rb.x = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(cb.x)));
                    rb.y = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(cb.y)));
                    const float rs = sc.has_medium ? __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(cs))) : 0.f;
                    if (slot == IDLE && rank < take) {
                        slot = q_next + rank;
                        const Ray r = make_ray(ra, rb, f, segment);
                        wo = r.o; wd = r.d;
                        inv = mk(fdiv(1.f, wd.x), fdiv(1.f, wd.y), fdiv(1.f, wd.z));
                        soft = PARK && soft_ray(f.ex, wd, sc.soft_shear);
                        path_id = __float_as_uint(rs);
                        cur = skip_ray(f.ex, r) ? REF_DONE : sc.tlas_root; st.sp = 0;      // a NaN ray's record comes from k_extend_exact
                        have = false; best_t = TMAX; best_obj = MISS; best_prim = 0; deferred = false; deferred_obj = 0;
                        if (!PARK && sc.n_hoisted) {      // objects are hoisted only in scenes without meshes
                            RngKey hkey{0, 0, 0};
                            if (sc.has_medium) hkey = key_of(f, path_id);
                            hoisted_hits(sc, r, inv, hkey, segment, blas_stack, have, best_t, best_obj, best_prim);
                        }
                    }
                    q_next += take;
                    if (q_next == cur_base + 64u && q_next < n) {      // cur is used up: nxt becomes cur, read 64 further ahead
                        ca = na; cb = nb; cs = ns; cur_base += 64u;
                        fetch(cur_base + 64u + lane, na, nb, ns);
                    }
                }
            } else if (n_idle == 64u) {
                break;                                                  // queue empty and every lane has retired its ray
            }

            // ---- one round: node steps (the wave stops once no more than a quarter of its busy lanes still walk), then objects
            const bool busy = slot != IDLE && cur != REF_DONE;
            const uint32_t n_busy = (uint32_t)__popcll(__ballot(busy));
            for (;;) {
                const bool walking = busy && !(cur & REF_LEAF);
                const uint32_t n_walk = (uint32_t)__popcll(__ballot(walking));
                if (n_walk == 0u || (n_walk < n_busy && n_walk * WALK_NUM <= n_busy * WALK_DEN)) break;
                if (walking) {
                    TS_TICK(0);
                    cur = pair_step(sc.tlas, cur, wo, inv, relaxed(inv), box_tmin(TMIN, soft), TMAX, soft ? NO_CULL : cull_bound(have ? best_t : TMAX), st);
                }
            }
            if (busy && (cur & REF_LEAF) && cur != REF_DONE) {
                const uint32_t item = cur & NODE_MASK;
                cur = st.sp ? st.pop() : REF_DONE;
                Obj o = load_obj_for_hit(sc.obj, item);
                if (PARK && obj_kind(o) == 5u && !deferred) {            // the first mesh: this ray goes to k_blas — if the reference's walk reaches the mesh (obj_gate_ok)
                    if (obj_gate_ok(sc, item, wo, inv)) { deferred = true; deferred_obj = item; }
                } else {
                    TS_TICK(2);
                    RngKey key{0, 0, 0};
                    if (sc.has_medium) key = key_of(f, path_id);
                    float t; uint32_t prim;
                    if (hit_object(sc, o, item, Ray{wo, wd}, TMIN, TMAX, blas_stack, key, segment, t, prim)) {
                        if ((!have || t < best_t || (t == best_t && sc.obj_rank[item] > sc.obj_rank[best_obj])) && obj_gate_ok(sc, item, wo, inv)) { have = true; best_t = t; best_obj = item; best_prim = prim; }
                    }
                }
            }

            // ---- retire the rays that are out of tree
            const bool done = slot != IDLE && cur == REF_DONE;
            if (done && !deferred) qst(&hits[base + slot], pack_hit(best_t, best_obj, best_prim, sc.prim_bits));
            if (PARK) {          // dense append to the wave's parked queue: the world ray and what the TLAS walk found so far
                const unsigned long long pmask = __ballot(done && deferred);
                if (pmask) {
                    const uint32_t prank = __builtin_amdgcn_mbcnt_hi((uint32_t)(pmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pmask, 0u));
                    if (done && deferred) {
                        const uint32_t e = w * park.stride + park_n + prank;
                        qst(&park.ray_a[e], make_float4(wo.x, wo.y, wo.z, wd.x));
                        qst(&park.ray_b[e], make_float2(wd.y, wd.z));
                        qst(&park.meta[e], make_float4(__uint_as_float(base + slot), __uint_as_float(deferred_obj), best_t, pack_hit(best_t, best_obj, best_prim, sc.prim_bits).y));
                    }
                    park_n += (uint32_t)__popcll(pmask);
                }
            }
            if (done) slot = IDLE;
        }
        if (PARK && lane == 0) park.pcount[w] = park_n;
        TS_END();
        return;
    }

    // software pipeline: the next chunk's ray is requested before the current chunk is traversed
    float4 ra_n = make_float4(0, 0, 0, 0); float2 rb_n = make_float2(0, 0);
    if (lane < n) { ra_n = qld(&in.ray_a[base + lane]); rb_n = load_ray_b(in, base + lane, f, segment); }
    for (uint32_t c0 = 0; c0 < n; c0 += 64u) {
        const uint32_t j = c0 + lane;
        const uint32_t i = base + j;
        float4 ra = ra_n; float2 rb = rb_n;
        if (j + 64u < n) { ra_n = qld(&in.ray_a[i + 64u]); rb_n = load_ray_b(in, i + 64u, f, segment); }
        const bool active = j < n;
        bool deferred = false; uint32_t deferred_obj = 0;
        float best_t = TMAX; uint32_t best_obj = MISS, best_prim = 0;
        if (active) {
            Ray r = make_ray(ra, rb, f, segment);
            RngKey key{0, 0, 0};
            if (sc.has_medium) key = key_of(f, __float_as_uint(load_home(in, i, f, segment)));
            if (!skip_ray(f.ex, r))     // a NaN ray's record comes from k_extend_exact
                closest_hit<USE_BVH, USE_BVH, MESH, MEDIUM, SIMPLE>(sc, r, key, segment, my_stack, blas_stack, best_t, best_obj, best_prim, deferred, deferred_obj,
                                              USE_BVH && sc.has_mesh && soft_ray(f.ex, r.d, sc.soft_shear));
            if (!USE_BVH && f.hit4) reinterpret_cast<uint32_t *>(hits)[i] = __float_as_uint(pack_hit(best_t, best_obj, best_prim, sc.prim_bits).y);   // the code alone: k_shade recomputes t
            else if (!deferred) qst(&hits[i], pack_hit(best_t, best_obj, best_prim, sc.prim_bits));
        }
        if (USE_BVH && sc.has_mesh) {
            unsigned long long mask = __ballot(deferred);
            if (mask) {
                uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                if (deferred) {
                    uint32_t e = list_n + rank;
                    e_slot[e] = i; e_obj[e] = deferred_obj; e_t[e] = best_t; e_bobj[e] = best_obj; e_bprim[e] = best_prim;
                }
                list_n += (uint32_t)__popcll(mask);
                if (list_n >= BLAS_RUN_MIN) blas_run();
            }
        }
    }
    if (USE_BVH && list_n) blas_run();
    TS_END();
}
// ------------------------------------------------------------------------------------------------
// K2''  k_extend_scan: scenes whose TLAS holds at most TLAS_SCAN_MAX objects (suzanne: floor, light, mesh; teapot: 4 meshes,
// floor, light).  Its own kernel since round 2: inside the one body of all BVH entry points it shared their 96 registers
// (5 waves per SIMD) although it only streams rays through a few wave-uniform tests and waits for HBM.
// ------------------------------------------------------------------------------------------------
#ifndef FW_DEFER_WAVES
#define FW_DEFER_WAVES 7
#endif
#ifndef FW_DEFER_LDS_PAD
#define FW_DEFER_LDS_PAD 0      // timing builds: LDS a workgroup of the box-list scan asks for beyond its lists (caps its waves per CU)
#endif
#ifndef FW_SCAN_WAVES
#define FW_SCAN_WAVES 6
#endif
#ifndef FW_SCAN_WAVES_PLAIN
#define FW_SCAN_WAVES_PLAIN 7      // the plain scan (no medium, meshes parked) fits 72 registers without spills: suzanne 64.9 -> 64.2 ms, teapot @128 48.6 -> 47.2 (gpurun_out/r04y/scan_variants.txt)
#endif
// PLAIN: the scene holds no ConstantMedium (the host checks) — meshes are parked, never walked here, so without a medium neither the mesh
// walk nor the medium's code is needed: compiled out (53 KB of code and 11 spilled registers with them)
template <bool PARK, bool PLAIN = false, bool SIMPLE = false>     // SIMPLE (with PLAIN): whatever is not a mesh is a sphere, a rectangle or a box (suzanne, teapot)
__global__ __launch_bounds__(WB) __attribute__((amdgpu_waves_per_eu(PLAIN ? FW_SCAN_WAVES_PLAIN : FW_SCAN_WAVES, 8))) void k_extend_scan(DScene sc, DFrame f, DPaths in, float2 *__restrict__ hits, DQueue q, int segment,
                                                    int tlas_levels, float4 *__restrict__ park_a, float2 *__restrict__ park_b, float4 *__restrict__ park_m,
                                                    uint32_t *__restrict__ park_count, uint32_t park_stride) {
    // the park arrays are `__restrict__` kernel arguments, not a DPark: only then can the compiler prove that the stores
    // into them leave the scene tables alone, and read the (wave-uniform) object records with scalar loads
    const uint32_t w = wave_index(), lane = threadIdx.x & 63u;
    if (w >= q.n_waves) return;
    const uint32_t n = q.wcount[(size_t)segment * q.n_waves + w];
    const uint32_t base = w * q.cap;
    uint32_t *blas_stack = lds_stack + threadIdx.x + (size_t)tlas_levels * WB;   // a medium around a mesh walks its BLAS in place
    const float TMIN = 0.001f, TMAX = 2e9f;                          // render.rs:19
    // ---- a TLAS of a handful of objects (suzanne: floor, light, mesh) is not walked but scanned: every lane tests every
    // object's box (the very box and slab test its leaf has in the tree, culled against the best t like pair_step) and, where
    // that passes, the object — with wave-uniform object indices, i.e. scalar loads and no divergence between lanes.  The
    // tree walk reaches an object iff its own box passes (its ancestors' boxes are supersets, and the slab arithmetic is
    // monotone), and ties are decided by rank, not by order, so the result is the tree walk's bit for bit
    // (suzanne's k_extend_tlas_park: 3.5 of the 11 ms went into walking three objects).
    uint32_t park_n = 0;
    float4 ra_n = make_float4(0, 0, 0, 0); float2 rb_n = make_float2(0, 0);
    if (lane < n) { ra_n = qld(&in.ray_a[base + lane]); rb_n = load_ray_b(in, base + lane, f, segment); }
    for (uint32_t c0 = 0; c0 < n; c0 += 64u) {
        const uint32_t j = c0 + lane, i = base + j;
        const float4 ra = ra_n; const float2 rb = rb_n;
        if (j + 64u < n) { ra_n = qld(&in.ray_a[i + 64u]); rb_n = load_ray_b(in, i + 64u, f, segment); }
        const bool active = j < n;
        bool deferred = false, have = false; uint32_t deferred_obj = 0, best_obj = MISS, best_prim = 0; float best_t = TMAX;
        Ray r{mk(0, 0, 0), mk(0, 0, 1)};
        if (active) {
            r = make_ray(ra, rb, f, segment);
            RngKey key{0, 0, 0};
            if (sc.has_medium) key = key_of(f, __float_as_uint(load_home(in, i, f, segment)));
            const V3 inv = mk(fdiv(1.f, r.d.x), fdiv(1.f, r.d.y), fdiv(1.f, r.d.z));
            const bool soft = sc.has_mesh && soft_ray(f.ex, r.d, sc.soft_shear);              // no culling against a t that may be rounding noise
            const uint32_t n_obj = skip_ray(f.ex, r) ? 0u : sc.n_objects;      // a NaN ray's record comes from k_extend_exact
            for (uint32_t k = 0; k < n_obj; k++) {
                const float4 lo = sc.obj_leaf[2 * (size_t)k], hi = sc.obj_leaf[2 * (size_t)k + 1];   // the leaf's box in the walked tree
                float entry;
                if (!hit_aabb_entry(lo, hi, r.o, inv, relaxed(inv), box_tmin(TMIN, soft), TMAX, entry) || entry > (soft ? NO_CULL : cull_bound(have ? best_t : TMAX))) continue;
                Obj o = load_obj(sc.obj, k);
                if (PARK && obj_kind(o) == 5u) {                // EVERY mesh the reference's walk reaches (obj_gate_ok) goes on the ray's list: k_blas* walks them one
                    if (obj_gate_ok(sc, k, r.o, inv)) { deferred = true; deferred_obj |= 1u << k; }      // after the other (round 4: until then only the first was
                    continue;                                                                            // parked, the others were walked here, in place, from L2)
                }
                float t; uint32_t prim;
                if (PLAIN ? hit_object<false, 0, false, SIMPLE>(sc, o, k, r, TMIN, TMAX, nullptr, key, segment, t, prim) : hit_object(sc, o, k, r, TMIN, TMAX, blas_stack, key, segment, t, prim)) {
                    if ((!have || t < best_t || (t == best_t && sc.obj_rank[k] > sc.obj_rank[best_obj])) && obj_gate_ok(sc, k, r.o, inv)) { have = true; best_t = t; best_obj = k; best_prim = prim; }
                }
            }
            if (!deferred) qst(&hits[i], pack_hit(best_t, best_obj, best_prim, sc.prim_bits));
        }
        if (PARK) {
            const unsigned long long pmask = __ballot(deferred);
            if (pmask) {
                const uint32_t prank = __builtin_amdgcn_mbcnt_hi((uint32_t)(pmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pmask, 0u));
                if (deferred) {
                    const uint32_t e = w * park_stride + park_n + prank;
                    qst(&park_a[e], make_float4(r.o.x, r.o.y, r.o.z, r.d.x));
                    qst(&park_b[e], make_float2(r.d.y, r.d.z));
                    qst(&park_m[e], make_float4(__uint_as_float(i), __uint_as_float(PARK_MASK | deferred_obj), best_t, pack_hit(best_t, best_obj, best_prim, sc.prim_bits).y));
                }
                park_n += (uint32_t)__popcll(pmask);
            }
        }
    }
    if (PARK && lane == 0) park_count[w] = park_n;
}

// ------------------------------------------------------------------------------------------------
// K2-lin'  k_extend_linear_defer: the linear scan with its trailing Rect3d objects taken out of the per-ray loop.
//
// cornell's two rotated boxes are 12 of a ray's 18 rectangle tests, and a bounce ray can reach either only about one time in
// four — but its 64 lanes never all miss, so the wave pays for both boxes for every ray.  Here a ray runs the other objects
// in line, then a conservative per-lane slab test against each box's inflated world box, culled against the hit it already
// holds; a ray that may reach a box is appended to that box's list in LDS (wave-private; the entry carries the ray's slot and
// the hit so far), and whenever a list holds 64 entries the wave tests THAT ONE box for 64 rays at once, every lane busy, the
// object record still wave-uniform (scalar loads).  A ray listed for both boxes goes from the first list to the second.
// The deferred objects are the LAST n_def of the scene and are run in index order with t_max = the hit so far, so every ray
// sees exactly the sequence of exact tests of scene.rs:137-149 minus tests that cannot succeed: the bits do not change.
// (Round 2's first attempt kept one list with a shape mask per ray and lost, 19.2 -> 24.4 ms: per-lane object records in
// the run phase, 18 spills.  One list per object keeps the records scalar.)
// Dynamic LDS per wave: two lists of 64 entries, DEFER_FIELDS dwords each (round 4, below).
// ------------------------------------------------------------------------------------------------
// Round 4: a list entry CARRIES ITS RAY.  Until then an entry was (slot, t, code) and the run gathered the ray again by slot — 24 useful
// bytes in two sectors, from beyond L2: the kernel's PMC traffic was 2.16 x its algorithmic bytes, and a timing build that skipped the
// gather (wrong frames) rendered cornell in 33.6 instead of 37.0 ms (gpurun_out/r04m/norefetch.txt: k_extend 17.1 -> 14.8 ms and k_shade,
// which shares HBM with it when two batches overlap, 19.8 -> 18.3).  Cache tricks did not get it back (non-temporal queue stores, earlier
// runs, fewer resident waves: gpurun_out/r04n).  With the ray in the entry — 36 bytes — the old lists (128 + 192 entries, sized for a run
// of exactly 64) would take 11.5 KB per wave: three waves per SIMD for a kernel whose division chains want eight.  So a list holds 64
// entries and never overflows (make_room, below): 2 x 64 x 36 B = 4.6 KB per wave, 147 KB at eight waves per SIMD.  The price: a run
// has 65 - (one chunk's candidates) to 64 of its lanes busy instead of all of them.
constexpr uint32_t DEFER_FIELDS = 9;       // slot | later-list bit 31, t, code, origin.xyz, direction.xyz — [field][64] per list
extern __shared__ uint32_t lds_defer[];
template <bool SIMPLE>      // the objects in front of the listed boxes are spheres / rectangles / boxes only (cornell): hit_shape's SIMPLE form
__global__ __launch_bounds__(WB) __attribute__((amdgpu_waves_per_eu(FW_DEFER_WAVES, 8)))
void k_extend_linear_defer(DScene sc, DFrame f, DPaths in, float2 *__restrict__ hits, DQueue q, int segment, uint32_t n_def) {
    const uint32_t w = wave_index(), lane = threadIdx.x & 63u;
    if (w >= q.n_waves) return;
    const uint32_t n = q.wcount[(size_t)segment * q.n_waves + w], base = w * q.cap;
    const float TMIN = 0.001f, TMAX = 2e9f;                          // render.rs:19
    const uint32_t n_first = sc.n_objects - n_def;
    uint32_t *const L0 = lds_defer, *const L1 = lds_defer + DEFER_FIELDS * 64u;
    uint32_t cnt0 = 0, cnt1 = 0;
    const RngKey nokey{0, 0, 0};
    auto write_final = [&](uint32_t slot, float t, uint32_t code) {
        if (f.hit4) reinterpret_cast<uint32_t *>(hits)[slot] = code; else qst(&hits[slot], make_float2(t, __uint_as_float(code)));
    };
    auto put = [&](uint32_t *L, uint32_t e, uint32_t sb, float t, uint32_t code, const Ray &r) {
        L[e] = sb; L[64u + e] = __float_as_uint(t); L[128u + e] = code;
        L[192u + e] = __float_as_uint(r.o.x); L[256u + e] = __float_as_uint(r.o.y); L[320u + e] = __float_as_uint(r.o.z);
        L[384u + e] = __float_as_uint(r.d.x); L[448u + e] = __float_as_uint(r.d.y); L[512u + e] = __float_as_uint(r.d.z);
    };
    auto get = [&](const uint32_t *L, uint32_t e, uint32_t &sb, float &t, uint32_t &code, Ray &r) {
        sb = L[e]; t = __uint_as_float(L[64u + e]); code = L[128u + e];
        r.o = mk(__uint_as_float(L[192u + e]), __uint_as_float(L[256u + e]), __uint_as_float(L[320u + e]));
        r.d = mk(__uint_as_float(L[384u + e]), __uint_as_float(L[448u + e]), __uint_as_float(L[512u + e]));
    };
    // the exact test of deferred object d for the first `take` entries of its list, one per lane, the object record wave-uniform
    auto test = [&](uint32_t d, const Ray &r, float &t, uint32_t &code) {
        const uint32_t k = n_first + d;
        const Obj o = load_obj(sc.obj, k);                            // scalar loads
        float tt; uint32_t prim;
        if (hit_rect3d(o.q3, o.q4, to_object_space(o, r), TMIN, t, tt, prim)) { t = tt; code = (k << sc.prim_bits) | prim; }   // a plain Rect3d (the host checks)
    };
    // Lists never overflow: before a chunk appends its candidates, make_room runs whichever list could not take them (and list 1 first when
    // it could not take what list 0's run passes on — the entries flagged in bit 31, counted beforehand).  A run so has between
    // 65 - (the chunk's candidates) and 64 lanes busy.  One textual copy of each run: the two places list 1 may have to run are the two
    // trips of a loop.
    auto append = [&](uint32_t *L, uint32_t &cnt, bool want, uint32_t sb, float t, uint32_t code, const Ray &r) {
        const unsigned long long m = __ballot(want);
        if (!m) return;
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (want) put(L, cnt + rank, sb, t, code, r);
        cnt += (uint32_t)__popcll(m);
    };
    // need0 / need1: entries the caller is about to append to list 0 / list 1 (65: run whatever is there — the end of the queue)
    auto make_room = [&](uint32_t need0, uint32_t need1) {
        const bool r0 = cnt0 != 0u && cnt0 + need0 > 64u;
        const uint32_t pass_on = r0 ? (uint32_t)__popcll(__ballot(lane < cnt0 && (L0[lane] >> 31) != 0u)) : 0u;
#pragma nounroll
        for (int phase = 0; phase < 2; phase++) {
            const bool r1 = cnt1 != 0u && (phase == 0 ? cnt1 + pass_on > 64u : cnt1 + need1 > 64u);
            if (r1) {
                uint32_t sb = 0, code = MISS; float t = TMAX; Ray r{mk(0, 0, 0), mk(0, 0, 1)};
                if (lane < cnt1) { get(L1, lane, sb, t, code, r); test(1, r, t, code); write_final(sb, t, code); }
                cnt1 = 0;
            }
            if (phase == 0 && r0) {
                uint32_t sb = 0, code = MISS; float t = TMAX; Ray r{mk(0, 0, 0), mk(0, 0, 1)};
                const bool on = lane < cnt0;
                if (on) { get(L0, lane, sb, t, code, r); test(0, r, t, code); }
                cnt0 = 0;
                const bool more = on && (sb >> 31) != 0u;              // listed for the second box too
                const uint32_t slot = sb & 0x7fffffffu;
                if (on && !more) write_final(slot, t, code);
                append(L1, cnt1, more, slot, t, code, r);
            }
        }
    };
    float4 ra_n = make_float4(0, 0, 0, 0); float2 rb_n = make_float2(0, 0);
    if (lane < n) { ra_n = qld(&in.ray_a[base + lane]); rb_n = load_ray_b(in, base + lane, f, segment); }
    for (uint32_t c0 = 0; c0 < n; c0 += 64u) {
        const uint32_t j = c0 + lane, i = base + j;
        const float4 ra = ra_n; const float2 rb = rb_n;
        if (j + 64u < n) { ra_n = qld(&in.ray_a[i + 64u]); rb_n = load_ray_b(in, i + 64u, f, segment); }
        const bool active = j < n;
        float best_t = TMAX; uint32_t best_obj = MISS, best_prim = 0;
        bool may0 = false, may1 = false;
        const Ray r = make_ray(ra, rb, f, segment);
        if (active) {
            const float4 *op = sc.obj;
            // A room's walls are PLAIN rectangles (no rotation, no degenerate interval), two or three per axis: they share the reciprocal of
            // their axis' direction component — v_rcp_f32 is a quarter-rate instruction, and the generic test computed it once per rectangle —
            // and skip the generic dispatch.  Wave-uniform branch (the record came through scalar loads); the quotients are hit_rect's, bit for
            // bit.  k_extend 16.2 -> 14.25 ms, cornell 36.0 -> 33.2 ms (gpurun_out/r04y/cornell_rcp.txt).  (The same in the generic linear scans
            // costs them registers they do not have — hdri 7.3 -> 7.8 ms, volume 13.8 -> 15.1 —, and spelled out inside hit_rect3d, where the
            // compiler already shares them, it lost as well: r04y/shared_rcp.txt.)
            const Rcp rcx = make_rcp(r.d.x), rcy = make_rcp(r.d.y), rcz = make_rcp(r.d.z);
            for (uint32_t k = 0; k < n_first; k++, op += OBJ_Q) {
                const Obj o = load_obj(op, 0);
                float t; uint32_t prim = 0; bool h;
                const uint32_t kind = obj_kind(o);
                if (kind >= 1u && kind <= 3u && !(obj_flags(o) & OF_ROTATED) && o.q4.y == 0.f) {
                    const Ray ro{r.o - mk(o.q0.w, o.q1.w, o.q2.w), r.d};                                 // to_object_space without a rotation
                    if (kind == 1u) h = hit_rect_rcp<0>(o.q3.x, o.q3.y, o.q3.z, o.q3.w, o.q4.x, ro, rcz, TMIN, best_t, t);
                    else if (kind == 2u) h = hit_rect_rcp<1>(o.q3.x, o.q3.y, o.q3.z, o.q3.w, o.q4.x, ro, rcy, TMIN, best_t, t);
                    else h = hit_rect_rcp<2>(o.q3.x, o.q3.y, o.q3.z, o.q3.w, o.q4.x, ro, rcx, TMIN, best_t, t);
                } else h = hit_object<false, 0, false, SIMPLE>(sc, o, k, r, TMIN, best_t, nullptr, nokey, segment, t, prim);
                if (h) { best_t = t; best_obj = k; best_prim = prim; }
            }
            // conservative pre-tests (see closest_hit's segment-0 cull: approximate reciprocals, boxes inflated by 1e-4 of the
            // scene and ray-origin scale, NaN-dropping min/max), here per lane and also culled against the hit so far
            const V3 ainv = mk(__builtin_amdgcn_rcpf(r.d.x), __builtin_amdgcn_rcpf(r.d.y), __builtin_amdgcn_rcpf(r.d.z));
            const float eps = 1e-4f * (fabsf(r.o.x) + fabsf(r.o.y) + fabsf(r.o.z));
            for (uint32_t d = 0; d < n_def; d++) {
                const uint32_t k = n_first + d;
                const float4 lo = sc.obj_cull[2 * (size_t)k], hi = sc.obj_cull[2 * (size_t)k + 1];
                const float m = eps + 1e-4f * (fmaxf(fmaxf(fabsf(lo.x), fabsf(lo.y)), fabsf(lo.z)) + fmaxf(fmaxf(fabsf(hi.x), fabsf(hi.y)), fabsf(hi.z))) + 1e-6f;
                float t0 = (lo.x - m - r.o.x) * ainv.x, t1 = (hi.x + m - r.o.x) * ainv.x;
                float tn = fminf(t0, t1), tf = fmaxf(t0, t1);
                t0 = (lo.y - m - r.o.y) * ainv.y; t1 = (hi.y + m - r.o.y) * ainv.y;
                tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
                t0 = (lo.z - m - r.o.z) * ainv.z; t1 = (hi.z + m - r.o.z) * ainv.z;
                tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
                const bool maybe = !(tf < tn * (1.f - 1e-4f) - 1e-4f) && !(tf < 0.f) && !(tn * (1.f - 1e-4f) - 1e-4f > best_t);
                if (d == 0) may0 = maybe; else may1 = maybe;
            }
        }
        const uint32_t code = best_obj == MISS ? MISS : ((best_obj << sc.prim_bits) | best_prim);
        const bool c1 = may1 && !may0;                                  // straight to list 1; a ray listed for box 0 carries box 1 in bit 31
        make_room((uint32_t)__popcll(__ballot(may0)), (uint32_t)__popcll(__ballot(c1)));
        if (active && !may0 && !may1) write_final(i, best_t, code);
        append(L0, cnt0, may0, i | (may1 ? 0x80000000u : 0u), best_t, code, r);
        append(L1, cnt1, c1, i, best_t, code, r);
    }
    make_room(65u, 65u);
}

// (The first attempt of round 2 to DEFER the expensive shapes of the linear scan — k_extend_linear_defer above is the second: cornell's two rotated boxes are 12 of a ray's 18 rectangle
// tests and 44 % of k_extend_linear (tools/cornell_parts.py), and a bounce ray can reach each only about one time in four.
// The scan tested the cheap shapes in line, appended the rays whose inflated world-box test passed to a wave-private LDS
// list with a bit mask of shapes, and ran the listed rays 64 at a time with every lane busy; equal t went to the later
// object by an explicit merge rule, so the bits were the in-order scan's (GPU tests green).  It lost: 19.2 -> 24.4 ms.  The
// run phase reads object records per lane (vector loads and VGPR operands instead of scalar ones), the list and the second
// copy of the shape tests cost registers (18 spills at 72), and scenes without such shapes paid 8 % for the bookkeeping.)
// Two entry points because the register budget that pays differs.  The linear scan is VALU-issue-bound and its dependent
// division chains want many waves: 7 per SIMD (72 VGPRs, no spills) 19.7 vs 20.2 ms at the compiler's own 73.  The BVH walk
// waits on dependent node fetches: 5 waves (96 VGPRs, no spills) instead of 4 (104): suzanne 18.7 vs 20.6 ms, part2 14.9 vs
// 16.2 ms at equal settings; 6 waves spill and lose again.
__global__ __launch_bounds__(WB) __attribute__((amdgpu_waves_per_eu(7, 8)))
void k_extend_linear(DScene sc, DFrame f, DPaths in, float2 *__restrict__ hits, DQueue q, int segment, int tlas_levels, int stack_levels) {
    extend_body<false, false, false>(sc, f, in, hits, q, segment, tlas_levels, stack_levels, DPark{});
}
// the same for a scene without meshes (hdri, volume): the mesh walk compiled out
__global__ __launch_bounds__(WB) __attribute__((amdgpu_waves_per_eu(7, 8)))
void k_extend_linear_nomesh(DScene sc, DFrame f, DPaths in, float2 *__restrict__ hits, DQueue q, int segment, int tlas_levels, int stack_levels) {
    extend_body<false, false, false, false>(sc, f, in, hits, q, segment, tlas_levels, stack_levels, DPark{});
}
// ... and without media either (hdri)
__global__ __launch_bounds__(WB) __attribute__((amdgpu_waves_per_eu(7, 8)))
void k_extend_linear_plain(DScene sc, DFrame f, DPaths in, float2 *__restrict__ hits, DQueue q, int segment, int tlas_levels, int stack_levels) {
    extend_body<false, false, false, false, false>(sc, f, in, hits, q, segment, tlas_levels, stack_levels, DPark{});
}
// round 5: the SIMPLE set (hit_shape): spheres, rectangles, boxes, and media around spheres — hdri (no medium) and volume (one)
template <bool MEDIUM>
__global__ __launch_bounds__(WB) __attribute__((amdgpu_waves_per_eu(7, 8)))
void k_extend_linear_simple(DScene sc, DFrame f, DPaths in, float2 *__restrict__ hits, DQueue q, int segment, int tlas_levels, int stack_levels) {
    extend_body<false, false, false, false, MEDIUM, true>(sc, f, in, hits, q, segment, tlas_levels, stack_levels, DPark{});
}
#if FW_AB
__global__ __launch_bounds__(WB) __attribute__((amdgpu_waves_per_eu(5, 8)))
void k_extend_bvh(DScene sc, DFrame f, DPaths in, float2 *__restrict__ hits, DQueue q, int segment, int tlas_levels, int stack_levels) {
    extend_body<true, false, false>(sc, f, in, hits, q, segment, tlas_levels, stack_levels, DPark{});
}
#endif
// scenes without meshes: the TLAS walk with in-wave refill (part2 @16 spp: 9.6 vs 10.5 ms)
__global__ __launch_bounds__(WB) __attribute__((amdgpu_waves_per_eu(FW_TLAS_WAVES, 8)))
void k_extend_tlas(DScene sc, DFrame f, DPaths in, float2 *__restrict__ hits, DQueue q, int segment, int tlas_levels, int stack_levels) {
    extend_body<true, true, false>(sc, f, in, hits, q, segment, tlas_levels, stack_levels, DPark{});
}
// scenes with meshes: the same walk, but a ray that reaches a mesh is handed to k_blas through the wave's parked queue in HBM
__global__ __launch_bounds__(WB) __attribute__((amdgpu_waves_per_eu(FW_TLAS_WAVES, 8)))
void k_extend_tlas_park(DScene sc, DFrame f, DPaths in, float2 *__restrict__ hits, DQueue q, int segment, int tlas_levels, int stack_levels, DPark park) {
    extend_body<true, true, true>(sc, f, in, hits, q, segment, tlas_levels, stack_levels, park);
}

// ------------------------------------------------------------------------------------------------
// K4  k_blas: the mesh BLAS walks of the rays k_extend_tlas_park handed over, with in-wave refill.
//
// Wave w walks the rays of its parked queue: 64 start, and whenever BLAS_REFILL_MIN lanes have finished theirs the idle
// lanes take the next ones — from a 64-entry register read-ahead of the queue (the parked entry carries the world ray, so
// a refill costs one object-record fetch and the transform, not a chain of HBM round trips).  Same rounds, same
// arithmetic and tie rules as hit_mesh; the finished ray is merged with what the TLAS walk held and its hit record written.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WB) __attribute__((amdgpu_waves_per_eu(FW_BLAS_WAVES, 8)))
void k_blas(DScene sc, DPark park, float2 *__restrict__ hits, DQueue q) {
    const uint32_t w = wave_index(), lane = threadIdx.x & 63u;
    if (w >= q.n_waves) return;
    TS_BEGIN();
    const uint32_t n = park.pcount[w];
    if (lane == 0 && n) park.ptotal[w] += n;                         // statistics (fw_stats.parked_rays): wave-private, no atomics
    const uint32_t base = w * park.stride;
    const float TMIN = 0.001f, TMAX = 2e9f;                          // render.rs:19
    float4 ca = make_float4(0, 0, 0, 0), na = ca, cm = ca, nm = ca; float2 cb = make_float2(0, 0), nb = cb;
    auto fetch = [&](uint32_t j, float4 &a, float2 &b, float4 &m) {
        if (j < n) { a = qld(&park.ray_a[base + j]); b = qld(&park.ray_b[base + j]); m = qld(&park.meta[base + j]); }
    };
    fetch(lane, ca, cb, cm); fetch(64u + lane, na, nb, nm);
    uint32_t cur_base = 0, q_next = 0;
    bool act = false, have = false;
    uint32_t slot = 0, obj = 0, tri_base = 0, bcode = MISS, mtri = 0, cur = REF_DONE;
    float bt = TMAX, mbest = TMAX;
    V3 ro = mk(0, 0, 0), inv = ro;
    bool soft = false;
    TriRay tr{mk(0, 0, 0), 0, 1, 2, 0.f, 0.f, 0.f};
    LdsStack st{lds_stack + threadIdx.x, 0};
    uint32_t more = 0, pidx = 0;            // the meshes this ray still has to walk after the current one (park_next_mesh), and where its parked entry lies
    auto start_walk = [&](uint32_t mesh_obj, float4 ra, float2 rb) {
        obj = mesh_obj;
        Obj o = load_obj(sc.obj, obj);
        Ray r = to_object_space(o, Ray{mk(ra.x, ra.y, ra.z), mk(ra.w, rb.x, rb.y)});
        ro = r.o; inv = mk(fdiv(1.f, r.d.x), fdiv(1.f, r.d.y), fdiv(1.f, r.d.z));
        soft = soft_direction(r.d.x, r.d.y, r.d.z, sc.soft_shear);
        tr = make_triray(r);
        tri_base = o.aux1; cur = o.aux0; st.sp = 0;
        have = false; mbest = TMAX; mtri = 0; act = true;
    };
    for (;;) {
        const unsigned long long idle_mask = __ballot(!act);
        const uint32_t n_idle = (uint32_t)__popcll(idle_mask);
        if (q_next < n) {
            if (n_idle >= BLAS_REFILL_MIN) {
                const uint32_t take = min(n_idle, min(n, cur_base + 64u) - q_next);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_mask, 0u));
                const int sel = (int)((((q_next - cur_base) + rank) & 63u) << 2);
                auto bp = [&](float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(v))); };
                const float4 ra = make_float4(bp(ca.x), bp(ca.y), bp(ca.z), bp(ca.w));
                const float2 rb = make_float2(bp(cb.x), bp(cb.y));
                const float4 me = make_float4(bp(cm.x), bp(cm.y), bp(cm.z), bp(cm.w));
                if (!act && rank < take) {
                    slot = __float_as_uint(me.x); more = __float_as_uint(me.y); bt = me.z; bcode = __float_as_uint(me.w);
                    pidx = base + q_next + rank;
                    start_walk(park_next_mesh(more), ra, rb);
                }
                q_next += take;
                if (q_next == cur_base + 64u && q_next < n) { ca = na; cb = nb; cm = nm; cur_base += 64u; fetch(cur_base + 64u + lane, na, nb, nm); }
            }
        } else if (n_idle == 64u) break;

        const uint32_t n_act = (uint32_t)__popcll(__ballot(act));
        for (;;) {   // node steps; the wave stops once no more than half of its busy lanes still walk (the others test their
                     // triangle now, the walkers go on next round).  Counting only true leaf holders, or leaving early for a
                     // refill, measured no better (11.1 vs 11.0 ms).
            const bool walking = act && !(cur & REF_LEAF);
            const uint32_t n_walk = (uint32_t)__popcll(__ballot(walking));
            if (n_walk == 0u || (n_walk < n_act && n_walk * BLAS_WALK_NUM <= n_act * BLAS_WALK_DEN)) break;
            if (walking) {
                TS_TICK(4);
                cur = pair_step(sc.blas, cur, ro, inv, relaxed(inv), box_tmin(TMIN, soft), TMAX, soft ? NO_CULL : cull_bound(have ? fminf(mbest, bt) : bt), st);
            }
        }
        if (act && (cur & REF_LEAF)) {
            if (cur != REF_DONE) {
                const uint32_t item = cur & NODE_MASK;
                cur = st.sp ? st.pop() : REF_DONE;
                TS_TICK(6);
                const float4 *tp = sc.tri + 3 * (size_t)(tri_base + item);
                float4 a = tp[0], b = tp[1], c = tp[2];
                float t, b0, b1, b2;
                if (hit_triangle(mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), tr, TMIN, TMAX, t, b0, b1, b2)) {
                    if ((!have || t < mbest || (t == mbest && sc.tri_rank[tri_base + item] > sc.tri_rank[tri_base + mtri])) &&
                        tri_gate_ok(sc, tri_base + item, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), ro, inv, TMIN, TMAX)) { have = true; mbest = t; mtri = item; }
                }
            }
            if (cur == REF_DONE) {                                      // finished: merge with what the TLAS walk held
                const uint32_t bobj = bcode == MISS ? MISS : (bcode >> sc.prim_bits);
                if (have && (bobj == MISS || mbest < bt || (mbest == bt && sc.obj_rank[obj] > sc.obj_rank[bobj]))) { bt = mbest; bcode = (obj << sc.prim_bits) | mtri; }
                if (more) start_walk(park_next_mesh(more), qld(&park.ray_a[pidx]), qld(&park.ray_b[pidx]));      // the ray's next mesh (k_extend_scan's mask)
                else { qst(&hits[slot], make_float2(bt, __uint_as_float(bcode))); act = false; }
            }
        }
    }
    TS_END();
}

// ------------------------------------------------------------------------------------------------
// K4'  k_blas_lds: the same walks with the WHOLE BLAS resident in LDS.
//
// k_blas is bound by its node fetches: every step gathers 64 bytes per lane from L2 (suzanne: 92 % L2 hits, 52 % of the
// wave time in s_waitcnt, and raising the lane utilisation from 55 to 80 % by streaming queues changed nothing).  MI355X has
// 160 KB of LDS per CU: a mesh of up to ~1 500 triangles (suzanne: 967 pair nodes = 62 KB) fits next to the traversal
// stacks of sixteen waves.  One 1024-thread workgroup per CU, persistent: it copies the pair nodes into LDS once per launch
// and its sixteen waves then walk the parked queues of the workgroup's share, taking the next queue from a counter in LDS
// whenever their stream runs dry (BlockStream, dynamic mode) — so the walks read nodes with ds_read_b128 (~100 cycles)
// instead of global loads, the waves of a workgroup balance themselves, and a wave's rays refill across queue boundaries.
// Stacks are 16-bit (LdsStack16).  Same arithmetic, same tie rules: the bits do not change.  The host launches it only
// when nodes + stacks fit (launch_extend); bigger meshes keep k_blas.
// Dynamic LDS: [pair nodes: 4 float4 each][16 stacks: levels x 64 x u16][queue counter].
// Round 3, both built, measured on suzanne and removed (tools/experiments/*.diff, profiles/r03f_*, r03h_*):
//  * rays handed out in windows of 512 sorted by direction octant x origin octant (counting sort in LDS): k_blas_lds 526 -> 534 us
//    per launch, the frame 77.8 -> 80.6 ms — six bits of key do not make 64 incoherent bounce rays walk the same nodes;
//  * TWO rays per lane (every lane holds two walks, a step issues both node reads before either's arithmetic; predicated, 111
//    VGPRs): 74.5 -> 85.4 ms.  The walk does not wait for LDS LATENCY that a second instruction stream could hide: a predicated
//    slot that does not walk still costs its 58 instructions, and the wave count per SIMD was never the limit either (r02: two
//    workgroups per CU lost as well).  What the LDS walks are short of is coherence, and neither change buys any.
// ------------------------------------------------------------------------------------------------
constexpr int LDS_WAVES = 16;                      // waves per workgroup of the LDS-resident walks
constexpr uint32_t LDS_UNIT = 512;                 // rays a wave of theirs takes from the workgroup's counter at a time (BlockStream unit mode).
// Measured (profiles/r03zi_lds_units_ab.txt, one batch in flight): whole queues -> 512: suzanne 73.8 -> 71.8 ms, part2 @16 10.6 -> 10.15;
// with two batches in flight the other batch filled those gaps already (64.4 -> 64.8, 10.55 -> 10.45).  128: a grab per two
// 64-ray rounds — counter, division, an emptied read-ahead — costs more than the balance gains: suzanne 64.8 -> 70 ms — but where the
// queues themselves hold 512 rays (small frames) 128 is the better unit: random_spheres 1.97 -> 1.88 ms, part2 @16 10.45 -> 10.3.
// Hence a quarter of the queue, between 128 and 512:
__host__ __device__ inline uint32_t lds_unit(uint32_t queue_slots) { const uint32_t u = (queue_slots / 4u) & ~63u; return u < 128u ? 128u : (u > LDS_UNIT ? LDS_UNIT : u); }
template <bool LDS_TRIS>
__global__ __launch_bounds__(LDS_WAVES * 64) void k_blas_lds(DScene sc, DPark park, float2 *__restrict__ hits, DQueue q,
                                                             uint32_t n_nodes, uint32_t n_tris, uint32_t levels) {
    extern __shared__ float4 lds_nodes[];
    const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6;
    for (uint32_t k = threadIdx.x; k < n_nodes * 4u; k += LDS_WAVES * 64) lds_nodes[k] = sc.blas[k];
    float4 *lds_tris = lds_nodes + (size_t)n_nodes * 4u;              // LDS_TRIS: the triangles too (3 float4 each), behind the nodes
    if (LDS_TRIS) for (uint32_t k = threadIdx.x; k < n_tris * 3u; k += LDS_WAVES * 64) lds_tris[k] = sc.tri[k];
    uint16_t *stacks = reinterpret_cast<uint16_t *>(lds_tris + (LDS_TRIS ? (size_t)n_tris * 3u : 0u));
    uint32_t *ctr = reinterpret_cast<uint32_t *>(stacks + (size_t)LDS_WAVES * levels * 64u);
    // this workgroup's share: the queues blockIdx.x + k * gridDim.x, k = 0 .. n_k - 1 — INTERLEAVED over the workgroups:
    // neighbouring queues hold neighbouring pixels, and a run of them is several times heavier where it covers the mesh.
    // Wave i starts with k = i, further k come from the counter in LDS.
    const uint32_t n_k = (q.n_waves + gridDim.x - 1u - blockIdx.x) / gridDim.x;
    uint32_t *lds_cnt = ctr + 4;                                          // the counts of this workgroup's queues
    for (uint32_t k = threadIdx.x; k < n_k; k += LDS_WAVES * 64) lds_cnt[k] = park.pcount[blockIdx.x + k * gridDim.x];
    if (threadIdx.x == 0) *ctr = 0u;
    __syncthreads();
    BlockStream bs;
    bs.init(park.pcount, 0u, 0u, park.stride, lane, blockIdx.x, gridDim.x);
    bs.init_units(ctr, n_k, lds_unit(park.stride), lds_cnt);               // units of rays from the counter in LDS
    bs.ptotal = park.ptotal;
    const float TMIN = 0.001f, TMAX = 2e9f;                          // render.rs:19
    float4 ca = make_float4(0, 0, 0, 0), na = ca, cm = ca, nm = ca; float2 cb = make_float2(0, 0), nb = cb;
    auto fetch = [&](uint32_t b_base, uint32_t b_n, float4 &a, float2 &b, float4 &m) {
        if (lane < b_n) { a = qld(&park.ray_a[b_base + lane]); b = qld(&park.ray_b[b_base + lane]); m = qld(&park.meta[b_base + lane]); }
    };
    uint32_t c_n = 0, c_pos = 0, n_n = 0, bb = 0, c_bb = 0, n_bb = 0;         // c_bb / n_bb: first parked entry of the current / next block
    if (bs.next(bb, c_n)) { fetch(bb, c_n, ca, cb, cm); c_bb = bb; } else c_n = 0;
    if (c_n && bs.next(bb, n_n)) { fetch(bb, n_n, na, nb, nm); n_bb = bb; } else n_n = 0;
    bool act = false, have = false;
    uint32_t slot = 0, obj = 0, tri_base = 0, bcode = MISS, mtri = 0, cur = REF_DONE;
    float bt = TMAX, mbest = TMAX;
    uint32_t pend = REF_DONE;                                       // a triangle put aside while its lane walks on (REF_DONE: none)
    V3 ro = mk(0, 0, 0), inv = ro;
    bool soft = false;
    TriRay tr{mk(0, 0, 0), 0, 1, 2, 0.f, 0.f, 0.f};
    LdsStack16 st{stacks + (size_t)wib * levels * 64u + lane, 0};
    uint32_t more = 0, pidx = 0;            // the meshes this ray still has to walk after the current one (park_next_mesh), and where its parked entry lies
    auto start_walk = [&](uint32_t mesh_obj, float4 ra, float2 rb) {
        obj = mesh_obj;
        Obj o = load_obj(sc.obj, obj);
        Ray r = to_object_space(o, Ray{mk(ra.x, ra.y, ra.z), mk(ra.w, rb.x, rb.y)});
        ro = r.o; inv = mk(fdiv(1.f, r.d.x), fdiv(1.f, r.d.y), fdiv(1.f, r.d.z));
        soft = soft_direction(r.d.x, r.d.y, r.d.z, sc.soft_shear);
        tr = make_triray(r);
        tri_base = o.aux1; cur = o.aux0; st.sp = 0;
        have = false; mbest = TMAX; mtri = 0; act = true;
    };
    for (;;) {
        const unsigned long long idle_mask = __ballot(!act);
        const uint32_t n_idle = (uint32_t)__popcll(idle_mask);
        if (c_pos < c_n) {
            if (n_idle >= BLAS_REFILL_MIN) {
                const uint32_t take = min(n_idle, c_n - c_pos);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_mask, 0u));
                const int sel = (int)(((c_pos + rank) & 63u) << 2);
                auto bp = [&](float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(v))); };
                const float4 ra = make_float4(bp(ca.x), bp(ca.y), bp(ca.z), bp(ca.w));
                const float2 rb = make_float2(bp(cb.x), bp(cb.y));
                const float4 me = make_float4(bp(cm.x), bp(cm.y), bp(cm.z), bp(cm.w));
                if (!act && rank < take) {
                    slot = __float_as_uint(me.x); more = __float_as_uint(me.y); bt = me.z; bcode = __float_as_uint(me.w);
                    pidx = c_bb + c_pos + rank;
                    start_walk(park_next_mesh(more), ra, rb);
                }
                c_pos += take;
                if (c_pos == c_n) {
                    ca = na; cb = nb; cm = nm; c_n = n_n; c_pos = 0; c_bb = n_bb;
                    if (c_n && bs.next(bb, n_n)) { fetch(bb, n_n, na, nb, nm); n_bb = bb; } else n_n = 0;
                }
            }
        } else if (n_idle == 64u) break;

        const uint32_t n_act = (uint32_t)__popcll(__ballot(act));
        for (;;) {
            // A lane that holds a triangle but has more of its tree on the stack puts the triangle aside and walks on (one aside
            // at most): more lanes walk in the node loop, more lanes hold a triangle when the wave turns to the triangles, for a few
            // node visits that the untested triangle would have culled.  Same tests, same winner.  suzanne 64.7 -> 63.4 ms, @64 9.99 ->
            // 9.78 (three interleaved pairs, profiles/r03zn_pending_leaf_ab.txt; other exit rules and refill thresholds with it: worse).
            if (act && (cur & REF_LEAF) && cur != REF_DONE && pend == REF_DONE && st.sp > 0) { pend = cur; cur = st.pop(); }
            const bool walking = act && !(cur & REF_LEAF);
            const uint32_t n_walk = (uint32_t)__popcll(__ballot(walking));
            if (n_walk == 0u || (n_walk < n_act && n_walk * BLAS_WALK_NUM <= n_act * BLAS_WALK_DEN)) break;
            if (walking) cur = pair_step(lds_nodes, cur, ro, inv, relaxed(inv), box_tmin(TMIN, soft), TMAX, soft ? NO_CULL : cull_bound(have ? fminf(mbest, bt) : bt), st);
            // two steps per exit check: the ballot, the count and the exit rule cost a third of a step's issue time
            // (suzanne @64 10.1 -> 9.9 ms; the TLAS walk of part2 gains nothing from the same and keeps one)
            if (act && !(cur & REF_LEAF)) cur = pair_step(lds_nodes, cur, ro, inv, relaxed(inv), box_tmin(TMIN, soft), TMAX, soft ? NO_CULL : cull_bound(have ? fminf(mbest, bt) : bt), st);
        }
        for (int pass = 0; pass < 2; pass++) {                         // the triangles put aside, then the ones held
            uint32_t item = REF_DONE;
            if (pass == 0) { if (act && pend != REF_DONE) { item = pend & NODE_MASK; pend = REF_DONE; } }
            else if (act && (cur & REF_LEAF) && cur != REF_DONE) { item = cur & NODE_MASK; cur = st.sp ? st.pop() : REF_DONE; }
            if (__ballot(item != REF_DONE) == 0ull) continue;
            if (item != REF_DONE) {
                const float4 *tp = (LDS_TRIS ? lds_tris : sc.tri) + 3 * (size_t)(tri_base + item);
                float4 a = tp[0], b = tp[1], c = tp[2];
                float t, b0, b1, b2;
                if (hit_triangle(mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), tr, TMIN, TMAX, t, b0, b1, b2)) {
                    if ((!have || t < mbest || (t == mbest && sc.tri_rank[tri_base + item] > sc.tri_rank[tri_base + mtri])) &&
                        tri_gate_ok(sc, tri_base + item, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), ro, inv, TMIN, TMAX)) { have = true; mbest = t; mtri = item; }
                }
            }
        }
        if (act && (cur & REF_LEAF)) {
            if (cur == REF_DONE) {
                const uint32_t bobj = bcode == MISS ? MISS : (bcode >> sc.prim_bits);
                if (have && (bobj == MISS || mbest < bt || (mbest == bt && sc.obj_rank[obj] > sc.obj_rank[bobj]))) { bt = mbest; bcode = (obj << sc.prim_bits) | mtri; }
                if (more) start_walk(park_next_mesh(more), qld(&park.ray_a[pidx]), qld(&park.ray_b[pidx]));      // the ray's next mesh (k_extend_scan's mask)
                else { qst(&hits[slot], make_float2(bt, __uint_as_float(bcode))); act = false; }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K2'  k_extend_tlas_lds: the TLAS walk of scenes without meshes (k_extend_tlas) with the whole TLAS resident in LDS —
// part2's 1 408 pair nodes are 90 KB.  Same organisation as k_blas_lds: one persistent 1024-thread workgroup per CU, nodes
// copied into LDS once per launch, 16-bit stacks, queues handed out dynamically inside the workgroup and streamed.  Object
// records stay in L2 (a ray tests ~1.5 objects for ~8 node visits).  Same tests, tie rules and culling: same bits.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LDS_WAVES * 64) void k_extend_tlas_lds(DScene sc, DFrame f, DPaths in, float2 *__restrict__ hits, DQueue q,
                                                                    int segment, uint32_t n_nodes, uint32_t levels) {
    extern __shared__ float4 lds_nodes[];
    const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6;
    for (uint32_t k = threadIdx.x; k < n_nodes * 4u; k += LDS_WAVES * 64) lds_nodes[k] = sc.tlas[k];
    uint16_t *stacks = reinterpret_cast<uint16_t *>(lds_nodes + (size_t)n_nodes * 4u);
    uint32_t *ctr = reinterpret_cast<uint32_t *>(stacks + (size_t)LDS_WAVES * levels * 64u);
    const uint32_t n_k = (q.n_waves + gridDim.x - 1u - blockIdx.x) / gridDim.x;     // queues blockIdx.x + k * gridDim.x (k_blas_lds)
    uint32_t *lds_cnt = ctr + 4;                                          // the counts of this workgroup's queues
    for (uint32_t k = threadIdx.x; k < n_k; k += LDS_WAVES * 64) lds_cnt[k] = q.wcount[(size_t)segment * q.n_waves + blockIdx.x + k * gridDim.x];
    if (threadIdx.x == 0) *ctr = 0u;
    __syncthreads();
    BlockStream bs;
    bs.init(q.wcount + (size_t)segment * q.n_waves, 0u, 0u, q.cap, lane, blockIdx.x, gridDim.x);
    bs.init_units(ctr, n_k, lds_unit(q.cap), lds_cnt);
    const float TMIN = 0.001f, TMAX = 2e9f;                          // render.rs:19
    float4 ca = make_float4(0, 0, 0, 0), na = ca; float2 cb = make_float2(0, 0), nb = cb; float cs = 0.f, ns = 0.f;
    auto fetch = [&](uint32_t b_base, uint32_t b_n, float4 &a, float2 &b, float &st) {
        if (lane < b_n) { a = qld(&in.ray_a[b_base + lane]); b = load_ray_b(in, b_base + lane, f, segment); if (sc.has_medium) st = load_home(in, b_base + lane, f, segment); }
    };
    uint32_t c_n = 0, c_pos = 0, c_base = 0, n_n = 0, n_base = 0;
    if (bs.next(c_base, c_n)) fetch(c_base, c_n, ca, cb, cs); else c_n = 0;
    if (c_n && bs.next(n_base, n_n)) fetch(n_base, n_n, na, nb, ns); else n_n = 0;
    // When a block becomes current, all of its (up to 64) rays sit one per lane: the per-ray start-up work — the reciprocal
    // direction (three divisions) and the hoisted objects (part2: the fog medium around everything, two sphere tests, a draw,
    // a log10 per ray) — is done HERE with every lane busy, not at refill time by the few lanes that refill; the results ride
    // in six more registers of the block and are handed out with the ray.
    float h_t = TMAX, hix = 0.f, hiy = 0.f, hiz = 0.f; uint32_t h_obj = MISS, h_prim = 0;
    auto prep_block = [&]() {
        h_t = TMAX; h_obj = MISS; h_prim = 0;
        if (lane < c_n) {
            const Ray r = make_ray(ca, cb, f, segment);
            const V3 iv = mk(fdiv(1.f, r.d.x), fdiv(1.f, r.d.y), fdiv(1.f, r.d.z));
            hix = iv.x; hiy = iv.y; hiz = iv.z;
            if (sc.n_hoisted) {
                RngKey hkey{0, 0, 0};
                if (sc.has_medium) hkey = key_of(f, __float_as_uint(cs));
                bool hv = false;
                hoisted_hits(sc, r, iv, hkey, segment, nullptr, hv, h_t, h_obj, h_prim);
            }
        }
    };
    prep_block();
    const uint32_t IDLE = 0xffffffffu;
    uint32_t slot = IDLE, cur = REF_DONE, path_id = 0, best_obj = MISS, best_prim = 0;
    float best_t = TMAX; bool have = false;
    V3 wo = mk(0, 0, 0), wd = wo, inv = wo;
    LdsStack16 st{stacks + (size_t)wib * levels * 64u + lane, 0};
    for (;;) {
        const unsigned long long idle_mask = __ballot(slot == IDLE);
        const uint32_t n_idle = (uint32_t)__popcll(idle_mask);
        if (c_pos < c_n) {
            if (n_idle >= TLAS_REFILL_MIN) {
                const uint32_t take = min(n_idle, c_n - c_pos);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_mask, 0u));
                const uint32_t src = c_pos + rank;
                const int sel = (int)((src & 63u) << 2);
                auto bp = [&](float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(v))); };
                const float4 ra = make_float4(bp(ca.x), bp(ca.y), bp(ca.z), bp(ca.w));
                const float2 rb = make_float2(bp(cb.x), bp(cb.y));
                const float rs = sc.has_medium ? bp(cs) : 0.f;
                const V3 rinv = mk(bp(hix), bp(hiy), bp(hiz));
                const float rt = bp(h_t); const uint32_t robj = __float_as_uint(bp(__uint_as_float(h_obj))), rprim = __float_as_uint(bp(__uint_as_float(h_prim)));
                if (slot == IDLE && rank < take) {
                    slot = c_base + src;
                    const Ray r = make_ray(ra, rb, f, segment);
                    wo = r.o; wd = r.d;
                    inv = rinv;
                    path_id = __float_as_uint(rs);
                    cur = skip_ray(f.ex, r) ? REF_DONE : sc.tlas_root; st.sp = 0;          // a NaN ray's record comes from k_extend_exact
                    have = robj != MISS; best_t = rt; best_obj = robj; best_prim = rprim;     // what the hoisted objects gave (prep_block)
                }
                c_pos += take;
                if (c_pos == c_n) {
                    ca = na; cb = nb; cs = ns; c_n = n_n; c_base = n_base; c_pos = 0;
                    if (c_n && bs.next(n_base, n_n)) fetch(n_base, n_n, na, nb, ns); else n_n = 0;
                    prep_block();
                }
            }
        } else if (n_idle == 64u) break;

        const bool busy = slot != IDLE && cur != REF_DONE;
        const uint32_t n_busy = (uint32_t)__popcll(__ballot(busy));
        for (;;) {
            // (An object put aside like k_blas_lds's triangle — FW_PEND_OBJ in round 3 — lost: part2 @64 33.8 -> 37.6 ms, random_spheres
            // 1.92 -> 2.05: an object found early culls much of the walk that its lane would do meanwhile.)
            const bool walking = busy && !(cur & REF_LEAF);
            const uint32_t n_walk = (uint32_t)__popcll(__ballot(walking));
            if (n_walk == 0u || (n_walk < n_busy && n_walk * WALK_NUM <= n_busy * WALK_DEN)) break;
            if (walking) cur = pair_step(lds_nodes, cur, wo, inv, relaxed(inv), box_tmin(TMIN, false), TMAX, cull_bound(have ? best_t : TMAX), st);
        }
        if (busy && (cur & REF_LEAF) && cur != REF_DONE) {
            const uint32_t item = cur & NODE_MASK;
            cur = st.sp ? st.pop() : REF_DONE;
            Obj o = load_obj_for_hit(sc.obj, item);
            RngKey key{0, 0, 0};
            if (sc.has_medium) key = key_of(f, path_id);
            float t; uint32_t prim;
            if (hit_object(sc, o, item, Ray{wo, wd}, TMIN, TMAX, nullptr, key, segment, t, prim)) {   // no meshes in these scenes: no BLAS stack
                if ((!have || t < best_t || (t == best_t && sc.obj_rank[item] > sc.obj_rank[best_obj])) && obj_gate_ok(sc, item, wo, inv)) { have = true; best_t = t; best_obj = item; best_prim = prim; }
            }
        }
        const bool done = slot != IDLE && cur == REF_DONE;
        if (done) { qst(&hits[slot], pack_hit(best_t, best_obj, best_prim, sc.prim_bits)); slot = IDLE; }
    }
}

// ------------------------------------------------------------------------------------------------
// WIDE walks (round 4): k_blas_wide / k_extend_tlas_wide = k_blas_lds / k_extend_tlas_lds over WIDE nodes (fw_device.h): four
// children per node.  Why: the pair walks issue vector instructions 70 % of the time with half their lanes (profiles/r03z_*_sq.json),
// and a third of what a wave issues per step is not box arithmetic — the stack operation, the ballots and exit rule of the round,
// the scalar bookkeeping of the loop.  A wide step decides four boxes for one of each, and the tree has a third of the pair tree's
// nodes: suzanne's BLAS 62 KB -> 36 KB as f32 nodes, part2's TLAS 90 -> 53 KB; teapot.yml's four meshes (404 KB of pair nodes: the
// L2 walk until now) are 101 KB as quantised nodes and walk out of LDS for the first time.
//   WIDE_F32: the children's own boxes, the slab test of aabb.rs:30-50 on them: an item is reached iff its own box passes and its
//             entry is not beyond the culling bound — the pair walk's decisions, bit for bit.
//   WIDE_Q8 : boxes rounded outward to 8 bits (a superset under the same monotone arithmetic): they only cull; an item reached
//             through one is tested like any other, ties go by reference rank, rays whose result depends on HOW the trees are
//             walked are on the exact list either way.
// The near / far planes are chosen by the ray's signs BEFORE the arithmetic (one offset per axis kept per ray), so a box costs
// 6 subtractions, 6 multiplications, max3 / min3 and two comparisons.  The children that pass are ordered by a 5-comparator
// network on 32-bit keys (upper half of the entry distance | reference): nearest next, the others pushed farthest first.
// ------------------------------------------------------------------------------------------------
// The device's ERROR WORD (A/B build, -DFW_AB=1: the build experiments are made in).  A walk kernel that would write past its LDS stack, or
// whose wave stops making progress, sets a bit here instead, drops the push / leaves its loop, and the render returns FW_ERR_HIP with the bit
// named in fw_last_error() — a fault or a hang inside a kernel cannot be turned into a status code after the fact: the runtime aborts the
// process (round 4's r04t: DESIGN.md §6).  The product build carries no check: its stack sizes are derived from the tree's depth on the
// host (3 * depth + 2 levels: at most three pushes per wide node on the path) and its loops retire a ray or consume one every round.
#if FW_AB
__device__ uint32_t g_err_word;
enum : uint32_t { ERR_STACK_OVERFLOW = 1u, ERR_NO_PROGRESS = 2u };
constexpr uint32_t WATCHDOG_ROUNDS = 1u << 24;     // rounds of one wave's walk loop; a wave handles a few thousand rays in a few rounds each
#define FW_WATCHDOG_DECL uint32_t watchdog_ = 0
#define FW_WATCHDOG_TICK if (++watchdog_ > WATCHDOG_ROUNDS) { if ((threadIdx.x & 63u) == 0) atomicOr(&g_err_word, ERR_NO_PROGRESS); break; }
#else
#define FW_WATCHDOG_DECL do { } while (0)
#define FW_WATCHDOG_TICK do { } while (0)
#endif
struct LdsStackW {   // 16-bit references, [level][lane] over the 64 lanes of one wave
    uint16_t *s; int sp;
#if FW_AB
    int cap;         // levels this stack owns: a push beyond them is dropped and flagged (the walk then misses a subtree: the frame is wrong, the call fails)
    __device__ __forceinline__ void push(uint32_t v) { if (sp >= cap) { atomicOr(&g_err_word, ERR_STACK_OVERFLOW); return; } s[sp * 64] = (uint16_t)v; sp++; }
#else
    __device__ __forceinline__ void push(uint32_t v) { s[sp * 64] = (uint16_t)v; sp++; }
#endif
    __device__ __forceinline__ uint32_t pop() { sp--; return s[sp * 64]; }
};
// per ray: WIDE_F32 byte offsets (inside a node) of the near planes of x, y, z (the far planes lie at 48 - q[0], 80 - q[1], 112 - q[2]); WIDE_Q8: q[0..2] = inv < 0
struct WideSel { uint32_t q[3]; };
template <int FMT> __device__ __forceinline__ WideSel wide_sel(V3 inv) {
    WideSel s;
    const bool nx = inv.x < 0.f, ny = inv.y < 0.f, nz = inv.z < 0.f;       // the selection of aabb.rs:36-38 (`if inv_d < 0 { swap }`)
    if (FMT == WIDE_F32) { s.q[0] = nx ? 48u : 0u; s.q[1] = ny ? 64u : 16u; s.q[2] = nz ? 80u : 32u; }
    else { s.q[0] = nx ? 1u : 0u; s.q[1] = ny ? 1u : 0u; s.q[2] = nz ? 1u : 0u; }
    return s;
}
#ifndef FW_WIDE_FMA
#define FW_WIDE_FMA 1
#endif
// FW_WIDE_FMA (round 5): a plane's distance as ONE fma — fma(plane, 1/d, -o/d) instead of (plane - o) * (1/d), and for quantised nodes
// fma(q, 2^e/d, (origin - o)/d) instead of decoding the plane first: 33 instead of 51 vector instructions for the four boxes of an f32
// step, 66 instead of 99 for a quantised one, in kernels that issue vector instructions 0.9 of the time.  The distances differ from the
// subtract-first form by ~2^-24 |o/d| (absolute) — the ulp of the origin's coordinate, orders of magnitude below what the walked boxes are
// relaxed by (hit_aabb_entry: exit planes x (1 + 2^-12), boxes grown by >= 2^-14 of their extent on the host) — and the walked boxes decide
// nothing: which item wins is decided by the exact item tests, the reference-rank rule and the reference's own box test on its leaf node's
// box (tri_gate_ok / obj_gate_ok).  Where a direction component is 0 the fma form meets inf - inf = NaN, which fmaxf / fminf drop: that axis
// then constrains nothing (conservative: a few more visits for axis-parallel rays).
template <int FMT>
__device__ __forceinline__ uint32_t wide_step(const uint32_t *__restrict__ nodes, uint32_t node, V3 o, V3 inv, V3 inv_hi, const WideSel &sel,
                                              float tmin, float tmax, float cull, LdsStackW &st) {
    float4 NX, NY, NZ, FX, FY, FZ; uint32_t r01, r23;
#if FW_WIDE_FMA
    V3 mn = inv, mf = inv_hi;                                            // multipliers and addends of the near / far planes' fma
    V3 an = mk(-o.x * inv.x, -o.y * inv.y, -o.z * inv.z), af = mk(-o.x * inv_hi.x, -o.y * inv_hi.y, -o.z * inv_hi.z);
#endif
    if (FMT == WIDE_F32) {
        const char *nd = reinterpret_cast<const char *>(nodes) + __umul24(node, WIDE_F32_DW * 4u);     // node < 2^15: the 24-bit multiply is full rate
        NX = *reinterpret_cast<const float4 *>(nd + sel.q[0]); NY = *reinterpret_cast<const float4 *>(nd + sel.q[1]); NZ = *reinterpret_cast<const float4 *>(nd + sel.q[2]);
        FX = *reinterpret_cast<const float4 *>(nd + (48u - sel.q[0])); FY = *reinterpret_cast<const float4 *>(nd + (80u - sel.q[1])); FZ = *reinterpret_cast<const float4 *>(nd + (112u - sel.q[2]));
        const uint2 rr = *reinterpret_cast<const uint2 *>(nd + 96);
        r01 = rr.x; r23 = rr.y;
    } else {
        const uint4 *nd = reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(nodes) + __umul24(node, WIDE_Q8_DW * 4u));
        const uint4 a = nd[0], lo = nd[1], hi = nd[2];
        const float ox = __uint_as_float(a.x), oy = __uint_as_float(a.y), oz = __uint_as_float(a.z);
        const float sx = __uint_as_float((a.w & 0xffu) << 23), sy = __uint_as_float(((a.w >> 8) & 0xffu) << 23), sz = __uint_as_float(((a.w >> 16) & 0xffu) << 23);
        const uint32_t nxw = sel.q[0] ? hi.x : lo.x, fxw = sel.q[0] ? lo.x : hi.x, nyw = sel.q[1] ? hi.y : lo.y, fyw = sel.q[1] ? lo.y : hi.y,
                       nzw = sel.q[2] ? hi.z : lo.z, fzw = sel.q[2] ? lo.z : hi.z;
        // plane = fma(q, 2^e, origin): q * 2^e is exact, so this is the one rounding the host checked its outward rounding with
#if FW_WIDE_FMA
        // the quanta themselves stand in for the planes: distance = fma(q, 2^e / d, (origin - o) / d)
        auto qf = [](uint32_t w) { return make_float4((float)(w & 0xffu), (float)((w >> 8) & 0xffu), (float)((w >> 16) & 0xffu), (float)(w >> 24)); };
        NX = qf(nxw); NY = qf(nyw); NZ = qf(nzw); FX = qf(fxw); FY = qf(fyw); FZ = qf(fzw);
        const V3 rel = mk(ox - o.x, oy - o.y, oz - o.z);
        an = mk(rel.x * inv.x, rel.y * inv.y, rel.z * inv.z); af = mk(rel.x * inv_hi.x, rel.y * inv_hi.y, rel.z * inv_hi.z);
        mn = mk(sx * inv.x, sy * inv.y, sz * inv.z); mf = mk(sx * inv_hi.x, sy * inv_hi.y, sz * inv_hi.z);
#else
        auto dq = [](uint32_t w, float s, float org) {
            return make_float4(fmaf((float)(w & 0xffu), s, org), fmaf((float)((w >> 8) & 0xffu), s, org), fmaf((float)((w >> 16) & 0xffu), s, org), fmaf((float)(w >> 24), s, org));
        };
        NX = dq(nxw, sx, ox); NY = dq(nyw, sy, oy); NZ = dq(nzw, sz, oz); FX = dq(fxw, sx, ox); FY = dq(fyw, sy, oy); FZ = dq(fzw, sz, oz);
#endif
        r01 = lo.w; r23 = hi.w;
    }
    const uint32_t NONE = 0xffffffffu;
    // hit_aabb_entry on one child with the planes already chosen; entry clamped to tmin.  key = upper half of the entry | reference; a child
    // that fails gets NONE, which sorts last and whose low half reads W_DONE.  The bits of a POSITIVE float order like unsigned integers, and
    // tmin > 0 for every ray but the SOFT class (box_tmin: -1/16, boxes entered from behind the origin), whose negative entries sort behind
    // the positive ones and among themselves in reverse.  That is harmless because — and only as long as — a SOFT ray walks with NO_CULL: it
    // visits every child that passes whatever the order, and the order of visits decides no hit (ties go by reference rank).  A change that
    // lets SOFT rays cull must build the key from fmaxf(tn, 0) (one more instruction per child) or flip the sign bit.
    auto key = [&](float nx, float ny, float nz, float fx, float fy, float fz, uint32_t ref16) -> uint32_t {
#if FW_WIDE_FMA
        const float tn = fmaxf(fmaxf(fmaxf(__builtin_fmaf(nx, mn.x, an.x), __builtin_fmaf(ny, mn.y, an.y)), __builtin_fmaf(nz, mn.z, an.z)), tmin);
        const float tf = fminf(fminf(fminf(__builtin_fmaf(fx, mf.x, af.x), __builtin_fmaf(fy, mf.y, af.y)), __builtin_fmaf(fz, mf.z, af.z)), tmax);   // relaxed exit (hit_aabb_entry)
#else
        const float tn = fmaxf(fmaxf(fmaxf((nx - o.x) * inv.x, (ny - o.y) * inv.y), (nz - o.z) * inv.z), tmin);
        const float tf = fminf(fminf(fminf((fx - o.x) * inv_hi.x, (fy - o.y) * inv_hi.y), (fz - o.z) * inv_hi.z), tmax);   // relaxed exit (hit_aabb_entry)
#endif
        const bool hit = tf > tn && !(tn > cull);
        return hit ? ((__float_as_uint(tn) & 0xffff0000u) | ref16) : NONE;
    };
    const uint32_t k0 = key(NX.x, NY.x, NZ.x, FX.x, FY.x, FZ.x, r01 & 0xffffu), k1 = key(NX.y, NY.y, NZ.y, FX.y, FY.y, FZ.y, r01 >> 16),
                   k2 = key(NX.z, NY.z, NZ.z, FX.z, FY.z, FZ.z, r23 & 0xffffu), k3 = key(NX.w, NY.w, NZ.w, FX.w, FY.w, FZ.w, r23 >> 16);
    const uint32_t a = min(k0, k1), b = max(k0, k1), c = min(k2, k3), d = max(k2, k3);
    const uint32_t s0 = min(a, c), m1 = max(a, c), m2 = min(b, d), s3 = max(b, d), s1 = min(m1, m2), s2 = max(m1, m2);
    // nearest next, the others pushed farthest first (sorted: a valid s3 implies valid s2 and s1)
    if (s1 != NONE) { if (s2 != NONE) { if (s3 != NONE) st.push(s3); st.push(s2); } st.push(s1); }
    uint32_t next = s0 & 0xffffu;
    if (s0 == NONE && st.sp) next = st.pop();
    return next;
}
// workgroup-wide copy of `dwords` (a multiple of 4) dwords into LDS
__device__ __forceinline__ void stage_lds(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, uint32_t dwords) {
    for (uint32_t k = threadIdx.x; k < dwords / 4u; k += blockDim.x) reinterpret_cast<uint4 *>(dst)[k] = reinterpret_cast<const uint4 *>(src)[k];
}
constexpr int WIDE_MAX_WAVES = 16;
#ifndef FW_WIDE_TWO_STEPS
#define FW_WIDE_TWO_STEPS 1     // a second step per exit check: suzanne 69.4 -> 67.1 ms, teapot @256 100 -> 98.3 (gpurun_out/r04m)
#endif
#ifndef FW_WIDE_WAVES
#define FW_WIDE_WAVES 4     // register budget of k_blas_wide (waves per SIMD it must leave room for): its own 16 waves per CU are 4 per SIMD; what is
#endif                      // left of the register file is where the other batch's kernels run
// Dynamic LDS: [wide nodes][triangles (LDS_TRIS)][stacks: waves x levels x 64 u16][counter, 4 dwords][this workgroup's queue counts]
template <int FMT, bool LDS_TRIS>
__global__ __launch_bounds__(WIDE_MAX_WAVES * 64) __attribute__((amdgpu_waves_per_eu(FW_WIDE_WAVES, 8))) void k_blas_wide(DScene sc, DPark park, float2 *__restrict__ hits, DQueue q,
                                                                   uint32_t n_nodes, uint32_t n_tris, uint32_t levels) {
    extern __shared__ uint32_t lds_w[];
    const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6, n_waves_wg = blockDim.x >> 6;
    const uint32_t node_dw = n_nodes * (FMT == WIDE_F32 ? WIDE_F32_DW : WIDE_Q8_DW);
    stage_lds(lds_w, sc.wblas, node_dw);
    float4 *lds_tris = reinterpret_cast<float4 *>(lds_w + node_dw);
    if (LDS_TRIS) stage_lds(reinterpret_cast<uint32_t *>(lds_tris), reinterpret_cast<const uint32_t *>(sc.tri), n_tris * 12u);
    uint16_t *stacks = reinterpret_cast<uint16_t *>(lds_tris + (LDS_TRIS ? (size_t)n_tris * 3u : 0u));
    uint32_t *ctr = reinterpret_cast<uint32_t *>(stacks + (size_t)n_waves_wg * levels * 64u);
    const uint32_t n_k = (q.n_waves + gridDim.x - 1u - blockIdx.x) / gridDim.x;      // this workgroup's queues: blockIdx.x + k * gridDim.x (k_blas_lds)
    uint32_t *lds_cnt = ctr + 4;
    for (uint32_t k = threadIdx.x; k < n_k; k += blockDim.x) lds_cnt[k] = park.pcount[blockIdx.x + k * gridDim.x];
    if (threadIdx.x == 0) *ctr = 0u;
    __syncthreads();
    BlockStream bs;
    bs.init(park.pcount, 0u, 0u, park.stride, lane, blockIdx.x, gridDim.x);
    bs.init_units(ctr, n_k, lds_unit(park.stride), lds_cnt);
    bs.ptotal = park.ptotal;
    const float TMIN = 0.001f, TMAX = 2e9f;                          // render.rs:19
    float4 ca = make_float4(0, 0, 0, 0), na = ca, cm = ca, nm = ca; float2 cb = make_float2(0, 0), nb = cb;
    auto fetch = [&](uint32_t b_base, uint32_t b_n, float4 &a, float2 &b, float4 &m) {
        if (lane < b_n) { a = qld(&park.ray_a[b_base + lane]); b = qld(&park.ray_b[b_base + lane]); m = qld(&park.meta[b_base + lane]); }
    };
    uint32_t c_n = 0, c_pos = 0, n_n = 0, bb = 0, c_bb = 0, n_bb = 0;         // c_bb / n_bb: first parked entry of the current / next block
    if (bs.next(bb, c_n)) { fetch(bb, c_n, ca, cb, cm); c_bb = bb; } else c_n = 0;
    if (c_n && bs.next(bb, n_n)) { fetch(bb, n_n, na, nb, nm); n_bb = bb; } else n_n = 0;
    bool act = false, have = false;
    uint32_t slot = 0, obj = 0, tri_base = 0, bcode = MISS, mtri = 0, cur = W_DONE, pend = W_DONE;
    float bt = TMAX, mbest = TMAX;
    V3 ro = mk(0, 0, 0), inv = ro;
    bool soft = false;
    WideSel sel = wide_sel<FMT>(inv);
    TriRay tr{mk(0, 0, 0), 0, 1, 2, 0.f, 0.f, 0.f};
#if FW_AB
    LdsStackW st{stacks + (size_t)wib * levels * 64u + lane, 0, (int)levels};
#else
    LdsStackW st{stacks + (size_t)wib * levels * 64u + lane, 0};
#endif
    FW_WATCHDOG_DECL;
    PH_DECL;
    uint32_t more = 0, pidx = 0;            // the meshes this ray still has to walk after the current one (park_next_mesh), and where its parked entry lies
    auto start_walk = [&](uint32_t mesh_obj, float4 ra, float2 rb) {
        obj = mesh_obj;
        Obj o = load_obj(sc.obj, obj);
        Ray r = to_object_space(o, Ray{mk(ra.x, ra.y, ra.z), mk(ra.w, rb.x, rb.y)});
        ro = r.o; inv = mk(fdiv(1.f, r.d.x), fdiv(1.f, r.d.y), fdiv(1.f, r.d.z));
        sel = wide_sel<FMT>(inv);
        soft = soft_direction(r.d.x, r.d.y, r.d.z, sc.soft_shear);
        tr = make_triray(r);
        tri_base = o.aux1; cur = sc.obj_wroot[obj]; st.sp = 0;
        have = false; mbest = TMAX; mtri = 0; act = true;
    };
    for (;;) {
        FW_WATCHDOG_TICK;
        const unsigned long long idle_mask = __ballot(!act);
        const uint32_t n_idle = (uint32_t)__popcll(idle_mask);
        if (c_pos < c_n) {
            if (n_idle >= BLAS_REFILL_MIN) {
                const uint32_t take = min(n_idle, c_n - c_pos);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_mask, 0u));
                const int bsel = (int)(((c_pos + rank) & 63u) << 2);
                auto bp = [&](float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(bsel, __float_as_int(v))); };
                const float4 ra = make_float4(bp(ca.x), bp(ca.y), bp(ca.z), bp(ca.w));
                const float2 rb = make_float2(bp(cb.x), bp(cb.y));
                const float4 me = make_float4(bp(cm.x), bp(cm.y), bp(cm.z), bp(cm.w));
                if (!act && rank < take) {
                    PH_T0;
                    slot = __float_as_uint(me.x); more = __float_as_uint(me.y); bt = me.z; bcode = __float_as_uint(me.w);
                    pidx = c_bb + c_pos + rank;
                    start_walk(park_next_mesh(more), ra, rb);
                    PH_ADD(0);
                }
                c_pos += take;
                if (c_pos == c_n) {
                    ca = na; cb = nb; cm = nm; c_n = n_n; c_pos = 0; c_bb = n_bb;
                    if (c_n && bs.next(bb, n_n)) { fetch(bb, n_n, na, nb, nm); n_bb = bb; } else n_n = 0;
                }
            }
        } else if (n_idle == 64u) break;

        const uint32_t n_act = (uint32_t)__popcll(__ballot(act));
        if (act) PH_COUNT(7);                                          // busy lanes per round
        for (;;) {
            // a triangle put aside while its lane walks on (k_blas_lds: one at most; same tests, same winner)
            if (act && (cur & W_LEAF) && cur != W_DONE && pend == W_DONE && st.sp > 0) { pend = cur; cur = st.pop(); }
            const bool walking = act && !(cur & W_LEAF);
            const uint32_t n_walk = (uint32_t)__popcll(__ballot(walking));
            if (n_walk == 0u || (n_walk < n_act && n_walk * BLAS_WALK_NUM <= n_act * BLAS_WALK_DEN)) break;
            if (walking) { PH_T0; cur = wide_step<FMT>(lds_w, cur, ro, inv, relaxed(inv), sel, box_tmin(TMIN, soft), TMAX, soft ? NO_CULL : cull_bound(have ? fminf(mbest, bt) : bt), st); PH_ADD(1); }
#if FW_WIDE_TWO_STEPS       // a second step per exit check (k_blas_lds gained 2 % from it on pair nodes)
            if (act && !(cur & W_LEAF)) { PH_T0; cur = wide_step<FMT>(lds_w, cur, ro, inv, relaxed(inv), sel, box_tmin(TMIN, soft), TMAX, soft ? NO_CULL : cull_bound(have ? fminf(mbest, bt) : bt), st); PH_ADD(2); }
#endif
        }
        // The round's triangle tests: the one put aside, then the one held.  A hit that would become the ray's best is a CANDIDATE; the
        // reference's gating rule (tri_gate_ok) is applied once per round, to the nearer of the two, and to the other only if that one
        // fails it (it costs as much as a third of a triangle test, and inside the loop it ran twice per round for the few lanes with a hit).
        float ct0 = 0.f, ct1 = 0.f; uint32_t ci0 = W_DONE, ci1 = W_DONE;
        for (int pass = 0; pass < 2; pass++) {
            uint32_t item = W_DONE;
            if (pass == 0) { if (act && pend != W_DONE) { item = pend & 0x7fffu; pend = W_DONE; } }
            else if (act && (cur & W_LEAF) && cur != W_DONE) { item = cur & 0x7fffu; cur = st.sp ? st.pop() : W_DONE; }
            if (__ballot(item != W_DONE) == 0ull) continue;
            if (item != W_DONE) {
                PH_T0;
                const float4 *tp = (LDS_TRIS ? lds_tris : sc.tri) + 3 * (size_t)(tri_base + item);
                float4 a = tp[0], b = tp[1], c = tp[2];
                float t, b0, b1, b2;
                if (hit_triangle(mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), tr, TMIN, TMAX, t, b0, b1, b2)) {
                    if (!have || t < mbest || (t == mbest && sc.tri_rank[tri_base + item] > sc.tri_rank[tri_base + mtri])) {
                        if (pass == 0) { ct0 = t; ci0 = item; } else { ct1 = t; ci1 = item; }
                    }
                }
                if (pass == 0) PH_ADD(3); else PH_ADD(4);
            }
        }
        if (ci1 != W_DONE && (ci0 == W_DONE || ct1 < ct0 || (ct1 == ct0 && sc.tri_rank[tri_base + ci1] > sc.tri_rank[tri_base + ci0]))) {
            const float tt = ct0; ct0 = ct1; ct1 = tt; const uint32_t ti = ci0; ci0 = ci1; ci1 = ti;      // the nearer (on a tie: the later in the reference's order) first
        }
        for (int k = 0; k < 2; k++) {
            const uint32_t ci = k ? ci1 : ci0; const float ct = k ? ct1 : ct0;
            const bool want = ci != W_DONE && (!have || ct < mbest || (ct == mbest && sc.tri_rank[tri_base + ci] > sc.tri_rank[tri_base + mtri]));
            if (__ballot(want) == 0ull) continue;
            if (want) {
                PH_T0;
                const float4 *tp = (LDS_TRIS ? lds_tris : sc.tri) + 3 * (size_t)(tri_base + ci);
                const float4 a = tp[0], b = tp[1], c = tp[2];
                if (tri_gate_ok(sc, tri_base + ci, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), ro, inv, TMIN, TMAX)) { have = true; mbest = ct; mtri = ci; }
                PH_ADD(5);
            }
        }
        if (act && cur == W_DONE) {
            PH_T0;
            const uint32_t bobj = bcode == MISS ? MISS : (bcode >> sc.prim_bits);
            if (have && (bobj == MISS || mbest < bt || (mbest == bt && sc.obj_rank[obj] > sc.obj_rank[bobj]))) { bt = mbest; bcode = (obj << sc.prim_bits) | mtri; }
            if (more) start_walk(park_next_mesh(more), qld(&park.ray_a[pidx]), qld(&park.ray_b[pidx]));      // the ray's next mesh (k_extend_scan's mask)
            else { qst(&hits[slot], make_float2(bt, __uint_as_float(bcode))); act = false; }
            PH_ADD(6);
        }
    }
    PH_FLUSH(0);
}

// k_extend_tlas_lds over WIDE_F32 nodes (scenes without meshes: part2's TLAS, random_spheres)
template <bool MEDIUM, bool SIMPLE = false>      // MEDIUM false: the scene holds no ConstantMedium (random_spheres): its code — two boundary tests and a log10 — compiled out;
                                                  // SIMPLE: spheres, rectangles, boxes and media around spheres only (part2): hit_shape's SIMPLE form
__global__ __launch_bounds__(WIDE_MAX_WAVES * 64) void k_extend_tlas_wide(DScene sc, DFrame f, DPaths in, float2 *__restrict__ hits, DQueue q,
                                                                          int segment, uint32_t n_nodes, uint32_t levels) {
    extern __shared__ uint32_t lds_w[];
    const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6, n_waves_wg = blockDim.x >> 6;
    const uint32_t node_dw = n_nodes * WIDE_F32_DW;
    stage_lds(lds_w, sc.wtlas, node_dw);
    uint16_t *stacks = reinterpret_cast<uint16_t *>(lds_w + node_dw);
    uint32_t *ctr = reinterpret_cast<uint32_t *>(stacks + (size_t)n_waves_wg * levels * 64u);
    const uint32_t n_k = (q.n_waves + gridDim.x - 1u - blockIdx.x) / gridDim.x;
    uint32_t *lds_cnt = ctr + 4;
    for (uint32_t k = threadIdx.x; k < n_k; k += blockDim.x) lds_cnt[k] = q.wcount[(size_t)segment * q.n_waves + blockIdx.x + k * gridDim.x];
    if (threadIdx.x == 0) *ctr = 0u;
    __syncthreads();
    BlockStream bs;
    bs.init(q.wcount + (size_t)segment * q.n_waves, 0u, 0u, q.cap, lane, blockIdx.x, gridDim.x);
    bs.init_units(ctr, n_k, lds_unit(q.cap), lds_cnt);
    const float TMIN = 0.001f, TMAX = 2e9f;                          // render.rs:19
    float4 ca = make_float4(0, 0, 0, 0), na = ca; float2 cb = make_float2(0, 0), nb = cb; float cs = 0.f, ns = 0.f;
    auto fetch = [&](uint32_t b_base, uint32_t b_n, float4 &a, float2 &b, float &st_) {
        if (lane < b_n) { a = qld(&in.ray_a[b_base + lane]); b = load_ray_b(in, b_base + lane, f, segment); if (sc.has_medium) st_ = load_home(in, b_base + lane, f, segment); }
    };
    uint32_t c_n = 0, c_pos = 0, c_base = 0, n_n = 0, n_base = 0;
    if (bs.next(c_base, c_n)) fetch(c_base, c_n, ca, cb, cs); else c_n = 0;
    if (c_n && bs.next(n_base, n_n)) fetch(n_base, n_n, na, nb, ns); else n_n = 0;
    // per-block start-up with every lane busy (k_extend_tlas_lds: prep_block): reciprocal direction + the hoisted objects
    float h_t = TMAX, hix = 0.f, hiy = 0.f, hiz = 0.f; uint32_t h_obj = MISS, h_prim = 0;
    auto prep_block = [&]() {
        h_t = TMAX; h_obj = MISS; h_prim = 0;
        if (lane < c_n) {
            const Ray r = make_ray(ca, cb, f, segment);
            const V3 iv = mk(fdiv(1.f, r.d.x), fdiv(1.f, r.d.y), fdiv(1.f, r.d.z));
            hix = iv.x; hiy = iv.y; hiz = iv.z;
            if (sc.n_hoisted) {
                RngKey hkey{0, 0, 0};
                if (sc.has_medium) hkey = key_of(f, __float_as_uint(cs));
                bool hv = false;
                hoisted_hits<SIMPLE>(sc, r, iv, hkey, segment, nullptr, hv, h_t, h_obj, h_prim);
            }
        }
    };
    prep_block();
    const uint32_t IDLE = 0xffffffffu;
    uint32_t slot = IDLE, cur = W_DONE, path_id = 0, best_obj = MISS, best_prim = 0;
    float best_t = TMAX; bool have = false;
    V3 wo = mk(0, 0, 0), wd = wo, inv = wo;
    WideSel sel = wide_sel<WIDE_F32>(inv);
#if FW_AB
    LdsStackW st{stacks + (size_t)wib * levels * 64u + lane, 0, (int)levels};
#else
    LdsStackW st{stacks + (size_t)wib * levels * 64u + lane, 0};
#endif
    FW_WATCHDOG_DECL;
    PH_DECL;
    for (;;) {
        FW_WATCHDOG_TICK;
        const unsigned long long idle_mask = __ballot(slot == IDLE);
        const uint32_t n_idle = (uint32_t)__popcll(idle_mask);
        if (c_pos < c_n) {
            if (n_idle >= TLAS_REFILL_MIN) {
                const uint32_t take = min(n_idle, c_n - c_pos);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_mask, 0u));
                const uint32_t src = c_pos + rank;
                const int bsel = (int)((src & 63u) << 2);
                auto bp = [&](float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(bsel, __float_as_int(v))); };
                const float4 ra = make_float4(bp(ca.x), bp(ca.y), bp(ca.z), bp(ca.w));
                const float2 rb = make_float2(bp(cb.x), bp(cb.y));
                const float rs = sc.has_medium ? bp(cs) : 0.f;
                const V3 rinv = mk(bp(hix), bp(hiy), bp(hiz));
                const float rt = bp(h_t); const uint32_t robj = __float_as_uint(bp(__uint_as_float(h_obj))), rprim = __float_as_uint(bp(__uint_as_float(h_prim)));
                if (slot == IDLE && rank < take) {
                    slot = c_base + src;
                    const Ray r = make_ray(ra, rb, f, segment);
                    wo = r.o; wd = r.d;
                    inv = rinv; sel = wide_sel<WIDE_F32>(inv);
                    path_id = __float_as_uint(rs);
                    cur = skip_ray(f.ex, r) ? W_DONE : sc.wtlas_root; st.sp = 0;          // a NaN ray's record comes from k_extend_exact
                    have = robj != MISS; best_t = rt; best_obj = robj; best_prim = rprim;     // what the hoisted objects gave (prep_block)
                }
                c_pos += take;
                if (c_pos == c_n) {
                    ca = na; cb = nb; cs = ns; c_n = n_n; c_base = n_base; c_pos = 0;
                    if (c_n && bs.next(n_base, n_n)) fetch(n_base, n_n, na, nb, ns); else n_n = 0;
                    { PH_T0; prep_block(); PH_ADD(0); }
                }
            }
        } else if (n_idle == 64u) break;

        const bool busy = slot != IDLE && cur != W_DONE;
        const uint32_t n_busy = (uint32_t)__popcll(__ballot(busy));
        if (busy) PH_COUNT(7);
        for (;;) {
            const bool walking = busy && !(cur & W_LEAF);
            const uint32_t n_walk = (uint32_t)__popcll(__ballot(walking));
            if (n_walk == 0u || (n_walk < n_busy && n_walk * WALK_NUM <= n_busy * WALK_DEN)) break;
            if (walking) { PH_T0; cur = wide_step<WIDE_F32>(lds_w, cur, wo, inv, relaxed(inv), sel, box_tmin(TMIN, false), TMAX, cull_bound(have ? best_t : TMAX), st); PH_ADD(1); }
        }
        if (busy && (cur & W_LEAF) && cur != W_DONE) {
            PH_T0;
            const uint32_t item = cur & 0x7fffu;
            cur = st.sp ? st.pop() : W_DONE;
            Obj o = load_obj_for_hit(sc.obj, item);
            { const uint32_t kd_ = obj_kind(o); if (kd_ == 0u) PH_COUNT(8); else if (kd_ == 4u) PH_COUNT(9); else if (kd_ == 6u) PH_COUNT(10); else PH_COUNT(11); }
            RngKey key{0, 0, 0};
            if (sc.has_medium) key = key_of(f, path_id);
            float t; uint32_t prim;
            // (with a medium the generic test stays: compiling the mesh walk out of the medium's boundary tests alone made part2 1 % slower, 82.6 -> 83.4 ms)
            if (SIMPLE ? hit_object<MEDIUM, 0, false, true>(sc, o, item, Ray{wo, wd}, TMIN, TMAX, nullptr, key, segment, t, prim)
                       : hit_object<MEDIUM, 0, MEDIUM>(sc, o, item, Ray{wo, wd}, TMIN, TMAX, nullptr, key, segment, t, prim)) {
                if ((!have || t < best_t || (t == best_t && sc.obj_rank[item] > sc.obj_rank[best_obj])) && obj_gate_ok(sc, item, wo, inv)) { have = true; best_t = t; best_obj = item; best_prim = prim; }
            }
            PH_ADD(3);
        }
        const bool done = slot != IDLE && cur == W_DONE;
        if (done) { PH_T0; qst(&hits[slot], pack_hit(best_t, best_obj, best_prim, sc.prim_bits)); slot = IDLE; PH_ADD(6); }
    }
    PH_FLUSH(0);
}

// ------------------------------------------------------------------------------------------------
// K2-exact  closest_hit_exact: one ray by the literal algorithm of the reference — scene.rs:137-149 (linear) or bvh.rs:115-151
// over the reference's OWN trees (use_bvh).  Used by k_extend_exact (behind k_shade below) for the rays on a segment's list
// (DExact, fw_device.h).  Nodes from L2, stacks in LDS.  With FIREWORK_EXACT_ALL=1 every ray takes it: the renderer then IS the
// reference's traversal.
// ------------------------------------------------------------------------------------------------
// The literal TLAS walk, one ray per wave (the wave form of k_extend_exact under use_bvh): hit_mesh_exact_wave's scheme one level up.
// The lanes pop up to 64 nodes of a shared stack per round; a lane that pops a leaf tests its objects itself — except objects with a
// mesh inside, whose own walk wants the whole wave: those go on a short list and are walked afterwards, one after the other, by
// all lanes together.  Winner: smallest t, the later item in depth-first order on a tie; a NaN t the way the reference's rule
// treats it (see hit_mesh_exact_wave: second pass behind the last NaN hit).  Returns false, having decided nothing, when more
// than EXACT_DEFER mesh objects are reached (the caller then walks node by node).
// ws: EXACT_STACK entries of stack + 2 * EXACT_DEFER of list, shared by the wave.
constexpr uint32_t EXACT_DEFER = 64;
__device__ __forceinline__ bool closest_hit_exact_tlas_wave(const DScene &sc, const Ray &r, const RngKey &key, int segment, uint32_t *ws, uint32_t *blas_stack,
                                                            float &best_t, uint32_t &best_obj, uint32_t &best_prim) {
    const float TMIN = 0.001f, TMAX = 2e9f;                          // render.rs:19
    const uint32_t lane = threadIdx.x & 63u;
    const V3 inv = mk(fdiv(1.f, r.d.x), fdiv(1.f, r.d.y), fdiv(1.f, r.d.z));
    uint32_t *dl = ws + EXACT_STACK;                                 // [EXACT_DEFER items][EXACT_DEFER positions]
    bool have = false; float best = TMAX; uint32_t bobj = MISS, bprim = 0, bpos = 0;
    int after = -1, nan_pos = -1; uint32_t nan_obj = 0, nan_prim = 0, nan_bits = 0;
    auto take = [&](float t, uint32_t item, uint32_t prim, uint32_t pos) {
        if (t != t) { if ((int)pos > nan_pos) { nan_pos = (int)pos; nan_obj = item; nan_prim = prim; nan_bits = __float_as_uint(t); } }
        else if ((int)pos > after && (!have || t < best || (t == best && pos > bpos))) { have = true; best = t; bobj = item; bprim = prim; bpos = pos; }
    };
    for (bool second = false;; second = true) {
        have = false; best = TMAX; bobj = MISS; bprim = 0; bpos = 0;
        uint32_t nd = 0;                                             // listed mesh objects (wave-uniform)
        uint32_t sp = 1;
        if (lane == 0) ws[0] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        while (sp > 0u) {
            const int room = EXACT_STACK - EXACT_LEVELS - 8 - (int)sp;
            uint32_t k = min(64u, sp);
            if ((int)k > room) k = (uint32_t)max(1, room);
            const bool active = lane < k;
            const uint32_t node = active ? ws[sp - 1u - lane] : 0u;
            sp -= k;
            float4 lo = make_float4(0, 0, 0, 0), hi = lo;
            if (active) { lo = sc.ref_tlas[2 * (size_t)node]; hi = sc.ref_tlas[2 * (size_t)node + 1]; }
            const uint32_t A = __float_as_uint(lo.w), B = __float_as_uint(hi.w);
            const bool hitb = active && hit_aabb(lo, hi, r.o, inv, TMIN, TMAX);
            const bool branch = hitb && (A >> 30) == 0u;
            const unsigned long long m = __ballot(branch);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            if (branch) {
                const uint32_t at = sp + 2u * (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                ws[at] = A & NODE_MASK; ws[at + 1u] = node + 1u;
            }
            sp += 2u * (uint32_t)__popcll(m);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            const bool leaf = hitb && !branch;
            for (uint32_t which = 0; which < 2u; which++) {          // a Leaf's item, then a DoubleLeaf's second
                const bool on = leaf && (which == 0u || (A >> 30) == NODE_DOUBLE);
                const uint32_t item = which ? B : (A & NODE_MASK);
                bool defer = false;
                if (on) {
                    const Obj o = load_obj(sc.obj, item);
                    defer = obj_kind(o) == 5u || (obj_kind(o) == 6u && obj_inner(o) == 5u);
                    float t; uint32_t prim;
                    if (!defer && hit_object<true, 1>(sc, o, item, r, TMIN, TMAX, blas_stack, key, segment, t, prim)) take(t, item, prim, 2u * node + which);
                }
                const unsigned long long dm = __ballot(defer);
                if (dm) {
                    const uint32_t e = nd + (uint32_t)__popcll(dm & ((1ull << lane) - 1ull));
                    if (defer && e < EXACT_DEFER) { dl[e] = item; dl[EXACT_DEFER + e] = 2u * node + which; }
                    nd += (uint32_t)__popcll(dm);
                }
            }
        }
        if (nd > EXACT_DEFER) return false;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        for (uint32_t e = 0; e < nd; e++) {                          // every lane the same object: its mesh walk is the wave's
            const uint32_t item = dl[e], pos = dl[EXACT_DEFER + e];
            const Obj o = load_obj(sc.obj, item);
            float t; uint32_t prim;
            if (hit_object<true, 2>(sc, o, item, r, TMIN, TMAX, blas_stack, key, segment, t, prim)) take(t, item, prim, pos);
        }
        if (second || __ballot(nan_pos >= 0) == 0ull) break;
        for (int d = 1; d < 64; d <<= 1) {        // the last NaN hit of the wave
            const int op = __shfl_xor(nan_pos, d);
            const uint32_t oo = (uint32_t)__shfl_xor((int)nan_obj, d), opr = (uint32_t)__shfl_xor((int)nan_prim, d), ob = (uint32_t)__shfl_xor((int)nan_bits, d);
            if (op > nan_pos) { nan_pos = op; nan_obj = oo; nan_prim = opr; nan_bits = ob; }
        }
        after = nan_pos;
    }
    for (int d = 1; d < 64; d <<= 1) {
        const bool oh = __shfl_xor((int)have, d) != 0;
        const float ot = __shfl_xor(best, d);
        const uint32_t oo = (uint32_t)__shfl_xor((int)bobj, d), opr = (uint32_t)__shfl_xor((int)bprim, d), opos = (uint32_t)__shfl_xor((int)bpos, d);
        if (oh && (!have || ot < best || (ot == best && opos > bpos))) { have = true; best = ot; bobj = oo; bprim = opr; bpos = opos; }
    }
    if (!have && after >= 0) { best_t = __uint_as_float(nan_bits); best_obj = nan_obj; best_prim = nan_prim; return true; }
    best_t = have ? best : TMAX; best_obj = have ? bobj : MISS; best_prim = have ? bprim : 0u;
    return true;
}

template <int EXACT>   // 1: one ray per lane, 2: one ray per wave (hit_mesh_exact_lane / _wave)
__device__ __forceinline__ void closest_hit_exact(const DScene &sc, const Ray &r, const RngKey &key, int segment, bool use_bvh, uint32_t *tlas_stack,
                                                  uint32_t *blas_stack, float &best_t, uint32_t &best_obj, uint32_t &best_prim) {
    const float TMIN = 0.001f, TMAX = 2e9f;                          // render.rs:19
    best_t = TMAX; best_obj = MISS; best_prim = 0;
    if (!use_bvh) {                                                  // scene.rs:137-149: in order, narrowing, a later object replaces
        for (uint32_t k = 0; k < sc.n_objects; k++) {
            const Obj o = load_obj(sc.obj, k);
            float t; uint32_t prim;
            if (hit_object<true, EXACT>(sc, o, k, r, TMIN, best_t, blas_stack, key, segment, t, prim)) { best_t = t; best_obj = k; best_prim = prim; }
        }
        return;
    }
    // a TLAS of a few objects (suzanne: 3) is walked faster node by node with every lane the same: the shared walk's rounds, its list
    // and its reductions cost more than they save (k_extend_exact 40 -> 51 us per launch on suzanne; part2, 1 409 objects: 62 -> ~25)
    if (EXACT == 2 && sc.n_objects > 16u && closest_hit_exact_tlas_wave(sc, r, key, segment, tlas_stack - (threadIdx.x & 63u), blas_stack, best_t, best_obj, best_prim)) return;
    best_t = TMAX; best_obj = MISS; best_prim = 0;
    const V3 inv = mk(fdiv(1.f, r.d.x), fdiv(1.f, r.d.y), fdiv(1.f, r.d.z));
    int sp = 0;
    uint32_t cur = 0; bool have = false;
    for (bool more = true; more;) {          // while-while, like hit_mesh_exact_lane: down to the next leaf node first, then the object tests together
        uint32_t A = 0, B = 0; bool leaf = false;
        for (;;) {
            const float4 lo = sc.ref_tlas[2 * (size_t)cur], hi = sc.ref_tlas[2 * (size_t)cur + 1];
            A = __float_as_uint(lo.w); B = __float_as_uint(hi.w);
            if (hit_aabb(lo, hi, r.o, inv, TMIN, TMAX)) {
                if ((A >> 30) == 0u) { if (sp < EXACT_LEVELS) { tlas_stack[sp * EXACT_WB] = A & NODE_MASK; sp++; } cur = cur + 1u; continue; }
                leaf = true; break;
            }
            if (sp == 0) { more = false; break; }
            sp--; cur = tlas_stack[sp * EXACT_WB];
        }
        if (leaf) {
            for (uint32_t k = 0; k < ((A >> 30) == NODE_DOUBLE ? 2u : 1u); k++) {
                const uint32_t item = k ? B : (A & NODE_MASK);
                const Obj o = load_obj(sc.obj, item);
                float t; uint32_t prim;
                if (hit_object<true, EXACT>(sc, o, item, r, TMIN, TMAX, blas_stack, key, segment, t, prim))
                    if (!have || !(best_t < t)) { have = true; best_t = t; best_obj = item; best_prim = prim; }
            }
            if (sp == 0) more = false;
            else { sp--; cur = tlas_stack[sp * EXACT_WB]; }
        }
    }
}
// ------------------------------------------------------------------------------------------------
// K5/K6  shading: textures, materials, environment
// ------------------------------------------------------------------------------------------------
__constant__ __attribute__((aligned(4))) uint8_t PERM[256] = {   // texture.rs:80-106
    151, 160, 137, 91, 90, 15, 131, 13, 201, 95, 96, 53, 194, 233, 7, 225, 140, 36, 103, 30, 69, 142, 8, 99, 37, 240, 21, 10,
    23, 190, 6, 148, 247, 120, 234, 75, 0, 26, 197, 62, 94, 252, 219, 203, 117, 35, 11, 32, 57, 177, 33, 88, 237, 149, 56, 87,
    174, 20, 125, 136, 171, 168, 68, 175, 74, 165, 71, 134, 139, 48, 27, 166, 77, 146, 158, 231, 83, 111, 229, 122, 60, 211,
    133, 230, 220, 105, 92, 41, 55, 46, 245, 40, 244, 102, 143, 54, 65, 25, 63, 161, 1, 216, 80, 73, 209, 76, 132, 187, 208,
    89, 18, 169, 200, 196, 135, 130, 116, 188, 159, 86, 164, 100, 109, 198, 173, 186, 3, 64, 52, 217, 226, 250, 124, 123, 5,
    202, 38, 147, 118, 126, 255, 82, 85, 212, 207, 206, 59, 227, 47, 16, 58, 17, 182, 189, 28, 42, 223, 183, 170, 213, 119,
    248, 152, 2, 44, 154, 163, 70, 221, 153, 101, 155, 167, 43, 172, 9, 129, 22, 39, 253, 19, 98, 108, 110, 79, 113, 224, 232,
    178, 185, 112, 104, 218, 246, 97, 228, 251, 34, 242, 193, 238, 210, 144, 12, 191, 179, 162, 241, 81, 51, 145, 235, 249,
    14, 239, 107, 49, 192, 214, 31, 181, 199, 106, 157, 184, 84, 204, 176, 115, 121, 50, 45, 127, 4, 150, 254, 138, 236, 205,
    93, 222, 114, 67, 29, 24, 72, 243, 141, 128, 195, 78, 66, 215, 61, 156, 180};
// Perlin's lattice walk is three dependent table reads deep and a turbulence texture runs it seven times: from
// __constant__ memory with per-lane indices each read is a vector load from L2 (~1 us under load), 21 in a row for the
// few lanes of a wave that hit such a texture while the others wait.  k_shade / k_bounce copy the table into LDS first
// (stage_perm) when the scene has such a texture: part2's k_shade 6.7 -> 5.7 ms, the frame @16 spp 10.19 -> 9.72 ms.
__shared__ uint8_t lds_perm[256];
__device__ __forceinline__ void stage_perm() {     // all threads of the workgroup; the caller's barrier publishes it
    for (uint32_t k = threadIdx.x; k < 64u; k += blockDim.x) reinterpret_cast<uint32_t *>(lds_perm)[k] = reinterpret_cast<const uint32_t *>(PERM)[k];
}
__device__ __forceinline__ uint32_t P(uint32_t i) { return lds_perm[i & 255u]; }
// Rust `f32 as usize & 255`: saturating cast (negatives and NaN -> 0, huge -> usize::MAX -> 255)
__device__ __forceinline__ uint32_t lattice(float fl) {
    if (!(fl > 0.f)) return 0u;
    if (fl >= 18446744073709551616.f) return 255u;
    return (uint32_t)((unsigned long long)fl & 255ull);
}
__device__ __forceinline__ float fade(float t) { return t * t * (3.f - 2.f * t); }
__device__ __forceinline__ float plerp(float t, float a, float b) { return a + t * (b - a); }
__device__ __forceinline__ float grad(uint32_t hash, float x, float y, float z) {
    uint32_t h = hash & 15u;
    float u = h < 8 ? x : y;
    float v = h < 4 ? y : ((h == 12 || h == 14) ? x : z);
    u = (h & 1) == 0 ? u : -u;
    v = (h & 2) == 0 ? v : -v;
    return u + v;
}
__device__ float perlin_noise(V3 p) {    // texture.rs:113-158
    float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
    uint32_t x0 = lattice(fx), y0 = lattice(fy), z0 = lattice(fz);
    float x = p.x - fx, y = p.y - fy, z = p.z - fz;
    float u = fade(x), v = fade(y), w = fade(z);
    uint32_t a = P(x0) + y0, aa = P(a) + z0, ab = P(a + 1) + z0;
    uint32_t b = P(x0 + 1) + y0, ba = P(b) + z0, bb = P(b + 1) + z0;
    return plerp(w,
        plerp(v, plerp(u, grad(P(aa), x, y, z), grad(P(ba), x - 1.f, y, z)),
                 plerp(u, grad(P(ab), x, y - 1.f, z), grad(P(bb), x - 1.f, y - 1.f, z))),
        plerp(v, plerp(u, grad(P(aa + 1), x, y, z - 1.f), grad(P(ba + 1), x - 1.f, y, z - 1.f)),
                 plerp(u, grad(P(ab + 1), x, y - 1.f, z - 1.f), grad(P(bb + 1), x - 1.f, y - 1.f, z - 1.f))));
}
__device__ float turb(uint32_t depth, V3 point) {    // texture.rs:206-217
    float accum = 0.f, weight = 1.f; V3 p = point;
    for (uint32_t i = 0; i < depth; i++) { accum += weight * perlin_noise(p); weight *= 0.5f; p = p * 2.f; }
    return accum;
}
__device__ __forceinline__ uint32_t sat_u32(float f) { if (!(f > 0.f)) return 0u; if (f >= 4294967296.f) return 0xffffffffu; return (uint32_t)f; }

// Texture::sample (texture.rs).  Checker recursion is unrolled into a bounded walk.
__device__ __forceinline__ V3 texture_sample(const float4 *texp, const uint8_t *images, uint32_t tex, float u, float v, V3 p) {
    for (int guard = 0; guard < 16; guard++) {
        float4 t0 = texp[2 * tex], t1 = texp[2 * tex + 1];
        uint32_t kind = __float_as_uint(t0.x);
        float scale = t0.y; uint32_t depth = __float_as_uint(t0.z);
        switch (kind) {
        case 0: return mk(t1.x, t1.y, t1.z);                                          // texture.rs:29-34
        case 1: {                                                                      // texture.rs:57-73
            float prod = 1.0f;
            prod = prod * fwlm::sinf_glibc(scale * p.x); prod = prod * fwlm::sinf_glibc(scale * p.y); prod = prod * fwlm::sinf_glibc(scale * p.z);
            bool positive = (__float_as_uint(prod) >> 31) == 0u;                       // is_sign_positive
            tex = positive ? __float_as_uint(t1.x) : __float_as_uint(t0.w);
            continue; }
        case 2: { float a = perlin_noise(p * scale); float c = fminf(a + 0.5f, 1.f); return mk(c, c, c); }   // texture.rs:161-168
        case 3: { float c = turb(depth, p * scale); return mk(c, c, c); }                                    // texture.rs:219-225
        case 4: { float c = 0.5f * (1.f + fwlm::sinf_glibc(scale * p.z + 10.f * turb(depth, p))); return mk(c, c, c); }  // texture.rs:239-249
        case 5: {                                                                      // texture.rs:296-309
            uint32_t off = __float_as_uint(t1.x), w = __float_as_uint(t1.y), h = __float_as_uint(t1.z);
            float fi = u * (float)w, fj = (1.f - v) * (float)h;
            uint32_t i = min(sat_u32(fi), w - 1), j = min(sat_u32(fj), h - 1);
            const uint8_t *c = images + off + 3 * ((size_t)j * w + i);
            return mk((float)c[0], (float)c[1], (float)c[2]) / 255.f; }
        default: return mk(0.f, 0.f, 0.f);
        }
    }
    return mk(0.f, 0.f, 0.f);
}

__device__ __forceinline__ void sphere_uv(V3 p, float &u, float &v) {   // objects/sphere.rs:22-29
    float phi = fwlm::atan2f_glibc(p.z, p.x);
    float theta = fwlm::asinf_glibc(p.y);
    u = 1.f - fdiv(phi + PI_F, 2.f * PI_F);
    v = fdiv(theta + PI_F / 2.f, PI_F);
}

__device__ V3 env_sample(const DEnv &e, V3 dir) {     // environment.rs:21-26,60-67; examples/hdri_test.rs:70-82
    if (e.kind == 0) return mk(e.color[0], e.color[1], e.color[2]);
    if (e.kind == 1) {
        float t = 0.5f * (dir.y + 1.0f);
        return (1.f - t) * mk(e.horizon[0], e.horizon[1], e.horizon[2]) + t * mk(e.zenith[0], e.zenith[1], e.zenith[2]);
    }
    float u, v; sphere_uv(dir, u, v);
    float width = (float)e.hdr_w, height = (float)e.hdr_h;
    auto sat64 = [](float f) -> unsigned long long { if (!(f > 0.f)) return 0ull; if (f >= 18446744073709551616.f) return ~0ull; return (unsigned long long)f; };
    unsigned long long x = sat64(u * width), y = sat64((1.f - v) * height);
    unsigned long long idx = sat64((float)y * width) + x;
    unsigned long long n = (unsigned long long)e.hdr_w * e.hdr_h;
    if (idx >= n) idx = n - 1;     // the reference indexes out of bounds at the poles; clamp
    const float4 c = reinterpret_cast<const float4 *>(e.hdr)[idx];     // 16-byte texels (the host pads: fw_runtime.cpp)
    return mk(c.x, c.y, c.z);
}

__device__ __forceinline__ V3 reflect(V3 v, V3 n) { return v - 2.f * dot(v, n) * n; }           // util.rs:54-56
__device__ __forceinline__ bool refract(V3 v, V3 n, float ni_over_nt, V3 &out) {                 // util.rs:58-67
    V3 uv = normalized(v);
    float dt = dot(uv, n);
    float disc = 1.f - ni_over_nt * ni_over_nt * (1.f - dt * dt);
    if (disc > 0.f) { out = ni_over_nt * (uv - n * dt) - n * fsqrt(disc); return true; }
    return false;
}
__device__ __forceinline__ float schlick(float cosine, float ref_idx) {                          // util.rs:69-73
    float r0 = fdiv(1.f - ref_idx, 1.f + ref_idx);
    r0 = r0 * r0;
    return r0 + (1.f - r0) * fwlm::powf_glibc(1.f - cosine, 5.f);
}

// rebuild the RaycastHit (render.rs:35-41) of the recorded (t, object, primitive) in world space
struct HitInfo { V3 point, normal; float u, v; uint32_t material; };

__device__ __forceinline__ void rect_hitinfo(uint32_t kind, float a_min, float a_max, float b_min, float b_max, bool flip,
                                             V3 p, V3 &normal, float &u, float &v, bool need_uv) {
    int a1 = (kind == 3) ? 1 : 0, a2 = (kind == 1) ? 1 : 2, ot = (kind == 1) ? 2 : (kind == 2 ? 1 : 0);
    V3 n = mk(ot == 0 ? 1.f : 0.f, ot == 1 ? 1.f : 0.f, ot == 2 ? 1.f : 0.f);
    normal = flip ? -n : n;
    if (need_uv) {   // uv feed ImageTexture only (texture.rs:296-309); every other texture ignores them
        u = fdiv(comp(p, a1) - a_min, a_max - a_min);
        v = fdiv(comp(p, a2) - b_min, b_max - b_min);
    }
}

// The t of a recorded hit on a sphere, rectangle or Rect3d face, recomputed with the very expressions k_extend used (4-byte
// hit records, DFrame.hit4).  A sphere's accepted root is t1 when t1 > t_min, else t2: the caller's t_max can only reject,
// and had it rejected t1 it would have rejected the larger t2 as well (sphere.rs:40-47).
__device__ __forceinline__ float recompute_t(const Obj &o, const Ray &r, uint32_t prim) {
    const uint32_t kind = obj_kind(o);
    if (kind == 0u) {
        const float a = dot(r.d, r.d), b = 2.f * dot(r.o, r.d), c = dot(r.o, r.o) - o.q3.x * o.q3.x;
        const float disc = b * b - 4.f * a * c;
        const Rcp two_a = make_rcp(2.f * a);
        if (disc == 0.f) return fdiv(-b, two_a);
        const float sq = fsqrt(disc), t1 = fdiv(-b - sq, two_a);
        return t1 > 0.001f ? t1 : fdiv(-b + sq, two_a);
    }
    float k = o.q4.x; uint32_t ot = kind == 1u ? 2u : (kind == 2u ? 1u : 0u);     // plane axis of XY / XZ / YZ
    if (kind == 4u) {                                                             // rect3d.rs:19-77: faces +z -z +y -y +x -x
        const float px = o.q3.x, py = o.q3.y, pz = o.q3.z, sx = o.q3.w, sy = o.q4.x, sz = o.q4.y;
        const uint32_t ax = prim >> 1;
        ot = ax == 0u ? 2u : (ax == 1u ? 1u : 0u);
        k = ax == 0u ? ((prim & 1u) ? pz : pz + sz) : (ax == 1u ? ((prim & 1u) ? py : py + sy) : ((prim & 1u) ? px : px + sx));
    }
    return fdiv(k - comp(r.o, (int)ot), comp(r.d, (int)ot));
}

__device__ __forceinline__ HitInfo rebuild_hit(const DScene &sc, const Obj &o, const Ray &world, float t, uint32_t prim, bool need_uv, bool hit4 = false) {
    Ray r = to_object_space(o, world);
    if (hit4) t = recompute_t(o, r, prim);
    uint32_t kind = obj_kind(o), flags = obj_flags(o);
    HitInfo h; h.material = o.material; h.u = 0.f; h.v = 0.f;
    V3 p = ray_point(r, t), n = mk(0.f, 1.f, 0.f);
    switch (kind) {
    case 0: {                                                                          // sphere.rs:49-56
        n = p / o.q3.x;
        if (need_uv) sphere_uv(n, h.u, h.v);
        break; }
    case 1: case 2: case 3:
        rect_hitinfo(kind, o.q3.x, o.q3.y, o.q3.z, o.q3.w, (flags & OF_RECT_FLIP) != 0, p, n, h.u, h.v, need_uv);
        break;
    case 4: {                                                                          // rect3d.rs:18-80
        float px = o.q3.x, py = o.q3.y, pz = o.q3.z, sx = o.q3.w, sy = o.q4.x, sz = o.q4.y;
        uint32_t fk = 1 + (prim >> 1);   // faces 0,1 -> XY(1); 2,3 -> XZ(2); 4,5 -> YZ(3)
        bool flip = (prim & 1) != 0;
        if (fk == 1) rect_hitinfo(1, px, px + sx, py, py + sy, flip, p, n, h.u, h.v, need_uv);
        else if (fk == 2) rect_hitinfo(2, px, px + sx, pz, pz + sz, flip, p, n, h.u, h.v, need_uv);
        else rect_hitinfo(3, py, py + sy, pz, pz + sz, flip, p, n, h.u, h.v, need_uv);
        break; }
    case 5: {                                                                          // mesh.rs:196-218
        const float4 *tp = sc.tri + 3 * (size_t)(o.aux1 + prim);
        float4 a = tp[0], b = tp[1], c = tp[2];
        V3 p0 = mk(a.x, a.y, a.z), p1 = mk(b.x, b.y, b.z), p2 = mk(c.x, c.y, c.z);
        float tt, b0 = 0.f, b1 = 0.f, b2 = 0.f;
        hit_triangle(p0, p1, p2, make_triray(r), -3.40282347e+38f, 3.40282347e+38f, tt, b0, b1, b2);   // same e0..e2/det as in k_extend
        p = b0 * p0 + b1 * p1 + b2 * p2;
        h.u = b0 * a.w + b1 * b.w + b2 * c.w;
        if (flags & OF_MESH_ATTR) {
            const float4 *np_ = sc.tri_nrm + 3 * (size_t)(o.aux1 + prim);
            float4 na = np_[0], nb = np_[1], nc = np_[2];
            h.v = b0 * na.w + b1 * nb.w + b2 * nc.w;
            if (flags & OF_MESH_NORMALS)
                n = normalized(b0 * mk(na.x, na.y, na.z) + b1 * mk(nb.x, nb.y, nb.z) + b2 * mk(nc.x, nc.y, nc.z));
            else
                n = cross(p0 - p2, p1 - p2);
        } else {
            h.v = b0 * 0.f + b1 * 0.f + b2 * 1.f;                                      // default uvs (mesh.rs:107)
            n = cross(p0 - p2, p1 - p2);                                               // unnormalised (mesh.rs:208)
        }
        break; }
    case 7: {                                                                          // cone.rs:62-77
        float radius = o.q3.x, height = o.q3.y;
        float v = fdiv(p.y, height);
        Rcp omv = make_rcp(1.f - v);
        V3 dpdu = mk(-p.z, 0.f, p.x);
        V3 dpdv = mk(fdiv(-p.x, omv), height, fdiv(-p.z, omv));
        n = normalized(cross(dpdv, dpdu));
        if (need_uv) { h.u = fdiv(fwlm::acosf_glibc(fdiv(p.x, radius * (1.f - v))), 2.f * PI_F); h.v = v; }
        break; }
    case 8: {                                                                          // cylinder.rs:66-78
        Rcp rr = make_rcp(o.q3.x);
        n = mk(fdiv(p.x, rr), 0.f, fdiv(p.z, rr));
        if (need_uv) {
            float phi = fwlm::atan2f_glibc(p.z, p.x);
            if (phi < 0.f) phi = phi + PI_F * 2.f;
            h.u = fdiv(phi, o.q3.z); h.v = fdiv(p.y, o.q3.y);
        }
        break; }
    case 9: {                                                                          // disk.rs:60-82
        if (need_uv) {
            float phi = fwlm::atan2f_glibc(p.z, p.x);
            if (phi < 0.f) phi = phi + 2.f * PI_F;
            h.u = fdiv(phi, o.q3.z);
            float dist = fsqrt(p.x * p.x + p.z * p.z);
            h.v = 1.f - fdiv(dist - o.q3.w, o.q3.x - o.q3.w);
        }
        break; }
    default: break;                                                                    // medium: normal +Y, uv 0 (volume.rs:72-78)
    }
    V3 pos = mk(o.q0.w, o.q1.w, o.q2.w);
    h.point = rot_fwd(o, p) + pos;                                                     // scene.rs:255-256 (always rotated)
    h.normal = rot_fwd(o, n);
    if (flags & OF_FLIP) h.normal = -h.normal;
    return h;
}

// ------------------------------------------------------------------------------------------------
// K5 + K7  shade + stream compaction
//
// k_shade is latency-bound (rocprofv3: 79 % of wave cycles in s_waitcnt in the first version): the chain
// hit -> object record -> material -> texture is three dependent fetches.  LDS_TAB = true stages the
// object / material / texture tables in LDS once per workgroup (scenes whose tables fit 16 KB: cornell,
// suzanne, hdri, volume), which turns those fetches into ~64-cycle ds_reads; materials whose texture is a
// ConstantTexture carry the colour inline, removing the third level entirely.
// ------------------------------------------------------------------------------------------------
// One path at its hit (or miss): emission / environment for paths that end here (written to sample_rad, every
// path writes exactly once), the scattered ray and throughput for those that continue (render.rs:19-31).
// "Expensive" shading (k_shade's deferred list): what a few lanes of a chunk do while the others wait — a texture that is not a
// constant (three sinf for a checker, 5-10 octaves of Perlin noise, an image lookup behind atan2f + asinf), the Fresnel branch
// of a dielectric (normalise, refract, powf), a miss into an HDR map (atan2f + asinf + a gather from 100 MB).
__device__ __forceinline__ bool expensive_shading(const DScene &sc, const float4 *objp, const float4 *matp, uint32_t hit_code) {
    if (hit_code == MISS) return sc.env.kind == 2;
    const uint32_t material = __float_as_uint(objp[(size_t)(hit_code >> sc.prim_bits) * OBJ_Q + 3].w);
    const uint32_t mbits = __float_as_uint(matp[2 * material].x), mkind = mbits & 0xffu;
    return mkind == 2u || (!(mbits & MF_TEX_CONST) && (mkind == 0u || mkind == 3u || mkind == 4u));
}
// CHEAP_ONLY: the caller has sent the expensive cases elsewhere (expensive_shading), their code is compiled out.
// CHAIN: the path's state is its chain of material ids (load_state_chain), `chain` in and `nchain` out; beta / nbeta are unused.
template <bool CHEAP_ONLY = false, bool CHAIN = false>
__device__ __forceinline__ bool shade_path(const DScene &sc, const DFrame &f, const float4 *objp, const float4 *matp,
                                           const float4 *texp, const Ray &r, V3 beta, uint32_t chain, uint32_t path_id, float t_hit,
                                           uint32_t hit_code, int segment, float4 *__restrict__ sample_rad, Ray &nr, V3 &nbeta, uint32_t &nchain PH_ARG,
                                           const RngKey *pre_key = nullptr) {      // pre_key: the path's RNG key, when the caller has fetched it already (k_shade, FW_SHADE_PIPE)
    bool alive = false;
    const uint32_t obj_index = hit_code == MISS ? MISS : (hit_code >> sc.prim_bits);
    V3 rad = mk(0.f, 0.f, 0.f);
    // what the path carries to the camera of a radiance x found at this segment: beta * x, or with the chain the reference's own
    // nesting a0 * (a1 * (... (a_{k-1} * x))) (render.rs:23-28: `emit + attenuation * color(..)`, emit = 0 on the way)
    auto carried = [&](V3 x) -> V3 {
        if (!CHAIN && f.atten) {        // option EXACT_PRODUCT: the attenuations themselves, one record per scattering at the path's home slot
            for (int sgm = segment - 1; sgm >= 0; sgm--) {
                const float4 c = f.atten[(size_t)sgm * f.atten_stride + path_id];
                x = mk(c.x, c.y, c.z) * x;
            }
            return x;
        }
        if (!CHAIN) return beta * x;
        const uint32_t mask = (1u << f.chain_bits) - 1u;
        for (int sgm = segment - 1; sgm >= 0; sgm--) {
            const float4 c = matp[2 * ((chain >> (f.chain_bits * (uint32_t)sgm)) & mask) + 1];
            x = mk(c.x, c.y, c.z) * x;
        }
        return x;
    };
    if (obj_index == MISS) {
        PH_T0;
        // render.rs:31; ColorEnv ignores the direction, so its normalisation (sqrt + 3 divisions) is skipped
        V3 dir = sc.env.kind == 0 ? r.d : normalized(r.d);
        if (CHEAP_ONLY) {      // ColorEnv or SkyEnv (environment.rs:21-26,60-67); an HdrEnvironment miss is an expensive case
            // (spelled out on sc.env's own fields: through a modified COPY of the struct the colour came back as a per-lane vector load from
            // the kernel-argument segment — a global load in the middle of k_shade's chunk, which on gfx9's in-order vmcnt completes the next
            // chunk's prefetch with it, round 5)
            if (sc.env.kind == 1) {
                const float t = 0.5f * (dir.y + 1.0f);
                rad = carried((1.f - t) * mk(sc.env.horizon[0], sc.env.horizon[1], sc.env.horizon[2]) + t * mk(sc.env.zenith[0], sc.env.zenith[1], sc.env.zenith[2]));
            } else if (!f.skip_zero_deposits) rad = carried(mk(sc.env.color[0], sc.env.color[1], sc.env.color[2]));
            // else: a black ColorEnv (the host checked): the path carries nothing, whatever it scattered on, and writes no record (below)
        } else rad = carried(env_sample(sc.env, dir));
        PH_ADD(1);
    } else {
        PH_T0;
        Obj o = load_obj(objp, obj_index);
        float4 m0 = matp[2 * o.material], m1 = matp[2 * o.material + 1];
        uint32_t mbits = __float_as_uint(m0.x), mkind = mbits & 0xffu, mtex = __float_as_uint(m0.y);
        bool need_uv = (mbits & MF_NEEDS_UV) != 0, tex_const = (mbits & MF_TEX_CONST) != 0;
        HitInfo h = rebuild_hit(sc, o, r, t_hit, hit_code & ((1u << sc.prim_bits) - 1u), need_uv, f.hit4 != 0u);
        V3 texc = mk(m1.x, m1.y, m1.z);                                      // inline ConstantTexture / Metal albedo
        PH_ADD(2);
        // (the chain state exists only where every such texture is a constant — the host checks —, so its kernels carry no texture code)
        if (!CHEAP_ONLY && !CHAIN && !tex_const && (mkind == 0 || mkind == 3 || mkind == 4)) { PH_T0; texc = texture_sample(texp, sc.images, mtex, h.u, h.v, h.point); PH_ADD(3); }
        if (mkind == 3) {                                                      // EmissiveMat: emit, never scatters
            PH_T0;
            rad = carried(texc);
            PH_ADD(4);
        } else if (segment < 10) {                                             // render.rs:21
            PH_T0;
            if (mkind == 0) PH_COUNT(8); else if (mkind == 1) PH_COUNT(9); else if (mkind == 2) PH_COUNT(10); else PH_COUNT(12);
            RngKey key = pre_key ? *pre_key : key_of(f, path_id);
            V3 atten = texc;
            switch (mkind) {
            case 0: {                                                          // Lambertian material.rs:64-75
                V3 target = h.point + h.normal + random_in_unit_sphere(key, segment PH_PASS);
                nr = Ray{h.point, target - h.point};
                alive = true; break; }
            case 1: {                                                          // Metal material.rs:90-107
                V3 reflected = reflect(r.d, h.normal);
                nr = Ray{h.point, reflected + m0.z * random_in_unit_sphere(key, segment PH_PASS)};
                alive = dot(nr.d, h.normal) > 0.f; break; }
            case 2: if (!CHEAP_ONLY) {                                         // Dielectric material.rs:121-151
                float ref_idx = m0.w;
                V3 reflected = reflect(r.d, h.normal);
                V3 outward; float ni_over_nt, cosine;
                float ddn = dot(r.d, h.normal);
                if (ddn > 0.f) { outward = -h.normal; ni_over_nt = ref_idx; cosine = fdiv(ref_idx * ddn, mag(r.d)); }
                else { outward = h.normal; ni_over_nt = fdiv(1.0f, ref_idx); cosine = fdiv(-ddn, mag(r.d)); }
                atten = mk(1.f, 1.f, 1.f);
                V3 refracted; bool took_refraction = false;
                if (refract(r.d, outward, ni_over_nt, refracted)) {
                    float xi = u2f(draw(key, P_FRESNEL, segment, 0).x);
                    if (xi > schlick(cosine, ref_idx)) { nr = Ray{h.point, refracted}; took_refraction = true; }
                }
                if (!took_refraction) nr = Ray{h.point, reflected};
                alive = true; } break;
            case 4: {                                                          // Isotropic material.rs:197-204
                nr = Ray{h.point, random_in_unit_sphere(key, segment PH_PASS)};
                alive = true; break; }
            default: break;
            }
            if (CHAIN) nchain = chain | (o.material << (f.chain_bits * (uint32_t)segment));   // atten IS the material's constant (host: chain_bits)
            else {
                nbeta = beta * atten;
                if (f.atten && alive) f.atten[(size_t)segment * f.atten_stride + path_id] = make_float4(atten.x, atten.y, atten.z, 0.f);
            }
            PH_ADD(5);
        }
    }
    // every path deposits exactly once — except zeros over a black environment: k_raygen has already written them, densely
    // (adding +0 is exact, and most indoor paths end black; the scattered 16-byte deposits are k_shade's costliest stores)
#ifdef FW_ABL_NO_DEPOSIT      // timing ablation (wrong frames): no radiance record, no bit
    if (false) {
#else
    if (!alive && !(f.skip_zero_deposits && rad.x == 0.f && rad.y == 0.f && rad.z == 0.f)) {
#endif
        PH_T0;
        // .w = the path's length in segments: k_accumulate sums it next to the colour, so accum.w of a pixel is its ray count
        // whenever every path deposits (any non-black environment, or FIREWORK_NO_ZERO_SKIP=1) — what tools/diverge.py
        // compares with the oracle's per-pixel counts to find a diverging path
        const float len = (float)(segment + 1);
        if (FW_NT_RAD) st_nt(&sample_rad[path_id], make_float4(rad.x, rad.y, rad.z, len));
        else sample_rad[path_id] = make_float4(rad.x, rad.y, rad.z, len);
        if (f.skip_zero_deposits) {                                  // "this path wrote a record" (3.6 % of cornell's paths)
            const uint32_t b = f.dep_pixel_major ? dep_bit_of(f, path_id) : path_id;
            atomicOr(&f.dep_bits[b >> 5], 1u << (b & 31u));
        }
        PH_ADD(6);
    }
    return alive;
}

extern __shared__ float4 lds_tables[];

// Round 3: the expensive materials of a chunk go to a LIST.  rocprofv3 on part2 (profiles/r02z_C5_part2_all_sq.json): k_shade ran at
// 30 % lane utilisation and was bound by instruction issue — a handful of lanes per chunk computed five octaves of Perlin noise,
// an earth-map lookup or a Fresnel branch while the rest waited.  MODE 2: a chunk shades its cheap paths in line (constant
// textures, Lambertian / Metal / Isotropic / Emissive, sky or colour misses) with the expensive code compiled OUT of that loop,
// appends the slots of the expensive ones (expensive_shading) to a wave-private list in LDS, and whenever the list holds 64 the
// wave shades 64 expensive paths at once, every lane busy (their rays, states and hits are gathered again by slot).  Survivors
// of both kinds are compacted into the same output queue; every result is keyed by the path's id, so the order does not matter.
// MODE 1: the scene has no expensive material or environment at all (cornell, suzanne): the cheap loop alone, the smaller kernel.
// MODE 0: everything in line (FIREWORK_NO_SHADE_DEFER=1: the A/B baseline, round 2's kernel).
template <int LDS_TAB, int MODE, bool CHAIN>   // CHAIN: 8-byte state (load_state_chain);  LDS_TAB 1: object + material + texture tables staged in LDS; 2: materials + textures only (part2: 1 409 objects are 135 KB, its 10 materials are not); 0: none
// 5 waves per SIMD (96 VGPRs, no spills) instead of the compiler's 4 (114): nothing while the scattered zero deposits bound
// the kernel (round 1), now cornell k_shade 20.55 -> 20.13 ms (four interleaved pairs), hdri 5.84 -> 5.56, suzanne 4.74 -> 4.57;
// 6 waves (80 VGPRs) spill three registers and gain nothing more.  Requesting the queue entries TWO chunks ahead (11 more
// registers; at 4 or at 5 waves) changes nothing that four interleaved runs can resolve (the kernel's own run-to-run spread
// is +-1 ms), nor does capping random_in_unit_sphere at one attempt: neither the prefetch distance nor that loop bounds it.
#ifndef FW_SHADE_WAVES
#define FW_SHADE_WAVES 5
#endif
#ifndef FW_SHADE_PIPE
#define FW_SHADE_PIPE 0            // the vmcnt-aware order of k_shade's loop (gathers, then prefetch, then shading, then stores by every lane)
#endif
#ifndef FW_SHADE_STATIC_STORES
#define FW_SHADE_STATIC_STORES FW_SHADE_PIPE
#endif
__attribute__((amdgpu_waves_per_eu(FW_SHADE_WAVES, 8)))
__global__ __launch_bounds__(WB) void k_shade(DScene sc, DFrame f, DPaths in, DPaths out, const float2 *__restrict__ hits,
                                                 float4 *__restrict__ sample_rad, DQueue q, int segment,
                                                 uint32_t n_mat, uint32_t n_tex) {
    const uint32_t w = wave_index(), lane = threadIdx.x & 63u;
    const float4 *objp = sc.obj, *matp = sc.mat, *texp = sc.tex;
    __shared__ uint16_t shade_list[128];                                 // MODE 2: queue positions of the expensive paths not yet shaded (< 64 + 64)
    if (sc.has_perlin && MODE != 1 && !CHAIN) { stage_perm(); if (!LDS_TAB) __syncthreads(); }
    if (LDS_TAB) {
        const uint32_t no = LDS_TAB == 1 ? sc.n_objects * OBJ_Q : 0u, nm = 2 * n_mat, nt = 2 * n_tex;
        if (LDS_TAB == 1) for (uint32_t k = threadIdx.x; k < no; k += WB) lds_tables[k] = sc.obj[k];
        for (uint32_t k = threadIdx.x; k < nm; k += WB) lds_tables[no + k] = sc.mat[k];
        for (uint32_t k = threadIdx.x; k < nt; k += WB) lds_tables[no + nm + k] = sc.tex[k];
        __syncthreads();
        if (LDS_TAB == 1) objp = lds_tables;
        matp = lds_tables + no; texp = lds_tables + no + nm;
    }
    if (w >= q.n_waves) return;
    const uint32_t n = q.wcount[(size_t)segment * q.n_waves + w];
    const uint32_t base = w * q.cap;
    uint32_t out_n = 0;                                                  // survivors written so far (wave-uniform)
    uint32_t list_n = 0;                                                 // MODE 2: entries of shade_list (wave-uniform)
    PH_DECL;
// ---- K7: compaction inside the wave's private queue: ballot -> mbcnt prefix -> dense stores --------
    auto compact = [&](bool alive, const Ray &nr, V3 nbeta, uint32_t nchain, uint32_t path_id, uint32_t c0 = 0xffffffffu) {
        const unsigned long long mask = __ballot(alive);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
#if FW_SHADE_STATIC_STORES
        // Round 5: EVERY lane stores, the dead ones into slots of the wave's own output window that no survivor of this chunk can take and
        // that a later chunk overwrites or nobody reads (position c0 + 64 + lane >= out_n + 64, clamped to the window's last slot, which a
        // survivor reaches only when all 64 lanes survive).  Why: on gfx9 loads and stores share ONE in-order counter (vmcnt), and with the
        // stores behind `if (alive)` — a branch the wave may skip — the compiler cannot know how many of them follow the next chunk's
        // prefetch loads, so the loop-top wait for those loads was `s_waitcnt vmcnt(0)`: every chunk of every wave waited for the L2's
        // acknowledgement of its own stores (54 % of k_shade's wave time in s_waitcnt, profiles/r04z_c2_sq.json).  With a static store count
        // the wait is vmcnt(3): the loads, nothing younger.
        if (c0 != 0xffffffffu) {
#if FW_SHADE_STATIC_STORES == 2     // timing variant: every dead lane into ONE slot (the window's last: no survivor takes it while any lane died) — the waits without the traffic
            const uint32_t dst = base + (alive ? out_n + rank : q.cap - 1u);
#else
            const uint32_t dst = base + (alive ? out_n + rank : min(c0 + 64u + lane, q.cap - 1u));
#endif
            qst(&out.ray_a[dst], make_float4(nr.o.x, nr.o.y, nr.o.z, nr.d.x));
            qst(&out.ray_b[dst], make_float2(nr.d.y, nr.d.z));
            if (CHAIN) qst(&reinterpret_cast<float2 *>(out.state)[dst], make_float2(__uint_as_float(nchain), __uint_as_float(path_id)));
            else qst(&out.state[dst], make_float4(nbeta.x, nbeta.y, nbeta.z, __uint_as_float(path_id)));
        } else
#endif
        if (alive) {
            const uint32_t dst = base + out_n + rank;
            qst(&out.ray_a[dst], make_float4(nr.o.x, nr.o.y, nr.o.z, nr.d.x));
            qst(&out.ray_b[dst], make_float2(nr.d.y, nr.d.z));
            if (CHAIN) qst(&reinterpret_cast<float2 *>(out.state)[dst], make_float2(__uint_as_float(nchain), __uint_as_float(path_id)));
            else qst(&out.state[dst], make_float4(nbeta.x, nbeta.y, nbeta.z, __uint_as_float(path_id)));
        }
        if (f.ex.mode)     // a new ray whose result depends on traversal order: k_extend_exact walks it literally (DExact)
            flag_exact(f.ex, alive && needs_exact(f.ex, nr.o.x, nr.o.y, nr.o.z, nr.d.x, nr.d.y, nr.d.z), base + out_n + rank, segment + 1);
        out_n += (uint32_t)__popcll(mask);
    };
    auto load_st = [&](uint32_t idx) { return CHAIN ? load_state_chain(in, idx, segment) : load_state(in, idx, segment); };
    auto load_hit = [&](uint32_t idx) { return f.hit4 ? make_float2(0.f, __uint_as_float(reinterpret_cast<const uint32_t *>(hits)[idx])) : qld(&hits[idx]); };
    // MODE 2: the last `take` listed paths, one per lane, with the full shading code
    auto run_list = [&](uint32_t take) {
        bool alive = false;
        Ray nr{mk(0, 0, 0), mk(0, 0, 0)}; V3 nbeta = mk(0, 0, 0); uint32_t path_id = 0, nchain = 0;
        if (lane < take) {
            const uint32_t i = base + shade_list[list_n - take + lane];
            const float4 ra = qld(&in.ray_a[i]), st = load_st(i); const float2 rb = load_ray_b(in, i, f, segment), hr = load_hit(i);
            path_id = __float_as_uint(st.w);
            alive = shade_path<false, CHAIN>(sc, f, objp, matp, texp, make_ray(ra, rb, f, segment), mk(st.x, st.y, st.z), __float_as_uint(st.x), path_id, hr.x, __float_as_uint(hr.y), segment, sample_rad, nr, nbeta, nchain PH_PASS);
        }
        list_n -= take;
        compact(alive, nr, nbeta, nchain, path_id);
    };
    // software pipeline: next chunk's ray / state / hit are in flight while the current chunk is shaded
    float4 ra_n = make_float4(0, 0, 0, 0), st_n = ra_n; float2 rb_n = make_float2(0, 0), hr_n = rb_n;
#if FW_SHADE_PIPE
    // Round 5: what the prefetch really needs on gfx9.  Loads and stores share ONE in-order counter (vmcnt): waiting for a load means
    // waiting for everything issued before it, and a wait's immediate is the number of YOUNGER operations that may stay outstanding — which
    // the compiler can only use when that number is the same on every path.  The loop used to issue the next chunk's loads at its top
    // behind `if (j + 64 < n)`, gather the path's pixel id (key_of) in the middle and store the survivors behind `if (alive)`: the gather's
    // wait (vmcnt(0): in-order) completed the prefetch a few hundred instructions after it was issued, and the next top's wait (vmcnt(0):
    // unknown store count) completed the stores — every chunk of every wave sat out an HBM round trip and an L2 write acknowledgement
    // (54 % of the wave time in s_waitcnt at 0.79 instruction issue, profiles/r04z_c2_sq.json; fewer instructions or a deeper prefetch
    // changed nothing: DESIGN §5 rounds 1-4).  Now: (A) the chunk's dependent gathers are issued FIRST (the pixel id), (B) then the next
    // chunk's loads, always the same four instructions (indices clamped to the queue's last entry, and where a stream is not read at all —
    // segment-0 state, ray_b of pinhole camera rays — one hot address stands in), (C) the shading, (D) the stores, every lane
    // (FW_SHADE_STATIC_STORES).  The waits become vmcnt(4) for the gather and vmcnt(3) at the top.
    auto prefetch = [&](uint32_t jn) {       // jn: queue position, clamped by the caller
        const uint32_t in_ = base + jn;
        ra_n = qld(&in.ray_a[in_]);
        rb_n = qld(&in.ray_b[short_rays(f, segment) ? base : in_]);
        if (CHAIN) { const float2 v = qld(&reinterpret_cast<const float2 *>(in.state)[segment == 0 ? base : in_]); st_n = make_float4(v.x, 0.f, 0.f, v.y); }
        else st_n = qld(&in.state[segment == 0 ? base : in_]);
        hr_n = load_hit(in_);
    };
    auto fix_implicit = [&](float4 &st_, float2 &rb_, uint32_t slot) {      // the streams that are not stored at segment 0 (load_state*, load_ray_b)
        if (segment == 0) st_ = CHAIN ? make_float4(0.f, 0.f, 0.f, __uint_as_float(slot)) : make_float4(1.f, 1.f, 1.f, __uint_as_float(slot));
        if (short_rays(f, segment)) rb_ = make_float2(0.f, 0.f);
    };
    if (n) {
        prefetch(min(lane, n - 1u));
        // ... and the loop is entered with the same operations behind the first chunk's loads as every later trip has behind its own: three
        // stores (here into the first 64 output slots, which chunk 0's survivors overwrite or nobody reads) — the wait at the top of the loop
        // is compiled once, for the entry and for the back edge, with the smaller of the two counts
        const uint32_t d0 = base + lane;
        qst(&out.ray_a[d0], make_float4(0.f, 0.f, 0.f, 0.f));
        qst(&out.ray_b[d0], make_float2(0.f, 0.f));
        if (CHAIN) qst(&reinterpret_cast<float2 *>(out.state)[d0], make_float2(0.f, 0.f)); else qst(&out.state[d0], make_float4(0.f, 0.f, 0.f, 0.f));
    }
#else
    if (lane < n) { ra_n = qld(&in.ray_a[base + lane]); rb_n = load_ray_b(in, base + lane, f, segment); st_n = load_st(base + lane); hr_n = load_hit(base + lane); }
#endif
    for (uint32_t c0 = 0; c0 < n; c0 += 64u) {
        const uint32_t j = c0 + lane;
        const uint32_t i = base + j;
        float4 ra = ra_n, st = st_n; float2 rb = rb_n, hr = hr_n;
#if FW_SHADE_PIPE
        fix_implicit(st, rb, i);
        const RngKey key_ = key_of(f, __float_as_uint(st.w));        // (A) the chunk's own gather, before (B) the prefetch
        prefetch(min(j + 64u, n - 1u));
        const RngKey *pre_key = &key_;
#else
        if (j + 64u < n) { ra_n = qld(&in.ray_a[i + 64u]); rb_n = load_ray_b(in, i + 64u, f, segment); st_n = load_st(i + 64u); hr_n = load_hit(i + 64u); }
        const RngKey *pre_key = nullptr;
#endif
        bool alive = false, later = false;
        Ray nr{mk(0, 0, 0), mk(0, 0, 0)}; V3 nbeta = mk(0, 0, 0); uint32_t path_id = 0, nchain = 0;
        if (j < n) {
            PH_T0;
            Ray r = make_ray(ra, rb, f, segment);
            V3 beta = mk(st.x, st.y, st.z);
            path_id = __float_as_uint(st.w);
            const uint32_t hit_code = __float_as_uint(hr.y);
#ifdef FW_ABL_SHADE_COPY     // timing ablation (wrong frames): the queue streams and the compaction alone — every hit path survives unchanged
            alive = hit_code != MISS && segment < 10; nr = r; nbeta = beta; nchain = __float_as_uint(st.x) + (pre_key ? pre_key->pixel & 1u : 0u);
#else
            if (MODE == 2 && expensive_shading(sc, objp, matp, hit_code)) later = true;
            else alive = shade_path<MODE != 0, CHAIN>(sc, f, objp, matp, texp, r, beta, __float_as_uint(st.x), path_id, hr.x, hit_code, segment, sample_rad, nr, nbeta, nchain PH_PASS, pre_key);
#endif
            PH_ADD(0);
        }
        { PH_T0; compact(alive, nr, nbeta, nchain, path_id, c0); PH_ADD(7); }
        if (MODE == 2) {
            const unsigned long long lm = __ballot(later);
            if (lm) {
                const uint32_t lr = __builtin_amdgcn_mbcnt_hi((uint32_t)(lm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lm, 0u));
                if (later) shade_list[list_n + lr] = (uint16_t)j;
                list_n += (uint32_t)__popcll(lm);
                if (list_n >= 64u) run_list(64u);
            }
        }
    }
    if (MODE == 2) while (list_n) run_list(min(list_n, 64u));
    if (lane == 0) q.wcount[(size_t)(segment + 1) * q.n_waves + w] = out_n;
    PH_FLUSH(1);
}

// ------------------------------------------------------------------------------------------------
// K-exact  k_extend_exact: the rays whose result depends on how the trees are walked (DExact), walked the reference's way.
//
// One launch per segment, after the ordinary k_extend (and k_blas) of the segment and before its k_shade: for every slot on the
// segment's list, closest_hit_exact over the reference's own trees, the hit record written over the ordinary one.  The rays stay
// in the wavefront — queue, shading, compaction and ray counts are everybody's — and a path pays for the literal walk only in
// the segments that need it (one in 2.75 on suzanne).
// A walk without culling is long (~150 us for a lane, whose wave alternates between node steps and triangle tests), and a launch
// lasts as long as its longest wave.  So a SHORT list (most of them: a few hundred rays per segment beyond the camera rays) is
// walked one ray per wave, the lanes sharing the walk (hit_mesh_exact_wave, ~25-50 us), and only a list too long for that
// (the flagged camera rays of a big batch) one ray per lane.
// (Earlier forms this round: a scan kernel + an exact kernel per segment, one ray per lane: suzanne @64 spp 9.7 -> 14.7 ms for
// 0.07 % of its rays.  Then the flagged paths LEFT the wavefront and one kernel per batch finished them, first one path per
// lane, then one per wave (k_exact_paths): 2.6 ms per batch either way — 64 different paths in a wave wait for each other, and
// one path per wave does every ordinary walk and every shading 64 times over: 9.7 -> 14.3 ms, suzanne at full size 71 -> 81-83 ms.)
// Dynamic LDS: [TLAS stacks: tlas_levels x 64 u32][BLAS stacks: max(blas_levels x 64, EXACT_STACK) u32].
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(EXACT_WB) void k_extend_exact(DScene sc, DFrame f, DPaths in, float2 *__restrict__ hits, int segment, int use_bvh, uint32_t tlas_levels, int form) {
    extern __shared__ float4 lds_exact[];
    const uint32_t n = min(f.ex.count[segment], f.ex.cap);
    if (n == 0u) return;
    const uint32_t *__restrict__ list = f.ex.slots[segment & 1];
    const uint32_t lane = threadIdx.x;                                  // a workgroup is one wave
    uint32_t *tlas_stack = reinterpret_cast<uint32_t *>(lds_exact) + lane, *blas_stack = tlas_stack + (size_t)tlas_levels * EXACT_WB;
    auto one = [&](uint32_t slot, auto form) {
        const Ray r = make_ray(qld(&in.ray_a[slot]), load_ray_b(in, slot, f, segment), f, segment);
        RngKey key{0, 0, 0};
        if (sc.has_medium) key = key_of(f, __float_as_uint(load_home(in, slot, f, segment)));
        float t = 2e9f; uint32_t obj = MISS, prim = 0;
        closest_hit_exact<decltype(form)::value>(sc, r, key, segment, use_bvh != 0, tlas_stack, blas_stack, t, obj, prim);
        const uint32_t code = obj == MISS ? MISS : ((obj << sc.prim_bits) | prim);
        if (decltype(form)::value == 1 || lane == 0u) {
            if (f.hit4) reinterpret_cast<uint32_t *>(hits)[slot] = code; else qst(&hits[slot], make_float2(t, __uint_as_float(code)));
        }
    };
    if (form == 2 || (form == 0 && n <= EXACT_WAVE_RAYS * gridDim.x)) {      // wave-uniform, the same for every wave of the launch (form: FIREWORK_EXACT_FORM, tests)
        for (uint32_t k = blockIdx.x; k < n; k += gridDim.x) one(list[k], std::integral_constant<int, 2>{});
    } else {
        for (uint32_t k = blockIdx.x * EXACT_WB + lane; k < n; k += gridDim.x * EXACT_WB) one(list[k], std::integral_constant<int, 1>{});
    }
}

#if FW_AB
// ------------------------------------------------------------------------------------------------
// K2 + K5 + K7 in one launch per segment ("bounce"): every lane intersects its ray and shades the hit while it
// is still in registers, so the hit record and the second read of the ray never touch HBM (80 B per ray
// instead of 120) and the VALU-bound intersection of one wave overlaps the HBM-bound state streaming of the
// others on the same CU.  Same device functions as k_extend / k_shade, hence the same bits.  Not used when a
// TLAS holds meshes (those rays are parked and finish out of order, see k_extend).
// Dynamic LDS: [tables (LDS_TAB)] [traversal stacks (USE_BVH)].
// ------------------------------------------------------------------------------------------------
template <bool USE_BVH, bool LDS_TAB>
__global__ __launch_bounds__(WB) void k_bounce(DScene sc, DFrame f, DPaths in, DPaths out, float4 *__restrict__ sample_rad,
                                                  DQueue q, int segment, int tlas_levels, uint32_t n_mat, uint32_t n_tex,
                                                  uint32_t table_quads) {
    const uint32_t w = wave_index(), lane = threadIdx.x & 63u;
    const float4 *objp = sc.obj, *matp = sc.mat, *texp = sc.tex;
    if (sc.has_perlin) { stage_perm(); if (!LDS_TAB) __syncthreads(); }
    if (LDS_TAB) {
        const uint32_t no = sc.n_objects * OBJ_Q, nm = 2 * n_mat, nt = 2 * n_tex;
        for (uint32_t k = threadIdx.x; k < no; k += WB) lds_tables[k] = sc.obj[k];
        for (uint32_t k = threadIdx.x; k < nm; k += WB) lds_tables[no + k] = sc.mat[k];
        for (uint32_t k = threadIdx.x; k < nt; k += WB) lds_tables[no + nm + k] = sc.tex[k];
        __syncthreads();
        objp = lds_tables; matp = lds_tables + no; texp = lds_tables + no + nm;
    }
    if (w >= q.n_waves) return;
    uint32_t *my_stack = reinterpret_cast<uint32_t *>(lds_tables + table_quads) + threadIdx.x;   // [level][lane]
    uint32_t *blas_stack = my_stack + (size_t)tlas_levels * WB;
    const uint32_t n = q.wcount[(size_t)segment * q.n_waves + w];
    const uint32_t base = w * q.cap;
    uint32_t out_n = 0;
    float4 ra_n = make_float4(0, 0, 0, 0), st_n = ra_n; float2 rb_n = make_float2(0, 0);
    if (lane < n) { ra_n = qld(&in.ray_a[base + lane]); rb_n = load_ray_b(in, base + lane, f, segment); st_n = load_state(in, base + lane, segment); }
    for (uint32_t c0 = 0; c0 < n; c0 += 64u) {
        const uint32_t j = c0 + lane;
        const uint32_t i = base + j;
        float4 ra = ra_n, st = st_n; float2 rb = rb_n;
        if (j + 64u < n) { ra_n = qld(&in.ray_a[i + 64u]); rb_n = load_ray_b(in, i + 64u, f, segment); st_n = load_state(in, i + 64u, segment); }
        bool alive = false;
        Ray nr{mk(0, 0, 0), mk(0, 0, 0)}; V3 nbeta = mk(0, 0, 0); uint32_t path_id = 0;
        if (j < n) {
            Ray r = make_ray(ra, rb, f, segment);
            path_id = __float_as_uint(st.w);
            RngKey key{0, 0, 0};
            if (sc.has_medium) key = key_of(f, path_id);
            float best_t = 2e9f; uint32_t best_obj = MISS, best_prim = 0;
            bool deferred = false; uint32_t deferred_obj = 0;
            closest_hit<USE_BVH, false>(sc, r, key, segment, my_stack, blas_stack, best_t, best_obj, best_prim, deferred, deferred_obj,
                                        USE_BVH && sc.has_mesh && soft_ray(f.ex, r.d, sc.soft_shear));
            const uint32_t hit_code = best_obj == MISS ? MISS : ((best_obj << sc.prim_bits) | best_prim);
            uint32_t nchain = 0;
            PH_DECL;
            alive = shade_path(sc, f, objp, matp, texp, r, mk(st.x, st.y, st.z), 0u, path_id, best_t, hit_code, segment, sample_rad, nr, nbeta, nchain PH_PASS);
        }
        unsigned long long mask = __ballot(alive);
        uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        if (alive) {
            uint32_t dst = base + out_n + rank;
            qst(&out.ray_a[dst], make_float4(nr.o.x, nr.o.y, nr.o.z, nr.d.x));
            qst(&out.ray_b[dst], make_float2(nr.d.y, nr.d.z));
            qst(&out.state[dst], make_float4(nbeta.x, nbeta.y, nbeta.z, __uint_as_float(path_id)));
        }
        out_n += (uint32_t)__popcll(mask);
    }
    if (lane == 0) q.wcount[(size_t)(segment + 1) * q.n_waves + w] = out_n;
}

#endif   // FW_AB

// per-segment queue totals of one batch (ray statistics): totals[s] += sum_w wcount[s][w]; grid = (slices, segments + 1);
// the last row sums the waves' parked-ray counts (ptotal, may be null) into totals[MAX_SEGMENTS]
__global__ __launch_bounds__(BLOCK) void k_queue_totals(DQueue q, uint32_t *totals, const uint32_t *ptotal) {
    __shared__ uint32_t part[BLOCK];
    const uint32_t seg = blockIdx.y;
    if (seg == (uint32_t)MAX_SEGMENTS && !ptotal) return;
    const uint32_t *src = seg == (uint32_t)MAX_SEGMENTS ? ptotal : q.wcount + (size_t)seg * q.n_waves;
    uint32_t acc = 0;
    for (uint32_t w = blockIdx.x * BLOCK + threadIdx.x; w < q.n_waves; w += gridDim.x * BLOCK) acc += src[w];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t s2 = BLOCK / 2; s2 > 0; s2 >>= 1) { if (threadIdx.x < s2) part[threadIdx.x] += part[threadIdx.x + s2]; __syncthreads(); }
    if (threadIdx.x == 0 && part[0]) atomicAdd(&totals[seg], part[0]);
}
// FW_FLAG_COUNT_DEPOSITS: how many radiance records k_shade wrote in this batch = the bits set in dep_bits.
__global__ __launch_bounds__(BLOCK) void k_count_deposits(const uint32_t *__restrict__ dep_bits, uint32_t n_words, uint32_t *total) {
    __shared__ uint32_t part[BLOCK];
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n_words; i += gridDim.x * BLOCK) acc += (uint32_t)__popc(dep_bits[i]);
    part[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t s2 = BLOCK / 2; s2 > 0; s2 >>= 1) { if (threadIdx.x < s2) part[threadIdx.x] += part[threadIdx.x + s2]; __syncthreads(); }
    if (threadIdx.x == 0 && part[0]) atomicAdd(total, part[0]);
}
// ------------------------------------------------------------------------------------------------
// K8  accumulate (sample order, deterministic) and resolve
// ------------------------------------------------------------------------------------------------
// One thread per pixel sums its samples in sample order (the reference's order; float addition does not reassociate).
// Single-wave workgroups and 16 loads in flight per thread keep a 1/8-frame tile share (32 Ki pixels = 512 waves for
// 1024 SIMDs) near the HBM rate: 0.49 -> see DESIGN §7.
__global__ __launch_bounds__(WB) void k_accumulate(DFrame f, const float4 *__restrict__ sample_rad, float4 *__restrict__ accum) {
    // linear id of (sample s, pixel p) = s * n_pixels + p = (c * n_waves + w) * 64 + lane -> home = w * cap + c * 64 + lane;
    // one sample further adds n_pixels = 64 * A + B to the linear id: (w, c, lane) are advanced without divisions
    const uint32_t A = f.n_pixels >> 6, B = f.n_pixels & 63u, A_div = A / f.q_n_waves, A_mod = A % f.q_n_waves;
    for (uint32_t p = blockIdx.x * WB + threadIdx.x; p < f.n_pixels; p += gridDim.x * WB) {
        float4 a = accum[p];
        uint32_t lane = p & 63u, c = (p >> 6) / f.q_n_waves, w = (p >> 6) % f.q_n_waves;
        auto home_then_advance = [&]() {
            const uint32_t home = (w << (f.q_shift + 6u)) | (c << 6) | lane;
            lane += B;
            const uint32_t carry = lane >> 6; lane &= 63u;
            w += A_mod + carry; c += A_div;
            if (w >= f.q_n_waves) { w -= f.q_n_waves; c++; }
            return home;
        };
        uint32_t s = 0;
        if (f.skip_zero_deposits) {
            // black environment: only the slots whose bit is set hold a record (k_shade), all others contribute an exact +0:
            // 1 bit instead of 16 bytes per sample is read, and nobody had to write the zeros
            if (!f.dep_pixel_major) {     // slot-major bits (whole frames): the word of sample s is that of its home slot
                for (; s + 16u <= f.spp_batch; s += 16u) {
                    uint32_t h[16], bw[16];
#pragma unroll
                    for (int k = 0; k < 16; k++) { h[k] = home_then_advance(); bw[k] = f.dep_bits[h[k] >> 5]; }
#pragma unroll
                    for (int k = 0; k < 16; k++) if ((bw[k] >> (h[k] & 31u)) & 1u) { const float4 v = sample_rad[h[k]]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
                }
                for (; s < f.spp_batch; s++) {
                    const uint32_t h = home_then_advance();
                    if ((f.dep_bits[h >> 5] >> (h & 31u)) & 1u) { const float4 v = sample_rad[h]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
                }
                accum[p] = a;
                continue;
            }
            // pixel-major bits (dep_bit_of): the pixel's bits are [p * spp, (p + 1) * spp) — eight words per round trip, then
            // one record per set bit, in sample order
            const uint32_t b0 = p * f.spp_batch, b1 = b0 + f.spp_batch, j_last = (b1 - 1u) >> 5;
            for (uint32_t j0 = b0 >> 5; j0 <= j_last; j0 += 8u) {
                uint32_t wd[8];
#pragma unroll
                for (uint32_t k = 0; k < 8u; k++) wd[k] = (j0 + k <= j_last) ? f.dep_bits[j0 + k] : 0u;
#pragma unroll
                for (uint32_t k = 0; k < 8u; k++) {
                    const uint32_t j = j0 + k;
                    uint32_t word = wd[k];
                    if (j == (b0 >> 5)) word &= ~0u << (b0 & 31u);
                    if (j == (b1 >> 5)) word &= (1u << (b1 & 31u)) - 1u;
                    while (word) {
                        const uint32_t bit = (uint32_t)__ffs((int)word) - 1u;
                        word &= word - 1u;
                        const uint32_t lin = (j * 32u + bit - b0) * f.n_pixels + p;       // s_local * n_pixels + p
                        const uint32_t g = lin >> 6, cc = g / f.q_n_waves, ww = g - cc * f.q_n_waves;
                        const float4 v = sample_rad[(ww << (f.q_shift + 6u)) | (cc << 6) | (lin & 63u)];
                        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
                    }
                }
            }
            accum[p] = a;
            continue;
        }
        for (; s + 16u <= f.spp_batch; s += 16u) {
            float4 v[16];
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = sample_rad[home_then_advance()];
#pragma unroll
            for (int k = 0; k < 16; k++) { a.x += v[k].x; a.y += v[k].y; a.z += v[k].z; a.w += v[k].w; }   // render.rs:181: total_color += color(...)
        }
        for (; s < f.spp_batch; s++) {
            float4 v = sample_rad[home_then_advance()];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        accum[p] = a;
    }
}

// A whole frame is traced in 16x16-tile order, not row order: the 64 paths of a chunk are then a 16x4 block of pixels whose
// camera rays hit the same objects (k_extend_linear's segment-0 box pre-test skips more, every kernel diverges less):
// cornell 512x512@1024 51 -> 45 ms against 64 consecutive pixels of one row.  ids[i] = the i-th pixel in that order
// (tiles row-major, pixels row-major inside a tile; the last tile row / column may be narrower).
constexpr uint32_t TILE = 16;
__global__ __launch_bounds__(BLOCK) void k_tile_order(uint32_t W, uint32_t H, uint32_t *__restrict__ ids) {
    const uint32_t n = W * H, tx = (W + TILE - 1) / TILE;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const uint32_t r = i / (W * TILE), j = i - r * W * TILE;           // tile row; offset inside it
        const uint32_t hr = min(TILE, H - r * TILE);                        // its height
        const uint32_t c = min(j / (TILE * hr), tx - 1u), k = j - c * TILE * hr;
        const uint32_t wc = min(TILE, W - c * TILE);                        // the tile's width
        const uint32_t y = k / wc, x = k - y * wc;
        ids[i] = (r * TILE + y) * W + c * TILE + x;
    }
}
void launch_tile_order(hipStream_t stream, uint32_t width, uint32_t height, uint32_t *ids) {
    const uint32_t n = width * height;
    hipLaunchKernelGGL(k_tile_order, dim3(std::min<uint32_t>((n + BLOCK - 1) / BLOCK, 4096u)), dim3(BLOCK), 0, stream, width, height, ids);
}
// Scene upload: a kernel reads the staging blob from pinned host memory.  The copy engines are kept out of the one-shot path:
// a 2.5 KB hipMemcpy(Async) + stream synchronisation there stalled for 10-30 ms every few calls (tools/oneshot.py with
// FIREWORK_TRACE=1), and a kernel is stream-ordered with the render that follows, so nothing has to wait on the host.
__global__ __launch_bounds__(BLOCK) void k_upload(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n16; i += (size_t)gridDim.x * BLOCK) dst[i] = src[i];
}
void launch_upload(hipStream_t stream, const void *pinned_src, void *dst, size_t bytes) {
    const size_t n16 = bytes / 16;
    hipLaunchKernelGGL(k_upload, dim3((unsigned)std::min<size_t>((n16 + BLOCK - 1) / BLOCK, 8192)), dim3(BLOCK), 0, stream,
                       (const uint4 *)pinned_src, (uint4 *)dst, n16);
}

// fw_render_scene_tiled's device-side gather: the tiles of every device, concatenated on the first one, go to their pixels
__global__ __launch_bounds__(BLOCK) void k_scatter_tiles(const uint32_t *__restrict__ ids, uint32_t n, const uint8_t *__restrict__ in8,
                                                         const float *__restrict__ ing, const float *__restrict__ inl,
                                                         uint8_t *__restrict__ out8, float *__restrict__ outg, float *__restrict__ outl) {
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const size_t s = 3 * (size_t)i, d = 3 * (size_t)ids[i];
        if (in8) { out8[d] = in8[s]; out8[d + 1] = in8[s + 1]; out8[d + 2] = in8[s + 2]; }
        if (ing) { outg[d] = ing[s]; outg[d + 1] = ing[s + 1]; outg[d + 2] = ing[s + 2]; }
        if (inl) { outl[d] = inl[s]; outl[d + 1] = inl[s + 1]; outl[d + 2] = inl[s + 2]; }
    }
}
void launch_scatter_tiles(hipStream_t stream, const uint32_t *ids, uint32_t n, const uint8_t *in8, const float *ing, const float *inl,
                          uint8_t *out8, float *outg, float *outl) {
    hipLaunchKernelGGL(k_scatter_tiles, dim3(std::min<uint32_t>((n + BLOCK - 1) / BLOCK, 4096u)), dim3(BLOCK), 0, stream, ids, n, in8, ing, inl, out8, outg, outl);
}

__device__ __forceinline__ uint8_t sat_u8(float f) { if (!(f > 0.f)) return 0; if (f >= 255.f) return 255; return (uint8_t)f; }

__global__ __launch_bounds__(BLOCK) void k_resolve(DFrame f, const float4 *__restrict__ accum, uint32_t total_spp, float gamma,
                                                   uint8_t *rgb8, float *gamma_rgb, float *linear_rgb) {
    for (uint32_t q = blockIdx.x * BLOCK + threadIdx.x; q < f.n_pixels; q += gridDim.x * BLOCK) {
        float4 a = accum[q];
        const uint32_t p = f.scatter_out ? f.pixel_ids[q] : q;                       // output index (the library's tile order is undone here)
        V3 total = mk(a.x, a.y, a.z) / (float)total_spp;                             // render.rs:184
        float ig = fdiv(1.f, gamma);
        V3 g = mk(fwlm::powf_glibc(total.x, ig), fwlm::powf_glibc(total.y, ig), fwlm::powf_glibc(total.z, ig));           // render.rs:186
        auto clamp01 = [](float x) { return (x != x) ? x : (x < 0.f ? 0.f : (x > 1.f ? 1.f : x)); };
        g = mk(clamp01(g.x), clamp01(g.y), clamp01(g.z));                             // render.rs:187
        if (linear_rgb) { linear_rgb[3 * (size_t)p] = total.x; linear_rgb[3 * (size_t)p + 1] = total.y; linear_rgb[3 * (size_t)p + 2] = total.z; }
        if (gamma_rgb) { gamma_rgb[3 * (size_t)p] = g.x; gamma_rgb[3 * (size_t)p + 1] = g.y; gamma_rgb[3 * (size_t)p + 2] = g.z; }
        if (rgb8) { rgb8[3 * (size_t)p] = sat_u8(g.x * 255.99f); rgb8[3 * (size_t)p + 1] = sat_u8(g.y * 255.99f); rgb8[3 * (size_t)p + 2] = sat_u8(g.z * 255.99f); }   // util.rs:14-23
    }
}

// ------------------------------------------------------------------------------------------------
// self-test: fdiv / fsqrt against the compiler's IEEE expansion, bit for bit, on hashed operands.
// mode 0: magnitudes 2^-40 .. 2^40 (what a renderer produces) — must agree exactly.
// mode 1: the whole float range incl. zeros, infinities, NaNs, denormals — counted, reported, not required.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_selftest_arith(uint32_t n, uint32_t seed, int mode, unsigned long long *out) {
    unsigned long long bad_div = 0, bad_sqrt = 0;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        uint4 h = pcg4d(make_uint4(i, seed, 0x51u, 0xA5A5u));
        float a, b;
        if (mode == 0) {
            uint32_t ea = 127u - 40u + (h.z % 81u), eb = 127u - 40u + ((h.z >> 8) % 81u);
            a = __uint_as_float((h.x & 0x807fffffu) | (ea << 23));
            b = __uint_as_float((h.y & 0x807fffffu) | (eb << 23));
        } else { a = __uint_as_float(h.x); b = __uint_as_float(h.y); }
        float q1 = fdiv(a, b), q0 = a / b;
        bool same = __float_as_uint(q0) == __float_as_uint(q1) || (q0 != q0 && q1 != q1);
        bad_div += same ? 0 : 1;
        float x = fabsf(a);
        float s1 = fsqrt(x), s0 = sqrtf(x);
        same = __float_as_uint(s0) == __float_as_uint(s1) || (s0 != s0 && s1 != s1);
        bad_sqrt += same ? 0 : 1;
    }
    if (bad_div) atomicAdd(&out[0], bad_div);
    if (bad_sqrt) atomicAdd(&out[1], bad_sqrt);
}
void launch_selftest_arith(hipStream_t stream, uint32_t n, uint32_t seed, int mode, unsigned long long *out) {
    hipLaunchKernelGGL(k_selftest_arith, dim3(1024), dim3(BLOCK), 0, stream, n, seed, mode, out);
}
// self-test: the libm-class functions of fw_libm.h evaluated on the device, element-wise: out[i] = fn(x[i] [, y[i]]).
// fn: 0 logf 1 log10f 2 sinf 3 asinf 4 acosf 5 atanf 6 atan2f(x = y-argument, y = x-argument) 7 powf(x, y)
__global__ __launch_bounds__(BLOCK) void k_selftest_libm(int fn, uint32_t n, const float *__restrict__ x, const float *__restrict__ y, float *__restrict__ out) {
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const float a = x[i], b = y ? y[i] : 0.f;
        float r;
        switch (fn) {
        case 0: r = fwlm::logf_glibc(a); break;
        case 1: r = fwlm::log10f_glibc(a); break;
        case 2: r = fwlm::sinf_glibc(a); break;
        case 3: r = fwlm::asinf_glibc(a); break;
        case 4: r = fwlm::acosf_glibc(a); break;
        case 5: r = fwlm::atanf_glibc(a); break;
        case 6: r = fwlm::atan2f_glibc(a, b); break;
        default: r = fwlm::powf_glibc(a, b); break;
        }
        out[i] = r;
    }
}
void launch_selftest_libm(hipStream_t stream, int fn, uint32_t n, const float *x, const float *y, float *out) {
    hipLaunchKernelGGL(k_selftest_libm, dim3(2048), dim3(BLOCK), 0, stream, fn, n, x, y, out);
}

// ------------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------------
static dim3 wave_grid(const LaunchCfg &c) { return dim3((c.q.n_waves + WB / 64 - 1) / (WB / 64)); }

void launch_raygen(const LaunchCfg &c, const DCamera &cam, const DFrame &f, DPaths out, float4 *sample_rad, uint32_t n_paths) {
    hipLaunchKernelGGL(k_raygen, wave_grid(c), dim3(WB), 0, c.stream, cam, f, out, sample_rad, c.q, n_paths);
}
// More than 64 KB of dynamic LDS has to be allowed per kernel AND per device (a process may drive several: fw_render_scene_tiled):
// true the first time kernel group `which` is about to be launched on the current device.
static bool lds_attr_needed(int which) {
    static std::atomic<unsigned char> done[4][64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
    return done[which][dev].exchange(1) == 0;
}
// LDS of an LDS-resident walk with `waves` waves per workgroup: tree (+ triangles) + 16-bit stacks + counter + the workgroup's queue counts
static size_t lds_walk_bytes(size_t tree_bytes, uint32_t waves, uint32_t levels, uint32_t n_queues, uint32_t grid) {
    return tree_bytes + (size_t)waves * levels * 64 * 2 + 16 + 4 * (size_t)((n_queues + grid - 1) / grid);
}
void launch_extend(const LaunchCfg &c, const DScene &sc, const DFrame &f, DPaths in, float2 *hits, int segment, bool use_bvh, DPark park) {
    int tl = use_bvh ? c.tlas_depth + 1 : 0;
    int levels = tl + c.blas_depth + 1;
    size_t lds = (size_t)levels * WB * sizeof(uint32_t) + (use_bvh && c.has_mesh ? 5 * DEFER_CAP * sizeof(uint32_t) : 0);
    dim3 eg = wave_grid(c);
    const dim3 sg = eg;
    auto walk_grid = [&](uint32_t waves) { return std::min<uint32_t>((uint32_t)c.n_cus, (c.q.n_waves + waves - 1) / waves); };
    if (use_bvh && c.tlas_refill && c.has_mesh) {
        // TLAS walk that parks mesh rays in HBM, then their BLAS walks; a medium around a mesh still walks it in place (blas levels)
        if (sc.n_objects <= TLAS_SCAN_MAX && !sc.has_medium && c.simple_but_meshes) hipLaunchKernelGGL((k_extend_scan<true, true, true>), eg, dim3(WB), (size_t)levels * WB * sizeof(uint32_t), c.stream, sc, f, in, hits, c.q, segment, tl, park.ray_a, park.ray_b, park.meta, park.pcount, park.stride);
        else if (sc.n_objects <= TLAS_SCAN_MAX && !sc.has_medium) hipLaunchKernelGGL((k_extend_scan<true, true>), eg, dim3(WB), (size_t)levels * WB * sizeof(uint32_t), c.stream, sc, f, in, hits, c.q, segment, tl, park.ray_a, park.ray_b, park.meta, park.pcount, park.stride);
        else if (sc.n_objects <= TLAS_SCAN_MAX) hipLaunchKernelGGL(k_extend_scan<true>, eg, dim3(WB), (size_t)levels * WB * sizeof(uint32_t), c.stream, sc, f, in, hits, c.q, segment, tl, park.ray_a, park.ray_b, park.meta, park.pcount, park.stride);
        else hipLaunchKernelGGL(k_extend_tlas_park, sg, dim3(WB), (size_t)levels * WB * sizeof(uint32_t), c.stream, sc, f, in, hits, c.q, segment, tl, levels, park);
        // The parked rays' BLAS walks.  WIDE nodes out of LDS where the scene has them (f32, or quantised for a BLAS too big for those):
        // as many waves per workgroup (16, 12, 8) as fit next to the tree, the triangles too when 16 waves still fit with them.
        if (c.lds_trees && c.wblas_fmt != WIDE_NONE && sc.wblas && c.max_tris < 0x7fffu) {
            const uint32_t wl = c.debug_wide_levels ? c.debug_wide_levels : 3u * c.wblas_depth + 2u;
            const size_t tree = (size_t)c.wblas_nodes * (c.wblas_fmt == WIDE_F32 ? WIDE_F32_DW : WIDE_Q8_DW) * 4, tris = (size_t)c.n_tris * 48;
            uint32_t waves = 0; bool lds_tris = false;
            if (!c.no_lds_tris && lds_walk_bytes(tree + tris, 16, wl, c.q.n_waves, walk_grid(16)) <= LDS_TREE_LIMIT) { waves = 16; lds_tris = true; }
            else for (uint32_t w : {16u, 12u, 8u}) if (lds_walk_bytes(tree, w, wl, c.q.n_waves, walk_grid(w)) <= LDS_TREE_LIMIT) { waves = w; break; }
            if (waves) {
                if (lds_attr_needed(2)) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_blas_wide<WIDE_F32, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_TREE_LIMIT);
                    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_blas_wide<WIDE_F32, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_TREE_LIMIT);
                    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_blas_wide<WIDE_Q8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_TREE_LIMIT);
                    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_blas_wide<WIDE_Q8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_TREE_LIMIT);
                }
                const dim3 lg(walk_grid(waves));
                const size_t bytes = lds_walk_bytes(tree + (lds_tris ? tris : 0), waves, wl, c.q.n_waves, lg.x);
#define FW_BLAS_WIDE(F, T) hipLaunchKernelGGL((k_blas_wide<F, T>), lg, dim3(waves * 64), bytes, c.stream, sc, park, hits, c.q, c.wblas_nodes, c.n_tris, wl)
                if (c.wblas_fmt == WIDE_F32) { if (lds_tris) FW_BLAS_WIDE(WIDE_F32, true); else FW_BLAS_WIDE(WIDE_F32, false); }
                else { if (lds_tris) FW_BLAS_WIDE(WIDE_Q8, true); else FW_BLAS_WIDE(WIDE_Q8, false); }
#undef FW_BLAS_WIDE
                return;
            }
        }
        // pair nodes: the whole BLAS in LDS when it fits next to sixteen 16-bit stacks (k_blas_lds), else node fetches from L2 (k_blas)
        const uint32_t bl = (uint32_t)c.blas_depth + 1u;
        const size_t lds_blas = lds_walk_bytes((size_t)c.blas_pair_nodes * 64, LDS_WAVES, bl, c.q.n_waves, walk_grid(LDS_WAVES)), lds_tris = (size_t)c.n_tris * 48;
        if (c.lds_trees && c.blas_pair_nodes > 0 && c.blas_pair_nodes < 32768u && c.max_tris < 32768u && lds_blas <= LDS_TREE_LIMIT) {
            if (lds_attr_needed(0)) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_blas_lds<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_TREE_LIMIT);
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_blas_lds<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_TREE_LIMIT);
            }
            const dim3 lg(walk_grid(LDS_WAVES));
            if (lds_blas + lds_tris <= LDS_TREE_LIMIT && !c.no_lds_tris)      // the triangles too, when they fit as well
                hipLaunchKernelGGL(k_blas_lds<true>, lg, dim3(LDS_WAVES * 64), lds_blas + lds_tris, c.stream, sc, park, hits, c.q, c.blas_pair_nodes, c.n_tris, bl);
            else
                hipLaunchKernelGGL(k_blas_lds<false>, lg, dim3(LDS_WAVES * 64), lds_blas, c.stream, sc, park, hits, c.q, c.blas_pair_nodes, c.n_tris, bl);
        }
        else hipLaunchKernelGGL(k_blas, sg, dim3(WB), (size_t)(c.blas_depth + 1) * WB * sizeof(uint32_t), c.stream, sc, park, hits, c.q);
    }
    else if (use_bvh && c.tlas_refill) {
        // scenes without meshes: the whole TLAS in LDS when it fits next to the walks' stacks — WIDE nodes where the scene has them
        if (c.lds_trees && !c.has_mesh && sc.n_objects > TLAS_SCAN_MAX && c.wtlas_fmt == WIDE_F32 && sc.wtlas && sc.n_objects < 0x7fffu) {
            const uint32_t wl = c.debug_wide_levels ? c.debug_wide_levels : 3u * c.wtlas_depth + 2u;
            const size_t tree = (size_t)c.wtlas_nodes * WIDE_F32_DW * 4;
            uint32_t waves = 0;
            for (uint32_t w : {16u, 12u, 8u}) if (lds_walk_bytes(tree, w, wl, c.q.n_waves, walk_grid(w)) <= LDS_TREE_LIMIT) { waves = w; break; }
            if (waves) {
                if (lds_attr_needed(3)) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_extend_tlas_wide<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_TREE_LIMIT);
                    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_extend_tlas_wide<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_TREE_LIMIT);
                    (void)hipFuncSetAttribute(reinterpret_cast<const void *>((k_extend_tlas_wide<true, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_TREE_LIMIT);
                    (void)hipFuncSetAttribute(reinterpret_cast<const void *>((k_extend_tlas_wide<false, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_TREE_LIMIT);
                }
                const dim3 lg(walk_grid(waves));
                if (!sc.has_medium && c.simple_set) hipLaunchKernelGGL((k_extend_tlas_wide<false, true>), lg, dim3(waves * 64), lds_walk_bytes(tree, waves, wl, c.q.n_waves, lg.x), c.stream, sc, f, in, hits, c.q, segment, c.wtlas_nodes, wl);
                else if (sc.has_medium && c.simple_set) hipLaunchKernelGGL((k_extend_tlas_wide<true, true>), lg, dim3(waves * 64), lds_walk_bytes(tree, waves, wl, c.q.n_waves, lg.x), c.stream, sc, f, in, hits, c.q, segment, c.wtlas_nodes, wl);
                else if (sc.has_medium) hipLaunchKernelGGL(k_extend_tlas_wide<true>, lg, dim3(waves * 64), lds_walk_bytes(tree, waves, wl, c.q.n_waves, lg.x), c.stream, sc, f, in, hits, c.q, segment, c.wtlas_nodes, wl);
                else hipLaunchKernelGGL(k_extend_tlas_wide<false>, lg, dim3(waves * 64), lds_walk_bytes(tree, waves, wl, c.q.n_waves, lg.x), c.stream, sc, f, in, hits, c.q, segment, c.wtlas_nodes, wl);
                return;
            }
        }
        const size_t lds_tlas = lds_walk_bytes((size_t)c.tlas_pair_nodes * 64, LDS_WAVES, (uint32_t)tl, c.q.n_waves, walk_grid(LDS_WAVES));
        if (c.lds_trees && !c.has_mesh && sc.n_objects > TLAS_SCAN_MAX && c.tlas_pair_nodes < 32768u && sc.n_objects < 32768u && lds_tlas <= LDS_TREE_LIMIT) {
            if (lds_attr_needed(1)) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_extend_tlas_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_TREE_LIMIT);
            hipLaunchKernelGGL(k_extend_tlas_lds, dim3(walk_grid(LDS_WAVES)), dim3(LDS_WAVES * 64), lds_tlas, c.stream,
                               sc, f, in, hits, c.q, segment, c.tlas_pair_nodes, (uint32_t)tl);
        }
        else if (sc.n_objects <= TLAS_SCAN_MAX) hipLaunchKernelGGL(k_extend_scan<false>, eg, dim3(WB), lds, c.stream, sc, f, in, hits, c.q, segment, tl, (float4 *)nullptr, (float2 *)nullptr, (float4 *)nullptr, (uint32_t *)nullptr, 0u);
        else hipLaunchKernelGGL(k_extend_tlas, sg, dim3(WB), lds, c.stream, sc, f, in, hits, c.q, segment, tl, levels);
    }
#if FW_AB
    else if (use_bvh) hipLaunchKernelGGL(k_extend_bvh, eg, dim3(WB), lds, c.stream, sc, f, in, hits, c.q, segment, tl, levels);
#endif
    else if (c.n_defer && c.simple_set) hipLaunchKernelGGL(k_extend_linear_defer<true>, eg, dim3(WB), (size_t)2 * 64 * DEFER_FIELDS * 4 + FW_DEFER_LDS_PAD, c.stream, sc, f, in, hits, c.q, segment, c.n_defer);
    else if (c.n_defer) hipLaunchKernelGGL(k_extend_linear_defer<false>, eg, dim3(WB), (size_t)2 * 64 * DEFER_FIELDS * 4 + FW_DEFER_LDS_PAD, c.stream, sc, f, in, hits, c.q, segment, c.n_defer);
    else if (c.simple_set && !sc.has_medium) hipLaunchKernelGGL(k_extend_linear_simple<false>, eg, dim3(WB), lds, c.stream, sc, f, in, hits, c.q, segment, tl, levels);
    else if (c.simple_set) hipLaunchKernelGGL(k_extend_linear_simple<true>, eg, dim3(WB), lds, c.stream, sc, f, in, hits, c.q, segment, tl, levels);
    else if (!c.has_mesh && !sc.has_medium) hipLaunchKernelGGL(k_extend_linear_plain, eg, dim3(WB), lds, c.stream, sc, f, in, hits, c.q, segment, tl, levels);
    else if (!c.has_mesh) hipLaunchKernelGGL(k_extend_linear_nomesh, eg, dim3(WB), lds, c.stream, sc, f, in, hits, c.q, segment, tl, levels);
    else hipLaunchKernelGGL(k_extend_linear, eg, dim3(WB), lds, c.stream, sc, f, in, hits, c.q, segment, tl, levels);
}
void launch_extend_exact(const LaunchCfg &c, const DScene &sc, const DFrame &f, const DPaths &in, float2 *hits, int segment, bool use_bvh) {
    // stacks sized by the reference trees' depths; the wave-shared mesh stack needs EXACT_STACK entries whatever the depth
    // (the wave-shared TLAS stack + its list of mesh objects: EXACT_STACK + 2 * EXACT_DEFER entries)
    const uint32_t tl = std::max<uint32_t>(std::min<uint32_t>(c.ref_tlas_depth + 2u, EXACT_LEVELS), (EXACT_STACK + 2u * EXACT_DEFER) / EXACT_WB);
    const uint32_t bl = std::max<uint32_t>(std::min<uint32_t>(c.ref_blas_depth + 2u, EXACT_LEVELS), EXACT_STACK / EXACT_WB);
    const size_t lds = (size_t)(tl + bl) * EXACT_WB * 4;
    // 8 single-wave workgroups per CU: most launches find an empty list, and dispatching 4 096 workgroups that only read a counter took 12 us
    hipLaunchKernelGGL(k_extend_exact, dim3((uint32_t)c.n_cus * 8u), dim3(EXACT_WB), lds, c.stream, sc, f, in, hits, segment, use_bvh ? 1 : 0, tl, c.exact_form);
}
void launch_shade(const LaunchCfg &c, const DScene &sc, const DFrame &f, DPaths in, DPaths out, const float2 *hits,
                  float4 *sample_rad, int segment) {
    size_t tab = ((size_t)sc.n_objects * OBJ_Q + 2 * (size_t)c.n_mat + 2 * (size_t)c.n_tex) * sizeof(float4);
    const size_t tab_mt = (2 * (size_t)c.n_mat + 2 * (size_t)c.n_tex) * sizeof(float4);
    // table mode: everything in LDS | materials + textures only (the two dependent fetches behind the object record: part2 k_shade
    // 6.7 -> 6.55 ms; a leaner per-kind object fetch on top — 3 loads instead of 6 for an unrotated sphere — did not pay: 6.8 ms) | none
    const int lt = (c.lds_tables && tab <= LDS_TABLE_LIMIT) ? 1 : ((c.lds_tables && tab_mt <= LDS_TABLE_LIMIT) ? 2 : 0);
    const size_t lds = lt == 1 ? tab : (lt == 2 ? tab_mt : 0);
    // shading mode (k_shade): 0 everything in line | 1 the scene has nothing expensive | 2 expensive paths through the list
    const int mode = c.shade_mode;
#define FW_SHADE(L, M, C) hipLaunchKernelGGL((k_shade<L, M, C>), wave_grid(c), dim3(WB), lds, c.stream, sc, f, in, out, hits, sample_rad, c.q, segment, c.n_mat, c.n_tex)
#define FW_SHADE_C(L, M) do { if (f.chain_bits) FW_SHADE(L, M, true); else FW_SHADE(L, M, false); } while (0)
#if FW_AB
#define FW_SHADE_L(L) do { if (mode == 2) FW_SHADE(L, 2, false); else if (mode == 1) FW_SHADE_C(L, 1); else FW_SHADE_C(L, 0); } while (0)
#else
#define FW_SHADE_L(L) do { if (mode == 1) FW_SHADE_C(L, 1); else FW_SHADE_C(L, 0); } while (0)
#endif
    if (lt == 1) FW_SHADE_L(1); else if (lt == 2) FW_SHADE_L(2); else FW_SHADE_L(0);
#undef FW_SHADE_L
#undef FW_SHADE_C
#undef FW_SHADE
}
#if FW_AB
void launch_bounce(const LaunchCfg &c, const DScene &sc, const DFrame &f, DPaths in, DPaths out, float4 *sample_rad,
                   int segment, bool use_bvh) {
    size_t tab = ((size_t)sc.n_objects * OBJ_Q + 2 * (size_t)c.n_mat + 2 * (size_t)c.n_tex) * sizeof(float4);
    const bool lds_tab = c.lds_tables && tab <= LDS_TABLE_LIMIT;
    if (!lds_tab) tab = 0;
    int tl = use_bvh ? c.tlas_depth + 1 : 0;
    int levels = tl + c.blas_depth + 1;
    size_t lds = tab + (size_t)levels * WB * sizeof(uint32_t);
    uint32_t tq = (uint32_t)(tab / sizeof(float4));
    dim3 g = wave_grid(c);
#define FW_BOUNCE(B, T) hipLaunchKernelGGL((k_bounce<B, T>), g, dim3(WB), lds, c.stream, sc, f, in, out, sample_rad, c.q, segment, tl, c.n_mat, c.n_tex, tq)
    if (use_bvh) { if (lds_tab) FW_BOUNCE(true, true); else FW_BOUNCE(true, false); }
    else { if (lds_tab) FW_BOUNCE(false, true); else FW_BOUNCE(false, false); }
#undef FW_BOUNCE
}
#endif   // FW_AB
void preload_kernels() {
    hipFuncAttributes a;
#define FW_TOUCH(k) (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(k))
    FW_TOUCH(k_raygen); FW_TOUCH(k_extend_linear); FW_TOUCH(k_extend_linear_nomesh); FW_TOUCH(k_extend_linear_plain); FW_TOUCH(k_extend_linear_defer<true>); FW_TOUCH(k_extend_linear_defer<false>); FW_TOUCH(k_extend_linear_simple<true>); FW_TOUCH(k_extend_linear_simple<false>); FW_TOUCH(k_extend_scan<true>); FW_TOUCH((k_extend_scan<true, true>)); FW_TOUCH((k_extend_scan<true, true, true>)); FW_TOUCH(k_extend_scan<false>);
    FW_TOUCH(k_extend_tlas); FW_TOUCH(k_extend_tlas_park); FW_TOUCH(k_blas); FW_TOUCH(k_blas_lds<true>); FW_TOUCH(k_blas_lds<false>); FW_TOUCH(k_extend_tlas_lds);
    FW_TOUCH((k_blas_wide<WIDE_F32, true>)); FW_TOUCH((k_blas_wide<WIDE_F32, false>)); FW_TOUCH((k_blas_wide<WIDE_Q8, true>)); FW_TOUCH((k_blas_wide<WIDE_Q8, false>));
    FW_TOUCH(k_extend_tlas_wide<true>); FW_TOUCH(k_extend_tlas_wide<false>); FW_TOUCH((k_extend_tlas_wide<true, true>)); FW_TOUCH((k_extend_tlas_wide<false, true>)); FW_TOUCH(k_extend_exact); FW_TOUCH(k_queue_totals); FW_TOUCH(k_count_deposits); FW_TOUCH(k_accumulate); FW_TOUCH(k_tile_order);
    FW_TOUCH(k_resolve); FW_TOUCH(k_scatter_tiles);
    FW_TOUCH((k_shade<0, 0, false>)); FW_TOUCH((k_shade<0, 0, true>)); FW_TOUCH((k_shade<0, 1, false>)); FW_TOUCH((k_shade<0, 1, true>));
    FW_TOUCH((k_shade<1, 0, false>)); FW_TOUCH((k_shade<1, 0, true>)); FW_TOUCH((k_shade<1, 1, false>)); FW_TOUCH((k_shade<1, 1, true>));
    FW_TOUCH((k_shade<2, 0, false>)); FW_TOUCH((k_shade<2, 0, true>)); FW_TOUCH((k_shade<2, 1, false>)); FW_TOUCH((k_shade<2, 1, true>));
#undef FW_TOUCH
    (void)hipGetLastError();
}
#if FW_AB
uint32_t take_error_word() {      // reads and clears the device's error word (after a frame's stream has drained); 0: none
    uint32_t v = 0, zero = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_err_word), 4) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (v) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_err_word), &zero, 4);
    return v;
}
#endif
#ifdef FW_PHASE_STATS
extern "C" int fw_debug_phase_stats(unsigned long long *out, int n) {   // debug builds only; reads and clears the sections' counters (3 per section)
    static unsigned long long zero[2 * 3 * PH_N];
    if (n < 2 * 3 * PH_N) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof zero) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phase), zero, sizeof zero) == hipSuccess ? 2 * 3 * PH_N : -1;
}
#endif
#ifdef FW_TRAV_STATS
extern "C" int fw_debug_trav_stats(unsigned long long out[8]) {   // debug builds only; reads and clears the counters
    unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trav), sizeof zero) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_trav), zero, sizeof zero) == hipSuccess ? 0 : -1;
}
#endif
void launch_queue_totals(const LaunchCfg &c, uint32_t *totals, const uint32_t *ptotal) {
    hipLaunchKernelGGL(k_queue_totals, dim3(32, MAX_SEGMENTS + 1), dim3(BLOCK), 0, c.stream, c.q, totals, ptotal);   // totals are zeroed per frame
}
void launch_count_deposits(const LaunchCfg &c, const uint32_t *dep_bits, uint32_t *total) {
    hipLaunchKernelGGL(k_count_deposits, dim3(1024), dim3(BLOCK), 0, c.stream, dep_bits, (c.q.n_waves * c.q.cap + 31u) / 32u, total);
}
void launch_accumulate(const LaunchCfg &c, const DFrame &f, const float4 *sample_rad, float4 *accum) {
    uint32_t blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(((uint64_t)f.n_pixels + WB - 1) / WB, 65536));
    hipLaunchKernelGGL(k_accumulate, dim3(blocks), dim3(WB), 0, c.stream, f, sample_rad, accum);
}
void launch_resolve(const LaunchCfg &c, const DFrame &f, const float4 *accum, uint32_t total_spp, float gamma,
                    uint8_t *rgb8, float *gamma_rgb, float *linear_rgb) {
    hipLaunchKernelGGL(k_resolve, dim3(c.blocks_other), dim3(BLOCK), 0, c.stream, f, accum, total_spp, gamma, rgb8, gamma_rgb, linear_rgb);
}

} // namespace fw
