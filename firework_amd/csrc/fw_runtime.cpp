// fw_runtime.cpp — host side of libfirework_hip.so: the C ABI of include/firework_hip.h.
//
//   fw_scene_create : `Scene -> SceneInternal` (reference src/scene.rs:111-135,279-292) flattened to the
//                     HBM layout of fw_device.h; TLAS/BLAS built with the reference's median split
//                     (src/bvh.rs:21-71) and stored in DFS order.
//   fw_render       : the wavefront loop that replaces the rayon pixel loop of src/render.rs:127-161:
//                     raygen -> 11 x (extend, shade+compact) -> accumulate, per batch of paths; resolve.
//
// No CPU rendering path exists here: without a HIP device every entry point fails with FW_ERR_NO_DEVICE.
#include "../../include/firework_hip.h"
#include "fw_device.h"

#include <algorithm>
#include <atomic>
#include <exception>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_last_error;

int fail(int status, const std::string &msg) { g_last_error = msg; return status; }

// ---- options: every runtime switch of the library.  The environment (FIREWORK_<NAME>) is read ONCE, when the library is loaded;
// fw_set_option changes one afterwards (tests, tools/).  Nothing on the render path calls getenv, and the product build carries
// no switch for a kernel it does not carry.
struct Options {
    bool bvh_median = false;      // BVH=median: walk the reference's own topology (parity / A-B mode)
    bool no_exact = false, exact_all = false;   // NO_EXACT: no literal walk at all; EXACT_ALL=1: every ray takes it (the renderer then IS bvh.rs:115-151)
    int exact_form = 0;           // EXACT_FORM=lane|wave
    bool no_defer = false, no_hit4 = false, no_hoist = false, no_lds_tables = false, no_lds_trees = false, no_lds_tris = false, no_short_rays = false,
         no_tile_order = false, no_zero_skip = false, dep_pixel_major = false, dep_slot_major = false, trace = false, no_chain = false,
         exact_product = false;   // EXACT_PRODUCT: textured scenes keep every scattering's attenuation and multiply back to front at deposit (render.rs:23-28's order)
    int streams = 0;              // STREAMS=n batches in flight (0: the library's choice)
    int graph = -1;               // GRAPH=0|1: a frame that repeats is replayed as one hipGraph (-1: frames of small batches, whose launches are short)
    int phase_lock = -1;          // PHASE_LOCK=0|1: the two batches in flight held in anti-phase, one's extend beside the other's shade (-1: where it pays: box-list scenes)
    int soft_shear_log2 = 5, exact_shear_log2 = 10; double exact_far_x = 256.0;   // SOFT_SHEAR_LOG2 (0: off), EXACT_SHEAR_LOG2, EXACT_FAR_X: the flag rules' thresholds (tools/flag_margin.py)
    double wide_node_cost = 0.0005;   // WIDE_NODE_COST: the constant a wide node costs in the collapse, in root areas (wide_convert)
    int wide = -1;                // WIDE=0|f32|q8: no wide nodes / force an encoding (-1: by size)
    long waves = 0;               // WAVES=n wave queues (0: the library's choice)
    long long paths_per_batch = 0;
    std::string dump_path;        // DUMP_PATH=file (tools/diverge.py)
#if FW_AB
    bool fused = false, tlas_refill_off = false, shade_list = false, no_shade_defer = false, stagger = false;
    int debug_wide_levels = 0;    // DEBUG_WIDE_LEVELS=n: undersized LDS stacks for the wide walks (the error word's test)
#endif
};
const char *const OPTION_NAMES[] = {"BVH", "NO_EXACT", "EXACT_ALL", "EXACT_FORM", "NO_DEFER", "NO_HIT4", "NO_HOIST", "NO_LDS_TABLES", "NO_LDS_TREES", "NO_LDS_TRIS",
                                    "NO_SHORT_RAYS", "NO_TILE_ORDER", "NO_ZERO_SKIP", "DEP_PIXEL_MAJOR", "DEP_SLOT_MAJOR", "NO_CHAIN", "EXACT_PRODUCT", "PHASE_LOCK", "GRAPH", "SOFT_SHEAR_LOG2", "EXACT_SHEAR_LOG2", "EXACT_FAR_X", "WIDE_NODE_COST", "TRACE", "STREAMS", "WIDE", "WAVES",
                                    "PATHS_PER_BATCH", "DUMP_PATH",
#if FW_AB
                                    "FUSED", "TLAS_REFILL", "SHADE_LIST", "NO_SHADE_DEFER", "STAGGER", "DEBUG_WIDE_LEVELS",
#endif
                                    nullptr};
bool option_apply(Options &o, const char *name, const char *v) {      // v == nullptr: back to the default
    const std::string n = name;
    auto on = [&] { return v != nullptr; };
    auto num = [&] { return v ? atoll(v) : 0ll; };
    if (n == "BVH") o.bvh_median = v && std::strcmp(v, "median") == 0;
    else if (n == "NO_EXACT") o.no_exact = on();
    else if (n == "EXACT_ALL") o.exact_all = num() != 0;
    else if (n == "EXACT_FORM") o.exact_form = v ? (std::strcmp(v, "lane") == 0 ? 1 : (std::strcmp(v, "wave") == 0 ? 2 : 0)) : 0;
    else if (n == "NO_DEFER") o.no_defer = on();
    else if (n == "NO_HIT4") o.no_hit4 = on();
    else if (n == "NO_HOIST") o.no_hoist = on();
    else if (n == "NO_LDS_TABLES") o.no_lds_tables = on();
    else if (n == "NO_LDS_TREES") o.no_lds_trees = on();
    else if (n == "NO_LDS_TRIS") o.no_lds_tris = on();
    else if (n == "NO_SHORT_RAYS") o.no_short_rays = on();
    else if (n == "NO_TILE_ORDER") o.no_tile_order = on();
    else if (n == "NO_ZERO_SKIP") o.no_zero_skip = on();
    else if (n == "DEP_PIXEL_MAJOR") o.dep_pixel_major = on();
    else if (n == "DEP_SLOT_MAJOR") o.dep_slot_major = on();
    else if (n == "NO_CHAIN") o.no_chain = on();
    else if (n == "EXACT_PRODUCT") o.exact_product = v && atoi(v) != 0;
    else if (n == "PHASE_LOCK") o.phase_lock = v ? (atoi(v) != 0 ? 1 : 0) : -1;
    else if (n == "GRAPH") o.graph = v ? (atoi(v) != 0 ? 1 : 0) : -1;
    else if (n == "SOFT_SHEAR_LOG2") o.soft_shear_log2 = v ? (int)num() : 5;
    else if (n == "EXACT_SHEAR_LOG2") o.exact_shear_log2 = v ? (int)num() : 10;
    else if (n == "EXACT_FAR_X") o.exact_far_x = v ? atof(v) : 256.0;
    else if (n == "WIDE_NODE_COST") o.wide_node_cost = v ? atof(v) : 0.0005;
    else if (n == "TRACE") o.trace = on();
    else if (n == "STREAMS") o.streams = (int)std::max<long long>(0, num());
    else if (n == "WIDE") o.wide = !v ? -1 : (std::strcmp(v, "0") == 0 ? 0 : (std::strcmp(v, "f32") == 0 ? 1 : (std::strcmp(v, "q8") == 0 ? 2 : -1)));
    else if (n == "WAVES") o.waves = (long)std::max<long long>(0, num());
    else if (n == "PATHS_PER_BATCH") o.paths_per_batch = std::max<long long>(0, num());
    else if (n == "DUMP_PATH") o.dump_path = v ? v : "";
#if FW_AB
    else if (n == "FUSED") o.fused = num() != 0;
    else if (n == "TLAS_REFILL") o.tlas_refill_off = v && atoi(v) == 0;
    else if (n == "SHADE_LIST") o.shade_list = on();
    else if (n == "NO_SHADE_DEFER") o.no_shade_defer = on();
    else if (n == "STAGGER") o.stagger = on();
    else if (n == "DEBUG_WIDE_LEVELS") o.debug_wide_levels = (int)std::max<long long>(0, num());
#endif
    else return false;
    return true;
}
Options options_from_env() {
    Options o;
    for (int k = 0; OPTION_NAMES[k]; k++) { const std::string e = std::string("FIREWORK_") + OPTION_NAMES[k]; if (const char *v = getenv(e.c_str())) option_apply(o, OPTION_NAMES[k], v); }
    return o;
}
std::mutex g_opt_mu;
Options g_opt = options_from_env();          // once, at load
Options options() { std::lock_guard<std::mutex> g(g_opt_mu); return g_opt; }

#define HIPCHK(expr)                                                                                           \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) {                                                                                \
            char buf_[512];                                                                                    \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return fail(e_ == hipErrorOutOfMemory ? FW_ERR_OOM : FW_ERR_HIP, buf_);                            \
        }                                                                                                      \
    } while (0)

// ---- host-side f32 vector math; same expressions as the reference (and -ffp-contract=off) ------------
struct V3 { float x, y, z; float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); } };
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(V3 a, V3 b) { return std::fmaf(a.x, b.x, std::fmaf(a.y, b.y, a.z * b.z)); }
inline V3 cross(V3 a, V3 b) { return {std::fmaf(a.y, b.z, -a.z * b.y), std::fmaf(a.z, b.x, -a.x * b.z), std::fmaf(a.x, b.y, -a.y * b.x)}; }
inline V3 normalized(V3 a) { float m = std::sqrt(dot(a, a)); return {a.x / m, a.y / m, a.z / m}; }
inline V3 vmin(V3 a, V3 b) { return {std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)}; }
inline V3 vmax(V3 a, V3 b) { return {std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)}; }
inline V3 tov(const fw_vec3 &v) { return {v.x, v.y, v.z}; }

struct Box { V3 mn, mx; };
inline Box box_union(const Box &a, const Box &b) { return {vmin(a.mn, b.mn), vmax(a.mx, b.mx)}; }   // aabb.rs:52-57
inline V3 box_center(const Box &b) { return 0.5f * b.mn + 0.5f * b.mx; }                             // aabb.rs:59-61

inline float bits_f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// ultraviolet Rotor3::into_matrix; rows[i] = row i of rotation_mat
void rotor_rows(const fw_rotor3 &r, float rows[3][3]) {
    float s2 = r.s * r.s, bxy2 = r.xy * r.xy, bxz2 = r.xz * r.xz, byz2 = r.yz * r.yz;
    float s_bxy = r.s * r.xy, s_bxz = r.s * r.xz, s_byz = r.s * r.yz;
    float bxz_byz = r.xz * r.yz, bxy_byz = r.xy * r.yz, bxy_bxz = r.xy * r.xz;
    float c0[3] = {s2 - bxy2 - bxz2 + byz2, -2.f * (bxz_byz + s_bxy), 2.f * (bxy_byz - s_bxz)};
    float c1[3] = {2.f * (s_bxy - bxz_byz), s2 - bxy2 + bxz2 - byz2, -2.f * (s_byz + bxy_bxz)};
    float c2[3] = {2.f * (s_bxz + bxy_byz), 2.f * (s_byz - bxy_bxz), s2 + bxy2 - bxz2 - byz2};
    for (int i = 0; i < 3; i++) { rows[i][0] = c0[i]; rows[i][1] = c1[i]; rows[i][2] = c2[i]; }
}
inline V3 mat_mul(const float rows[3][3], V3 v) {   // Mat3 * Vec3: c0*x + c1*y + c2*z
    return {rows[0][0] * v.x + rows[0][1] * v.y + rows[0][2] * v.z,
            rows[1][0] * v.x + rows[1][1] * v.y + rows[1][2] * v.z,
            rows[2][0] * v.x + rows[2][1] * v.y + rows[2][2] * v.z};
}

// ---- K10: flat BVH builder reproducing bvh.rs:21-71 ---------------------------------------------------
struct FlatBvh {
    std::vector<float> nodes;   // 8 floats per node
    uint32_t depth = 0;
    uint32_t count() const { return (uint32_t)(nodes.size() / 8); }
};
struct NanError {};

// ---- parallel host builds (round 5).  The reference builds its trees inside its timed region (main.rs:40-44) on one thread; so did this
// library until round 4: 1.0 s for the median tree and 0.5 s for the SAH tree of a million triangles (profiles/r05_big_mesh.txt).  Both
// recursions are independent below a node, so a node with enough items hands its left subtree to another thread, which builds it into a
// tree of its own; the two are appended behind the node in depth-first order (child indices shifted) — the same nodes in the same order as
// the sequential build — and the big sorts and binning passes near the root are split over the threads that are still free.  A stable
// sort's result does not depend on how it was computed, box unions (fmin / fmax) and counts are exact: the trees are bit-identical.
struct BuildPool {
    std::atomic<int> free_threads;
    explicit BuildPool(int n) : free_threads(n) {}
    int take(int want) {      // up to `want` helper threads (possibly 0)
        int got = 0;
        while (got < want) { int f = free_threads.load(); if (f <= 0) break; if (free_threads.compare_exchange_weak(f, f - 1)) got++; }
        return got;
    }
    void give(int n) { free_threads.fetch_add(n); }
};
inline int host_build_threads() {
    static const int n = [] { const char *e = getenv("FIREWORK_BUILD_THREADS"); int v = e ? atoi(e) : (int)std::thread::hardware_concurrency(); return std::max(1, std::min(v, 64)); }();
    return n;
}
constexpr size_t PAR_SUBTREE_MIN = 4096, PAR_PASS_MIN = 1 << 15;     // items below which a subtree / a pass over the items stays on its thread

// f(begin, end, part) over [0, n) in `parts` contiguous parts, part 0 on the calling thread
template <class F> void par_parts(size_t n, int parts, F f) {
    if (parts <= 1) { f((size_t)0, n, 0); return; }
    std::vector<std::thread> th;
    th.reserve((size_t)parts - 1);
    for (int k = 1; k < parts; k++) th.emplace_back([&, k] { f(n * (size_t)k / (size_t)parts, n * (size_t)(k + 1) / (size_t)parts, k); });
    f((size_t)0, n / (size_t)parts, 0);
    for (auto &t : th) t.join();
}
// std::stable_sort's result by any route: sorted parts, merged pairwise (inplace_merge is stable)
template <class Cmp> void par_stable_sort(uint32_t *b, size_t n, Cmp cmp, BuildPool &pool) {
    int helpers = n >= PAR_PASS_MIN ? pool.take(15) : 0;
    int parts = 1; while (parts * 2 <= helpers + 1) parts *= 2;           // a power of two
    pool.give(helpers - (parts - 1)); helpers = parts - 1;
    if (parts == 1) { std::stable_sort(b, b + n, cmp); return; }
    auto cut = [&](int k) { return n * (size_t)k / (size_t)parts; };
    par_parts(n, parts, [&](size_t lo, size_t hi, int) { std::stable_sort(b + lo, b + hi, cmp); });
    for (int width = 1; width < parts; width *= 2) {
        const int pairs = parts / (2 * width);
        par_parts((size_t)pairs, pairs, [&](size_t lo, size_t hi, int) {
            for (size_t q = lo; q < hi; q++) std::inplace_merge(b + cut((int)q * 2 * width), b + cut((int)q * 2 * width + width), b + cut((int)(q + 1) * 2 * width), cmp);
        });
    }
    pool.give(helpers);
}
// appends tree `t` (root at its node 0) to `out`, child indices shifted; returns where its root went
inline uint32_t append_tree(FlatBvh &out, const FlatBvh &t) {
    const uint32_t off = out.count();
    out.nodes.insert(out.nodes.end(), t.nodes.begin(), t.nodes.end());
    for (uint32_t i = 0; i < t.count(); i++) {
        float *p = &out.nodes[(size_t)(off + i) * 8];
        uint32_t A; std::memcpy(&A, p + 3, 4);
        if ((A >> 30) == 0u) { A += off; std::memcpy(p + 3, &A, 4); }      // a Branch: its right child
    }
    out.depth = std::max(out.depth, t.depth);
    return off;
}
// build(left, into) and build(right, into): the left one on a helper thread into a tree of its own when the node is big enough and a thread is
// free, else both in place.  Returns the right child's index; lb / rb = the children's boxes.
template <class Build> uint32_t build_children(FlatBvh &out, size_t n, size_t half, BuildPool &pool, Build build, Box &lb, Box &rb) {
    if (n >= PAR_SUBTREE_MIN && pool.take(1) == 1) {
        FlatBvh L, R;
        std::exception_ptr ep;
        std::thread th([&] { try { build(L, (size_t)0, half, lb); } catch (...) { ep = std::current_exception(); } });
        try { build(R, half, n - half, rb); } catch (...) { th.join(); pool.give(1); throw; }
        th.join(); pool.give(1);
        if (ep) std::rethrow_exception(ep);
        append_tree(out, L);
        return append_tree(out, R);
    }
    build(out, (size_t)0, half, lb);                                    // left child = me + 1
    const uint32_t before = out.count();
    build(out, half, n - half, rb);
    return before;
}

struct Centers { std::vector<float> c[3]; };     // box_center per item and axis, computed once per build (the comparators read them millions of times)
inline Centers centers_of(const std::vector<Box> &boxes, BuildPool &pool) {
    Centers ce;
    const size_t n = boxes.size();
    for (int a = 0; a < 3; a++) ce.c[a].resize(n);
    const int helpers = n >= PAR_PASS_MIN ? pool.take(7) : 0;
    par_parts(n, helpers + 1, [&](size_t lo, size_t hi, int) { for (size_t i = lo; i < hi; i++) { const V3 c = box_center(boxes[i]); ce.c[0][i] = c.x; ce.c[1][i] = c.y; ce.c[2][i] = c.z; } });
    pool.give(helpers);
    return ce;
}

uint32_t bvh_build_rec(FlatBvh &out, const std::vector<Box> &boxes, const Centers &ce, uint32_t *idx, size_t n, uint32_t depth, Box &node_box, BuildPool &pool) {
    int axis = (int)(depth % 3);
    const float *key = ce.c[axis].data();
    for (size_t i = 0; i < n; i++) { float c = key[idx[i]]; if (c != c) throw NanError(); }
    // Rust's sort_by is stable; the sub-slice is re-sorted at every level (bvh.rs:29-35)
    par_stable_sort(idx, n, [key](uint32_t a, uint32_t b) { return key[a] < key[b]; }, pool);
    uint32_t me = out.count();
    out.nodes.resize(out.nodes.size() + 8);
    out.depth = std::max(out.depth, depth);
    uint32_t A, B = 0;
    if (n == 1) { node_box = boxes[idx[0]]; A = (fw::NODE_LEAF << 30) | idx[0]; }
    else if (n == 2) { node_box = box_union(boxes[idx[0]], boxes[idx[1]]); A = (fw::NODE_DOUBLE << 30) | idx[0]; B = idx[1]; }
    else {
        size_t half = n / 2;
        Box lb, rb;
        const uint32_t right = build_children(out, n, half, pool, [&](FlatBvh &into, size_t first, size_t count, Box &b) {
            bvh_build_rec(into, boxes, ce, idx + first, count, depth + 1, b, pool); }, lb, rb);
        node_box = box_union(lb, rb);
        A = right;
        B = (uint32_t)axis;     // split axis of this Branch (depth % 3): picks the near child during traversal
    }
    float *p = &out.nodes[(size_t)me * 8];
    p[0] = node_box.mn.x; p[1] = node_box.mn.y; p[2] = node_box.mn.z; p[3] = bits_f(A);
    p[4] = node_box.mx.x; p[5] = node_box.mx.y; p[6] = node_box.mx.z; p[7] = bits_f(B);
    return me;
}
Box bvh_build(FlatBvh &out, const std::vector<Box> &boxes, BuildPool *shared = nullptr) {
    std::vector<uint32_t> idx(boxes.size());
    for (size_t i = 0; i < idx.size(); i++) idx[i] = (uint32_t)i;
    BuildPool own(host_build_threads() - 1);
    BuildPool &pool = shared ? *shared : own;
    const Centers ce = centers_of(boxes, pool);
    Box root;
    bvh_build_rec(out, boxes, ce, idx.data(), idx.size(), 0, root, pool);
    return root;
}

// In-order rank of every item in the reference (median-split) tree: leaves appear left to right in DFS order.
std::vector<uint32_t> reference_ranks(const FlatBvh &ref, size_t n_items) {
    std::vector<uint32_t> rank(n_items, 0);
    uint32_t next = 0;
    for (uint32_t i = 0; i < ref.count(); i++) {
        uint32_t A, B; std::memcpy(&A, &ref.nodes[(size_t)i * 8 + 3], 4); std::memcpy(&B, &ref.nodes[(size_t)i * 8 + 7], 4);
        uint32_t kind = A >> 30;
        if (kind == fw::NODE_LEAF) rank[A & fw::NODE_MASK] = next++;
        else if (kind == fw::NODE_DOUBLE) { rank[A & fw::NODE_MASK] = next++; rank[B] = next++; }
    }
    return rank;
}

// The boxes of the WALKED trees are the items' own boxes grown by 2^-14 of their largest extent (round 4).  The walks only have to
// reach every item the reference's rule admits and the item test accepts (fw_kernels.hip: hit_aabb_entry; the rule itself is applied
// to every hit that counts: tri_gate_ok / obj_gate_ok).  A triangle test carries (max|d| / |d_kz|) ulps of the distance from the ray's
// origin to the triangle's far vertices — at most distance + extent — so a ray can "hit" a triangle it passes by that much: the part
// in proportion to the distance is what the walks' relaxed exit planes admit (2^-12), the part in proportion to the extent is this.
// How much: an item test's error SIDEWAYS, for a ray that runs along a face of the box, is not helped by relaxed exit planes and has to
// be in the box itself (part2 pixel 1049389, sample 106: a camera ray 17 units away grazes a sphere of radius 0.1 past the face of its
// own box, gpurun_out/r04j).  Every ray that is not on the exact list starts within far_r = 256 x the smallest item (DExact.far_r):
//   triangle: shear x 2^-24 x distance, shear < 2^10 (beyond: the exact list) -> 2^-14 far_r = 2^-6 of a typical triangle of the mesh;
//   sphere / cone / cylinder: the discriminant's rounding lets a ray that misses by 2^-24 L^2 / r still hit -> 2^-22 far_r^2 / size;
//   rectangles and boxes: their tests are exact in the plane; 2^-14 of the extent for the rounding of the slab test itself.
inline float box_extent(const Box &b) { const V3 e = b.mx - b.mn; return std::fmax(std::fabs(e.x), std::fmax(std::fabs(e.y), std::fabs(e.z))); }
inline Box grown_by(const Box &b, float g) {
    if (!(g > 0.f) || !std::isfinite(g)) return b;
    return Box{{b.mn.x - g, b.mn.y - g, b.mn.z - g}, {b.mx.x + g, b.mx.y + g, b.mx.z + g}};
}
inline Box grown(const Box &b) { return grown_by(b, std::ldexp(box_extent(b), -14)); }

// item boxes := the box of the reference leaf node that holds the item (its own box for a Leaf, the union for a DoubleLeaf)
void leaf_node_boxes(const FlatBvh &ref, std::vector<Box> &boxes) {
    for (uint32_t i = 0; i < ref.count(); i++) {
        const float *nd = &ref.nodes[(size_t)i * 8];
        uint32_t A, B; std::memcpy(&A, nd + 3, 4); std::memcpy(&B, nd + 7, 4);
        if ((A >> 30) != fw::NODE_DOUBLE) continue;
        const Box nb{{nd[0], nd[1], nd[2]}, {nd[4], nd[5], nd[6]}};
        boxes[A & fw::NODE_MASK] = nb; boxes[B] = nb;
    }
}

// ---- traversal tree: binned-SAH top-down build, same node format (leaves of 1 or 2 items, DFS order).
// The reference's tree (bvh_build above) fixes WHICH hit wins ties (in-order rank); it does not have to be the
// tree that is walked.  SAH isolates large items near the root (part2's r=5000 fog sphere otherwise inflates
// every ancestor box along its spine) and splits on the axis that actually separates the items.
constexpr uint32_t SAH_MAX_DEPTH = 40;
inline float box_area(const Box &b) { V3 d = b.mx - b.mn; return 2.f * (d.x * d.y + d.y * d.z + d.z * d.x); }

uint32_t sah_build_rec(FlatBvh &out, const std::vector<Box> &boxes, const Centers &ce, uint32_t *idx, size_t n, uint32_t depth, Box &node_box, BuildPool &pool) {
    uint32_t me = out.count();
    out.nodes.resize(out.nodes.size() + 8);
    out.depth = std::max(out.depth, depth);
    uint32_t A, B = 0;
    if (n == 1) { node_box = boxes[idx[0]]; A = (fw::NODE_LEAF << 30) | idx[0]; }
    else if (n == 2) { node_box = box_union(boxes[idx[0]], boxes[idx[1]]); A = (fw::NODE_DOUBLE << 30) | idx[0]; B = idx[1]; }
    else {
        const Box EMPTY{{1e30f, 1e30f, 1e30f}, {-1e30f, -1e30f, -1e30f}};
        constexpr int NB = 16;
        // the passes over the node's items (the centroids' bounds, then the bins of the three axes) in parts on the free threads: unions
        // by fmin / fmax and counts are exact, so the merged bins are the sequential ones
        const int helpers = n >= PAR_PASS_MIN ? pool.take(15) : 0, parts = helpers + 1;
        Box cb = EMPTY;
        {
            std::vector<Box> pcb((size_t)parts, EMPTY);
            par_parts(n, parts, [&](size_t lo, size_t hi, int k) {
                Box b = EMPTY;
                for (size_t i = lo; i < hi; i++) { const uint32_t it = idx[i]; const V3 c{ce.c[0][it], ce.c[1][it], ce.c[2][it]}; b.mn = vmin(b.mn, c); b.mx = vmax(b.mx, c); }
                pcb[(size_t)k] = b; });
            for (const Box &b : pcb) { cb.mn = vmin(cb.mn, b.mn); cb.mx = vmax(cb.mx, b.mx); }
        }
        int best_axis = -1; size_t best_split = 0; float best_cost = 1e38f;
        if (depth < SAH_MAX_DEPTH) {
            struct Bins { Box bb[3][NB]; size_t bc[3][NB]; };
            std::vector<Bins> pb((size_t)parts);
            par_parts(n, parts, [&](size_t lo_i, size_t hi_i, int k) {
                Bins &bn = pb[(size_t)k];
                for (int axis = 0; axis < 3; axis++) for (int q = 0; q < NB; q++) { bn.bb[axis][q] = EMPTY; bn.bc[axis][q] = 0; }
                for (int axis = 0; axis < 3; axis++) {
                    const float lo = cb.mn[axis], ext = cb.mx[axis] - lo;
                    if (!(ext > 0.f)) continue;
                    const float *key = ce.c[axis].data();
                    for (size_t i = lo_i; i < hi_i; i++) {
                        const uint32_t it = idx[i];
                        const int q = std::min(NB - 1, std::max(0, (int)((key[it] - lo) / ext * NB)));
                        bn.bb[axis][q] = box_union(bn.bb[axis][q], boxes[it]); bn.bc[axis][q]++;
                    }
                } });
            for (int axis = 0; axis < 3; axis++) {
                float lo = cb.mn[axis], ext = cb.mx[axis] - lo;
                if (!(ext > 0.f)) continue;
                Box bb[NB]; size_t bc[NB];
                for (int k = 0; k < NB; k++) { bb[k] = EMPTY; bc[k] = 0; }
                for (const Bins &bn : pb) for (int k = 0; k < NB; k++) if (bn.bc[axis][k]) { bb[k] = box_union(bb[k], bn.bb[axis][k]); bc[k] += bn.bc[axis][k]; }
                float la[NB], ra[NB]; size_t lc[NB], rc[NB];
                Box acc = EMPTY; size_t cnt = 0;
                for (int k = 0; k < NB; k++) { if (bc[k]) acc = box_union(acc, bb[k]); cnt += bc[k]; la[k] = cnt ? box_area(acc) : 0.f; lc[k] = cnt; }
                acc = EMPTY; cnt = 0;
                for (int k = NB - 1; k >= 0; k--) { if (bc[k]) acc = box_union(acc, bb[k]); cnt += bc[k]; ra[k] = cnt ? box_area(acc) : 0.f; rc[k] = cnt; }
                for (int k = 0; k + 1 < NB; k++) {
                    if (lc[k] == 0 || rc[k + 1] == 0) continue;
                    float cost = la[k] * (float)lc[k] + ra[k + 1] * (float)rc[k + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_split = (size_t)k; }
                }
            }
        }
        pool.give(helpers);
        size_t half;
        int axis;
        if (best_axis >= 0) {
            axis = best_axis;
            float lo = cb.mn[axis], ext = cb.mx[axis] - lo;
            const float *key = ce.c[axis].data();
            auto mid = std::stable_partition(idx, idx + n, [&](uint32_t a) {
                int k = std::min(NB - 1, std::max(0, (int)((key[a] - lo) / ext * NB)));
                return (size_t)k <= best_split; });
            half = (size_t)(mid - idx);
        } else {   // all centroids coincide (or depth cap): median split keeps the tree balanced
            V3 e = cb.mx - cb.mn;
            axis = e.x >= e.y ? (e.x >= e.z ? 0 : 2) : (e.y >= e.z ? 1 : 2);
            const float *key = ce.c[axis].data();
            par_stable_sort(idx, n, [key](uint32_t a, uint32_t b) { return key[a] < key[b]; }, pool);
            half = n / 2;
        }
        if (half == 0 || half == n) half = n / 2;
        Box lb, rb;
        const uint32_t right = build_children(out, n, half, pool, [&](FlatBvh &into, size_t first, size_t count, Box &b) {
            sah_build_rec(into, boxes, ce, idx + first, count, depth + 1, b, pool); }, lb, rb);
        node_box = box_union(lb, rb);
        A = right;
        B = (uint32_t)axis;     // left child holds the smaller centroids along this axis
    }
    float *p = &out.nodes[(size_t)me * 8];
    p[0] = node_box.mn.x; p[1] = node_box.mn.y; p[2] = node_box.mn.z; p[3] = bits_f(A);
    p[4] = node_box.mx.x; p[5] = node_box.mx.y; p[6] = node_box.mx.z; p[7] = bits_f(B);
    return me;
}
void sah_build(FlatBvh &out, const std::vector<Box> &boxes, BuildPool *shared = nullptr) {
    std::vector<uint32_t> idx(boxes.size());
    for (size_t i = 0; i < idx.size(); i++) idx[i] = (uint32_t)i;
    BuildPool own(host_build_threads() - 1);
    BuildPool &pool = shared ? *shared : own;
    const Centers ce = centers_of(boxes, pool);
    Box root;
    sah_build_rec(out, boxes, ce, idx.data(), idx.size(), 0, root, pool);
}

// ---- the tree as walked on the device: pair nodes (fw_device.h), converted from a FlatBvh.  A DoubleLeaf becomes a
// pair node over two single-item leaves, each with its own box.  `depth` = inner nodes on the longest root-to-leaf path
// = the most references a walk can have pushed.
struct PairBvh {
    std::vector<float> nodes;   // 16 floats per pair node
    uint32_t depth = 0;
    uint32_t count() const { return (uint32_t)(nodes.size() / 16); }
};
static uint32_t pair_convert_rec(const FlatBvh &src, uint32_t i, const std::vector<Box> &item_boxes, PairBvh &out, uint32_t base,
                                 uint32_t depth, Box &box) {
    const float *nd = &src.nodes[(size_t)i * 8];
    uint32_t A, B; std::memcpy(&A, nd + 3, 4); std::memcpy(&B, nd + 7, 4);
    const uint32_t kind = A >> 30;
    if (kind == fw::NODE_LEAF) { box = item_boxes[A & fw::NODE_MASK]; return fw::REF_LEAF | (A & fw::NODE_MASK); }
    const uint32_t me = out.count();
    out.nodes.resize(out.nodes.size() + 16, 0.f);
    out.depth = std::max(out.depth, depth + 1);
    uint32_t rl, rr; Box bl, br;
    if (kind == fw::NODE_DOUBLE) {
        rl = fw::REF_LEAF | (A & fw::NODE_MASK); bl = item_boxes[A & fw::NODE_MASK];
        rr = fw::REF_LEAF | B; br = item_boxes[B];
    } else {
        rl = pair_convert_rec(src, i + 1, item_boxes, out, base, depth + 1, bl);
        rr = pair_convert_rec(src, A & fw::NODE_MASK, item_boxes, out, base, depth + 1, br);
    }
    box = box_union(bl, br);       // (= the node's own box when `src` was built over item_boxes; the reference tree walked as it is — BVH=median — was built over the items' exact boxes)
    float *p = &out.nodes[(size_t)me * 16];
    p[0] = bl.mn.x; p[1] = bl.mn.y; p[2] = bl.mn.z; p[3] = bits_f(rl);
    p[4] = bl.mx.x; p[5] = bl.mx.y; p[6] = bl.mx.z; p[7] = bits_f(rr);
    p[8] = br.mn.x; p[9] = br.mn.y; p[10] = br.mn.z; p[12] = br.mx.x; p[13] = br.mx.y; p[14] = br.mx.z;
    return base + me;
}
// appends the converted tree to `out` (whose nodes already hold `base` = out.count() pair nodes of other trees) and returns its root reference
static uint32_t pair_convert(const FlatBvh &src, const std::vector<Box> &item_boxes, PairBvh &out) {
    Box root;
    PairBvh local;
    uint32_t base = out.count();
    uint32_t ref = pair_convert_rec(src, 0, item_boxes, local, base, 0, root);
    out.nodes.insert(out.nodes.end(), local.nodes.begin(), local.nodes.end());
    out.depth = std::max(out.depth, local.depth);
    return ref;
}
inline bool use_sah() { return !options().bvh_median; }

// ---- WIDE nodes (fw_device.h): the tree the LDS-resident walks step through since round 4 — four children per node, collapsed
// from the same binary tree the pair nodes come from (a child is opened, largest box first, until four are held; a DoubleLeaf
// opens into its two items with their own boxes).  One step decides four boxes for one stack operation and one trip round the
// walk's loop, and the tree has a third of the pair tree's nodes.  Two encodings of the same topology:
//   WIDE_F32 (112 B)  the children's boxes as they are, SoA by plane: the slab test is the pair walk's, an item is reached iff its
//                     own box passes aabb.rs:30-50 — bit for bit what the pair walk decides;
//   WIDE_Q8  (48 B)   boxes quantised to 8 bits per plane relative to the node (origin + q * 2^e per axis), rounded OUTWARD and
//                     checked here with the device's own dequantisation (one fma): a superset of the exact box under the same
//                     monotone slab arithmetic, so no item the exact boxes admit is ever culled.  For trees whose f32 nodes do
//                     not fit a CU's LDS (teapot.yml: 6 320 triangles = 236 KB as f32 nodes, 101 KB quantised).
struct WideBvh {
    std::vector<uint32_t> words;   // fw::WIDE_F32_DW or fw::WIDE_Q8_DW dwords per node
    int fmt = 0;
    uint32_t depth = 0;            // wide nodes on the longest root-to-leaf path
    uint32_t dw() const { return fmt == fw::WIDE_Q8 ? fw::WIDE_Q8_DW : fw::WIDE_F32_DW; }
    uint32_t count() const { return (uint32_t)(words.size() / dw()); }
};
struct WideChild { bool leaf; uint32_t id; Box box; };   // id: item, or node index in the BinTree
// The binary tree the wide nodes are collapsed from: the FlatBvh with a DoubleLeaf opened into a node over two single items, and the
// collapse chosen by dynamic programming over the surface-area cost (Ylitie, Karras, Laine 2017, section 4.1, for 4 slots): cost[k] =
// the cheapest way to hang the subtree into AT MOST k slots of a parent — as one wide node of its own (its box's area x the cost of
// a step, its children spread over four slots), or with its two children spread over the k slots directly.  Opening the child
// with the largest box until four are held (the first version) left a third of the slots empty: a node's small children stayed
// nodes of two leaves (suzanne: 458 nodes for 968 triangles, teapot 3 116 for 6 320: 150 KB quantised, too big for a CU's LDS).
struct BinNode { int left = -1, right = -1; uint32_t item = 0; Box box{}; float cost[5] = {0, 0, 0, 0, 0}; uint8_t split[5] = {0, 0, 0, 0, 0}; bool inherit[5] = {false, false, false, false, false}; };
constexpr float WIDE_COST_STEP = 1.0f, WIDE_COST_ITEM = 0.7f;     // one wide step (four boxes, a stack operation) against one item test
static int bin_build(const FlatBvh &src, uint32_t i, const std::vector<Box> &item_boxes, std::vector<BinNode> &t, float node_cost) {
    const float *nd = &src.nodes[(size_t)i * 8];
    uint32_t A, B; std::memcpy(&A, nd + 3, 4); std::memcpy(&B, nd + 7, 4);
    const uint32_t kind = A >> 30;
    auto leaf = [&](uint32_t item) { BinNode n; n.item = item; n.box = item_boxes[item]; const float c = box_area(n.box) * WIDE_COST_ITEM; for (int k = 1; k <= 4; k++) n.cost[k] = c; t.push_back(n); return (int)t.size() - 1; };
    if (kind == fw::NODE_LEAF) return leaf(A & fw::NODE_MASK);
    BinNode n;
    if (kind == fw::NODE_DOUBLE) { n.left = leaf(A & fw::NODE_MASK); n.right = leaf(B); }
    else { n.left = bin_build(src, i + 1, item_boxes, t, node_cost); n.right = bin_build(src, A & fw::NODE_MASK, item_boxes, t, node_cost); }
    const BinNode &l = t[n.left], &r = t[n.right];
    n.box = box_union(l.box, r.box);
    float dist[5] = {0, 0, 0, 0, 0};
    for (int k = 2; k <= 4; k++) {
        dist[k] = 3.0e38f;
        for (int j = 1; j < k; j++) { const float c = l.cost[j] + r.cost[k - j]; if (c < dist[k]) { dist[k] = c; n.split[k] = (uint8_t)j; } }
    }
    n.cost[1] = box_area(n.box) * WIDE_COST_STEP + node_cost + dist[4];
    for (int k = 2; k <= 4; k++) { n.inherit[k] = !(dist[k] < n.cost[k - 1]); n.cost[k] = n.inherit[k] ? n.cost[k - 1] : dist[k]; }
    t.push_back(n);
    return (int)t.size() - 1;
}
static void wide_place(const std::vector<BinNode> &t, int m, int k, std::vector<WideChild> &out);
static void wide_spread(const std::vector<BinNode> &t, int n, int k, std::vector<WideChild> &out) {     // n's two children over k slots
    const int j = t[n].split[k];
    wide_place(t, t[n].left, j, out);
    wide_place(t, t[n].right, k - j, out);
}
static void wide_place(const std::vector<BinNode> &t, int m, int k, std::vector<WideChild> &out) {      // subtree m into at most k slots
    if (t[m].left < 0) { out.push_back({true, t[m].item, t[m].box}); return; }
    while (k > 1 && t[m].inherit[k]) k--;
    if (k == 1) out.push_back({false, (uint32_t)m, t[m].box});
    else wide_spread(t, m, k, out);
}
inline uint32_t f_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
// one axis of a WIDE_Q8 node: exponent byte e (scale 2^(e-127)) and the children's quantised planes, rounded outward
static bool wide_quantise_axis(const float *lo, const float *hi, int n, float org, uint32_t &e_out, uint8_t *qlo, uint8_t *qhi) {
    float ext = 0.f;
    for (int c = 0; c < n; c++) { if (!(lo[c] >= org) || !(hi[c] >= lo[c]) || !std::isfinite(hi[c])) return false; ext = std::fmax(ext, hi[c] - org); }
    int e = 1;                                               // smallest scale with every plane <= 254 quanta from the origin (255 is kept for the rounding step)
    if (ext > 0.f) { int ex; std::frexp(ext / 254.f, &ex); e = std::max(1, std::min(254, ex + 127)); }
    if (org != 0.f) { int eo; std::frexp(std::fabs(org), &eo); e = std::max(e, std::min(254, eo + 127 - 22)); }   // a quantum the origin can see: 255 quanta must move it (a free slot's inverted box)
    for (; e <= 254; e++) {
        const float s = bits_f((uint32_t)e << 23);
        bool ok = true;
        for (int c = 0; c < n && ok; c++) {
            double ql = std::floor(((double)lo[c] - (double)org) / (double)s), qh = std::ceil(((double)hi[c] - (double)org) / (double)s);
            if (ql < 0) ql = 0;
            if (ql > 255 || qh > 255) { ok = false; break; }
            int a = (int)ql, b = (int)qh;
            while (a > 0 && std::fmaf((float)a, s, org) > lo[c]) a--;          // the device's own dequantisation: one fma
            while (b < 255 && std::fmaf((float)b, s, org) < hi[c]) b++;
            if (std::fmaf((float)a, s, org) > lo[c] || std::fmaf((float)b, s, org) < hi[c]) { ok = false; break; }
            qlo[c] = (uint8_t)a; qhi[c] = (uint8_t)b;
        }
        if (ok) { e_out = (uint32_t)e; return true; }
    }
    return false;
}
// returns the node's index in `out` (relative to `base` nodes of other trees already there), or 0xffffffff when the tree cannot be encoded
static uint32_t wide_build_rec(const std::vector<BinNode> &t, int bin, WideBvh &out, uint32_t base, uint32_t depth) {
    std::vector<WideChild> ch;
    wide_spread(t, bin, 4, ch);
    const uint32_t me = out.count(), dw = out.dw();
    out.words.resize(out.words.size() + dw, 0u);
    out.depth = std::max(out.depth, depth + 1);
    uint32_t refs[4];
    const int n = (int)ch.size();
    for (int c = 0; c < n; c++) {
        if (ch[c].leaf) { if (ch[c].id >= 0x7fffu) return 0xffffffffu; refs[c] = fw::W_LEAF | ch[c].id; }
        else {
            const uint32_t r = wide_build_rec(t, (int)ch[c].id, out, base, depth + 1);
            if (r == 0xffffffffu || base + r >= 0x8000u) return 0xffffffffu;
            refs[c] = base + r;
        }
    }
    for (int c = n; c < 4; c++) refs[c] = refs[0];           // a free slot repeats child 0 behind a box no finite ray can hit
    uint32_t *w = &out.words[(size_t)me * dw];
    if (out.fmt == fw::WIDE_F32) {                            // planes: lo.x[4] lo.y[4] lo.z[4] hi.x[4] hi.y[4] hi.z[4], then the refs
        const float INF = std::numeric_limits<float>::infinity();
        for (int c = 0; c < 4; c++) {
            const bool on = c < n;
            const Box &b = ch[on ? c : 0].box;
            w[0 + c] = f_bits(on ? b.mn.x : INF); w[4 + c] = f_bits(on ? b.mn.y : INF); w[8 + c] = f_bits(on ? b.mn.z : INF);
            w[12 + c] = f_bits(on ? b.mx.x : -INF); w[16 + c] = f_bits(on ? b.mx.y : -INF); w[20 + c] = f_bits(on ? b.mx.z : -INF);
        }
        w[24] = refs[0] | (refs[1] << 16); w[25] = refs[2] | (refs[3] << 16); w[26] = w[27] = 0u;
    } else {
        V3 org = ch[0].box.mn;
        for (int c = 1; c < n; c++) org = vmin(org, ch[c].box.mn);
        uint32_t e[3]; uint8_t ql[3][4], qh[3][4];
        for (int a = 0; a < 3; a++) {
            float lo[4], hi[4];
            for (int c = 0; c < n; c++) { lo[c] = ch[c].box.mn[a]; hi[c] = ch[c].box.mx[a]; }
            if (!wide_quantise_axis(lo, hi, n, org[a], e[a], ql[a], qh[a])) return 0xffffffffu;
            for (int c = n; c < 4; c++) { ql[a][c] = 255; qh[a][c] = 0; }
        }
        w[0] = f_bits(org.x); w[1] = f_bits(org.y); w[2] = f_bits(org.z); w[3] = e[0] | (e[1] << 8) | (e[2] << 16);
        for (int a = 0; a < 3; a++) {
            w[4 + a] = (uint32_t)ql[a][0] | ((uint32_t)ql[a][1] << 8) | ((uint32_t)ql[a][2] << 16) | ((uint32_t)ql[a][3] << 24);
            w[8 + a] = (uint32_t)qh[a][0] | ((uint32_t)qh[a][1] << 8) | ((uint32_t)qh[a][2] << 16) | ((uint32_t)qh[a][3] << 24);
        }
        w[7] = refs[0] | (refs[1] << 16); w[11] = refs[2] | (refs[3] << 16);
    }
    return me;
}
// appends the wide form of `src` to `out`; returns its root reference (a node, or W_LEAF | item for a one-item tree), 0xffffffff on failure
static uint32_t wide_convert(const FlatBvh &src, const std::vector<Box> &item_boxes, WideBvh &out) {
    uint32_t A; std::memcpy(&A, &src.nodes[3], 4);
    if ((A >> 30) == fw::NODE_LEAF) return (A & fw::NODE_MASK) < 0x7fffu ? (fw::W_LEAF | (A & fw::NODE_MASK)) : 0xffffffffu;
    if (item_boxes.size() >= 0x7fffu) return 0xffffffffu;          // 15-bit item references: not encodable (found only at the last leaf otherwise: 90 ms for a million triangles)
    WideBvh local; local.fmt = out.fmt;
    const uint32_t base = out.count();
    std::vector<BinNode> bin;
    bin.reserve(2 * item_boxes.size());
    // a node also costs LDS: a constant per node (WIDE_COST_NODE of the root's area) makes the collapse prefer full nodes where the
    // surface-area terms are indifferent (teapot: 2 453 -> FILL nodes, which is what lets sixteen waves' stacks fit beside them)
    const float *rn = &src.nodes[0];
    const float node_cost = (float)options().wide_node_cost * box_area(Box{{rn[0], rn[1], rn[2]}, {rn[4], rn[5], rn[6]}});
    const int root = bin_build(src, 0, item_boxes, bin, node_cost);
    const uint32_t r = wide_build_rec(bin, root, local, base, 0);
    if (r == 0xffffffffu) return r;
    out.words.insert(out.words.end(), local.words.begin(), local.words.end());
    out.depth = std::max(out.depth, local.depth);
    return base + r;
}
// Every invariant the walks rely on, checked on the finished tree (fw_selftest_wide_bvh; the CPU test suite runs it on the
// reference's meshes): each item is the leaf of exactly one slot, a child's box as the DEVICE decodes it contains the exact box,
// node references point forward, free slots cannot be hit.  Returns the number of violations.
static uint32_t wide_check(const WideBvh &t, uint32_t root, const std::vector<Box> &item_boxes, uint32_t stats[4]) {
    uint32_t bad = 0, leaves = 0, free_slots = 0;
    std::vector<uint32_t> seen(item_boxes.size(), 0);
    if (root & fw::W_LEAF) { stats[0] = 0; stats[1] = 1; stats[2] = 0; stats[3] = 0; return (root & 0x7fffu) == 0 && item_boxes.size() == 1 ? 0u : 1u; }
    struct Rec { uint32_t node; Box bound; bool has_bound; };
    std::vector<Rec> todo{{root, Box{}, false}};
    std::vector<uint32_t> visited(t.count(), 0);
    auto subtree_box = [&](auto &&self, uint32_t ref) -> Box {
        if (ref & fw::W_LEAF) return item_boxes[ref & 0x7fffu];
        const uint32_t *w = &t.words[(size_t)ref * t.dw()];
        const uint32_t r[4] = {(t.fmt == fw::WIDE_F32 ? w[24] : w[7]) & 0xffffu, (t.fmt == fw::WIDE_F32 ? w[24] : w[7]) >> 16,
                               (t.fmt == fw::WIDE_F32 ? w[25] : w[11]) & 0xffffu, (t.fmt == fw::WIDE_F32 ? w[25] : w[11]) >> 16};
        Box b = self(self, r[0]);
        for (int c = 1; c < 4; c++) if (r[c] != r[0]) b = box_union(b, self(self, r[c]));
        return b;
    };
    while (!todo.empty()) {
        const Rec rec = todo.back(); todo.pop_back();
        if (rec.node >= t.count() || visited[rec.node]++) { bad++; continue; }
        const uint32_t *w = &t.words[(size_t)rec.node * t.dw()];
        uint32_t r[4]; Box dec[4];
        if (t.fmt == fw::WIDE_F32) {
            r[0] = w[24] & 0xffffu; r[1] = w[24] >> 16; r[2] = w[25] & 0xffffu; r[3] = w[25] >> 16;
            for (int c = 0; c < 4; c++) dec[c] = Box{{bits_f(w[c]), bits_f(w[4 + c]), bits_f(w[8 + c])}, {bits_f(w[12 + c]), bits_f(w[16 + c]), bits_f(w[20 + c])}};
        } else {
            r[0] = w[7] & 0xffffu; r[1] = w[7] >> 16; r[2] = w[11] & 0xffffu; r[3] = w[11] >> 16;
            const float org[3] = {bits_f(w[0]), bits_f(w[1]), bits_f(w[2])};
            for (int c = 0; c < 4; c++) {
                float lo[3], hi[3];
                for (int a = 0; a < 3; a++) {
                    const float s = bits_f(((w[3] >> (8 * a)) & 0xffu) << 23);
                    lo[a] = std::fmaf((float)((w[4 + a] >> (8 * c)) & 0xffu), s, org[a]);
                    hi[a] = std::fmaf((float)((w[8 + a] >> (8 * c)) & 0xffu), s, org[a]);
                }
                dec[c] = Box{{lo[0], lo[1], lo[2]}, {hi[0], hi[1], hi[2]}};
            }
        }
        for (int c = 0; c < 4; c++) {
            const bool is_free = c > 0 && r[c] == r[0];
            if (is_free) { free_slots++; if (!(dec[c].mn.x > dec[c].mx.x && dec[c].mn.y > dec[c].mx.y && dec[c].mn.z > dec[c].mx.z)) bad++; continue; }
            const Box exact = subtree_box(subtree_box, r[c]);
            if (!(dec[c].mn.x <= exact.mn.x && dec[c].mn.y <= exact.mn.y && dec[c].mn.z <= exact.mn.z &&
                  dec[c].mx.x >= exact.mx.x && dec[c].mx.y >= exact.mx.y && dec[c].mx.z >= exact.mx.z)) bad++;
            if (t.fmt == fw::WIDE_F32 && (r[c] & fw::W_LEAF) && std::memcmp(&dec[c], &item_boxes[r[c] & 0x7fffu], sizeof(Box)) != 0) bad++;   // an item's own box, bit for bit
            if (r[c] & fw::W_LEAF) { const uint32_t it = r[c] & 0x7fffu; if (it >= seen.size() || seen[it]++) bad++; leaves++; }
            else { if (r[c] <= rec.node) bad++; todo.push_back({r[c], Box{}, false}); }
        }
    }
    for (uint32_t s : seen) if (s != 1) bad++;
    for (uint32_t v : visited) if (v != 1) bad++;
    stats[0] = t.count(); stats[1] = leaves; stats[2] = free_slots; stats[3] = t.depth;
    return bad;
}

// ---- device allocations owned by a scene / workspace ----------------------------------------------------
struct DevBuf {
    void *p = nullptr; size_t bytes = 0;
    int alloc(size_t n) {
        if (n <= bytes && p) return FW_OK;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        if (n == 0) return FW_OK;
        HIPCHK(hipMalloc(&p, n));
        bytes = n;
        return FW_OK;
    }
    int upload(const void *src, size_t n) {
        int rc = alloc(n ? n : 16);
        if (rc) return rc;
        if (n) {   // async copy + explicit wait: the blocking hipMemcpy of pageable memory showed 20-30 ms stalls
            HIPCHK(hipMemcpyAsync(p, src, n, hipMemcpyHostToDevice, nullptr));
            HIPCHK(hipStreamSynchronize(nullptr));
        }
        return FW_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};


// Per-device wavefront workspace: the path pools, accumulators and events.  It is the library's only global state
// (created lazily, kept until fw_release_workspace or process exit): a one-shot `Renderer::render(scene)` would
// otherwise hipMalloc/hipFree tens of GB per call, which costs from 8 ms to seconds (fresh pages are cleared).
// Optionally (FIREWORK_STREAMS=n) up to MAX_LANES batches are in flight on their own streams, each with its own
// path pool, so that one batch's k_extend (VALU-bound) overlaps another's k_shade (HBM-bound).  Results do not
// depend on n (batches are accumulated in order).  Default 1: the measured gain is ~3 %.
struct Workspace {
    static constexpr int MAX_LANES = 4;
    // exact_slots: the two slot lists of the exact walk (DExact.slots) + their counters behind them
    // A lane's buffers are slices of ONE device allocation per workspace (`arena`, laid out per call by render_impl): with a DevBuf
    // each, a call that needed other sizes than the last one — one batch in flight after two, no parked rays after some — freed
    // and allocated two dozen multi-GB buffers, 2 s in the caller's timed region (profiles/r03z_oneshot_trace.txt: the first
    // volume frame after suzanne).  The arena only grows, in steps of 1 GiB.
    struct Lane { void *ray_a[2] = {nullptr, nullptr}, *ray_b[2] = {nullptr, nullptr}, *state[2] = {nullptr, nullptr}, *hits = nullptr, *sample_rad = nullptr,
                       *wcount = nullptr, *park_a = nullptr, *park_b = nullptr, *park_m = nullptr, *pcount = nullptr, *dep_bits = nullptr, *exact_slots = nullptr, *atten = nullptr;
                  hipStream_t stream = nullptr;
                  std::vector<hipEvent_t> events; };
    DevBuf arena;
    std::mutex mu;                        // one fw_render at a time per device
    Lane lanes[MAX_LANES];
    DevBuf accum, totals, pixel_ids, out_rgb8, out_gamma, out_linear;
    DevBuf scene_cache;                   // the last destroyed scene's allocation, reused by the next fw_scene_create
    std::vector<hipEvent_t> events;       // [0] frame start, [1] frame stop, [2] fork, [3..] per-batch "accumulated" events
    hipEvent_t ev_d2h = nullptr;          // after the device -> host copies of the outputs
    // GRAPH: the launches of one frame between fork and join, captured the second time the same frame is asked for and replayed from then on.
    // `key` is a hash of every by-value kernel argument struct of the frame (camera, frame, queue, launch configuration, scene) and of
    // the buffers they do not name: anything that changes what a launch would be changes it, and a changed key is only ever a miss.
    struct FrameGraph { uint64_t key = 0, seen = 0; hipGraphExec_t exec = nullptr; hipStream_t origin = nullptr; bool broken = false; fw::DFrame fr_after{}; } fg;
    std::vector<hipEvent_t> phase_events; // PHASE_LOCK: [lane-in-group][segment] "this batch's extend of the segment has finished"
    DevBuf tile_ids; uint32_t tile_w = 0, tile_h = 0;   // the library's own 16x16-tile pixel order of a (tile_w x tile_h) frame
    void *staging = nullptr; size_t staging_bytes = 0;  // pinned host memory the scene blob is assembled in (k_upload reads it)
    void *host_out = nullptr; size_t host_out_bytes = 0; // pinned host memory the counters and output frames are copied into
    hipEvent_t ev_upload = nullptr;       // after the latest scene upload on this device: renders wait for it in stream order
    hipStream_t upload_stream = nullptr;  // the upload kernel's own non-blocking stream: a launch on the legacy NULL stream would
                                          // synchronise with every blocking stream of the process (torch's default stream included)
    bool inited = false;                  // init_device_locked has run on this device (fw_init, or the first call that needed the device)
    void release() {
        inited = false;
        if (ev_d2h) { (void)hipEventDestroy(ev_d2h); ev_d2h = nullptr; }
        if (fg.exec) { (void)hipGraphExecDestroy(fg.exec); fg.exec = nullptr; } if (fg.origin) { (void)hipStreamDestroy(fg.origin); fg.origin = nullptr; } fg.key = fg.seen = 0; fg.broken = false;
        if (ev_upload) { (void)hipEventSynchronize(ev_upload); (void)hipEventDestroy(ev_upload); ev_upload = nullptr; }
        if (upload_stream) { (void)hipStreamDestroy(upload_stream); upload_stream = nullptr; }
        if (staging) { (void)hipHostFree(staging); staging = nullptr; staging_bytes = 0; }
        if (host_out) { (void)hipHostFree(host_out); host_out = nullptr; host_out_bytes = 0; }
        tile_ids.release(); tile_w = tile_h = 0;
        for (DevBuf *b : {&accum, &totals, &pixel_ids, &out_rgb8, &out_gamma, &out_linear, &scene_cache, &arena}) b->release();
        for (Lane &l : lanes) {
            for (hipEvent_t e : l.events) (void)hipEventDestroy(e);
            l.events.clear();
            if (l.stream) (void)hipStreamDestroy(l.stream);
            l.stream = nullptr;
        }
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
        events.clear();
        for (hipEvent_t e : phase_events) (void)hipEventDestroy(e);
        phase_events.clear();
    }
};
constexpr int MAX_DEVICES = 64;
Workspace *workspace_for(int device) {
    static Workspace *table[MAX_DEVICES] = {};
    static std::mutex table_mu;
    if (device < 0 || device >= MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> g(table_mu);
    if (!table[device]) table[device] = new (std::nothrow) Workspace();   // intentionally never deleted (process lifetime)
    return table[device];
}

// hipGetDeviceProperties costs up to ~25 ms per call: ask once per device
int device_cus(int device) {
    static int cu_cache[MAX_DEVICES] = {};
    static std::mutex mu;
    std::lock_guard<std::mutex> g(mu);
    if (device < 0 || device >= MAX_DEVICES) return 256;
    if (cu_cache[device] == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) return -1;
        cu_cache[device] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return cu_cache[device];
}

// ---- initialisation (round 5: fw_init).  A process's first call pays for things that are no part of any frame: the HIP context, the
// device query, the load of this library's code objects (its first kernel launch), every kernel's first resolution, the pinned
// staging buffer, the streams — 150-180 ms of the 213 ms of a cold cornell frame (profiles/r03z_oneshot_trace.txt) — and, where a
// big frame follows, the path arena, whose hipMalloc takes 0.4 ms on a device whose memory is clean and 0.2-1.4 s where the driver
// has pages to clear (profiles/r04z_bench.json).  Round 4 did all of it in a static initialiser, so that merely loading the library
// created a context and took 64 GiB on LOCAL_RANK's device.  Now LOADING THE LIBRARY MAKES NO HIP CALL: the host calls
// fw_init(device, arena_bytes) where it wants the cost (the CLI and bench.py do, before their timed regions), and a host that never
// does gets the same initialisation — without an arena — from its first fw_scene_create / fw_render on that device.
// Caller holds ws->mu and has made `dev` current.
int init_device_locked(Workspace *ws, int dev) {
    if (ws->inited) return FW_OK;
    if (device_cus(dev) <= 0) return fail(FW_ERR_HIP, "hipGetDeviceProperties failed");
    if (!ws->staging) { HIPCHK(hipHostMalloc(&ws->staging, 1 << 20, hipHostMallocDefault)); ws->staging_bytes = 1 << 20; }
    if (!ws->upload_stream) HIPCHK(hipStreamCreateWithFlags(&ws->upload_stream, hipStreamNonBlocking));
    if (!ws->ev_upload) HIPCHK(hipEventCreateWithFlags(&ws->ev_upload, hipEventDisableTiming));
    for (int l = 0; l < 2; l++) if (!ws->lanes[l].stream) HIPCHK(hipStreamCreateWithFlags(&ws->lanes[l].stream, hipStreamNonBlocking));
    void *tmp = nullptr;
    HIPCHK(hipMalloc(&tmp, 256));
    std::memset(ws->staging, 0, 256);
    fw::launch_upload(ws->upload_stream, ws->staging, tmp, 256);          // any launch loads the code objects of the whole library on this device
    fw::preload_kernels();                                                // ... and every kernel's first launch resolves it: 60 ms of a first frame's enqueue (gpurun_out/r04j)
    (void)hipEventRecord(ws->ev_upload, ws->upload_stream);
    const hipError_t e = hipStreamSynchronize(ws->upload_stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(FW_ERR_HIP, std::string("first launch failed: ") + hipGetErrorString(e)); }
    (void)hipGetLastError();
    ws->inited = true;
    return FW_OK;
}
// The arena grows by allocating the new one FIRST and freeing the old one on a background thread: hipFree of a 35 GB arena took 2.4 s of
// the caller's time (gpurun_out/r04j/oneshot.txt) while the allocation itself is lazy.  Only when both do not fit is the old one freed
// in line.  Caller holds ws->mu; no render of this workspace is in flight (fw_render returns after its stream has drained).
int arena_reserve_locked(Workspace *ws, int dev, size_t want) {
    if (want <= ws->arena.bytes && ws->arena.p) return FW_OK;
    void *old_p = ws->arena.p;
    void *np = nullptr;
    hipError_t e = hipMalloc(&np, want);
    if (e != hipSuccess && old_p) {         // no room for both
        (void)hipGetLastError();
        (void)hipFree(old_p); old_p = nullptr; ws->arena.p = nullptr; ws->arena.bytes = 0;
        e = hipMalloc(&np, want);
    }
    if (e != hipSuccess) { (void)hipGetLastError(); ws->arena.p = old_p; if (!old_p) ws->arena.bytes = 0; return fail(FW_ERR_OOM, "path arena allocation failed: " + std::to_string(want >> 20) + " MiB"); }
    ws->arena.p = np; ws->arena.bytes = want;
    if (old_p) std::thread([old_p, dev] { if (hipSetDevice(dev) == hipSuccess) (void)hipFree(old_p); }).detach();
    return FW_OK;
}
size_t default_arena_bytes(const Workspace *ws) {      // what fw_init(device, 0) reserves: every default-budget frame of the BASELINE configs fits (the largest needs 52 GB)
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return std::min<size_t>((size_t)64 << 30, (free_b + ws->arena.bytes) / 3) & ~(((size_t)1 << 30) - 1);
}

} // namespace

struct fw_scene {
    int device = 0;
    int n_cus = 256;
    fw::DScene d{};
    DevBuf data;   // every scene array in one allocation (sections 256-byte aligned)
    uint32_t tlas_nodes = 0, blas_nodes = 0, tlas_depth = 0, blas_depth = 0, n_mat = 0, n_tex = 0;
    uint32_t blas_pair_nodes = 0, tlas_pair_nodes = 0, max_tris = 0, n_tris = 0;
    uint32_t n_defer = 0;
    uint32_t chain_bits = 0;                // != 0: the scene's paths can carry their material ids instead of a running product (DFrame.chain_bits)
    int wblas_fmt = 0, wtlas_fmt = 0;       // fw::WIDE_*: the encoding of the wide trees uploaded with this scene (0: none)
    uint32_t wblas_nodes = 0, wtlas_nodes = 0, wblas_depth = 0, wtlas_depth = 0;
    bool has_expensive = false;   // some material is a dielectric or carries a non-constant texture, or the environment is an HDR map (k_shade's list)
    bool simple_shapes = false;   // every object is a sphere, an axis-aligned rect or a Rect3d (no medium, mesh, cone, cylinder, disk)
    bool simple_set = false;      // ... or a medium around a sphere (LaunchCfg.simple_set)
    bool simple_but_meshes = false;   // every object is a sphere, a rect, a Rect3d or a TriangleMesh (LaunchCfg.simple_but_meshes)
    bool hdr_env = false;
    fw::DExact ex{};              // flag rule of the exact walk (bits pointer is per render)
    uint32_t ref_tlas_depth = 0, ref_blas_depth = 0;
    ~fw_scene() {
        data.release();
    }
};

namespace {

struct ShapeParams { float q3[4] = {0, 0, 0, 0}, q4[4] = {0, 0, 0, 0}; uint32_t kind = 0, flags = 0, aux0 = 0, aux1 = 0; Box box{};
                     uint32_t wroot_f32 = 0xffffffffu, wroot_q8 = 0xffffffffu;   // meshes: root reference of the wide tree in either encoding
                     uint32_t ref_root = 0xffffffffu, n_tris = 0;   // meshes: first node of the reference tree in Flattener::ref_blas
                     Box true_box{}; };   // OF_GATE shapes: a box that really encloses the geometry (object space)

struct Flattener {
    const fw_scene_desc *d;
    std::vector<float> tri, tri_attr;     // 12 floats per triangle each
    std::vector<float> tri_gate;          // 8 floats per triangle: the box of its leaf node in the REFERENCE tree of its mesh (own box, or a DoubleLeaf's union)
    std::vector<uint32_t> tri_rank;       // in-order rank of each triangle in the reference tree of its mesh
    bool any_attr = false;
    PairBvh blas;                 // all meshes' trees, as walked on the device
    std::vector<float> ref_blas;  // all meshes' REFERENCE trees (FlatBvh nodes, 8 floats each; child indices relative to the mesh's first node)
    uint32_t ref_blas_depth = 0;
    uint32_t blas_depth = 0, ref_blas_nodes = 0, max_tris = 0;
    WideBvh wblas_f32, wblas_q8;  // all meshes' trees as WIDE nodes, both encodings
    bool wide_ok[2] = {true, true};
    std::vector<ShapeParams> mesh_cache;  // per shape index: a TriangleMesh shape referenced by several objects (or by a medium
    std::vector<uint8_t> mesh_cached;     // and an object) is flattened and built once, every user shares its triangles and BLAS
    double ms_gather = 0, ms_ref = 0, ms_gate = 0, ms_sah = 0, ms_pair = 0, ms_wide = 0;   // where mesh_params spends its time (FIREWORK_TRACE=1, tools/big_mesh.py)

    int check_material(int32_t m) const { return (m < 0 || (uint32_t)m >= d->n_materials) ? FW_ERR_BAD_ARG : FW_OK; }

    // object-space shape -> parameters + bounding box (Hitable::bounding_box of each shape)
    int shape_params(int32_t si, ShapeParams &sp, int nest) {
        if (si < 0 || (uint32_t)si >= d->n_shapes) return fail(FW_ERR_BAD_ARG, "shape index out of range");
        const fw_shape &s = d->shapes[si];
        if (s.kind != FW_SHAPE_CONSTANT_MEDIUM || nest > 0) { if (check_material(s.material)) return fail(FW_ERR_BAD_ARG, "material index out of range"); }
        sp.kind = (uint32_t)s.kind;
        switch (s.kind) {
        case FW_SHAPE_SPHERE:                                        // sphere.rs:62-64
            sp.q3[0] = s.radius;
            sp.box = {{-s.radius, -s.radius, -s.radius}, {s.radius, s.radius, s.radius}};
            // the reference builds the box as -Vec3::one()*r .. Vec3::one()*r: same values
            return FW_OK;
        case FW_SHAPE_XYRECT: case FW_SHAPE_XZRECT: case FW_SHAPE_YZRECT: {   // rect.rs:75-85
            sp.q3[0] = s.a_min; sp.q3[1] = s.a_max; sp.q3[2] = s.b_min; sp.q3[3] = s.b_max; sp.q4[0] = s.k;
            sp.q4[1] = (s.a_min <= s.a_max && s.b_min <= s.b_max) ? 0.f : 1.f;      // an interval with lo > hi admits nothing (hit_rect)
            if (s.flip_normal) sp.flags |= fw::OF_RECT_FLIP;
            int a1 = s.kind == FW_SHAPE_YZRECT ? 1 : 0, a2 = s.kind == FW_SHAPE_XYRECT ? 1 : 2, ot = s.kind == FW_SHAPE_XYRECT ? 2 : (s.kind == FW_SHAPE_XZRECT ? 1 : 0);
            float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
            mn[a1] = s.a_min; mn[a2] = s.b_min; mn[ot] = s.k - 0.01f;
            mx[a1] = s.a_max; mx[a2] = s.b_max; mx[ot] = s.k + 0.01f;
            sp.box = {{mn[0], mn[1], mn[2]}, {mx[0], mx[1], mx[2]}};
            if (sp.q4[1] != 0.f) {   // inverted box: like Disk, reachable under use_bvh only through the reference's leaf-node box
                sp.flags |= fw::OF_GATE;
                sp.true_box = {vmin(sp.box.mn, sp.box.mx), vmax(sp.box.mn, sp.box.mx)};
            }
            return FW_OK; }
        case FW_SHAPE_RECT3D:                                        // rect3d.rs:102-104
            sp.q3[0] = s.pos.x; sp.q3[1] = s.pos.y; sp.q3[2] = s.pos.z; sp.q3[3] = s.size.x; sp.q4[0] = s.size.y; sp.q4[1] = s.size.z;
            // a negative size turns some faces' bounds into lo > hi (hit_rect); pos + size is what the faces use (rect3d.rs:30-75)
            sp.q4[2] = (s.pos.x <= s.pos.x + s.size.x && s.pos.y <= s.pos.y + s.size.y && s.pos.z <= s.pos.z + s.size.z) ? 0.f : 1.f;
            sp.box = {tov(s.pos), tov(s.pos) + tov(s.size)};
            if (sp.q4[2] != 0.f) { sp.flags |= fw::OF_GATE; sp.true_box = {vmin(sp.box.mn, sp.box.mx), vmax(sp.box.mn, sp.box.mx)}; }   // as above
            return FW_OK;
        case FW_SHAPE_CONE: case FW_SHAPE_CYLINDER:                 // cone.rs:90-95, cylinder.rs:92-97
            sp.q3[0] = s.radius; sp.q3[1] = s.height; sp.q3[2] = s.phi_max;
            sp.box = {{-s.radius, 0.f, -s.radius}, {s.radius, s.height, s.radius}};
            return FW_OK;
        case FW_SHAPE_DISK:                                          // disk.rs:85-90 — degenerate box, kept as written
            sp.q3[0] = s.radius; sp.q3[2] = s.phi_max; sp.q3[3] = s.inner_radius;
            sp.box = {{-s.radius, 0.f, s.radius}, {-s.radius, 0.001f, s.radius}};
            sp.flags |= fw::OF_GATE;
            sp.true_box = {{-s.radius, -0.001f, -s.radius}, {s.radius, 0.001f, s.radius}};
            return FW_OK;
        case FW_SHAPE_TRIANGLE_MESH: {
            if (mesh_cached.empty()) { mesh_cached.assign(d->n_shapes, 0); mesh_cache.resize(d->n_shapes); }
            if (mesh_cached[si]) { sp = mesh_cache[si]; return FW_OK; }
            int rc = mesh_params(s, sp);
            if (rc == FW_OK) { mesh_cache[si] = sp; mesh_cached[si] = 1; }
            return rc; }
        case FW_SHAPE_CONSTANT_MEDIUM: {                             // volume.rs:84-86: bbox of the inner shape
            if (nest > 0) return fail(FW_ERR_UNSUPPORTED, "ConstantMedium nested in a ConstantMedium");
            ShapeParams in;
            int rc = shape_params(s.inner, in, nest + 1);
            if (rc) return rc;
            if (in.kind == FW_SHAPE_CONSTANT_MEDIUM) return fail(FW_ERR_UNSUPPORTED, "ConstantMedium nested in a ConstantMedium");
            if (check_material(s.material)) return fail(FW_ERR_BAD_ARG, "material index out of range");
            sp = in;
            sp.kind = FW_SHAPE_CONSTANT_MEDIUM | (in.kind << 16);   // inner kind travels in bits 16..23 until packing
            sp.q4[3] = s.density;
            return FW_OK; }
        default: return fail(FW_ERR_BAD_ARG, "unknown shape kind");
        }
    }

    // mesh.rs:21-30,221-242: gather triangles, build the mesh's own BVH (always, even with use_bvh=false)
    int mesh_params(const fw_shape &s, ShapeParams &sp) {
        if (!s.verts || !s.indices || s.n_indices % 3) return fail(FW_ERR_BAD_ARG, "TriangleMesh needs verts and 3*k indices");
        if (s.n_indices == 0) return fail(FW_ERR_EMPTY_SCENE, "TriangleMesh with no triangles (reference: unbounded recursion in bvh.rs:29-70)");
        uint32_t n_tris = s.n_indices / 3, tri_base = (uint32_t)(tri.size() / 12);
        max_tris = std::max(max_tris, n_tris);
        if ((uint64_t)tri_base + n_tris > fw::NODE_MASK) return fail(FW_ERR_UNSUPPORTED, "too many triangles");
        bool attr = s.normals || s.uvs;
        auto tm = std::chrono::steady_clock::now();
        auto lap = [&](double &acc) { const auto t = std::chrono::steady_clock::now(); acc += std::chrono::duration<double, std::milli>(t - tm).count(); tm = t; };
        std::vector<Box> boxes(n_tris);
        tri.resize(tri.size() + (size_t)n_tris * 12);
        if (attr) any_attr = true;
        tri_attr.resize(tri.size(), 0.f);
        BuildPool pool(host_build_threads() - 1);
        std::atomic<bool> bad_index{false};
        {
            const int helpers = n_tris >= PAR_PASS_MIN ? pool.take(15) : 0;
            par_parts(n_tris, helpers + 1, [&](size_t t_lo, size_t t_hi, int) {
                for (size_t t = t_lo; t < t_hi; t++) {
                    V3 p[3];
                    for (int k = 0; k < 3; k++) {
                        uint32_t vi = s.indices[3 * t + k];
                        if (vi >= s.n_verts) { bad_index = true; return; }
                        p[k] = {s.verts[3 * vi], s.verts[3 * vi + 1], s.verts[3 * vi + 2]};
                        float *o = &tri[(size_t)(tri_base + t) * 12 + 4 * k];
                        o[0] = p[k].x; o[1] = p[k].y; o[2] = p[k].z;
                        o[3] = s.uvs ? s.uvs[2 * vi] : (k == 1 ? 1.f : 0.f);        // default uvs (0,0),(1,0),(0,1) mesh.rs:107
                        float *a = &tri_attr[(size_t)(tri_base + t) * 12 + 4 * k];
                        if (s.normals) { a[0] = s.normals[3 * vi]; a[1] = s.normals[3 * vi + 1]; a[2] = s.normals[3 * vi + 2]; }
                        a[3] = s.uvs ? s.uvs[2 * vi + 1] : (k == 2 ? 1.f : 0.f);
                    }
                    Box b{vmin(p[0], p[1]), vmax(p[0], p[1])};               // from_two_points(p0,p1).expand_to_point(p2)
                    b = {vmin(b.mn, p[2]), vmax(b.mx, p[2])};
                    V3 size = b.mx - b.mn;
                    if (std::fabs(size.x) < 0.001f) { b.mn.x -= 0.001f; b.mx.x += 0.001f; }
                    if (std::fabs(size.y) < 0.001f) { b.mn.y -= 0.001f; b.mx.y += 0.001f; }
                    if (std::fabs(size.z) < 0.001f) { b.mn.z -= 0.001f; b.mx.z += 0.001f; }
                    boxes[t] = b;
                } });
            pool.give(helpers);
        }
        if (bad_index) return fail(FW_ERR_BAD_ARG, "vertex index out of range");
        lap(ms_gather);
        // Round 5: the reference tree and the walked tree are built at the same time, each in parallel below its big nodes (BuildPool: one budget
        // of host threads for both), and the three forms of the walked tree — pair nodes, wide f32, wide q8 — are converted side by side.
        FlatBvh local, walked;
        bool nan_error = false;
        std::exception_ptr ref_error;
        const bool two = n_tris >= PAR_SUBTREE_MIN && pool.take(1) == 1;
        auto build_reference = [&] { try { (void)bvh_build(local, boxes, &pool); } catch (NanError &) { nan_error = true; } catch (...) { ref_error = std::current_exception(); } };
        std::thread ref_thread;
        if (two) ref_thread = std::thread(build_reference); else build_reference();
        // the mesh's own box = the reference root's: the union of every triangle's box (fmin / fmax: the same in any order)
        Box all = boxes[0];
        for (const Box &b : boxes) all = box_union(all, b);
        sp.box = all;
        // the boxes of the walked trees (grown_by: 2^-6 of a typical triangle, at least 2^-14 of the triangle's own extent)
        std::vector<Box> wboxes = boxes;
        {
            const float typ = box_extent(sp.box) / std::sqrt((float)std::max(1u, n_tris));
            for (Box &b : wboxes) b = grown_by(b, std::fmax(std::ldexp(typ, -6), std::ldexp(box_extent(b), -14)));
        }
        const bool sah = use_sah();
        std::exception_ptr sah_error;
        if (sah) { try { sah_build(walked, wboxes, &pool); } catch (...) { sah_error = std::current_exception(); } }
        lap(ms_sah);
        if (two) { ref_thread.join(); pool.give(1); }
        if (nan_error) return fail(FW_ERR_NAN_BBOX, "Float comparison failed in BVH constructor");
        if (ref_error) std::rethrow_exception(ref_error);
        if (sah_error) std::rethrow_exception(sah_error);
        {   // ties are resolved by the reference tree's in-order rank, whatever tree is traversed
            std::vector<uint32_t> rk = reference_ranks(local, n_tris);
            tri_rank.insert(tri_rank.end(), rk.begin(), rk.end());
        }
        ref_blas_nodes += local.count();
        sp.ref_root = (uint32_t)(ref_blas.size() / 8); sp.n_tris = n_tris;
        ref_blas.insert(ref_blas.end(), local.nodes.begin(), local.nodes.end());
        ref_blas_depth = std::max(ref_blas_depth, local.depth);
        lap(ms_ref);
        // Round 4: the reference tests a triangle iff the ray passes every box down to its LEAF NODE, i.e. iff it passes that node's box
        // (bvh.rs:44-52: a DoubleLeaf's two triangles sit behind the union; the ancestors' boxes are supersets under the monotone slab
        // arithmetic).  The walked trees keep the triangles' own, tight boxes — walking the unions costs 30 % (gpurun_out/r04d) — and
        // every hit that would become a ray's best is checked against this rule before it counts (fw_kernels.hip: tri_gate_ok):
        // tri_gate holds each triangle's reference leaf-node box.
        {
            std::vector<Box> gboxes = boxes;
            leaf_node_boxes(local, gboxes);
            tri_gate.resize(tri.size() / 12 * 8, 0.f);
            const int helpers = n_tris >= PAR_PASS_MIN ? pool.take(15) : 0;
            par_parts(n_tris, helpers + 1, [&](size_t t_lo, size_t t_hi, int) {
                for (size_t t = t_lo; t < t_hi; t++) {
                    float *g = &tri_gate[(size_t)(tri_base + t) * 8];
                    g[0] = gboxes[t].mn.x; g[1] = gboxes[t].mn.y; g[2] = gboxes[t].mn.z; g[4] = gboxes[t].mx.x; g[5] = gboxes[t].mx.y; g[6] = gboxes[t].mx.z;
                } });
            pool.give(helpers);
        }
        lap(ms_gate);
        const FlatBvh &wt = sah ? walked : local;      // BVH=median walks the reference's own topology (over the grown boxes)
        uint32_t root = 0, wr[2] = {0xffffffffu, 0xffffffffu};
        for (int f = 0; f < 2; f++) { WideBvh &wb = f == 0 ? wblas_f32 : wblas_q8; if (wb.fmt == 0) wb.fmt = f == 0 ? fw::WIDE_F32 : fw::WIDE_Q8; }
        {
            std::exception_ptr conv_error[3];
            auto convert = [&](int which) {
                try {
                    if (which == 0) { root = pair_convert(wt, wboxes, blas); blas_depth = blas.depth; }      // root reference into the shared BLAS array
                    else if (wide_ok[which - 1]) wr[which - 1] = wide_convert(wt, wboxes, which == 1 ? wblas_f32 : wblas_q8);   // the same tree as WIDE nodes, in both encodings (create_scene_impl keeps one, or none)
                } catch (...) { conv_error[which] = std::current_exception(); }
            };
            const int helpers = n_tris >= PAR_SUBTREE_MIN ? pool.take(2) : 0;
            std::vector<std::thread> th;
            for (int k = 0; k < helpers; k++) th.emplace_back(convert, k + 1);
            convert(0);
            for (int k = helpers; k < 2; k++) convert(k + 1);
            for (auto &t : th) t.join();
            pool.give(helpers);
            for (auto &e : conv_error) if (e) std::rethrow_exception(e);
        }
        lap(ms_pair);
        sp.aux0 = root; sp.aux1 = tri_base;
        for (int f = 0; f < 2; f++) {
            if (!wide_ok[f]) continue;
            if (wr[f] == 0xffffffffu) wide_ok[f] = false;
            (f == 0 ? sp.wroot_f32 : sp.wroot_q8) = wr[f];
        }
        lap(ms_wide);
        if (s.normals) sp.flags |= fw::OF_MESH_NORMALS;
        if (attr) sp.flags |= fw::OF_MESH_ATTR;
        return FW_OK;
    }
};

int create_scene_impl(const fw_scene_desc *desc, int device, fw_scene **out) {
    if (!desc || !out) return fail(FW_ERR_BAD_ARG, "null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(FW_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(FW_ERR_BAD_ARG, "device index out of range");
    if (desc->n_objects == 0 || !desc->objects) return fail(FW_ERR_EMPTY_SCENE, "No render objects added to scene!");
    if (desc->n_objects > fw::NODE_MASK) return fail(FW_ERR_UNSUPPORTED, "too many objects");
    HIPCHK(hipSetDevice(device));
    const int n_cus_dev = device_cus(device);
    if (n_cus_dev <= 0) return fail(FW_ERR_HIP, "hipGetDeviceProperties failed");
    // FIREWORK_TRACE=1: where a scene creation spends its time (host flatten + BVH builds | staging blob | alloc | copy)
    const Options O = options();
    const bool trace = O.trace;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(now() - t).count(); };
    const auto tr0 = now();

    Flattener fl{desc};
    std::vector<float> objs((size_t)desc->n_objects * fw::OBJ_Q * 4, 0.f);
    std::vector<Box> world(desc->n_objects), true_world(desc->n_objects);
    std::vector<uint32_t> obj_ref_blas(desc->n_objects, 0xffffffffu);
    std::vector<uint32_t> obj_wroot_f32(desc->n_objects, 0xffffffffu), obj_wroot_q8(desc->n_objects, 0xffffffffu);
    std::vector<float> obj_size(desc->n_objects, 0.f);       // the exact walk's far rule: extent of an object (a mesh: of a typical triangle)
    fw::DExact ex{};
    bool has_medium = false, has_perlin = false;
    for (uint32_t i = 0; i < desc->n_objects; i++) {
        const fw_object &o = desc->objects[i];
        ShapeParams sp;
        int rc = fl.shape_params(o.shape, sp, 0);
        if (rc) return rc;
        float rows[3][3];
        rotor_rows(o.rotation, rows);
        uint32_t flags = sp.flags;
        float cos_trace = 0.5f * ((rows[0][0] + rows[1][1] + rows[2][2]) - 1.f);     // scene.rs:180-185
        if (cos_trace < 0.999f) flags |= fw::OF_ROTATED;
        auto to_world = [&](const Box &ob) {                                          // scene.rs:177-212
            Box rb = ob;
            if (cos_trace < 0.999f) {
                V3 mn = 10e9f * V3{1, 1, 1}, mx = -10e9f * V3{1, 1, 1};               // scene.rs:188-203
                for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) for (int c = 0; c < 2; c++) {
                    V3 corner{a == 0 ? ob.mn.x : ob.mx.x, b == 0 ? ob.mn.y : ob.mx.y, c == 0 ? ob.mn.z : ob.mx.z};
                    V3 np = mat_mul(rows, corner);
                    mx = {std::fmax(np.x, mx.x), std::fmax(np.y, mx.y), std::fmax(np.z, mx.z)};
                    mn = {std::fmin(np.x, mn.x), std::fmin(np.y, mn.y), std::fmin(np.z, mn.z)};
                }
                rb = {mn, mx};
            }
            return Box{rb.mn + tov(o.position), rb.mx + tov(o.position)};             // scene.rs:207-210
        };
        world[i] = to_world(sp.box);
        if (flags & fw::OF_GATE) true_world[i] = to_world(sp.true_box);
        obj_ref_blas[i] = sp.ref_root;
        if ((sp.kind & 0xffu) == FW_SHAPE_TRIANGLE_MESH) { obj_wroot_f32[i] = sp.wroot_f32; obj_wroot_q8[i] = sp.wroot_q8; }
        {
            const Box &tb = (flags & fw::OF_GATE) ? sp.true_box : sp.box;
            const V3 e = tb.mx - tb.mn;
            obj_size[i] = std::fmax(std::fabs(e.x), std::fmax(std::fabs(e.y), std::fabs(e.z)));
            if (sp.ref_root != 0xffffffffu) {
                obj_size[i] /= std::sqrt((float)std::max(1u, sp.n_tris));
                if (cos_trace < 0.999f) {      // a rotated mesh: its frame is one the ill-direction test has to look at
                    bool known = false;
                    for (uint32_t f = 0; f < ex.n_frames && !known; f++) known = std::memcmp(ex.frames[f], rows, 36) == 0;
                    if (!known && ex.n_frames < 4) { std::memcpy(ex.frames[ex.n_frames], rows, 36); ex.n_frames++; }
                }
            }
        }
        if (o.flip_normals) flags |= fw::OF_FLIP;
        uint32_t kind = sp.kind & 0xffu, inner = (sp.kind >> 16) & 0xffu;
        if (kind == FW_SHAPE_CONSTANT_MEDIUM) has_medium = true;
        int32_t material = desc->shapes[o.shape].material;
        float *q = &objs[(size_t)i * fw::OBJ_Q * 4];
        const float pos[3] = {o.position.x, o.position.y, o.position.z};
        if ((kind == FW_SHAPE_RECT3D && sp.q4[2] == 0.f) || kind == FW_SHAPE_TRIANGLE_MESH || kind == FW_SHAPE_CONE || kind == FW_SHAPE_CYLINDER) flags |= fw::OF_CULL0;
        q[0] = pos[0]; q[1] = pos[1]; q[2] = pos[2]; q[3] = bits_f(kind | (flags << 8) | (inner << 24));
        std::memcpy(q + 4, sp.q3, 16);
        std::memcpy(q + 8, sp.q4, 16);
        const uint32_t tail[3] = {(uint32_t)material, sp.aux0, sp.aux1};
        for (int r = 0; r < 3; r++) { q[12 + 4 * r] = rows[r][0]; q[13 + 4 * r] = rows[r][1]; q[14 + 4 * r] = rows[r][2]; q[15 + 4 * r] = bits_f(tail[r]); }
    }
    // packed hit records: object << prim_bits | primitive must fit 32 bits with all-ones left for MISS
    uint32_t prim_bits = 3;                                   // rect3d faces 0..5
    while (prim_bits < 31 && (1ull << prim_bits) < (uint64_t)fl.max_tris) prim_bits++;
    {
        uint32_t obj_bits = 1; while (obj_bits < 32 && (1ull << obj_bits) < (uint64_t)desc->n_objects + 1) obj_bits++;
        if (obj_bits + prim_bits > 32) return fail(FW_ERR_UNSUPPORTED, "objects x triangles-per-mesh exceed the 32-bit hit code");
    }
    FlatBvh tlas;
    try { bvh_build(tlas, world); } catch (NanError &) { return fail(FW_ERR_NAN_BBOX, "Float comparison failed in BVH constructor"); }
    std::vector<uint32_t> obj_rank = reference_ranks(tlas, desc->n_objects);
    uint32_t ref_tlas_nodes = tlas.count();
    const std::vector<float> ref_tlas = tlas.nodes;          // the reference's own tree: what k_extend_exact walks
    const uint32_t ref_tlas_depth = tlas.depth;
    {   // the exact walk's flag rule (fw_device.h DExact)
        if (!fl.tri.empty()) ex.mode |= 1u;
        float min_size = 3.0e38f;
        for (uint32_t i = 0; i < desc->n_objects; i++) if (obj_size[i] > 0.f) min_size = std::fmin(min_size, obj_size[i]);
        if (min_size < 3.0e38f) {
            Box cl{{3e38f, 3e38f, 3e38f}, {-3e38f, -3e38f, -3e38f}};
            for (uint32_t i = 0; i < desc->n_objects; i++)
                if (obj_size[i] > 0.f && obj_size[i] <= 16.f * min_size) cl = box_union(cl, world[i].mn.x <= world[i].mx.x ? world[i] : Box{vmin(world[i].mn, world[i].mx), vmax(world[i].mn, world[i].mx)});
            const V3 c = box_center(cl), h = cl.mx - c;
            const float radius = std::fmax(h.x, std::fmax(h.y, h.z));
            ex.far_c[0] = c.x; ex.far_c[1] = c.y; ex.far_c[2] = c.z;
            ex.far_r = std::fmax((float)O.exact_far_x * min_size, 2.f * radius);  // noise / signal of a sphere's discriminant = 2^-23 (|o| / r)^2 = 2^-23 (2 |o| / size)^2: 1/2 at this distance
                                                                    // (128 x flagged every camera ray of teapot.rs, whose camera sits 16 units from triangles of 0.08)
            const float pad = 2.f * min_size + 1e-3f * radius;
            ex.box_lo[0] = cl.mn.x - pad; ex.box_lo[1] = cl.mn.y - pad; ex.box_lo[2] = cl.mn.z - pad;
            ex.box_hi[0] = cl.mx.x + pad; ex.box_hi[1] = cl.mx.y + pad; ex.box_hi[2] = cl.mx.z + pad;
            ex.mode |= 2u;                                          // used under use_bvh only (render_impl)
        }
        ex.shear = std::ldexp(1.f, -O.exact_shear_log2);
        if (O.no_exact) ex.mode = 0;
        if (O.exact_all) ex.mode |= 4u;   // every ray takes the exact walk (parity tool / tests)
    }
    // gate boxes: the box of each object's leaf node in the reference tree (own box for a Leaf, the union for a
    // DoubleLeaf).  In the reference an object is tested iff the ray hits that box (ancestors are supersets), which
    // matters for shapes whose own box does not enclose them (OF_GATE): they stay exactly as (in)visible as there.
    // Round 4: the reference tests an object iff the ray passes the box of its LEAF NODE in the reference tree (the union for a
    // DoubleLeaf: bvh.rs:44-52; the ancestors' boxes are supersets).  The walked trees keep the objects' own boxes (the unions cost
    // part2 26 %: gpurun_out/r04d); every object hit is checked against its reference leaf-node box (obj_gate) before it counts, and
    // a mesh before its rays are parked (fw_kernels.hip: obj_gate_ok).
    std::vector<float> gate((size_t)desc->n_objects * 8, 0.f);
    std::vector<Box> build_boxes = world, own_boxes = world;
    for (uint32_t i = 0; i < tlas.count(); i++) {
        const float *nd = &tlas.nodes[(size_t)i * 8];
        uint32_t A, B; std::memcpy(&A, nd + 3, 4); std::memcpy(&B, nd + 7, 4);
        uint32_t kind = A >> 30;
        if (kind == 0) continue;
        uint32_t items[2] = {A & fw::NODE_MASK, B};
        for (uint32_t q = 0; q < (kind == fw::NODE_DOUBLE ? 2u : 1u); q++) {
            float *g = &gate[(size_t)items[q] * 8];
            g[0] = nd[0]; g[1] = nd[1]; g[2] = nd[2]; g[4] = nd[4]; g[5] = nd[5]; g[6] = nd[6];
            uint32_t kf; std::memcpy(&kf, &objs[(size_t)items[q] * fw::OBJ_Q * 4 + 3], 4);
            const Box node_box{{nd[0], nd[1], nd[2]}, {nd[4], nd[5], nd[6]}};
            // a gated object (its own box does not enclose it): the leaf node's box (so the gate test is reachable) united with bounds
            // that really enclose the geometry (so culling against the best t so far stays valid)
            if ((kf >> 8) & fw::OF_GATE) build_boxes[items[q]] = own_boxes[items[q]] = box_union(node_box, true_world[items[q]]);
        }
    }
    for (uint32_t i = 0; i < desc->n_objects; i++) {             // the walked trees' boxes (grown_by: what a walk must still reach)
        uint32_t kf; std::memcpy(&kf, &objs[(size_t)i * fw::OBJ_Q * 4 + 3], 4);
        const uint32_t kind = kf & 0xffu, inner = kf >> 24, shape = kind == FW_SHAPE_CONSTANT_MEDIUM ? inner : kind;
        const float ext = box_extent(build_boxes[i]);
        float g = std::ldexp(ext, -14);
        if (shape == FW_SHAPE_TRIANGLE_MESH) g = std::fmax(g, std::ldexp(obj_size[i], -5));                        // its triangles' boxes grew by 2^-6 of this, in the mesh's frame
        else if ((shape == FW_SHAPE_SPHERE || shape == FW_SHAPE_CONE || shape == FW_SHAPE_CYLINDER) && (ex.mode & 2u) && ext > 0.f)
            g = std::fmax(g, std::fmin(ext, std::ldexp(ex.far_r * ex.far_r / ext, -22)));
        build_boxes[i] = grown_by(build_boxes[i], g);
    }
    auto pack_boxes = [&](const std::vector<Box> &bs) {
        std::vector<float> out((size_t)desc->n_objects * 8, 0.f);
        for (uint32_t i = 0; i < desc->n_objects; i++) {
            const Box &b = bs[i];
            float *c = &out[(size_t)i * 8];
            c[0] = b.mn.x; c[1] = b.mn.y; c[2] = b.mn.z; c[4] = b.mx.x; c[5] = b.mx.y; c[6] = b.mx.z;
        }
        return out;
    };
    const std::vector<float> cull = pack_boxes(own_boxes);       // enclosing world boxes of the objects themselves: the pre-tests of the linear scan (k_extend_linear*)
    const std::vector<float> leafb = pack_boxes(build_boxes);    // the objects' boxes in the walked trees (k_extend_scan, hoisted_hits)
    // Hoisting: an object whose box covers most of the scene (part2's r = 5000 fog medium) is met by nearly every ray, so in
    // the tree its leaf is one more divergent leaf test per ray.  Scenes without meshes and too many objects for the scan keep
    // such objects out of the WALKED tree; the kernels test them for every ray before the walk, with a wave-uniform index
    // (hoisted_hits in fw_kernels.hip: same own-box test, same tie rule, so the same result as the leaf would give).
    std::vector<uint32_t> hoisted;
    if (use_sah() && desc->n_objects > 8 && fl.tri.empty() && !O.no_hoist) {
        Box root = build_boxes[0];
        for (const Box &b : build_boxes) root = box_union(root, b);
        const float ra = box_area(root);
        for (uint32_t i = 0; i < desc->n_objects && hoisted.size() < 4; i++)
            if (box_area(build_boxes[i]) >= 0.5f * ra) hoisted.push_back(i);
        if (desc->n_objects - hoisted.size() < 2) hoisted.clear();
    }
    if (use_sah()) {
        FlatBvh sah;
        if (hoisted.empty()) sah_build(sah, build_boxes);
        else {
            std::vector<Box> sub; std::vector<uint32_t> ids;
            for (uint32_t i = 0; i < desc->n_objects; i++)
                if (std::find(hoisted.begin(), hoisted.end(), i) == hoisted.end()) { sub.push_back(build_boxes[i]); ids.push_back(i); }
            sah_build(sah, sub);
            for (uint32_t i = 0; i < sah.count(); i++) {      // leaf items: positions in `sub` -> object ids
                float *nd = &sah.nodes[(size_t)i * 8];
                uint32_t A, B; std::memcpy(&A, nd + 3, 4); std::memcpy(&B, nd + 7, 4);
                const uint32_t kind = A >> 30;
                if (kind == fw::NODE_LEAF) { A = (kind << 30) | ids[A & fw::NODE_MASK]; std::memcpy(nd + 3, &A, 4); }
                else if (kind == fw::NODE_DOUBLE) { A = (kind << 30) | ids[A & fw::NODE_MASK]; B = ids[B]; std::memcpy(nd + 3, &A, 4); std::memcpy(nd + 7, &B, 4); }
            }
        }
        tlas = std::move(sah);
    }
    PairBvh tlas_p;
    const uint32_t tlas_root = pair_convert(tlas, build_boxes, tlas_p);
    // WIDE nodes (fw_device.h) for the LDS-resident walks, where they fit a CU's LDS next to the walks' stacks: f32 nodes first,
    // quantised ones for a BLAS too big for those.  Option WIDE=0: none (the pair-node kernels, A/B); =f32 / =q8 force an encoding.
    const bool wide_on = use_sah() && O.wide != 0;
    WideBvh wtlas; wtlas.fmt = fw::WIDE_F32;
    uint32_t wtlas_root = 0xffffffffu;
    auto wide_lds_bytes = [&](const WideBvh &t, uint32_t waves) { return (size_t)t.words.size() * 4 + (size_t)waves * (3 * t.depth + 2) * 128 + 4096; };
    if (wide_on && fl.tri.empty() && desc->n_objects > 8) {
        wtlas_root = wide_convert(tlas, build_boxes, wtlas);     // (hoisted objects are not in `tlas`; its leaves hold object ids)
        if (wtlas_root == 0xffffffffu || wide_lds_bytes(wtlas, 8) > fw::LDS_TREE_LIMIT) { wtlas.words.clear(); wtlas_root = 0xffffffffu; }
    }
    int wblas_fmt = fw::WIDE_NONE;
    if (wide_on && !fl.tri.empty()) {
        const bool force_q8 = O.wide == 2, force_f32 = O.wide == 1;
        if (fl.wide_ok[0] && !force_q8 && wide_lds_bytes(fl.wblas_f32, 8) <= fw::LDS_TREE_LIMIT) wblas_fmt = fw::WIDE_F32;
        else if (fl.wide_ok[1] && !force_f32 && wide_lds_bytes(fl.wblas_q8, 8) <= fw::LDS_TREE_LIMIT) wblas_fmt = fw::WIDE_Q8;
    }
    const WideBvh &wblas = wblas_fmt == fw::WIDE_Q8 ? fl.wblas_q8 : fl.wblas_f32;
    std::vector<uint32_t> obj_wroot(desc->n_objects, fw::W_DONE);
    for (uint32_t i = 0; i < desc->n_objects; i++) obj_wroot[i] = wblas_fmt == fw::WIDE_Q8 ? obj_wroot_q8[i] : obj_wroot_f32[i];

    // materials / textures / images
    std::vector<float> mats((size_t)std::max(1u, desc->n_materials) * 8, 0.f), texs((size_t)std::max(1u, desc->n_textures) * 8, 0.f);
    std::vector<uint8_t> images;
    for (uint32_t i = 0; i < desc->n_textures; i++) {
        const fw_texture &t = desc->textures[i];
        float *q = &texs[(size_t)i * 8];
        q[0] = bits_f((uint32_t)t.kind); q[1] = t.scale; q[2] = bits_f(t.depth);
        switch (t.kind) {
        case FW_TEX_CONSTANT: q[4] = t.color.x; q[5] = t.color.y; q[6] = t.color.z; break;
        case FW_TEX_CHECKER:
            if (t.odd < 0 || t.even < 0 || (uint32_t)t.odd >= desc->n_textures || (uint32_t)t.even >= desc->n_textures) return fail(FW_ERR_BAD_ARG, "checker child texture out of range");
            q[3] = bits_f((uint32_t)t.odd); q[4] = bits_f((uint32_t)t.even); break;
        case FW_TEX_PERLIN: case FW_TEX_TURBULENCE: case FW_TEX_MARBLE: has_perlin = true; break;
        case FW_TEX_IMAGE: {
            if (!t.img_rgb8 || !t.img_w || !t.img_h) return fail(FW_ERR_BAD_ARG, "ImageTexture without pixels");
            size_t off = images.size(), nb = (size_t)t.img_w * t.img_h * 3;
            if (off + nb > 0xffffffffull) return fail(FW_ERR_UNSUPPORTED, "image textures exceed 4 GiB");
            images.insert(images.end(), t.img_rgb8, t.img_rgb8 + nb);
            q[4] = bits_f((uint32_t)off); q[5] = bits_f(t.img_w); q[6] = bits_f(t.img_h); break; }
        default: return fail(FW_ERR_BAD_ARG, "unknown texture kind");
        }
    }
    // CheckerTexture children must form a finite tree no deeper than the device's bounded walk (the reference would
    // recurse forever on a cycle: texture.rs:59-72)
    for (uint32_t i = 0; i < desc->n_textures; i++) {
        std::vector<std::pair<int32_t, int>> todo{{(int32_t)i, 0}};
        size_t visited = 0;
        while (!todo.empty()) {
            auto [ti, depth] = todo.back(); todo.pop_back();
            if (depth > 14 || ++visited > 65536) return fail(FW_ERR_BAD_ARG, "CheckerTexture nesting too deep or cyclic");
            const fw_texture &t = desc->textures[ti];
            if (t.kind == FW_TEX_CHECKER) { todo.push_back({t.odd, depth + 1}); todo.push_back({t.even, depth + 1}); }
        }
    }
    for (uint32_t i = 0; i < desc->n_materials; i++) {
        const fw_material &m = desc->materials[i];
        float *q = &mats[(size_t)i * 8];
        if (m.kind < FW_MAT_LAMBERTIAN || m.kind > FW_MAT_ISOTROPIC) return fail(FW_ERR_BAD_ARG, "unknown material kind");
        bool needs_tex = m.kind == FW_MAT_LAMBERTIAN || m.kind == FW_MAT_EMISSIVE || m.kind == FW_MAT_ISOTROPIC;
        if (needs_tex && (m.texture < 0 || (uint32_t)m.texture >= desc->n_textures)) return fail(FW_ERR_BAD_ARG, "material texture out of range");
        uint32_t mbits = (uint32_t)m.kind;
        q[4] = m.albedo.x; q[5] = m.albedo.y; q[6] = m.albedo.z;
        if (needs_tex) {
            const fw_texture &t = desc->textures[m.texture];
            if (t.kind == FW_TEX_CONSTANT) { mbits |= fw::MF_TEX_CONST; q[4] = t.color.x; q[5] = t.color.y; q[6] = t.color.z; }
            // uv are consumed by ImageTexture only: walk the (checker) tree
            std::vector<int32_t> todo{m.texture};
            for (int guard = 0; !todo.empty() && guard < 4096; guard++) {
                const fw_texture &c = desc->textures[todo.back()];
                todo.pop_back();
                if (c.kind == FW_TEX_IMAGE) mbits |= fw::MF_NEEDS_UV;
                if (c.kind == FW_TEX_CHECKER) { todo.push_back(c.odd); todo.push_back(c.even); }
            }
        }
        if (m.kind == FW_MAT_DIELECTRIC) { q[4] = q[5] = q[6] = 1.f; }          // its attenuation (material.rs:128), for the chain state's product
        q[0] = bits_f(mbits); q[1] = bits_f((uint32_t)(needs_tex ? m.texture : 0)); q[2] = m.roughness; q[3] = m.ref_idx;
    }
    const fw_environment &e = desc->environment;
    if (e.kind < FW_ENV_COLOR || e.kind > FW_ENV_HDR) return fail(FW_ERR_BAD_ARG, "unknown environment kind");
    if (e.kind == FW_ENV_HDR && (!e.hdr_rgb || !e.hdr_w || !e.hdr_h)) return fail(FW_ERR_BAD_ARG, "HdrEnv without pixels");

    // stack depth the kernels will be given
    if (tlas_p.depth + 1 + fl.blas_depth + 1 > 120) return fail(FW_ERR_BVH_DEPTH, "BVH deeper than the LDS traversal stack (120 levels)");

    fw_scene *sc = new (std::nothrow) fw_scene();
    if (!sc) return fail(FW_ERR_OOM, "host allocation failed");
    sc->device = device;
    sc->n_cus = n_cus_dev;
    int rc = FW_OK;
    // one device allocation + one copy for the whole scene (12 separate hipMalloc/hipFree pairs cost up to 30 ms of a
    // one-shot render): sections are 256-byte aligned inside a host staging blob
    struct Sec { const void *src; size_t bytes, off; };
    Sec secs[21] = {
        {objs.data(), objs.size() * 4, 0}, {tlas_p.nodes.data(), tlas_p.nodes.size() * 4, 0}, {fl.blas.nodes.data(), fl.blas.nodes.size() * 4, 0},
        {fl.tri.data(), fl.tri.size() * 4, 0}, {fl.any_attr ? fl.tri_attr.data() : nullptr, fl.any_attr ? fl.tri_attr.size() * 4 : 0, 0},
        {fl.tri_rank.data(), fl.tri_rank.size() * 4, 0}, {obj_rank.data(), obj_rank.size() * 4, 0}, {gate.data(), gate.size() * 4, 0},
        {mats.data(), mats.size() * 4, 0}, {texs.data(), texs.size() * 4, 0}, {images.data(), images.size(), 0},
        // the HDR map as 16-byte texels (rgb + pad), written straight into the blob below: a 12-byte texel straddles a 32-byte
        // sector in two offsets of eight, and a miss's lookup is one gather per path out of 100 MB
        {nullptr, e.kind == FW_ENV_HDR ? (size_t)e.hdr_w * e.hdr_h * 4 * 4 : 0, 0},
        {cull.data(), cull.size() * 4, 0},
        {ref_tlas.data(), ref_tlas.size() * 4, 0}, {fl.ref_blas.data(), fl.ref_blas.size() * 4, 0}, {obj_ref_blas.data(), obj_ref_blas.size() * 4, 0},
        {leafb.data(), leafb.size() * 4, 0},
        {wblas_fmt ? wblas.words.data() : nullptr, wblas_fmt ? wblas.words.size() * 4 : 0, 0}, {wtlas.words.data(), wtlas.words.size() * 4, 0},
        {obj_wroot.data(), obj_wroot.size() * 4, 0}, {fl.tri_gate.data(), fl.tri_gate.size() * 4, 0}};
    size_t total = 0;
    for (Sec &x : secs) { x.off = total; total += (x.bytes + 255) & ~(size_t)255; }
    total = std::max<size_t>(total, 256);                  // a multiple of 256: k_upload copies 16-byte words
    const double tr_build = ms_since(tr0);
    const auto tr1 = now();
    Workspace *ws = workspace_for(device);
    if (!ws) { delete sc; return fail(FW_ERR_OOM, "no workspace for this device"); }
    std::lock_guard<std::mutex> ws_guard(ws->mu);
    if ((rc = init_device_locked(ws, device)) != FW_OK) { delete sc; return rc; }      // a host that never called fw_init pays for the device here, once
    // the blob is assembled in pinned host memory (grown on demand, kept per device) and copied by a kernel on the null
    // stream; renders of this scene wait for ws->ev_upload in stream order, the host never blocks here
    (void)hipEventSynchronize(ws->ev_upload);          // the previous upload has read the staging buffer (long done)
    if (ws->staging_bytes < total) {
        if (ws->staging) (void)hipHostFree(ws->staging);
        ws->staging = nullptr; ws->staging_bytes = 0;
        const size_t want = std::max<size_t>(total + total / 4, 1 << 20);
        if (hipHostMalloc(&ws->staging, want, hipHostMallocDefault) != hipSuccess) { delete sc; return fail(FW_ERR_OOM, "pinned staging allocation failed"); }
        ws->staging_bytes = want;
    }
    uint8_t *blob = (uint8_t *)ws->staging;
    size_t prev_end = 0;
    for (const Sec &x : secs) {       // sections + zeroed padding between them
        if (x.off > prev_end) std::memset(blob + prev_end, 0, x.off - prev_end);
        if (&x == &secs[11] && x.bytes) {
            float *dst = reinterpret_cast<float *>(blob + x.off);
            const float *src = e.hdr_rgb;
            for (size_t k = 0, n = (size_t)e.hdr_w * e.hdr_h; k < n; k++) { dst[4 * k] = src[3 * k]; dst[4 * k + 1] = src[3 * k + 1]; dst[4 * k + 2] = src[3 * k + 2]; dst[4 * k + 3] = 0.f; }
        }
        else if (x.bytes) std::memcpy(blob + x.off, x.src, x.bytes);
        prev_end = x.off + x.bytes;
    }
    if (total > prev_end) std::memset(blob + prev_end, 0, total - prev_end);
    const double tr_blob = ms_since(tr1);
    if (ws->scene_cache.p && ws->scene_cache.bytes >= total) { sc->data = ws->scene_cache; ws->scene_cache = DevBuf{}; }   // reuse the previous scene's allocation
    const auto tr2 = now();
    const bool reused = sc->data.p != nullptr;
    rc = sc->data.alloc(total);
    const double tr_alloc = ms_since(tr2);
    const auto tr3 = now();
    if (!rc && !ws->upload_stream && hipStreamCreateWithFlags(&ws->upload_stream, hipStreamNonBlocking) != hipSuccess) rc = fail(FW_ERR_HIP, "upload stream creation failed");
    if (!rc) {
        fw::launch_upload(ws->upload_stream, blob, sc->data.p, total);
        if (hipEventRecord(ws->ev_upload, ws->upload_stream) != hipSuccess || hipGetLastError() != hipSuccess) rc = fail(FW_ERR_HIP, "scene upload failed");
    }
    if (trace) fprintf(stderr, "[firework] scene_create: build %.2f ms, blob %.2f ms (%zu B), alloc %.2f ms (%s), upload launch %.2f ms\n",
                       tr_build, tr_blob, total, tr_alloc, reused ? "cached" : "hipMalloc", ms_since(tr3));
    if (trace && !fl.tri.empty()) fprintf(stderr, "[firework] scene_create: meshes (%zu triangles, %d host threads): gather %.2f ms | SAH tree, the reference tree beside it %.2f | wait for the reference tree + ranks %.2f | gate boxes %.2f | pair + wide f32 + wide q8 side by side %.2f\n",
                                          fl.tri.size() / 12, host_build_threads(), fl.ms_gather, fl.ms_sah, fl.ms_ref, fl.ms_gate, fl.ms_pair + fl.ms_wide);
    if (rc) { delete sc; return rc; }
    const uint8_t *base = (const uint8_t *)sc->data.p;
    fw::DScene &d = sc->d;
    d.obj = (const float4 *)(base + secs[0].off); d.tlas = (const float4 *)(base + secs[1].off); d.blas = (const float4 *)(base + secs[2].off);
    d.tri = (const float4 *)(base + secs[3].off); d.tri_nrm = (const float4 *)(base + secs[4].off);
    d.tri_rank = (const uint32_t *)(base + secs[5].off); d.obj_rank = (const uint32_t *)(base + secs[6].off); d.obj_gate = (const float4 *)(base + secs[7].off);
    d.mat = (const float4 *)(base + secs[8].off); d.tex = (const float4 *)(base + secs[9].off); d.images = base + secs[10].off;
    const float *hdr_dev = (const float *)(base + secs[11].off);
    d.obj_cull = (const float4 *)(base + secs[12].off);
    d.ref_tlas = (const float4 *)(base + secs[13].off); d.ref_blas = (const float4 *)(base + secs[14].off);
    d.obj_ref_blas = (const uint32_t *)(base + secs[15].off);
    d.obj_leaf = (const float4 *)(base + secs[16].off);
    d.wblas = wblas_fmt ? (const uint32_t *)(base + secs[17].off) : nullptr;
    d.wtlas = wtlas.words.empty() ? nullptr : (const uint32_t *)(base + secs[18].off);
    d.obj_wroot = (const uint32_t *)(base + secs[19].off);
    d.tri_gate = (const float4 *)(base + secs[20].off);
    d.wtlas_root = wtlas_root;
    sc->wblas_fmt = wblas_fmt; sc->wtlas_fmt = wtlas.words.empty() ? fw::WIDE_NONE : fw::WIDE_F32;
    sc->wblas_nodes = wblas_fmt ? wblas.count() : 0; sc->wtlas_nodes = wtlas.count();
    sc->wblas_depth = wblas_fmt ? wblas.depth : 0; sc->wtlas_depth = wtlas.depth;
    sc->ex = ex; sc->ref_tlas_depth = ref_tlas_depth; sc->ref_blas_depth = fl.ref_blas_depth;
    d.n_objects = desc->n_objects; d.has_medium = has_medium ? 1u : 0u; d.has_perlin = has_perlin ? 1u : 0u; d.has_mesh = fl.tri.empty() ? 0u : 1u;
    d.prim_bits = prim_bits; d.tlas_root = tlas_root;
    d.soft_shear = O.soft_shear_log2 > 0 ? std::ldexp(1.f, -O.soft_shear_log2) : 0.f;
    d.n_hoisted = (uint32_t)hoisted.size();
    for (size_t i = 0; i < 4; i++) d.hoisted[i] = i < hoisted.size() ? hoisted[i] : 0u;
    d.env.kind = e.kind;
    d.env.color[0] = e.color.x; d.env.color[1] = e.color.y; d.env.color[2] = e.color.z;
    d.env.zenith[0] = e.zenith.x; d.env.zenith[1] = e.zenith.y; d.env.zenith[2] = e.zenith.z;
    d.env.horizon[0] = e.horizon.x; d.env.horizon[1] = e.horizon.y; d.env.horizon[2] = e.horizon.z;
    d.env.hdr = hdr_dev; d.env.hdr_w = e.hdr_w; d.env.hdr_h = e.hdr_h;
    sc->hdr_env = e.kind == FW_ENV_HDR;
    sc->has_expensive = sc->hdr_env;
    for (uint32_t i = 0; i < desc->n_materials; i++) {
        uint32_t mb; std::memcpy(&mb, &mats[(size_t)i * 8], 4);
        const uint32_t mk = mb & 0xffu;
        if (mk == (uint32_t)FW_MAT_DIELECTRIC || (!(mb & fw::MF_TEX_CONST) && (mk == (uint32_t)FW_MAT_LAMBERTIAN || mk == (uint32_t)FW_MAT_EMISSIVE || mk == (uint32_t)FW_MAT_ISOTROPIC))) sc->has_expensive = true;
    }
    // chain state (fw_kernels.hip: load_state_chain): every attenuation a constant of its material, and ten material ids in 32 bits
    sc->chain_bits = 0;
    {
        bool constant = desc->n_materials > 0;
        for (uint32_t i = 0; i < desc->n_materials; i++) {
            uint32_t mb; std::memcpy(&mb, &mats[(size_t)i * 8], 4);
            const uint32_t mk = mb & 0xffu;
            if (!(mb & fw::MF_TEX_CONST) && (mk == (uint32_t)FW_MAT_LAMBERTIAN || mk == (uint32_t)FW_MAT_EMISSIVE || mk == (uint32_t)FW_MAT_ISOTROPIC)) constant = false;
        }
        uint32_t bits = 1; while ((1u << bits) < desc->n_materials) bits++;
        if (constant && 10u * bits <= 32u) sc->chain_bits = bits;
    }
    sc->simple_shapes = true;
    sc->simple_set = true; sc->simple_but_meshes = true;
    for (uint32_t i = 0; i < desc->n_objects; i++) {
        uint32_t kf; std::memcpy(&kf, &objs[(size_t)i * fw::OBJ_Q * 4 + 3], 4);
        if ((kf & 0xffu) > 4u) sc->simple_shapes = false;
        if ((kf & 0xffu) > 5u) sc->simple_but_meshes = false;
        if ((kf & 0xffu) > 4u && !((kf & 0xffu) == FW_SHAPE_CONSTANT_MEDIUM && (kf >> 24) == FW_SHAPE_SPHERE)) sc->simple_set = false;
    }
    // the trailing plain boxes of a linear scene (k_extend_linear_defer): at most two, no media or meshes anywhere in the scene
    sc->n_defer = 0;
    if (!has_medium && fl.tri.empty())
        for (uint32_t i = desc->n_objects; i-- > 0 && sc->n_defer < 2u;) {
            uint32_t kf; std::memcpy(&kf, &objs[(size_t)i * fw::OBJ_Q * 4 + 3], 4);
            if ((kf & 0xffu) == (uint32_t)FW_SHAPE_RECT3D && (((kf >> 8) & 0xffffu) & fw::OF_CULL0) && !(((kf >> 8) & 0xffffu) & fw::OF_GATE)) sc->n_defer++; else break;
        }
    sc->tlas_nodes = ref_tlas_nodes; sc->blas_nodes = fl.ref_blas_nodes;   // reported: the reference topology (bvh.rs)
    sc->tlas_depth = tlas_p.depth; sc->blas_depth = fl.blas_depth;
    sc->blas_pair_nodes = fl.blas.count(); sc->tlas_pair_nodes = tlas_p.count(); sc->max_tris = fl.max_tris; sc->n_tris = (uint32_t)(fl.tri.size() / 12);
    sc->n_mat = desc->n_materials; sc->n_tex = desc->n_textures;
    *out = sc;
    return FW_OK;
}

// camera.rs:74-107
fw::DCamera make_camera(const fw_camera_settings &s, uint32_t width, uint32_t height) {
    const float PI_F = 3.14159265358979323846f;
    float theta = s.vfov * PI_F / 180.f;
    V3 cam_pos = tov(s.cam_pos), look_at = tov(s.look_at);
    V3 w = normalized(cam_pos - look_at);
    V3 u = normalized(cross(V3{0, 1, 0}, w));
    V3 v = cross(w, u);
    float half_height = std::tan(theta / 2.0f);
    float half_width = half_height * (float)width / (float)height;
    V3 lower_left = cam_pos - half_width * s.focus_dist * u - half_height * s.focus_dist * v - w * s.focus_dist;
    V3 horizontal = 2.0f * half_width * s.focus_dist * u;
    V3 vertical = 2.0f * half_height * s.focus_dist * v;
    fw::DCamera c;
    auto put = [](float *d, V3 a) { d[0] = a.x; d[1] = a.y; d[2] = a.z; };
    put(c.position, cam_pos); put(c.horizontal, horizontal); put(c.vertical, vertical); put(c.lower_left, lower_left); put(c.u, u); put(c.v, v);
    c.lens_radius = s.aperture / 2.f;
    return c;
}

// Wavefront pool size.  Bigger is better on this part: fewer, fuller launches and longer wave-private queues
// (measured on cornell 512x512@1024: 4 Mi paths 85 ms/frame, 16 Mi 64 ms, 256 Mi = the whole frame 55 ms).
// Default: up to 2^28 paths (104 B per slot + 40 B for parked mesh rays, and up to twice as many slots as paths because a
// wave's queue capacity is a power of two: 28-77 GB), never more than half of the free HBM.
uint32_t default_paths_per_batch(const Options &O, size_t arena_bytes) {
    if (O.paths_per_batch > 0) return (uint32_t)std::min<long long>(O.paths_per_batch, 0x7fffffffll);
    size_t free_b = 0, total_b = 0;
    uint64_t budget = 1ull << 28;
    // (the arena's own bytes count as available: the pools are carved out of it)
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b + arena_bytes > 0) budget = std::min<uint64_t>(budget, (uint64_t)((free_b + arena_bytes) / 2) / 288u);
    return (uint32_t)std::max<uint64_t>(budget, 1u << 16);
}

// first_sample / user_accum: fw_render_progressive (0 / nullptr for a plain render)
int render_impl(fw_scene *sc, const fw_render_params *p, uint8_t *rgb8, float *gamma_rgb, float *linear_rgb, fw_stats *stats,
                uint32_t first_sample = 0, float *user_accum = nullptr) {
    if (!sc || !p) return fail(FW_ERR_BAD_ARG, "null argument");
    if (p->width == 0 || p->height == 0 || p->samples == 0) return fail(FW_ERR_BAD_ARG, "width, height and samples must be > 0");
    if (!(p->gamma > 0.f)) return fail(FW_ERR_BAD_ARG, "gamma must be > 0");
    if (p->rng_mode != FW_RNG_CTR) return fail(FW_ERR_UNSUPPORTED, "the HIP path implements FW_RNG_CTR only (FW_RNG_LCG is a sequential stream)");
    uint64_t full = (uint64_t)p->width * p->height;
    if (full > 0xffffffffull) return fail(FW_ERR_UNSUPPORTED, "image too large");
    uint32_t n_pix = p->pixel_ids ? p->n_pixels : (uint32_t)full;
    if (n_pix == 0) return fail(FW_ERR_BAD_ARG, "no pixels to render");
    if ((uint64_t)first_sample + p->samples > 0xffffffffull) return fail(FW_ERR_BAD_ARG, "first_sample + samples overflows");
    if (p->pixel_ids) for (uint32_t i = 0; i < n_pix; i++) if (p->pixel_ids[i] >= full) return fail(FW_ERR_BAD_ARG, "pixel id out of range");
    const auto wall0 = std::chrono::steady_clock::now();
    HIPCHK(hipSetDevice(sc->device));
    hipStream_t stream = (hipStream_t)p->stream;
    Workspace *ws = workspace_for(sc->device);
    if (!ws) return fail(FW_ERR_OOM, "no workspace for this device");
    std::lock_guard<std::mutex> ws_guard(ws->mu);
    { const int irc = init_device_locked(ws, sc->device); if (irc) return irc; }     // after fw_release_workspace, or a scene made before it
    const Options O = options();

    // ---- batches and lanes -----------------------------------------------------------------------------------
    // Two batches in flight on two streams under use_bvh: the tree walks leave issue slots and HBM idle (the LDS-resident
    // ones run 4 waves per SIMD) that the other batch's k_shade / k_extend_scan fills.  Full-size configs, interleaved on one
    // box (tools/configs.sh): suzanne 92.2 -> 81.1 ms, random_spheres 2.0 -> 1.8, part2 2287 -> 2271; 3 or 4 lanes gain less.
    // The linear scan stays at one batch in flight: cornell gains 4-5 % (42.2 -> 40.3 ms), hdri and volume lose 1-3 %
    // (both of their kernels wait for HBM), and with one lane every kernel's HIP-event time is its own — what bench.py's
    // roofline object divides by.  FIREWORK_STREAMS=n overrides.  Results do not depend on n (batches accumulate in order).
    // Round 3: a linear scene whose scan runs the box lists (k_extend_linear_defer: cornell) takes two lanes as well — its scan is
    // bound by instruction issue, its k_shade by HBM, and they overlap (42.2 -> 40.3 ms in round 2's A/B); per-kernel times for
    // the roofline come from an exclusive pass (FIREWORK_STREAMS=1) that bench.py runs next to the timed loop.
    // (Small frames too: random_spheres, 5.8 M paths, 1.83 ms in two batches against 1.92-1.97 in one, round 3.)
    // Round 5: two lanes for every scene.  hdri and volume (linear scans without box lists) had lost 1-3 % with two in rounds 2-4; with the
    // XCD-contiguous queues and the SIMPLE-set scans they gain: hdri 32.4-33.4 -> 31.0-32.0 ms, volume 33.8-36.1 -> 32.4-34.2 (each setting twice in a
    // row on a box whose processes alternate between two k_shade modes: profiles/r05k_lanes2.txt).
    int n_lanes = 2;
    if (O.streams >= 1) n_lanes = std::min(O.streams, (int)Workspace::MAX_LANES);
    n_lanes = (int)std::min<uint32_t>((uint32_t)n_lanes, p->samples);
    // EXACT_PRODUCT: 160 more bytes per slot (ten attenuation records) where the scene has no chain state: half the default batch
    const bool exact_product = O.exact_product && (sc->chain_bits == 0 || O.no_chain);
    uint32_t budget = p->paths_per_batch ? p->paths_per_batch : default_paths_per_batch(O, ws->arena.bytes) / (uint32_t)n_lanes / (exact_product ? 2u : 1u);
    uint32_t spp_b = std::max<uint32_t>(1u, budget / n_pix);
    spp_b = std::min(spp_b, (p->samples + (uint32_t)n_lanes - 1) / (uint32_t)n_lanes);     // at least one batch per lane
    uint64_t paths64 = (uint64_t)n_pix * spp_b;
    if (paths64 > 0x7fffffffull) return fail(FW_ERR_UNSUPPORTED, "too many paths per batch");
    uint32_t max_paths = (uint32_t)paths64;
    uint32_t n_batches = (p->samples + spp_b - 1) / spp_b;
    // equal batches (round 4): 512 samples at up to 145 per batch were 145 + 145 + 145 + 77; four times 128 keep the two lanes level, and the
    // pools — sized by the largest batch, with a power of two of chunks per wave — shrink with it (suzanne: 73 -> 37 GB; the arena no longer
    // has to grow between cornell and suzanne: hipFree of a 35 GB arena cost 2.4 s of the caller's time)
    spp_b = (p->samples + n_batches - 1) / n_batches;
    max_paths = n_pix * spp_b;
    n_lanes = (int)std::min<uint32_t>((uint32_t)n_lanes, n_batches);

    // wave-private queues: many more waves than are resident, each owning >= 8 chunks of 64 paths when the batch allows
    fw::DQueue q{};
    // How many: whole rounds of resident waves for BOTH queue kernels (k_extend_linear holds 7 waves per SIMD, k_extend_bvh 5,
    // k_shade 4 -> multiples of lcm x SIMDs), and 16-50 chunks per wave so that the half-empty last chunk of a queue stays
    // small; measured on cornell (tools/waves_sweep.sh): whole frame 86 016 waves 41.1 ms vs 131 072: 42.0; the 1/4 share of
    // a 4-GPU frame 57 344: 12.7 vs 13.7 ms; the 1/8 share 28 672: 6.30 vs 6.67 ms.
    const uint32_t unit = (uint32_t)sc->n_cus * 4u * (p->use_bvh ? 20u : 28u);
    const uint64_t chunks = ((uint64_t)max_paths + 63u) / 64u;
    uint32_t want_waves = unit * (uint32_t)std::min<uint64_t>(3u, std::max<uint64_t>(1u, chunks / ((uint64_t)unit * 16u)));
    // Round 4 (gpurun_out/r04i/share_waves.txt, r04j/cornell_waves.txt): with TWO batches in flight on a linear scene (the box lists: cornell)
    // fewer, longer queues overlap better — whole frame 86 016 waves 36.4-37.0 ms, 28 672: 35.2-35.3 (its exclusive pass is slower: 40.6 vs
    // 37.8) — and a small share keeps its queues long with 12 288: rank 0's eighth of the frame 5.9 -> 5.5 ms (79 -> 85 % of ideal).
    if (!p->use_bvh && n_lanes > 1) want_waves = chunks >= (1u << 20) ? unit : (uint32_t)sc->n_cus * 48u;
    if (O.waves > 0) want_waves = (uint32_t)O.waves;
    q.n_waves = std::max(4u, std::min(want_waves, (max_paths + 511u) / 512u));
    q.n_waves = (q.n_waves + 7u) & ~7u;       // a multiple of 8: one contiguous eighth of the queues per XCD (fw_kernels.hip: wave_index)
    uint32_t chunks_per_wave = (max_paths + q.n_waves * 64u - 1) / (q.n_waves * 64u);
    q.cpw_shift = 0;
    while ((1u << q.cpw_shift) < chunks_per_wave) q.cpw_shift++;      // power of two: chunk -> (wave, row) is a shift
    q.cap = 64u << q.cpw_shift;
    uint64_t cap64 = (uint64_t)q.cap * q.n_waves;
    if (cap64 > 0x7fffffffull) return fail(FW_ERR_UNSUPPORTED, "too many path slots");
    uint32_t cap = (uint32_t)cap64;

    // the exact walk: ill-conditioned mesh rays in any mode, far origins and "every ray" (FIREWORK_EXACT_ALL) under use_bvh only
    // (the linear scan tests every object anyway: only a mesh's BLAS is walked there)
    // (FIREWORK_FUSED=1, the one-launch-per-segment A/B kernel, has no second pass: it runs without the exact walk)
#if FW_AB
    const bool fused_req = O.fused && !(p->use_bvh && sc->d.has_mesh);
    const bool tlas_refill = !O.tlas_refill_off;
#else
    const bool fused_req = false, tlas_refill = true;
#endif
    const uint32_t exact_mode = fused_req ? 0u : ((sc->ex.mode & 1u) | (p->use_bvh ? (sc->ex.mode & 6u) : ((sc->ex.mode & 4u) && sc->d.has_mesh ? 4u : 0u)));
    const bool park_meshes = p->use_bvh && sc->d.has_mesh != 0 && tlas_refill;
    int rc = FW_OK;
    auto need = [&](DevBuf &b, size_t n) { if (!rc) rc = b.alloc(n); };
    // the lanes' buffers: slices of the workspace's arena (Workspace::Lane), laid out twice — sizes first, then pointers
    auto layout = [&](uint8_t *arena_base) -> size_t {
        size_t off = 0;
        auto put = [&](void *&dst, size_t n) { dst = arena_base ? arena_base + off : nullptr; off += (n + 255) & ~(size_t)255; };
        for (int l = 0; l < n_lanes; l++) {
            Workspace::Lane &L = ws->lanes[l];
            for (int k = 0; k < 2; k++) { put(L.ray_a[k], (size_t)cap * 16); put(L.ray_b[k], (size_t)cap * 8); put(L.state[k], (size_t)cap * 16); }
            put(L.hits, (size_t)cap * 8);
            put(L.sample_rad, (size_t)cap * 16);              // indexed by home slot
            put(L.dep_bits, ((size_t)cap + 31) / 32 * 4);     // one bit per slot: "a radiance record was written here" (black environments)
            if (exact_product && !fused_req) put(L.atten, (size_t)cap * (fw::MAX_SEGMENTS - 1) * fw::B_ATTEN); else L.atten = nullptr;
            if (exact_mode) put(L.exact_slots, 2 * (size_t)max_paths * 4 + 64);   // two lists of at most every ray of a segment, + the counters
            put(L.wcount, (size_t)(fw::MAX_SEGMENTS + 1) * q.n_waves * 4);
            if (park_meshes) {     // rays handed from k_extend_scan / k_extend_tlas_park to k_blas*: 40 B per slot
                const size_t pcap = (size_t)(q.cap + 64u) * q.n_waves;     // park regions: q.cap + 64 entries per queue (DPark.stride)
                put(L.park_a, pcap * 16); put(L.park_b, pcap * 8); put(L.park_m, pcap * 16); put(L.pcount, (size_t)q.n_waves * 8);   // pcount[n_waves] + ptotal[n_waves]
            }
        }
        return off;
    };
    {
        size_t want = (layout(nullptr) + ((size_t)1 << 30) - 1) & ~(((size_t)1 << 30) - 1);
        // Round 5: the arena is sized by what is asked of it — fw_init(device, bytes) where the host reserved one, else this call's own
        // layout — and grows by a quarter at least, the new allocation made first and the old one freed on a background thread
        // (arena_reserve_locked): hipFree + hipMalloc in line cost 2.8 s for 35 -> 36 GiB (gpurun_out/r04j/oneshot.txt).
        if (want > ws->arena.bytes && ws->arena.bytes) want = std::max(want, (ws->arena.bytes + ws->arena.bytes / 4 + ((size_t)1 << 30) - 1) & ~(((size_t)1 << 30) - 1));
        const auto ta = std::chrono::steady_clock::now();
        const bool grow = want > ws->arena.bytes;
        if (!rc && grow) rc = arena_reserve_locked(ws, sc->device, want);
        if (O.trace && grow) fprintf(stderr, "[firework] render: path arena grown to %.1f GiB in %.2f ms\n", (double)want / (double)(1 << 30),
                                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ta).count());
        if (!rc) layout((uint8_t *)ws->arena.p);
    }
    for (int l = 0; l < n_lanes && !rc; l++)
        if (!ws->lanes[l].stream && n_lanes > 1) HIPCHK(hipStreamCreateWithFlags(&ws->lanes[l].stream, hipStreamNonBlocking));
    need(ws->accum, (size_t)n_pix * 16);
    need(ws->totals, (size_t)n_batches * fw::COUNT_STRIDE * 4);
    if (p->pixel_ids) need(ws->pixel_ids, (size_t)n_pix * 4);
    // a whole frame is traced in the library's own 16x16-tile order (k_tile_order); k_resolve undoes it.  Not for progressive
    // renders: their accumulation buffer belongs to the caller and stays in pixel order.
    const bool own_order = !p->pixel_ids && !user_accum && n_pix >= 1024 && !O.no_tile_order;
    if (own_order) { const void *before = ws->tile_ids.p; need(ws->tile_ids, (size_t)n_pix * 4); if (ws->tile_ids.p != before) ws->tile_w = ws->tile_h = 0; }
    uint8_t *d_rgb8 = rgb8; float *d_gamma = gamma_rgb, *d_linear = linear_rgb;
    if (!p->outputs_on_device) {
        if (rgb8) { need(ws->out_rgb8, (size_t)n_pix * 3); d_rgb8 = (uint8_t *)ws->out_rgb8.p; }
        if (gamma_rgb) { need(ws->out_gamma, (size_t)n_pix * 12); d_gamma = (float *)ws->out_gamma.p; }
        if (linear_rgb) { need(ws->out_linear, (size_t)n_pix * 12); d_linear = (float *)ws->out_linear.p; }
    }
    if (rc) return rc;
    while (ws->events.size() < 3 + 2 * (size_t)n_batches) {      // [3 + b] accumulated, [3 + n_batches + b] batch b's first k_extend finished (stagger)
         hipEvent_t e; HIPCHK(hipEventCreateWithFlags(&e, ws->events.size() < 2 ? hipEventDefault : hipEventDisableTiming)); ws->events.push_back(e); }

    if (ws->ev_upload) HIPCHK(hipStreamWaitEvent(stream, ws->ev_upload, 0));     // the scene's upload kernel (null stream)
    if (p->pixel_ids) HIPCHK(hipMemcpyAsync(ws->pixel_ids.p, p->pixel_ids, (size_t)n_pix * 4, hipMemcpyHostToDevice, stream));
    if (own_order && (ws->tile_w != p->width || ws->tile_h != p->height)) {
        ws->tile_w = ws->tile_h = 0;                                               // invalid until the launch below has been queued
        fw::launch_tile_order(stream, p->width, p->height, (uint32_t *)ws->tile_ids.p);
        ws->tile_w = p->width; ws->tile_h = p->height;
    }
    if (user_accum)     // resume: the sums of the samples rendered so far (host or device memory, like the outputs)
        HIPCHK(hipMemcpyAsync(ws->accum.p, user_accum, (size_t)n_pix * 16, p->outputs_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
    else HIPCHK(hipMemsetAsync(ws->accum.p, 0, (size_t)n_pix * 16, stream));
    HIPCHK(hipMemsetAsync(ws->totals.p, 0, (size_t)n_batches * fw::COUNT_STRIDE * 4, stream));

    fw::LaunchCfg cfg{};
    cfg.q = q;
    int max_blocks = sc->n_cus * 8;
    cfg.blocks_other = (int)std::max<uint64_t>(1, std::min<uint64_t>(((uint64_t)n_pix + fw::BLOCK - 1) / fw::BLOCK, (uint64_t)max_blocks));
    cfg.tlas_depth = (int)sc->tlas_depth; cfg.blas_depth = (int)sc->blas_depth;
    cfg.n_mat = sc->n_mat; cfg.n_tex = sc->n_tex;
    cfg.lds_tables = !O.no_lds_tables;
    cfg.has_mesh = sc->d.has_mesh != 0;
    cfg.simple_set = sc->simple_set; cfg.simple_but_meshes = sc->simple_but_meshes;
#ifdef FW_NO_SIMPLE      // A/B build: the kernels of round 4 (every shape's code in every kernel that calls hit_object)
    cfg.simple_set = cfg.simple_but_meshes = false;
#endif
    cfg.tlas_refill = tlas_refill;
    cfg.n_cus = sc->n_cus;
    cfg.blas_pair_nodes = sc->blas_pair_nodes; cfg.tlas_pair_nodes = sc->tlas_pair_nodes; cfg.max_tris = sc->max_tris; cfg.n_tris = sc->n_tris;
    cfg.no_lds_tris = O.no_lds_tris;
    cfg.n_defer = (!p->use_bvh && !O.no_defer) ? sc->n_defer : 0u;
    cfg.lds_trees = !O.no_lds_trees;
    cfg.wblas_fmt = sc->wblas_fmt; cfg.wtlas_fmt = sc->wtlas_fmt; cfg.wblas_nodes = sc->wblas_nodes; cfg.wtlas_nodes = sc->wtlas_nodes;
    cfg.wblas_depth = sc->wblas_depth; cfg.wtlas_depth = sc->wtlas_depth;
    cfg.exact_form = O.exact_form;
    cfg.debug_wide_levels = 0;
#if FW_AB
    cfg.debug_wide_levels = (uint32_t)O.debug_wide_levels;
#endif
    cfg.ref_tlas_nodes = sc->tlas_nodes; cfg.ref_blas_nodes = sc->blas_nodes; cfg.ref_tlas_depth = sc->ref_tlas_depth; cfg.ref_blas_depth = sc->ref_blas_depth;
    // k_shade's list entries are 16-bit queue positions: longer queues (cap > 65536: never with the default geometry) shade in line
    // Default: the cheap loop alone where the scene has nothing expensive (cornell k_shade -5 %), everything in line otherwise — the
    // list (mode 2) is slower wherever it was measured (gpurun_out/r03h: part2@16 11.7 -> 12.0 ms, hdri@64 5.5 -> 6.1, random_spheres
    // 1.90 -> 1.98, volume@64 6.08 -> 6.12) and stays behind FIREWORK_SHADE_LIST=1; FIREWORK_NO_SHADE_DEFER=1 forces mode 0.
    cfg.shade_mode = !sc->has_expensive ? 1 : 0;
#if FW_AB
    if (O.no_shade_defer) cfg.shade_mode = 0; else if (sc->has_expensive && O.shade_list && q.cap <= 65536u) cfg.shade_mode = 2;
#endif

    fw::DCamera cam = make_camera(p->camera, p->width, p->height);
    fw::DFrame fr{};         // (zeroed: the frame graph's key hashes these structs, padding and not-yet-set per-batch fields included)
    fr.width = p->width; fr.height = p->height; fr.n_pixels = n_pix; fr.inv_n_pixels = 1.0f / (float)n_pix; fr.inv_width = 1.0f / (float)p->width;
    fr.pixel_ids = p->pixel_ids ? (const uint32_t *)ws->pixel_ids.p : (own_order ? (const uint32_t *)ws->tile_ids.p : nullptr);
    fr.scatter_out = own_order ? 1u : 0u;
    fr.seed32 = (uint32_t)p->seed ^ ((uint32_t)(p->seed >> 32) * 0x9E3779B9u);
    fr.q_n_waves = q.n_waves; fr.q_shift = q.cpw_shift;
    fr.cam_pos[0] = cam.position[0]; fr.cam_pos[1] = cam.position[1]; fr.cam_pos[2] = cam.position[2];
    // (what must not occur in the position is a NEGATIVE zero: -0 + (+0) = +0 is not the position any more, while +0 + (+-0) = +0 is.  Until round 5
    //  this read "!= 0.f", which kept hdri_test's and volume_test's cameras — at x = 0.0 — on 24-byte camera rays)
    auto not_negative_zero = [](float x) { uint32_t b; std::memcpy(&b, &x, 4); return b != 0x80000000u; };
    fr.pinhole0 = (cam.lens_radius == 0.f && not_negative_zero(cam.position[0]) && not_negative_zero(cam.position[1]) && not_negative_zero(cam.position[2]) &&
                   !O.no_short_rays) ? 1u : 0u;
    // 4-byte hit records where k_shade can recompute t cheaply and exactly: the linear scan over spheres, rects and Rect3d
    fr.hit4 = (!p->use_bvh && sc->simple_shapes && !exact_mode && !O.no_hit4 && !fused_req) ? 1u : 0u;
    const fw::DEnv &env = sc->d.env;
    // (pixel-major bits need the sample index of a path from a float quotient that is exact only while spp_batch < 2^21: dep_bit_of)
    fr.dep_pixel_major = (((n_pix <= 65536u && !O.dep_slot_major) || O.dep_pixel_major) && spp_b < (1u << 21)) ? 1u : 0u;
    fr.ex = sc->ex; fr.ex.mode = exact_mode;
    fr.chain_bits = (O.no_chain || fused_req || cfg.shade_mode == 2) ? 0u : sc->chain_bits;      // k_bounce and k_shade's list mode (A/B build) carry the running product
    fr.skip_zero_deposits = (env.kind == 0 && env.color[0] == 0.f && env.color[1] == 0.f && env.color[2] == 0.f && !O.no_zero_skip) ? 1u : 0u;

    // per-launch timing (FW_FLAG_TIME_KERNELS): one event after every launch on the launch's own stream; the end of
    // launch k is the start of launch k+1 of that lane.  With several lanes the intervals overlap in wall time.
    const bool timing = (p->flags & FW_FLAG_TIME_KERNELS) != 0;
    const bool count_deposits = (p->flags & FW_FLAG_COUNT_DEPOSITS) != 0 && fr.skip_zero_deposits != 0;   // otherwise every terminated path writes one
    const size_t per_batch_launches = 1 + 3 * fw::MAX_SEGMENTS + 2;     // raygen, 11 x (extend, exact extend, shade), queue totals, accumulate
    std::vector<std::vector<int>> ev_class(n_lanes);   // per lane: class of the launch that ENDS at events[1 + k]
    std::vector<size_t> ev_next(n_lanes, 0);
    if (timing) for (int l = 0; l < n_lanes; l++) {
        size_t want = 1 + per_batch_launches * ((n_batches + n_lanes - 1) / n_lanes);
        auto &ev = ws->lanes[l].events;
        while (ev.size() < want) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); ev.push_back(e); }
    }
    const bool use_bvh = p->use_bvh != 0;
    // FIREWORK_FUSED=1: one launch per segment (k_bounce = intersect + shade in registers, 80 instead of 120 B per ray).
    // Off by default: the frame is VALU-bound, not HBM-bound, and the fused kernel's lower occupancy costs more than the
    // bytes save (cornell 62.0 vs 55.5 ms, hdri 35.5 vs 38.5 ms, 1/8-frame shares 9.2 vs 8.3 ms).  Never with parked mesh rays.
    const bool fused = fused_req;
#if FW_AB
    const bool stagger = n_lanes > 1 && O.stagger;   // experiment: batch b's first k_extend waits for batch b-1's
#else
    const bool stagger = false;
#endif

    // FIREWORK_DUMP_PATH=file, one pixel x one sample: after every k_extend the path's ray, state and hit record are copied out
    // and written to `file` as 11 x 16 floats (ray_a[4] ray_b[2] state[4] hit[2] alive pad[3]) behind a header of 8 u32
    // (magic, pinhole0, hit4, prim_bits, bits of cam_pos[3], n_defer).  Debug aid of tools/diverge.py; never on a timed path.
    const char *dump_file = O.dump_path.c_str();
    const bool dump_one = *dump_file && n_pix == 1 && p->samples == 1 && !fused;
    std::vector<float> dump_rec(dump_one ? (size_t)fw::MAX_SEGMENTS * 16 : 0, 0.f);

    HIPCHK(hipEventRecord(ws->events[0], stream));
    hipStream_t origin = stream;       // the stream the frame's launches fork from and join: the caller's, or the capture's own (GRAPH)
    bool graph_replayed = false;       // this frame's launches went out as one hipGraphLaunch (fw_stats.reserved bit 31)
    auto fork_lanes = [&]() -> int {
        if (n_lanes > 1) {     // fork: the lane streams start after everything queued on the origin stream so far
            HIPCHK(hipEventRecord(ws->events[2], origin));
            for (int l = 0; l < n_lanes; l++) HIPCHK(hipStreamWaitEvent(ws->lanes[l].stream, ws->events[2], 0));
        }
        return FW_OK;
    };
    // The batches of a frame, n_lanes at a time.  Each batch's launches go to its lane's stream in order; the HOST enqueues a group segment by
    // segment (A.extend(s) A.shade(s) B.extend(s) B.shade(s) A.extend(s+1) ...), which changes nothing for independent streams and is what lets
    // PHASE_LOCK tie them: B's extend of a segment waits for A's extend of that segment, A's next extend for B's — so that an issue-bound
    // extend always runs beside the other batch's memory-bound shade (left alone, the two lanes drift INTO phase within three segments:
    // profiles/r05a_share_trace.txt).
    struct BatchCtx { fw::DFrame fr; fw::LaunchCfg cfg; fw::DPaths buf[2]; float2 *hits; float4 *srad; fw::DPark park; uint32_t *totals; uint32_t n_paths; int cur; int lane; hipStream_t ls; };
    // Measured (profiles/r05j_phase_lock.txt, three interleaved pairs): cornell 33.4-33.8 -> 32.2-32.4 ms — the lock holds the frame in the faster
    // of the two phases it otherwise lands in by chance (profiles/r05h_layout_pad.txt) —, where extend and shade last about as long as each
    // other.  Under use_bvh an extend lasts three shades and waiting for the other batch's costs: suzanne 62.7 -> 67.7, part2 @256 119.6 ->
    // 124.5, teapot @128 46.4 -> 47.9, random_spheres 1.64 -> 1.98.  And short launches pay for every wait: a rank's share of a cornell frame
    // (profiles/r05n_share_lock.txt, each setting twice) 1/4: 8.8 ms without the lock, 10.0 with it; 1/8: 4.5 vs 5.3; 1/2 (67 M paths per
    // batch): 16.7-16.8 vs 16.6 on one box, 16.3 vs 16.8-17.2 on another; the whole frame (134 M per batch) 33.4-33.8 vs 32.3-32.5.  Hence:
    // on for the box-list scenes of the linear scan when a batch holds PHASE_LOCK_MIN_CHUNKS chunks of 64 paths (between the two sizes it
    // was measured to pay and not to pay at), off elsewhere.
    constexpr uint64_t PHASE_LOCK_MIN_CHUNKS = 3u << 19;      // 100 M paths
    const bool phase_lock = n_lanes == 2 && !fused && (O.phase_lock == 1 || (O.phase_lock < 0 && !p->use_bvh && cfg.n_defer > 0 && chunks >= PHASE_LOCK_MIN_CHUNKS));
    std::vector<hipEvent_t> &pe = ws->phase_events;
    while (phase_lock && pe.size() < 2 * (size_t)fw::MAX_SEGMENTS) { hipEvent_t e; HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); pe.push_back(e); }
    float4 *const accum = (float4 *)ws->accum.p;
    auto begin_batch = [&](uint32_t b, BatchCtx &c) -> int {
        const int l = (int)(b % (uint32_t)n_lanes);
        Workspace::Lane &L = ws->lanes[l];
        c.lane = l; c.ls = n_lanes > 1 ? L.stream : origin;
        c.cfg = cfg; c.cfg.stream = c.ls; c.cfg.q.wcount = (uint32_t *)L.wcount;
        c.fr = fr;
        if (timing && ev_next[l] == 0) (void)hipEventRecord(L.events[0], c.ls);
        c.fr.sample0 = first_sample + b * spp_b;
        c.fr.dep_bits = (uint32_t *)L.dep_bits;
        c.fr.atten = (c.fr.chain_bits == 0 && exact_product && !fused_req) ? (float4 *)L.atten : nullptr; c.fr.atten_stride = cap;
        if (c.fr.skip_zero_deposits) HIPCHK(hipMemsetAsync(c.fr.dep_bits, 0, ((size_t)cap + 31) / 32 * 4, c.ls));
        if (exact_mode) {
            c.fr.ex.slots[0] = (uint32_t *)L.exact_slots; c.fr.ex.slots[1] = c.fr.ex.slots[0] + max_paths;
            c.fr.ex.count = c.fr.ex.slots[1] + max_paths; c.fr.ex.cap = max_paths;
            HIPCHK(hipMemsetAsync(c.fr.ex.count, 0, 64, c.ls));
        }
        c.fr.spp_batch = std::min(spp_b, p->samples - b * spp_b);
        c.n_paths = n_pix * c.fr.spp_batch;
        c.totals = (uint32_t *)ws->totals.p + (size_t)b * fw::COUNT_STRIDE;
        for (int k = 0; k < 2; k++) c.buf[k] = {(float4 *)L.ray_a[k], (float2 *)L.ray_b[k], (float4 *)L.state[k]};
        c.hits = (float2 *)L.hits;
        c.srad = (float4 *)L.sample_rad;
        c.park = fw::DPark{(float4 *)L.park_a, (float2 *)L.park_b, (float4 *)L.park_m, q.cap + 64u, (uint32_t *)L.pcount,
                           park_meshes ? (uint32_t *)L.pcount + q.n_waves : nullptr};
        if (park_meshes) HIPCHK(hipMemsetAsync(c.park.ptotal, 0, (size_t)q.n_waves * 4, c.ls));
        c.cur = 0;
        return FW_OK;
    };
    auto timed = [&](BatchCtx &c, int cls, auto &&launch) {
        launch();
        if (timing) { (void)hipEventRecord(ws->lanes[c.lane].events[1 + ev_next[c.lane]], c.ls); ev_next[c.lane]++; ev_class[c.lane].push_back(cls); }
    };
    // which: index of the batch inside its group (0 or 1 under PHASE_LOCK)
    auto segment = [&](uint32_t b, BatchCtx &c, int seg, int which, int group_size) -> int {
#if FW_AB
        if (fused) { timed(c, 2, [&] { fw::launch_bounce(c.cfg, sc->d, c.fr, c.buf[c.cur], c.buf[c.cur ^ 1], c.srad, seg, use_bvh); }); c.cur ^= 1; return FW_OK; }
#endif
        if (stagger && seg == 0 && b > 0) HIPCHK(hipStreamWaitEvent(c.ls, ws->events[3 + n_batches + b - 1], 0));   // start half a segment behind the batch before
        if (phase_lock && group_size == 2) {
            if (which == 1) HIPCHK(hipStreamWaitEvent(c.ls, pe[(size_t)seg], 0));                                     // B.extend(s) beside A.shade(s)
            else if (seg > 0) HIPCHK(hipStreamWaitEvent(c.ls, pe[(size_t)fw::MAX_SEGMENTS + (size_t)seg - 1], 0));   // A.extend(s) beside B.shade(s - 1)
        }
        timed(c, 1, [&] { fw::launch_extend(c.cfg, sc->d, c.fr, c.buf[c.cur], c.hits, seg, use_bvh, c.park); });
        if (phase_lock && group_size == 2) HIPCHK(hipEventRecord(pe[(size_t)which * fw::MAX_SEGMENTS + (size_t)seg], c.ls));
        if (stagger && seg == 0) HIPCHK(hipEventRecord(ws->events[3 + n_batches + b], c.ls));
        // the rays whose result depends on how the trees are walked (DExact), walked the reference's way: their hit records replaced
        if (exact_mode) timed(c, 1, [&] { fw::launch_extend_exact(c.cfg, sc->d, c.fr, c.buf[c.cur], c.hits, seg, use_bvh); });
        if (dump_one) {     // debug (tools/diverge.py): the one path of this call sits in slot 0 of wave 0 in every segment
            float *r = &dump_rec[(size_t)seg * 16];
            uint32_t alive = 0;
            HIPCHK(hipMemcpyAsync(&alive, c.cfg.q.wcount + (size_t)seg * q.n_waves, 4, hipMemcpyDeviceToHost, c.ls));
            HIPCHK(hipMemcpyAsync(r, c.buf[c.cur].ray_a, 16, hipMemcpyDeviceToHost, c.ls));
            HIPCHK(hipMemcpyAsync(r + 4, c.buf[c.cur].ray_b, 8, hipMemcpyDeviceToHost, c.ls));
            HIPCHK(hipMemcpyAsync(r + 6, c.buf[c.cur].state, 16, hipMemcpyDeviceToHost, c.ls));
            HIPCHK(hipMemcpyAsync(r + 10, c.hits, 8, hipMemcpyDeviceToHost, c.ls));
            HIPCHK(hipStreamSynchronize(c.ls));
            r[12] = (float)alive;
        }
        timed(c, 2, [&] { fw::launch_shade(c.cfg, sc->d, c.fr, c.buf[c.cur], c.buf[c.cur ^ 1], c.hits, c.srad, seg); });
        c.cur ^= 1;
        return FW_OK;
    };
    auto end_batch = [&](uint32_t b, BatchCtx &c) -> int {
        timed(c, 3, [&] { fw::launch_queue_totals(c.cfg, c.totals, c.park.ptotal); });       // reads this batch's queue counts: before the lane's next batch overwrites them
        if (count_deposits) fw::launch_count_deposits(c.cfg, c.fr.dep_bits, c.totals + 12);   // not a kernel class: after the last timed event of its neighbours
        // `total_color += color(..)` in sample order (render.rs:181): batch b is accumulated after batch b-1, whichever
        // lanes they ran on, so the image does not depend on the number of lanes or batches
        if (n_lanes > 1 && b > 0) HIPCHK(hipStreamWaitEvent(c.ls, ws->events[3 + b - 1], 0));
        timed(c, 3, [&] { fw::launch_accumulate(c.cfg, c.fr, c.srad, accum); });
        if (n_lanes > 1) HIPCHK(hipEventRecord(ws->events[3 + b], c.ls));
        return FW_OK;
    };
    auto enqueue_frame = [&]() -> int {
    if (int frc = fork_lanes()) return frc;
    for (uint32_t b0 = 0; b0 < n_batches; b0 += (uint32_t)n_lanes) {
        const int group = (int)std::min<uint32_t>((uint32_t)n_lanes, n_batches - b0);
        BatchCtx ctx[Workspace::MAX_LANES];
        for (int g = 0; g < group; g++) {
            if (int brc = begin_batch(b0 + (uint32_t)g, ctx[g])) return brc;
            timed(ctx[g], 0, [&] { fw::launch_raygen(ctx[g].cfg, cam, ctx[g].fr, ctx[g].buf[0], ctx[g].srad, ctx[g].n_paths); });
        }
        for (int seg = 0; seg < fw::MAX_SEGMENTS; seg++)
            for (int g = 0; g < group; g++)
                if (int src = segment(b0 + (uint32_t)g, ctx[g], seg, g, group)) return src;
        for (int g = 0; g < group; g++)
            if (int erc = end_batch(b0 + (uint32_t)g, ctx[g])) return erc;
        fr = ctx[group - 1].fr;        // (the frame-level fields the code below reads are the same in every batch)
    }
    if (n_lanes > 1) HIPCHK(hipStreamWaitEvent(origin, ws->events[3 + n_batches - 1], 0));    // join
    return FW_OK;
    };
    // GRAPH: launch-bound frames (random_spheres: 37 launches of 10-80 us with ~6 us between them; a rank's eighth of a frame) replayed as
    // one graph.  The first time a frame is asked for it runs as ever; the second time in a row it is captured (on a stream of the
    // workspace's own: the caller's may be the legacy stream, which cannot be captured), instantiated and launched; from then on launched.
    // Any failure on the way turns the feature off for the workspace and the frame runs as ever.
    {
        Workspace::FrameGraph &fg = ws->fg;
        constexpr uint64_t GRAPH_MAX_CHUNKS = 1u << 19;      // batches below 33 M paths
        const bool graph_ok = O.graph != 0 && !fg.broken && !timing && !dump_one && !phase_lock && !stagger && (O.graph == 1 || chunks < GRAPH_MAX_CHUNKS);   // (a capture with PHASE_LOCK's events crashed inside the runtime: the two never meet by default — the lock wants batches of 100 M paths)
        uint64_t key = 0;
        if (graph_ok) {
            uint64_t h = 1469598103934665603ull;
            auto mix = [&](const void *ptr, size_t n) { const unsigned char *b = (const unsigned char *)ptr; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } };
            fw::LaunchCfg kc = cfg; kc.stream = nullptr;
            mix(&cam, sizeof cam); mix(&fr, sizeof fr); mix(&kc, sizeof kc); mix(&sc->d, sizeof sc->d); mix(&q, sizeof q);
            const uint64_t scalars[] = {n_batches, (uint64_t)n_lanes, spp_b, first_sample, p->samples, n_pix, max_paths, cap, exact_mode, (uint64_t)park_meshes, (uint64_t)phase_lock,
                                        (uint64_t)count_deposits, (uint64_t)exact_product, (uint64_t)use_bvh, (uint64_t)fused, (uint64_t)stagger, (uint64_t)(uintptr_t)accum, (uint64_t)(uintptr_t)ws->totals.p,
                                        (uint64_t)(uintptr_t)ws->arena.p, (uint64_t)ws->arena.bytes};
            mix(scalars, sizeof scalars);
            for (int l = 0; l < n_lanes; l++) { const Workspace::Lane &L = ws->lanes[l]; const void *ptrs[] = {L.ray_a[0], L.ray_a[1], L.ray_b[0], L.ray_b[1], L.state[0], L.state[1], L.hits, L.sample_rad, L.wcount, L.park_a, L.park_b, L.park_m, L.pcount, L.dep_bits, L.exact_slots, L.atten}; mix(ptrs, sizeof ptrs); }
            key = h | 1ull;
        }
        bool done = false;
        if (graph_ok && fg.exec && fg.key == key) {
            HIPCHK(hipGraphLaunch(fg.exec, stream));
            fr = fg.fr_after; done = true; graph_replayed = true;
        } else if (graph_ok && fg.seen == key) {
            if (!fg.origin && hipStreamCreateWithFlags(&fg.origin, hipStreamNonBlocking) != hipSuccess) { fg.origin = nullptr; fg.broken = true; }
            if (!fg.broken && hipStreamBeginCapture(fg.origin, hipStreamCaptureModeRelaxed) == hipSuccess) {
                origin = fg.origin;
                if (O.trace) fprintf(stderr, "[firework] GRAPH: capturing (%u batches, %d lanes, phase lock %d)\n", n_batches, n_lanes, (int)phase_lock);
                const int crc = enqueue_frame();
                origin = stream;
                hipGraph_t g = nullptr;
                const hipError_t ee = hipStreamEndCapture(fg.origin, &g);
                if (O.trace) fprintf(stderr, "[firework] GRAPH: capture ended: enqueue %d, end %s\n", crc, hipGetErrorString(ee));
                if (fg.exec) { (void)hipGraphExecDestroy(fg.exec); fg.exec = nullptr; fg.key = 0; }
                if (crc == FW_OK && ee == hipSuccess && g && hipGraphInstantiate(&fg.exec, g, nullptr, nullptr, 0) == hipSuccess) {
                    fg.key = key; fg.fr_after = fr;
                    (void)hipGraphDestroy(g);
                    if (O.trace) fprintf(stderr, "[firework] GRAPH: instantiated\n");
                    HIPCHK(hipGraphLaunch(fg.exec, stream));
                    if (O.trace) fprintf(stderr, "[firework] GRAPH: launched\n");
                    graph_replayed = true;
                    done = true;
                } else {
                    if (g) (void)hipGraphDestroy(g);
                    fg.exec = nullptr; fg.broken = true; (void)hipGetLastError();
                    if (O.trace) fprintf(stderr, "[firework] GRAPH: capture failed (%d, %s): frames run as plain launches from here on\n", crc, hipGetErrorString(ee));
                }
            } else { fg.broken = true; (void)hipGetLastError(); }
        }
        if (graph_ok) fg.seen = key;
        if (!done) { if (int frc = enqueue_frame()) return frc; }
    }
    cfg.stream = stream;
    fw::launch_resolve(cfg, fr, (const float4 *)ws->accum.p, first_sample + p->samples, p->gamma, d_rgb8, d_gamma, d_linear);
    if (user_accum) HIPCHK(hipMemcpyAsync(user_accum, ws->accum.p, (size_t)n_pix * 16, p->outputs_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, stream));
    HIPCHK(hipEventRecord(ws->events[1], stream));
    HIPCHK(hipGetLastError());

    if (dump_one) {
        if (FILE *fp = fopen(dump_file, "wb")) {
            uint32_t hdr[8] = {0x46574450u, fr.pinhole0, fr.hit4, sc->d.prim_bits, 0, 0, 0, cfg.n_defer};
            std::memcpy(&hdr[4], fr.cam_pos, 12);
            fwrite(hdr, 4, 8, fp); fwrite(dump_rec.data(), 4, dump_rec.size(), fp); fclose(fp);
        }
    }
    const bool trace = O.trace;
    const auto tq0 = std::chrono::steady_clock::now();
    auto since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    // Device -> host through PINNED memory, then a host memcpy into the caller's buffers.  A hipMemcpyAsync into pageable
    // memory blocks inside the runtime until the stream has drained, and that wait returned 15-30 ms late every few calls
    // (FIREWORK_TRACE=1, tools/oneshot.py: "counters copy returned +69 ms" for 43 ms of device work); with pinned
    // destinations the copies are queued and the only host wait is the hipStreamSynchronize below.
    const size_t n_counts = (size_t)n_batches * fw::COUNT_STRIDE;
    const size_t off_c = 0, off_8 = (n_counts * 4 + 255) & ~(size_t)255;
    const size_t off_g = off_8 + ((!p->outputs_on_device && rgb8 ? (size_t)n_pix * 3 : 0) + 255 & ~(size_t)255);
    const size_t off_l = off_g + ((!p->outputs_on_device && gamma_rgb ? (size_t)n_pix * 12 : 0) + 255 & ~(size_t)255);
    const size_t host_need = off_l + (!p->outputs_on_device && linear_rgb ? (size_t)n_pix * 12 : 0) + 256;
    if (ws->host_out_bytes < host_need) {
        if (ws->host_out) (void)hipHostFree(ws->host_out);
        ws->host_out = nullptr; ws->host_out_bytes = 0;
        if (hipHostMalloc(&ws->host_out, host_need + host_need / 4, hipHostMallocDefault) != hipSuccess) return fail(FW_ERR_OOM, "pinned output staging allocation failed");
        ws->host_out_bytes = host_need + host_need / 4;
    }
    uint8_t *ho = (uint8_t *)ws->host_out;
    const uint32_t *h_counts = (const uint32_t *)(ho + off_c);
    HIPCHK(hipMemcpyAsync(ho + off_c, ws->totals.p, n_counts * 4, hipMemcpyDeviceToHost, stream));
    const double t_counts = since(tq0);
    if (!p->outputs_on_device) {
        if (rgb8) HIPCHK(hipMemcpyAsync(ho + off_8, d_rgb8, (size_t)n_pix * 3, hipMemcpyDeviceToHost, stream));
        if (gamma_rgb) HIPCHK(hipMemcpyAsync(ho + off_g, d_gamma, (size_t)n_pix * 12, hipMemcpyDeviceToHost, stream));
        if (linear_rgb) HIPCHK(hipMemcpyAsync(ho + off_l, d_linear, (size_t)n_pix * 12, hipMemcpyDeviceToHost, stream));
    }
    const double t_outs = since(tq0);
    if (!ws->ev_d2h) HIPCHK(hipEventCreate(&ws->ev_d2h));
    HIPCHK(hipEventRecord(ws->ev_d2h, stream));
    HIPCHK(hipStreamSynchronize(stream));
    if (!p->outputs_on_device) {
        if (rgb8) std::memcpy(rgb8, ho + off_8, (size_t)n_pix * 3);
        if (gamma_rgb) std::memcpy(gamma_rgb, ho + off_g, (size_t)n_pix * 12);
        if (linear_rgb) std::memcpy(linear_rgb, ho + off_l, (size_t)n_pix * 12);
    }
#if FW_AB
    if (const uint32_t ew = fw::take_error_word())      // the A/B build's device-side guards (fw_kernels.hip: g_err_word): the frame is not to be trusted
        return fail(FW_ERR_HIP, std::string("device error word: ") + ((ew & 1u) ? "LDS traversal stack overflow (a push beyond the levels the launch reserved) " : "") +
                                    ((ew & 2u) ? "a walk kernel's wave made no progress for 2^24 rounds (left its loop) " : ""));
#endif
    if (trace) fprintf(stderr, "[firework] render: enqueue %.2f ms | counters copy queued +%.2f | output copies queued +%.2f | sync + host memcpy returned +%.2f\n",
                       std::chrono::duration<double, std::milli>(tq0 - wall0).count(), t_counts, t_outs, since(tq0));
    const auto wall1 = std::chrono::steady_clock::now();

    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        stats->samples = (uint64_t)n_pix * p->samples;
        uint64_t deposits = 0;
        for (uint32_t b = 0; b < n_batches; b++) {
            for (int s = 0; s < fw::MAX_SEGMENTS; s++) { uint64_t c = h_counts[(size_t)b * fw::COUNT_STRIDE + s]; stats->rays_per_depth[s] += c; stats->rays += c; }
            stats->parked_rays += h_counts[(size_t)b * fw::COUNT_STRIDE + fw::MAX_SEGMENTS];
            deposits += h_counts[(size_t)b * fw::COUNT_STRIDE + 12];
        }
        stats->deposits = count_deposits ? deposits : stats->samples;      // every path ends exactly once (render.rs:19-31)
        {   // HBM bytes this layout moves (fw_device.h B_*): queue streams only, scene tables are cache-resident
            const uint64_t *R = stats->rays_per_depth;
            const uint64_t S = stats->samples, ray0 = fr.pinhole0 ? fw::B_RAY_PINHOLE0 : fw::B_RAY, b_hit = fr.hit4 ? fw::B_HIT4 : fw::B_HIT;
            uint64_t rd_ray = R[0] * ray0, later = 0, survivors = 0;
            for (int s = 1; s < fw::MAX_SEGMENTS; s++) { rd_ray += R[s] * fw::B_RAY; later += R[s]; survivors += R[s]; }
            stats->bytes_raygen = S * ray0 + (fr.pixel_ids ? S * 4 : 0);
            const uint64_t medium = sc->d.has_medium ? later * 4 : 0;       // the path's home slot (RNG key of the medium's draw)
            const uint64_t b_state = fr.chain_bits ? fw::B_STATE_CHAIN : fw::B_STATE;
            // (EXACT_PRODUCT: one attenuation record written per survivor; a depositing path reads back its own — at most its length: not counted)
            const uint64_t shade_in = rd_ray + later * b_state, shade_out = survivors * (fw::B_RAY + b_state + (fr.atten ? fw::B_ATTEN : 0u)) + stats->deposits * fw::B_DEPOSIT;
            if (fused) { stats->bytes_extend = 0; stats->bytes_shade = shade_in + shade_out; }
            else {
                stats->bytes_extend = rd_ray + medium + stats->rays * b_hit + stats->parked_rays * 2 * fw::B_PARK;
                stats->bytes_shade = shade_in + stats->rays * b_hit + shade_out;
            }
            stats->bytes_accumulate = (fr.skip_zero_deposits ? stats->deposits * fw::B_DEPOSIT + S / 8 : S * fw::B_DEPOSIT) + (uint64_t)n_batches * n_pix * 2 * fw::B_ACCUM;
        }
        stats->ms_wall = std::chrono::duration<double, std::milli>(wall1 - wall0).count();
        float ms_copy = 0.f;
        HIPCHK(hipEventElapsedTime(&ms_copy, ws->events[1], ws->ev_d2h));   // counters (a few hundred bytes) + the outputs
        stats->ms_d2h = ms_copy;
        stats->algorithmic_bytes = 160 * stats->rays + 24 * stats->samples;   // SURVEY §8(d); HDR env misses are added by the caller that knows them
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, ws->events[0], ws->events[1]));
        stats->ms_render = ms;
        if (timing) {
            double acc[4] = {0, 0, 0, 0};
            for (int l = 0; l < n_lanes; l++)
                for (size_t i = 0; i < ev_class[l].size(); i++) {
                    float t = 0.f;
                    HIPCHK(hipEventElapsedTime(&t, ws->lanes[l].events[i], ws->lanes[l].events[i + 1]));
                    acc[ev_class[l][i]] += t;
                }
            stats->ms_raygen = acc[0]; stats->ms_extend = acc[1]; stats->ms_shade = acc[2]; stats->ms_accumulate = acc[3];
        }
        stats->n_extend_launches = fused ? 0 : n_batches * fw::MAX_SEGMENTS; stats->n_shade_launches = n_batches * fw::MAX_SEGMENTS;
        stats->n_batches = n_batches; stats->tlas_nodes = sc->tlas_nodes; stats->blas_nodes = sc->blas_nodes;
        stats->reserved = ((sc->tlas_depth & 0x7fffu) << 16) | (sc->blas_depth & 0xffffu) | (graph_replayed ? 0x80000000u : 0u);   // depths of the trees actually walked; bit 31: the frame ran as a hipGraph (GRAPH)
    }
    return FW_OK;
}

} // namespace

// =========================================================================================================
extern "C" {

int fw_abi_version(void) { return FW_ABI_VERSION; }

const char *fw_strerror(int s) {
    switch (s) {
    case FW_OK: return "ok";
    case FW_ERR_BAD_ARG: return "bad argument";
    case FW_ERR_EMPTY_SCENE: return "No render objects added to scene!";
    case FW_ERR_NAN_BBOX: return "Float comparison failed in BVH constructor";
    case FW_ERR_MESH_NORMALS: return "TriangleMesh::new() -- normals.len() must equal verts.len()";
    case FW_ERR_MESH_UVS: return "TriangleMesh::new() -- uvs.len() must equal verts.len()";
    case FW_ERR_UNSUPPORTED: return "unsupported on the HIP path";
    case FW_ERR_HIP: return "HIP runtime error";
    case FW_ERR_NO_DEVICE: return "no HIP device (no CPU fallback exists)";
    case FW_ERR_BVH_DEPTH: return "BVH deeper than the traversal stack";
    case FW_ERR_OOM: return "out of memory";
    default: return "unknown error";
    }
}

const char *fw_last_error(void) { return g_last_error.c_str(); }

int fw_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int fw_selftest_arith(int device, uint32_t n, uint32_t seed, int mode, uint64_t *div_mismatches, uint64_t *sqrt_mismatches) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(FW_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev || !div_mismatches || !sqrt_mismatches) return fail(FW_ERR_BAD_ARG, "bad argument");
    HIPCHK(hipSetDevice(device));
    unsigned long long *d = nullptr, h[2] = {0, 0};
    HIPCHK(hipMalloc(&d, sizeof h));
    HIPCHK(hipMemset(d, 0, sizeof h));
    fw::launch_selftest_arith(nullptr, n, seed, mode, d);
    hipError_t e = hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(FW_ERR_HIP, hipGetErrorString(e));
    *div_mismatches = h[0]; *sqrt_mismatches = h[1];
    return FW_OK;
}

int fw_selftest_libm(int device, int fn, uint32_t n, const float *x, const float *y, float *out) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(FW_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev || fn < 0 || fn > 7 || !x || !out || n == 0 || ((fn == 6 || fn == 7) && !y)) return fail(FW_ERR_BAD_ARG, "bad argument");
    HIPCHK(hipSetDevice(device));
    DevBuf dx, dy, dout;
    int rc = dx.alloc((size_t)n * 4);
    if (!rc && y) rc = dy.alloc((size_t)n * 4);
    if (!rc) rc = dout.alloc((size_t)n * 4);
    hipError_t e = hipSuccess;
    if (!rc) {
        e = hipMemcpy(dx.p, x, (size_t)n * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess && y) e = hipMemcpy(dy.p, y, (size_t)n * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            fw::launch_selftest_libm(nullptr, fn, n, (const float *)dx.p, y ? (const float *)dy.p : nullptr, (float *)dout.p);
            e = hipMemcpy(out, dout.p, (size_t)n * 4, hipMemcpyDeviceToHost);
        }
    }
    dx.release(); dy.release(); dout.release();
    if (rc) return rc;
    if (e != hipSuccess) return fail(FW_ERR_HIP, hipGetErrorString(e));
    return FW_OK;
}

int fw_set_option(const char *name, const char *value) {
    std::lock_guard<std::mutex> g(g_opt_mu);
    if (!name) { g_opt = options_from_env(); return FW_OK; }          // back to what the environment said
    const char *n = std::strncmp(name, "FIREWORK_", 9) == 0 ? name + 9 : name;
    if (!option_apply(g_opt, n, value)) return fail(FW_ERR_BAD_ARG, std::string("unknown option ") + name);
    return FW_OK;
}

// CPU-only diagnostic: the WIDE-node builder (wide_convert) on a caller's item boxes, its invariants checked on the finished tree
// (every item exactly once, every decoded child box a superset of the exact one, free slots unhittable, f32 item boxes bit for bit).
// boxes: n x 6 floats (min.xyz max.xyz); format: FW_WIDE_F32 (1) or FW_WIDE_Q8 (2); stats: nodes, leaves, free slots, depth.
int fw_selftest_wide_bvh(const float *boxes, uint32_t n, int format, uint32_t *violations, uint32_t stats[4]) {
    if (!boxes || n == 0 || !violations || !stats || (format != fw::WIDE_F32 && format != fw::WIDE_Q8)) return fail(FW_ERR_BAD_ARG, "bad argument");
    try {
        std::vector<Box> b(n);
        for (uint32_t i = 0; i < n; i++) b[i] = Box{{boxes[6 * i], boxes[6 * i + 1], boxes[6 * i + 2]}, {boxes[6 * i + 3], boxes[6 * i + 4], boxes[6 * i + 5]}};
        FlatBvh sah; sah_build(sah, b);
        WideBvh w; w.fmt = format;
        const uint32_t root = wide_convert(sah, b, w);
        if (root == 0xffffffffu) return fail(FW_ERR_UNSUPPORTED, "tree not encodable as wide nodes (more than 32767 items or nodes, or a box that is not finite)");
        *violations = wide_check(w, root, b, stats);
        return FW_OK;
    }
    catch (std::bad_alloc &) { return fail(FW_ERR_OOM, "host allocation failed"); }
    catch (...) { return fail(FW_ERR_BAD_ARG, "unexpected exception in fw_selftest_wide_bvh"); }
}

// CPU-only diagnostic (ABI v7): the host builders — the reference's median-split tree (bvh.rs:21-71) and the binned-SAH tree that is walked —
// over n item boxes with `threads` host threads (1 = the sequential recursion); hashes[0..1] = FNV-1a of the two node arrays, stats = nodes and
// depth of each.  The parallel builds must reproduce the sequential ones bit for bit (tests/test_host_build_cpu.py).
int fw_selftest_bvh_build(const float *boxes, uint32_t n, int threads, uint64_t hashes[2], uint32_t stats[4]) {
    if (!boxes || n == 0 || threads < 1 || !hashes || !stats) return fail(FW_ERR_BAD_ARG, "bad argument");
    try {
        std::vector<Box> b(n);
        for (uint32_t i = 0; i < n; i++) b[i] = Box{{boxes[6 * i], boxes[6 * i + 1], boxes[6 * i + 2]}, {boxes[6 * i + 3], boxes[6 * i + 4], boxes[6 * i + 5]}};
        auto fnv = [](const std::vector<float> &v) { uint64_t h = 1469598103934665603ull; const uint8_t *p = (const uint8_t *)v.data(); for (size_t i = 0; i < v.size() * 4; i++) { h ^= p[i]; h *= 1099511628211ull; } return h; };
        FlatBvh ref, sah;
        { BuildPool pool(threads - 1); try { (void)bvh_build(ref, b, &pool); } catch (NanError &) { return fail(FW_ERR_NAN_BBOX, "Float comparison failed in BVH constructor"); } }
        { BuildPool pool(threads - 1); sah_build(sah, b, &pool); }
        hashes[0] = fnv(ref.nodes); hashes[1] = fnv(sah.nodes);
        stats[0] = ref.count(); stats[1] = ref.depth; stats[2] = sah.count(); stats[3] = sah.depth;
        return FW_OK;
    }
    catch (std::bad_alloc &) { return fail(FW_ERR_OOM, "host allocation failed"); }
    catch (...) { return fail(FW_ERR_BAD_ARG, "unexpected exception in fw_selftest_bvh_build"); }
}

#if FW_AB
int fw_debug_ab(void) { return 1; }      // present only in the A/B build: tests of the alternative kernels look for it
#endif

int fw_init(int device, uint64_t arena_bytes) {
    try {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return fail(FW_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU path)"); }
        if (device < 0 || device >= ndev) return fail(FW_ERR_BAD_ARG, "device index out of range");
        HIPCHK(hipSetDevice(device));
        Workspace *ws = workspace_for(device);
        if (!ws) return fail(FW_ERR_OOM, "no workspace for this device");
        std::lock_guard<std::mutex> g(ws->mu);
        int rc = init_device_locked(ws, device);
        if (rc) return rc;
        size_t want = arena_bytes == FW_INIT_NO_ARENA ? 0 : (arena_bytes ? (size_t)arena_bytes : default_arena_bytes(ws));
        want = (want + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
        if (want > ws->arena.bytes) {
            const auto ta = std::chrono::steady_clock::now();
            rc = arena_reserve_locked(ws, device, want);
            if (options().trace) fprintf(stderr, "[firework] fw_init: path arena of %.1f GiB in %.2f ms\n", (double)want / (double)(1 << 30),
                                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ta).count());
        }
        return rc;
    }
    catch (std::bad_alloc &) { return fail(FW_ERR_OOM, "host allocation failed"); }
    catch (...) { return fail(FW_ERR_BAD_ARG, "unexpected exception in fw_init"); }
}

int fw_scene_create(const fw_scene_desc *desc, int device, fw_scene **out) {
    try { return create_scene_impl(desc, device, out); }
    catch (std::bad_alloc &) { return fail(FW_ERR_OOM, "host allocation failed"); }
    catch (...) { return fail(FW_ERR_BAD_ARG, "unexpected exception in fw_scene_create"); }
}

void fw_release_workspace(int device) {
    Workspace *ws = workspace_for(device);
    if (!ws) return;
    std::lock_guard<std::mutex> g(ws->mu);
    if (hipSetDevice(device) == hipSuccess) ws->release();
}

void fw_scene_destroy(fw_scene *scene) {
    if (!scene) return;
    (void)hipSetDevice(scene->device);
    if (Workspace *ws = workspace_for(scene->device)) {   // keep the allocation for the next scene (one-shot renders)
        std::lock_guard<std::mutex> g(ws->mu);
        if (scene->data.p && scene->data.bytes > ws->scene_cache.bytes) { ws->scene_cache.release(); ws->scene_cache = scene->data; scene->data = DevBuf{}; }
    }
    delete scene;
}

int fw_render(fw_scene *scene, const fw_render_params *params, uint8_t *rgb8, float *gamma_rgb, float *linear_rgb, fw_stats *stats) {
    try { return render_impl(scene, params, rgb8, gamma_rgb, linear_rgb, stats); }
    catch (std::bad_alloc &) { return fail(FW_ERR_OOM, "host allocation failed"); }
    catch (...) { return fail(FW_ERR_BAD_ARG, "unexpected exception in fw_render"); }
}

// ---- single-process multi-GPU: one host thread per device, 16x16 tiles dealt diagonally (the scheme of firework_amd/tiles.py),
// each device renders its pixels with the keys one GPU would use, results are scattered into the caller's buffers.
int fw_render_scene_tiled(const fw_scene_desc *desc, const fw_render_params *params, const int *devices, int n_devices,
                          uint8_t *rgb8, float *gamma_rgb, float *linear_rgb, fw_stats *stats) {
    if (!desc || !params || !devices || n_devices <= 0) return fail(FW_ERR_BAD_ARG, "null argument");
    if (params->pixel_ids || params->outputs_on_device) return fail(FW_ERR_BAD_ARG, "fw_render_scene_tiled renders whole frames into host buffers");
    if (params->width == 0 || params->height == 0) return fail(FW_ERR_BAD_ARG, "width, height and samples must be > 0");
    const auto call0 = std::chrono::steady_clock::now();
    try {
        const uint32_t W = params->width, H = params->height, TILE = 16, tx = (W + TILE - 1) / TILE, ty = (H + TILE - 1) / TILE;
        const int N = n_devices;
        std::vector<std::vector<uint32_t>> ids(N);
        for (uint32_t t = 0; t < tx * ty; t++) {
            const uint32_t owner = (t % tx + t / tx) % (uint32_t)N, y0 = (t / tx) * TILE, x0 = (t % tx) * TILE;
            for (uint32_t y = y0; y < std::min(y0 + TILE, H); y++) for (uint32_t x = x0; x < std::min(x0 + TILE, W); x++) ids[owner].push_back(y * W + x);
        }
        // Device-side gather (round 2): every device renders its tiles into its OWN memory, copies them peer-to-peer
        // (hipMemcpyPeer: over xGMI where the devices are linked) into one buffer on the first device, which scatters them to
        // their pixels and sends the finished frames to the host once — instead of one D2H per device and a host-side scatter.
        struct Part { int rc = FW_OK; std::string err; fw_stats st{}; double ms_scene = 0; };
        std::vector<Part> parts(N);
        std::vector<size_t> first(N + 1, 0);                       // part r holds the concatenated entries [first[r], first[r+1])
        for (int r = 0; r < N; r++) first[r + 1] = first[r] + ids[r].size();
        const size_t n_total = first[N];
        const int dev0 = devices[0];
        HIPCHK(hipSetDevice(dev0));
        // on the first device: [ids | gathered rgb8 | gamma | linear | frame rgb8 | gamma | linear], 256-byte aligned sections
        auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
        const size_t b8 = rgb8 ? n_total * 3 : 0, bg = gamma_rgb ? n_total * 12 : 0, bl = linear_rgb ? n_total * 12 : 0;
        const size_t o_ids = 0, o_g8 = al(n_total * 4), o_gg = o_g8 + al(b8), o_gl = o_gg + al(bg), o_f8 = o_gl + al(bl), o_fg = o_f8 + al(b8),
                     o_fl = o_fg + al(bg), dev_bytes = o_fl + al(bl) + 256;
        struct GatherBuf : DevBuf { int dev; explicit GatherBuf(int d) : dev(d) {} ~GatherBuf() { if (p) { (void)hipSetDevice(dev); release(); } } };
        GatherBuf gather(dev0);                                    // released on every way out of this function, exceptions included
        int grc = gather.alloc(dev_bytes);
        if (grc) return grc;
        uint8_t *gb = (uint8_t *)gather.p;
        {
            std::vector<uint32_t> all_ids; all_ids.reserve(n_total);
            for (int r = 0; r < N; r++) all_ids.insert(all_ids.end(), ids[r].begin(), ids[r].end());
            if (hipMemcpy(gb + o_ids, all_ids.data(), n_total * 4, hipMemcpyHostToDevice) != hipSuccess) return fail(FW_ERR_HIP, "tile id upload failed");
        }
        // Peer access to the first device, asked for once per pair (hipMemcpyPeer works without it, staged through the host by the
        // runtime; with it the copy goes over the xGMI link).  Where a pair has no peer path the tiles are staged through pinned
        // host memory here, and fw_last_error() says so after a successful call.
        std::vector<char> peer_ok(N, 1);
        std::string peer_note;
        for (int r = 0; r < N; r++) {
            if (devices[r] == dev0) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[r], dev0) != hipSuccess) can = 0;
            if (can) {
                (void)hipSetDevice(devices[r]);
                const hipError_t pe = hipDeviceEnablePeerAccess(dev0, 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) can = 0;
                (void)hipGetLastError();
            }
            if (!can) { peer_ok[r] = 0; peer_note += (peer_note.empty() ? "no peer access to device " : ", ") + std::to_string(devices[r]); }
        }
        (void)hipSetDevice(dev0);
        if (!peer_note.empty()) peer_note = peer_note + " from device " + std::to_string(dev0) + ": those tiles were staged through host memory";
        std::vector<std::thread> threads;
        struct Joiner { std::vector<std::thread> &t; ~Joiner() { for (auto &x : t) if (x.joinable()) x.join(); } } joiner{threads};   // a failed emplace_back must not destroy joinable threads
        threads.reserve((size_t)N);
        for (int r = 0; r < N; r++) threads.emplace_back([&, r] {
            Part &pt = parts[r];
            const size_t n = ids[r].size();
            if (n == 0) return;
            try {
                auto t0 = std::chrono::steady_clock::now();
                fw_scene *sc = nullptr;
                pt.rc = fw_scene_create(desc, devices[r], &sc);
                if (pt.rc) { pt.err = g_last_error; return; }
                pt.ms_scene = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                DevBuf part;                                            // this device's tiles, in its own HBM
                const size_t p8 = rgb8 ? n * 3 : 0, pg = gamma_rgb ? n * 12 : 0, pl = linear_rgb ? n * 12 : 0;
                const size_t q8 = 0, qg = al(p8), ql = qg + al(pg);
                (void)hipSetDevice(devices[r]);
                pt.rc = part.alloc(ql + al(pl) + 256);
                if (!pt.rc) {
                    uint8_t *pb = (uint8_t *)part.p;
                    fw_render_params p = *params;
                    p.pixel_ids = ids[r].data(); p.n_pixels = (uint32_t)n; p.stream = nullptr; p.outputs_on_device = 1;
                    pt.rc = fw_render(sc, &p, rgb8 ? pb + q8 : nullptr, gamma_rgb ? (float *)(pb + qg) : nullptr, linear_rgb ? (float *)(pb + ql) : nullptr, &pt.st);
                    if (pt.rc) pt.err = g_last_error;
                    auto peer = [&](size_t dst_off, size_t src_off, size_t bytes) {
                        if (pt.rc || !bytes) return;
                        if (peer_ok[r]) {
                            if (hipMemcpyPeer(gb + dst_off, dev0, pb + src_off, devices[r], bytes) != hipSuccess) { pt.rc = FW_ERR_HIP; pt.err = "hipMemcpyPeer of a device's tiles failed"; }
                            return;
                        }
                        void *host = nullptr;                            // no peer path: device -> pinned host -> first device
                        if (hipHostMalloc(&host, bytes, hipHostMallocDefault) != hipSuccess) { pt.rc = FW_ERR_OOM; pt.err = "pinned staging for a device's tiles failed"; return; }
                        hipError_t e = hipMemcpy(host, pb + src_off, bytes, hipMemcpyDeviceToHost);
                        if (e == hipSuccess) { (void)hipSetDevice(dev0); e = hipMemcpy(gb + dst_off, host, bytes, hipMemcpyHostToDevice); (void)hipSetDevice(devices[r]); }
                        (void)hipHostFree(host);
                        if (e != hipSuccess) { pt.rc = FW_ERR_HIP; pt.err = "host-staged copy of a device's tiles failed"; }
                    };
                    peer(o_g8 + first[r] * 3, q8, p8); peer(o_gg + first[r] * 12, qg, pg); peer(o_gl + first[r] * 12, ql, pl);
                } else pt.err = g_last_error;
                part.release();
                fw_scene_destroy(sc);
            } catch (...) { pt.rc = FW_ERR_OOM; pt.err = "host allocation failed in a tile worker"; }
        });
        for (auto &t : threads) t.join();
        for (int r = 0; r < N; r++) if (parts[r].rc) return fail(parts[r].rc, parts[r].err);
        // the one scatter and the one device -> host transfer
        (void)hipSetDevice(dev0);
        fw::launch_scatter_tiles(nullptr, (const uint32_t *)(gb + o_ids), (uint32_t)n_total, rgb8 ? gb + o_g8 : nullptr, gamma_rgb ? (const float *)(gb + o_gg) : nullptr,
                                 linear_rgb ? (const float *)(gb + o_gl) : nullptr, gb + o_f8, (float *)(gb + o_fg), (float *)(gb + o_fl));
        hipError_t ce = hipGetLastError();                          // the scatter launch itself
        if (rgb8 && ce == hipSuccess) ce = hipMemcpy(rgb8, gb + o_f8, b8, hipMemcpyDeviceToHost);
        if (gamma_rgb && ce == hipSuccess) ce = hipMemcpy(gamma_rgb, gb + o_fg, bg, hipMemcpyDeviceToHost);
        if (linear_rgb && ce == hipSuccess) ce = hipMemcpy(linear_rgb, gb + o_fl, bl, hipMemcpyDeviceToHost);
        if (ce != hipSuccess) return fail(FW_ERR_HIP, hipGetErrorString(ce));
        if (!peer_note.empty()) g_last_error = peer_note;          // informational: the call succeeded
        if (stats) std::memset(stats, 0, sizeof *stats);
        for (int r = 0; r < N; r++) {
            const Part &pt = parts[r];
            if (stats && !ids[r].empty()) {
                stats->samples += pt.st.samples; stats->rays += pt.st.rays; stats->algorithmic_bytes += pt.st.algorithmic_bytes;
                stats->bytes_raygen += pt.st.bytes_raygen; stats->bytes_extend += pt.st.bytes_extend; stats->bytes_shade += pt.st.bytes_shade;
                stats->bytes_accumulate += pt.st.bytes_accumulate; stats->deposits += pt.st.deposits; stats->parked_rays += pt.st.parked_rays;
                stats->ms_d2h = std::max(stats->ms_d2h, pt.st.ms_d2h);
                for (int k = 0; k < FW_MAX_SEGMENTS; k++) stats->rays_per_depth[k] += pt.st.rays_per_depth[k];
                stats->ms_render = std::max(stats->ms_render, pt.st.ms_render); stats->ms_scene = std::max(stats->ms_scene, pt.ms_scene);
                stats->n_batches = std::max(stats->n_batches, pt.st.n_batches);
                stats->tlas_nodes = pt.st.tlas_nodes; stats->blas_nodes = pt.st.blas_nodes; stats->reserved = pt.st.reserved;
            }
        }
        if (stats) stats->ms_wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - call0).count();
        return FW_OK;
    }
    catch (std::bad_alloc &) { return fail(FW_ERR_OOM, "host allocation failed"); }
    catch (...) { return fail(FW_ERR_BAD_ARG, "unexpected exception in fw_render_scene_tiled"); }
}

int fw_render_progressive(fw_scene *scene, const fw_render_params *params, uint32_t first_sample, float *accum,
                          uint8_t *rgb8, float *gamma_rgb, float *linear_rgb, fw_stats *stats) {
    if (!accum) return fail(FW_ERR_BAD_ARG, "fw_render_progressive needs an accumulation buffer");
    try { return render_impl(scene, params, rgb8, gamma_rgb, linear_rgb, stats, first_sample, accum); }
    catch (std::bad_alloc &) { return fail(FW_ERR_OOM, "host allocation failed"); }
    catch (...) { return fail(FW_ERR_BAD_ARG, "unexpected exception in fw_render_progressive"); }
}

int fw_render_scene(const fw_scene_desc *desc, const fw_render_params *params, int device, uint8_t *rgb8, float *gamma_rgb,
                    float *linear_rgb, fw_stats *stats) {
    auto t0 = std::chrono::steady_clock::now();
    fw_scene *sc = nullptr;
    int rc = fw_scene_create(desc, device, &sc);
    if (rc) return rc;
    double ms_scene = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    rc = fw_render(sc, params, rgb8, gamma_rgb, linear_rgb, stats);
    fw_scene_destroy(sc);
    if (!rc && stats) {
        stats->ms_scene = ms_scene;
        stats->ms_wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();   // main.rs:40-44's region
    }
    return rc;
}

} // extern "C"
