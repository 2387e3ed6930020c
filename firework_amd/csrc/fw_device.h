// fw_device.h — HBM data layout shared by the host runtime (fw_runtime.cpp) and the kernels
// (fw_kernels.hip).  See DESIGN.md §"Data layout in HBM".
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// FW_AB=1 (make ab): the A/B build, which also carries the alternatives that were measured slower and taken out of the product —
// k_bounce (one launch per segment), the chunked k_extend_bvh, k_shade's list of expensive materials — with their switches, so
// that the bit-identity tests and tools/ can still run them.  The product build (0) has neither the kernels nor the switches.
#ifndef FW_AB
#define FW_AB 0
#endif

namespace fw {

// ---- object record: 6 x float4 = 96 B, 16-B aligned (one RenderObjectInternal, scene.rs:165-174,
//      with its shape inlined).  Linear scenes read it with a wave-uniform index (scalar loads),
//      TLAS leaves with a per-lane index (float4 vector loads).
//   r0 = (pos.xyz, bits: kind | flags<<8 | inner_kind<<24)      <- all an unrotated sphere needs with r1
//   r1 = q3, r2 = q4 = shape parameters (below)
//   r3 = (R00 R01 R02, bits material)   rows of rotation_mat (inv_rotation_mat is exactly the transpose:
//   r4 = (R10 R11 R12, bits aux0)        scene.rs:285)
//   r5 = (R20 R21 R22, bits aux1)
constexpr int OBJ_Q = 6;

enum : uint32_t {
    OF_ROTATED = 1u << 0,      // 0.5*(trace R - 1) < 0.999  (scene.rs:187,242)
    OF_FLIP = 1u << 1,         // RenderObject.flip_normals
    OF_RECT_FLIP = 1u << 2,    // AARect.flip_normal
    OF_MESH_NORMALS = 1u << 3, // TriangleMesh.normals is Some
    OF_MESH_ATTR = 1u << 4,    // the per-triangle attribute array (normals and/or v coordinates) is valid
    OF_CULL0 = 1u << 6,        // expensive shape (Rect3d, mesh, cone, cylinder): camera-ray chunks test its inflated world box first
    OF_GATE = 1u << 5,         // the shape's own bounding box does not enclose it (Disk: disk.rs:85-90): under use_bvh the
                               // object is tested only if the ray hits the box of its leaf node in the REFERENCE tree
};
// shape parameters:
//   sphere : q3.x = radius
//   cone / cylinder / disk : q3 = (radius height phi_max inner_radius)
//   rect   : q3 = (a_min a_max b_min b_max)  q4.x = k
//   rect3d : q3 = (pos.x pos.y pos.z size.x) q4 = (size.y size.z - -)
//   mesh   : aux0 = BLAS root reference, aux1 = first triangle index
//   medium : inner shape parameters in q3/q4 as above, q4.w = density, material = the Isotropic
//            material, aux0/aux1 as for a mesh when the inner shape is one

// ---- BVH as walked on the device: PAIR NODES, 4 x float4 = 64 B, each holding the boxes of BOTH its children:
//   (L.min.xyz, ref L) (L.max.xyz, ref R) (R.min.xyz, -) (R.max.xyz, -)
//   a reference is the index of another pair node, or REF_LEAF | item (one object / one triangle per leaf), REF_DONE ends
//   a walk.  One fetch decides two boxes (a node that carries only its own box makes every test wait for its own
//   dependent fetch), and one item per leaf keeps the leaf phases uniform.  A tree of a single item has no node: its root
//   reference is the leaf itself.
// ---- BVH as built on the host (fw_runtime.cpp, FlatBvh: the reference's topology and the SAH tree before conversion):
//   2 x float4 per node, DFS order (left child of node i is i+1):  lo = (min.xyz, bits A)   hi = (max.xyz, bits B)
//   A>>30 == 0: Branch      right child = A & 0x3fffffff, B = split axis
//   A>>30 == 1: Leaf        item = A & 0x3fffffff
//   A>>30 == 2: DoubleLeaf  items = A & 0x3fffffff, B
constexpr uint32_t REF_LEAF = 0x80000000u, REF_DONE = 0xffffffffu;
// ---- BVH as walked by the LDS-resident kernels since round 4: WIDE NODES, four children each (fw_runtime.cpp: wide_convert,
//   collapsed from the same binary tree as the pair nodes).  References are 16 bits: bit 15 = leaf (item in the low 15 bits),
//   otherwise a wide node's index; W_DONE ends a walk.  A free slot repeats child 0's reference behind a box no finite ray hits.
//   WIDE_F32, 28 dwords (112 B), SoA by plane so that a lane fetches the near / far plane of all four children with one
//   ds_read_b128 at an offset its ray's signs select (no per-box selects):
//       lo.x[4] lo.y[4] lo.z[4] hi.x[4] hi.y[4] hi.z[4] | ref0|ref1<<16  ref2|ref3<<16  -  -
//   WIDE_Q8, 12 dwords (48 B): planes quantised to 8 bits, plane = fma(q, 2^(e-127), origin) per axis, rounded outward on the host
//   with that very expression (a superset of the exact box under the same slab arithmetic: such boxes only cull):
//       origin.xyz  e.x|e.y<<8|e.z<<16 | qlo.x[4] qlo.y[4] qlo.z[4]  ref0|ref1<<16 | qhi.x[4] qhi.y[4] qhi.z[4]  ref2|ref3<<16
constexpr int WIDE_NONE = 0, WIDE_F32 = 1, WIDE_Q8 = 2;
constexpr uint32_t WIDE_F32_DW = 28, WIDE_Q8_DW = 12;
constexpr uint32_t W_LEAF = 0x8000u, W_DONE = 0xffffu;
constexpr uint32_t MF_NEEDS_UV = 1u << 8;    // the material's texture tree contains an ImageTexture
constexpr uint32_t MF_TEX_CONST = 1u << 9;   // texture is a ConstantTexture, colour inlined in the record
constexpr uint32_t NODE_LEAF = 1u, NODE_DOUBLE = 2u, NODE_MASK = 0x3fffffffu;

// ---- triangle: 3 x float4 = 48 B: (p0.xyz u0) (p1.xyz u1) (p2.xyz u2), vertices pre-gathered through
//      the index buffer.  Meshes with vertex normals and/or uvs add an attribute array of 3 x float4:
//      (n0.xyz v0) (n1.xyz v1) (n2.xyz v2).  Without uvs the defaults (0,0),(1,0),(0,1) of mesh.rs:99-109
//      are u = (0,1,0) in the .w lanes and v = (0,0,1) as constants.
// ---- material: 2 x float4: (bits kind | MF_*, bits texture, roughness, ref_idx) (colour.rgb, -)
//      colour = Metal albedo, or the ConstantTexture colour when MF_TEX_CONST (no texture fetch needed)
// ---- texture : 2 x float4: (bits kind, scale, bits depth, bits odd) then by kind:
//        constant (color.rgb -) | checker (bits even - - -) | image (bits byte offset, bits w, bits h, -)

struct DEnv {
    int32_t kind;
    float color[3];
    float zenith[3];
    float horizon[3];
    const float *hdr;      // equirect map in HBM, 16 bytes per texel (r g b -)
    uint32_t hdr_w, hdr_h;
};

struct DScene {
    const float4 *obj;
    const float4 *tlas;
    const float4 *blas;
    const float4 *tri;
    const float4 *tri_nrm;   // same indexing as tri; valid only for meshes with OF_MESH_ATTR
    const float4 *tri_gate;   // per triangle (same indexing as tri): box (2 x float4) of its leaf node in the REFERENCE tree of its mesh — the box
                              // the reference's walk must pass to test the triangle (tri_gate_ok)
    const uint32_t *tri_rank; // in-order rank of each triangle in the REFERENCE tree of its mesh (tie-breaking)
    const uint32_t *obj_rank; // same for objects in the reference TLAS
    const float4 *obj_gate;   // per object: box (2 x float4) of its leaf node in the reference TLAS (used with OF_GATE)
    const float4 *obj_cull;   // per object: enclosing world box (2 x float4) of the object itself: the pre-tests of the linear scan
    const float4 *obj_leaf;   // per object: its box in the WALKED trees = its own world box (a gated object: its reference leaf-node box united
                              // with bounds that enclose it): what k_extend_scan / hoisted_hits test before the object
    const float4 *mat;
    const float4 *tex;
    const uint8_t *images;
    uint32_t n_objects;
    uint32_t tlas_root;       // root reference of the TLAS (a pair node, or REF_LEAF | object for a one-object scene)
    uint32_t has_medium;
    uint32_t has_perlin;          // some texture is PerlinNoise / Turbulence / Marble: k_shade stages the permutation table in LDS
    uint32_t has_mesh;
    uint32_t prim_bits;       // hit code = object << prim_bits | primitive
    const float4 *ref_tlas;   // the REFERENCE trees (host FlatBvh format below: 2 x float4 per node, DFS order), walked only by
    const float4 *ref_blas;   // k_extend_exact; ref_blas holds every mesh's tree, child indices relative to the mesh's first node
    const uint32_t *obj_ref_blas;   // per object: first node of its mesh's reference tree in ref_blas (meshes and media around meshes)
    const uint32_t *wblas;    // WIDE nodes of every mesh (one array, node indices global to it) and of the TLAS; null when the scene
    const uint32_t *wtlas;    // has no tree the LDS-resident wide walks take (launch_extend then keeps the pair-node kernels)
    const uint32_t *obj_wroot;    // per object: root reference of its mesh's wide tree (16-bit convention)
    uint32_t wtlas_root;
    float soft_shear;         // the SOFT class (fw_kernels.hip: soft_direction): |d_kz| < soft_shear * max|d| walks without culling; 0: off
    uint32_t n_hoisted;       // objects kept OUT of the walked TLAS because nearly every ray meets their box (part2's fog sphere
    uint32_t hoisted[4];      // around the whole scene): tested for every ray, wave-uniformly, before the walk (hoisted_hits)
    DEnv env;
};

// ---- the exact walk (k_extend_exact).  A ray whose result depends on HOW the trees are walked — not on where it goes — is
// flagged when it is made (k_raygen / k_shade): its queue slot goes on the list of its segment, and after the segment's ordinary
// k_extend one more kernel traces the listed rays with the literal algorithm of bvh.rs:115-151 over
// the reference's own trees (median split, node boxes, DoubleLeaf = both items behind one box, both children always, no
// culling) and writes its hit record over the ordinary one.  Two classes, both found by tools/diverge.py:
//   mode & 1  ill-conditioned triangle shears (scenes with meshes): mesh.rs:147-162 permutes the axes by the SIGNED largest
//             direction component (util.rs:104-118), so a ray like (-0.88, 0.00016, -0.41) divides by 0.00016: shear factors
//             of ~5000, edge functions that are rounding noise, "hits" whose t lies outside the triangle's box.  Which of
//             those a walk finds depends on the boxes it passes and on culling.  Flagged: |d_kz| < 2^-10 max|d_i| in the world
//             frame or in the frame of a rotated mesh (up to four distinct rotations).
//   mode & 2  far origins (use_bvh): from 1000 units away a sphere of radius 0.1 has a discriminant that is the difference of two
//             numbers near 2*10^6 — noise — and the reference tests it whenever the ray passes the box of its DoubleLeaf node,
//             a front-to-back walk when it passes the sphere's own box.  Flagged: origin farther than far_r from the cluster
//             of the scene's small objects AND the ray passes that cluster's (inflated) box.
struct DExact {
    uint32_t mode;
    uint32_t n_frames;            // distinct rotations of mesh objects (<= 4), rows of rotation_mat
    float frames[4][9];
    float shear;                  // mode & 1: |d_kz| < shear * max|d_i| is ill-conditioned (2^-10)
    float far_c[3], far_r;        // centre of the small-object cluster, distance beyond which an origin is "far"
    float box_lo[3], box_hi[3];   // the cluster's box, inflated
    uint32_t *slots[2];           // the flagged rays' queue slots: list of segment s in slots[s & 1] (filled by k_raygen / k_shade(s-1), read by k_extend_exact(s))
    uint32_t *count;              // [MAX_SEGMENTS + 1] entries per list, zeroed per batch; bumped once per wave and chunk that flags anything
    uint32_t cap;                 // = the batch's paths
};

struct DCamera {   // camera.rs:7-16
    float position[3], horizontal[3], vertical[3], lower_left[3], u[3], v[3];
    float lens_radius;
};

struct DFrame {
    uint32_t width, height;
    uint32_t n_pixels;            // pixels this call renders
    const uint32_t *pixel_ids;    // device copy of fw_render_params.pixel_ids (or the library's own tile order, below), or nullptr
    uint32_t scatter_out;         // pixel_ids is the library's own 16x16-tile order of a whole frame: k_resolve writes pixel
                                  // pixel_ids[p] at output index pixel_ids[p] (the caller sees index order)
    uint32_t seed32;
    uint32_t sample0;             // first sample index of this batch
    uint32_t spp_batch;           // samples per pixel in this batch
    float inv_width;              // 1.0f / width
    float inv_n_pixels;           // 1.0f / n_pixels (path_id -> sample index without an integer division)
    uint32_t q_n_waves, q_shift;  // the batch's queue geometry (DQueue n_waves, cpw_shift): home slot <-> linear path id
    uint32_t skip_zero_deposits;  // black environment: k_shade writes only non-zero radiance records and sets their bit in dep_bits
    uint32_t dep_pixel_major;     // dep_bits is indexed p_local * spp_batch + s_local (small pixel sets) instead of by home slot
    uint32_t *dep_bits;           // one bit per path slot (zeroed per batch): k_accumulate reads a record only where it is set
    uint32_t hit4;                // 4-byte hit records (the code only): k_shade recomputes t.  Linear scan, scenes of spheres / rects / Rect3d only
    uint32_t pinhole0;            // aperture 0: every camera ray starts at cam_pos, segment-0 rays are stored as 16 B (direction only)
    float cam_pos[3];
    uint32_t chain_bits;          // != 0: 8-byte path state (fw_kernels.hip: load_state_chain): bits per material id in the chain; every material's
                                  // attenuation is a constant of the material and 10 ids fit 32 bits (host: fw_scene.chain_bits)
    float4 *atten;                // option EXACT_PRODUCT (scenes without a chain state, i.e. with a varying texture): the attenuation of every scattering, [segment][home
    uint32_t atten_stride;        // slot] (stride = slots per segment); a path that ends in light multiplies them back to front like the reference's recursion.  nullptr: off
    DExact ex;                    // which rays are traced a second time by the literal reference walk
};

// ---- wavefront path state, SoA over path slots (DESIGN.md §"Path state")
struct DPaths {
    float4 *ray_a;   // (o.x o.y o.z d.x)
    float2 *ray_b;   // (d.y d.z)
    float4 *state;   // (beta.r beta.g beta.b, bits home slot = the path's slot in the segment-0 queue)
};

// wave-private queues: wave w owns slots [w*cap, (w+1)*cap) of every path array;
// wcount[segment * n_waves + w] = live paths of wave w entering that segment (segment 0..MAX_SEGMENTS)
struct DQueue {
    uint32_t n_waves, cap;      // cap = 64 << cpw_shift slots per wave
    uint32_t cpw_shift;         // log2(chunks of 64 per wave)
    uint32_t *wcount;
};

// Rays that reached a mesh leaf of the TLAS, parked by k_extend_tlas_park for k_blas: wave w appends to its own
// [w*cap, (w+1)*cap) like every queue; pcount[w] = how many.  40 B per parked ray: the world ray + (slot, mesh object,
// best t so far, best hit code so far).
struct DPark {
    float4 *ray_a; float2 *ray_b; float4 *meta;
    uint32_t stride;    // slots per wave region: q.cap + 64 (a wave that streams several queues parks the rays still in flight
                        // from the previous queue into the region of the one it has open)
    uint32_t *pcount;
    uint32_t *ptotal;   // per wave: parked rays over all segments of the batch (statistics; zeroed per batch)
};

// Bytes per record of the streams above: what fw_stats.bytes_* (the layout's own algorithmic HBM bytes) are computed from,
// kept next to the layout so that the two change together.
constexpr uint32_t B_RAY = 24, B_RAY_PINHOLE0 = 16, B_STATE = 16, B_STATE_CHAIN = 8, B_HIT = 8, B_HIT4 = 4, B_DEPOSIT = 16, B_PARK = 40, B_ACCUM = 16, B_ATTEN = 16;

constexpr uint32_t MISS = 0xffffffffu;
constexpr int MAX_SEGMENTS = 11;
constexpr int COUNT_STRIDE = 16;  // u32 counters per batch: [0..10] queue sizes, [11] parked rays, [12] deposits written (FW_FLAG_COUNT_DEPOSITS)

// launch wrappers (fw_kernels.hip)
struct LaunchCfg {
    hipStream_t stream;
    DQueue q;
    int blocks_other;
    int tlas_depth, blas_depth;   // LDS traversal-stack levels needed
    uint32_t n_mat, n_tex;        // table sizes (for the LDS-resident copy in k_shade)
    bool lds_tables;              // stage small scene tables in LDS (debug switch: FIREWORK_NO_LDS_TABLES)
    bool has_mesh;
    bool simple_but_meshes;   // ... where meshes do not count (k_extend_scan parks them): suzanne, teapot
    bool simple_set;      // every object is a sphere, an axis-aligned rectangle, a Rect3d, or a medium around a sphere: kernels with hit_shape's SIMPLE form (round 5)
    int n_cus;
    uint32_t blas_pair_nodes, tlas_pair_nodes, max_tris, n_tris;   // n_tris: all meshes together
    bool no_lds_tris;     // A/B switch FIREWORK_NO_LDS_TRIS   // sizes of the walked trees (pair nodes) and of the biggest mesh
    uint32_t n_defer;     // linear scan: the scene's last n_defer (1 or 2) objects are plain Rect3d boxes handled by k_extend_linear_defer; 0: k_extend_linear
    bool lds_trees;       // walk trees that fit out of LDS (k_blas_lds); FIREWORK_NO_LDS_TREES=1 switches it off
    int shade_mode;       // k_shade: 0 everything in line, 1 the scene has no expensive material / environment (the cheap loop alone), 2 expensive paths go through a list (FIREWORK_SHADE_LIST=1: measured slower, kept for A/B)
    int exact_form;               // k_extend_exact: 0 = by list length, 1 = one ray per lane, 2 = one ray per wave (FIREWORK_EXACT_FORM=lane|wave: tests)
    uint32_t ref_tlas_nodes, ref_blas_nodes, ref_tlas_depth, ref_blas_depth;   // the reference trees k_extend_exact walks (nodes of 32 B)
    int wblas_fmt, wtlas_fmt;     // WIDE_NONE / WIDE_F32 / WIDE_Q8: the encoding of DScene.wblas / wtlas (FIREWORK_WIDE=0: none; =q8 / =f32 force one)
    uint32_t wblas_nodes, wtlas_nodes, wblas_depth, wtlas_depth;   // wide nodes; wide nodes on the longest root-to-leaf path
    uint32_t debug_wide_levels;   // A/B build, option DEBUG_WIDE_LEVELS: LDS stack levels of the wide walks instead of 3 * depth + 2 (the error word's test); 0: off
    bool tlas_refill;     // refilling walks: k_extend_tlas (no meshes) / k_extend_tlas_park + k_blas (meshes); FIREWORK_TLAS_REFILL=0: the chunked k_extend_bvh
};
constexpr size_t LDS_TREE_LIMIT = 160 * 1024;   // the whole LDS of a CU: one workgroup of the LDS-resident walks per CU
constexpr size_t LDS_TABLE_LIMIT = 16 * 1024;   // object+material+texture tables up to this size are staged in LDS

void launch_raygen(const LaunchCfg &, const DCamera &, const DFrame &, DPaths out, float4 *sample_rad, uint32_t n_paths);
void launch_extend(const LaunchCfg &, const DScene &, const DFrame &, DPaths in, float2 *hits, int segment, bool use_bvh, DPark park);
void launch_extend_exact(const LaunchCfg &, const DScene &, const DFrame &, const DPaths &in, float2 *hits, int segment, bool use_bvh);
void launch_shade(const LaunchCfg &, const DScene &, const DFrame &, DPaths in, DPaths out, const float2 *hits,
                  float4 *sample_rad, int segment);
void launch_bounce(const LaunchCfg &, const DScene &, const DFrame &, DPaths in, DPaths out, float4 *sample_rad, int segment,
                   bool use_bvh);
void launch_queue_totals(const LaunchCfg &, uint32_t *totals, const uint32_t *ptotal);
void launch_scatter_tiles(hipStream_t stream, const uint32_t *ids, uint32_t n, const uint8_t *in8, const float *ing, const float *inl,
                          uint8_t *out8, float *outg, float *outl);
void launch_tile_order(hipStream_t stream, uint32_t width, uint32_t height, uint32_t *ids);
void launch_upload(hipStream_t stream, const void *pinned_src, void *dst, size_t bytes);
#if FW_AB
uint32_t take_error_word();   // the device's error word (fw_kernels.hip: g_err_word), read and cleared
#endif
void preload_kernels();    // resolves every kernel of the default paths on the current device (a first launch pays ~2 ms for it otherwise)
void launch_count_deposits(const LaunchCfg &, const uint32_t *dep_bits, uint32_t *total);
void launch_selftest_arith(hipStream_t stream, uint32_t n, uint32_t seed, int mode, unsigned long long *out);
void launch_selftest_libm(hipStream_t stream, int fn, uint32_t n, const float *x, const float *y, float *out);
void launch_accumulate(const LaunchCfg &, const DFrame &, const float4 *sample_rad, float4 *accum);
void launch_resolve(const LaunchCfg &, const DFrame &, const float4 *accum, uint32_t total_spp, float gamma,
                    uint8_t *rgb8, float *gamma_rgb, float *linear_rgb);

constexpr int BLOCK = 256;

} // namespace fw
