/*
 * firework_hip.h — C ABI of the MI355X-native path-tracing core.
 *
 * This is the drop-in boundary for ONE path of ritobanrc/firework: the
 * `Renderer::render(&self, scene: Scene) -> Vec<Color>` call
 * (reference src/render.rs:109-161) and the data it consumes
 * (`Scene`, `RenderObject`, shapes, materials, textures, environments,
 * `CameraSettings`).  Everything is plain C: fixed-width scalars, plain
 * pointers and sizes.  No torch / C++ / HIP types appear in a signature.
 *
 * The reference keeps its scene as open sets of trait objects
 * (`Box<dyn SerializableShape>`, `Box<dyn Material>`, `Box<dyn Texture>`,
 * `Box<dyn Environment>`; src/scene.rs:19-24,270-277).  A GPU cannot call a
 * user's trait impl, so the ABI carries CLOSED tagged unions whose tags are
 * exactly the reference's typetag names (src/serde_compat.rs:22,
 * src/material.rs:9, src/texture.rs:7, src/environment.rs:5).
 *
 * Ownership: the caller owns every buffer it passes in or receives results
 * in; the library copies what it needs during the call.  No call throws or
 * aborts across this boundary: errors are negative `fw_status` codes
 * (the reference panics instead: src/bvh.rs:34, src/scene.rs:161).
 *
 * The SAME structs are the input format of the CPU oracle under oracle/
 * (test infrastructure, not part of the product).
 */
#ifndef FIREWORK_HIP_H
#define FIREWORK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FW_ABI_VERSION 7   /* 7: + fw_init (loading the library no longer touches the GPU); options EXACT_PRODUCT, PHASE_LOCK, GRAPH; 6: + fw_set_option (the runtime switches leave the environment: read once at load), fw_selftest_wide_bvh; 3: + fw_render_progressive; 4: fw_stats carries this layout's own HBM bytes per kernel class and the one-shot timings;
                              5: + fw_selftest_libm; the 4th float of an accumulation record counts the path segments of the samples that deposited */

/* ---- status codes ------------------------------------------------------ */
typedef enum fw_status {
    FW_OK = 0,
    FW_ERR_BAD_ARG = -1,       /* null pointer, zero size, index out of range      */
    FW_ERR_EMPTY_SCENE = -2,   /* reference: panic "No render objects added to scene!" (src/scene.rs:161)
                                  and unbounded recursion on an empty BVH (src/bvh.rs:29-70)          */
    FW_ERR_NAN_BBOX = -3,      /* reference: panic "Float comparison failed in BVH constructor" (src/bvh.rs:34) */
    FW_ERR_MESH_NORMALS = -4,  /* "TriangleMesh::new() -- normals.len() must equal verts.len()" (src/objects/mesh.rs:46) */
    FW_ERR_MESH_UVS = -5,      /* "TriangleMesh::new() -- uvs.len() must equal verts.len()"     (src/objects/mesh.rs:51) */
    FW_ERR_UNSUPPORTED = -6,   /* valid scene, but a feature the GPU path does not implement yet */
    FW_ERR_HIP = -7,           /* a HIP runtime call failed; see fw_last_error()                  */
    FW_ERR_NO_DEVICE = -8,     /* no gfx950 device visible: the product path never falls back to CPU */
    FW_ERR_BVH_DEPTH = -9,     /* tree deeper than the LDS traversal stack                        */
    FW_ERR_OOM = -10
} fw_status;

/* ---- small value types ------------------------------------------------- */
typedef struct fw_vec3 { float x, y, z; } fw_vec3;

/* ultraviolet::Rotor3 as serialised by src/serde_compat.rs:6-20: {s, bv:{xy,xz,yz}} */
typedef struct fw_rotor3 { float s, xy, xz, yz; } fw_rotor3;

/* ---- textures: src/texture.rs ------------------------------------------ */
typedef enum fw_texture_kind {
    FW_TEX_CONSTANT = 0,    /* ConstantTexture   texture.rs:16-34  */
    FW_TEX_CHECKER = 1,     /* CheckerTexture    texture.rs:36-73  */
    FW_TEX_PERLIN = 2,      /* PerlinNoiseTexture texture.rs:75-168 */
    FW_TEX_TURBULENCE = 3,  /* TurbulenceTexture texture.rs:195-225 */
    FW_TEX_MARBLE = 4,      /* MarbleTexture     texture.rs:227-249 */
    FW_TEX_IMAGE = 5        /* ImageTexture      texture.rs:270-310 */
} fw_texture_kind;

typedef struct fw_texture {
    int32_t kind;
    fw_vec3 color;           /* Constant */
    float scale;             /* Checker / Perlin / Turbulence / Marble */
    uint32_t depth;          /* Turbulence / Marble */
    int32_t odd, even;       /* Checker: indices into fw_scene_desc.textures */
    uint32_t img_w, img_h;   /* Image: RGB8, row 0 = top, tightly packed */
    const uint8_t *img_rgb8;
} fw_texture;

/* ---- materials: src/material.rs ---------------------------------------- */
typedef enum fw_material_kind {
    FW_MAT_LAMBERTIAN = 0,  /* material.rs:25-75   {albedo: texture}   */
    FW_MAT_METAL = 1,       /* material.rs:77-107  {albedo, roughness} */
    FW_MAT_DIELECTRIC = 2,  /* material.rs:109-151 {ref_idx}           */
    FW_MAT_EMISSIVE = 3,    /* material.rs:153-181 {albedo: texture}   */
    FW_MAT_ISOTROPIC = 4    /* material.rs:183-204 {texture}           */
} fw_material_kind;

typedef struct fw_material {
    int32_t kind;
    int32_t texture;   /* Lambertian / Emissive / Isotropic: index into textures */
    fw_vec3 albedo;    /* Metal */
    float roughness;   /* Metal */
    float ref_idx;     /* Dielectric */
} fw_material;

/* ---- shapes: src/objects/ ---------------------------------------------- */
typedef enum fw_shape_kind {
    FW_SHAPE_SPHERE = 0,          /* objects/sphere.rs:10-20  {radius, material}            */
    FW_SHAPE_XYRECT = 1,          /* objects/rect.rs:9-45     AARect<X,Y>                   */
    FW_SHAPE_XZRECT = 2,          /*                          AARect<X,Z>                   */
    FW_SHAPE_YZRECT = 3,          /*                          AARect<Y,Z>                   */
    FW_SHAPE_RECT3D = 4,          /* objects/rect3d.rs:9-86   {pos, size} + 6 derived faces */
    FW_SHAPE_TRIANGLE_MESH = 5,   /* objects/mesh.rs:12-63                                  */
    FW_SHAPE_CONSTANT_MEDIUM = 6, /* objects/volume.rs:10-54  {obj, density, material}      */
    FW_SHAPE_CONE = 7,            /* objects/cone.rs:9-25     {radius, height, material}    */
    FW_SHAPE_CYLINDER = 8,        /* objects/cylinder.rs:10-38 {radius, height, max_phi, material} */
    FW_SHAPE_DISK = 9             /* objects/disk.rs:10-37    {radius, phi_max, inner_radius, material} */
} fw_shape_kind;

typedef struct fw_shape {
    int32_t kind;
    int32_t material;          /* MaterialIdx (src/scene.rs:13) */
    float radius;              /* Sphere, Cone, Cylinder, Disk */
    float height;              /* Cone, Cylinder */
    float phi_max;             /* Cylinder.max_phi / Disk.phi_max, radians */
    float inner_radius;        /* Disk */
    /* AARect<A1,A2>: min=(a_min,b_min) max=(a_max,b_max) on axes (A1,A2), plane at k
       on the third axis (rect.rs:14-20).  XY: a=x b=y; XZ: a=x b=z; YZ: a=y b=z. */
    float a_min, a_max, b_min, b_max, k;
    int32_t flip_normal;       /* AARect.flip_normal (rect.rs:18) */
    fw_vec3 pos, size;         /* Rect3d (rect3d.rs:10-14); faces are derived as in Rect3d::new */
    /* TriangleMesh (mesh.rs:12-18): verts = 3*n_verts floats, indices = n_indices (multiple of 3),
       normals = NULL or 3*n_verts floats, uvs = NULL or 2*n_verts floats */
    const float *verts;
    uint32_t n_verts;
    const uint32_t *indices;
    uint32_t n_indices;
    const float *normals;
    const float *uvs;
    /* ConstantMedium (volume.rs:11-15): inner = index into fw_scene_desc.shapes */
    int32_t inner;
    float density;
} fw_shape;

/* RenderObject (src/scene.rs:270-277): shape + position + rotation + flip_normals */
typedef struct fw_object {
    int32_t shape;             /* index into fw_scene_desc.shapes */
    fw_vec3 position;
    fw_rotor3 rotation;
    int32_t flip_normals;
} fw_object;

/* ---- environments: src/environment.rs (+ examples/hdri_test.rs:22-82) --- */
typedef enum fw_env_kind {
    FW_ENV_COLOR = 0,  /* ColorEnv environment.rs:10-26 (Scene::new default: black, scene.rs:36) */
    FW_ENV_SKY = 1,    /* SkyEnv   environment.rs:28-67                                          */
    FW_ENV_HDR = 2     /* HdrEnvironment: user-side plugin in examples/hdri_test.rs:22-82, promoted
                          to a built-in because a user trait impl cannot cross to the GPU        */
} fw_env_kind;

typedef struct fw_environment {
    int32_t kind;
    fw_vec3 color;             /* ColorEnv */
    fw_vec3 zenith, horizon;   /* SkyEnv   */
    uint32_t hdr_w, hdr_h;     /* HdrEnv: equirect, f32 RGB, row 0 = top */
    const float *hdr_rgb;
} fw_environment;

/* Scene (src/scene.rs:19-24) */
typedef struct fw_scene_desc {
    const fw_object *objects;
    uint32_t n_objects;
    const fw_shape *shapes;
    uint32_t n_shapes;
    const fw_material *materials;
    uint32_t n_materials;
    const fw_texture *textures;
    uint32_t n_textures;
    fw_environment environment;
} fw_scene_desc;

/* CameraSettings (src/camera.rs:18-36); defaults there: (0,0,-10) -> 0, vfov 30, aperture 0, focus 10 */
typedef struct fw_camera_settings {
    fw_vec3 cam_pos, look_at;
    float vfov, aperture, focus_dist;
} fw_camera_settings;

typedef enum fw_rng_mode {
    FW_RNG_CTR = 0,  /* counter-based, keyed (seed,pixel,sample,dimension): the GPU's RNG (spec: DESIGN.md §RNG) */
    FW_RNG_LCG = 1   /* sequential per-pixel LCG seeded with the pixel index (render.rs:172); CPU oracle only    */
} fw_rng_mode;

/* Renderer (src/render.rs:59-77) + what the GPU path needs on top */
typedef struct fw_render_params {
    uint32_t width, height, samples;   /* Default: 1920, 1080, 128 (render.rs:207-209) */
    float gamma;                       /* Default 2.2 (render.rs:214) */
    int32_t use_bvh;                   /* Default false (render.rs:213) */
    int32_t multithreaded;             /* reference: rayon on/off; ignored by the HIP path */
    fw_camera_settings camera;
    uint64_t seed;                     /* CTR mode key; 0 by default */
    int32_t rng_mode;                  /* fw_rng_mode; the HIP path accepts FW_RNG_CTR only */
    /* Pixel subset for framebuffer tiling across GPUs: linear pixel indices
       (idx as in render.rs:127) this call renders, in output order.
       NULL => all width*height pixels in index order. */
    const uint32_t *pixel_ids;
    uint32_t n_pixels;
    uint32_t paths_per_batch;          /* wavefront pool size; 0 = library default */
    uint32_t flags;                    /* FW_FLAG_* */
    int32_t outputs_on_device;         /* !=0: the three output pointers are device pointers */
    void *stream;                      /* hipStream_t to launch on, NULL = default stream    */
} fw_render_params;

#define FW_FLAG_TIME_KERNELS 1u  /* bracket every launch with HIP events and fill fw_stats.ms_<class> */
#define FW_FLAG_COUNT_DEPOSITS 2u /* count the radiance records k_shade really wrote (one extra pass over the sample buffer per
                                     batch, outside the kernel classes' times): makes fw_stats.bytes_shade exact when zero
                                     deposits are elided over a black environment; without it they are counted as written */

#define FW_MAX_SEGMENTS 11  /* depths 0..10: render.rs:21 */

typedef struct fw_stats {
    uint64_t samples;                        /* camera samples = pixels * spp (render.rs:177)      */
    uint64_t rays;                           /* root.hit() calls = path segments (render.rs:19)    */
    uint64_t rays_per_depth[FW_MAX_SEGMENTS];
    uint64_t algorithmic_bytes;              /* 160*rays + 24*samples (+12*env misses for HDR), SURVEY §8(d) */
    double ms_scene;                         /* host: flatten + BVH build + LAUNCH of the upload (one-shot call only); the upload kernel itself is asynchronous and the render that follows waits for it, so its time is part of ms_render */
    double ms_render;                        /* device: first launch to last, HIP events on the launch stream */
    double ms_raygen, ms_extend, ms_shade, ms_accumulate; /* per-kernel-class device time (HIP events) */
    uint32_t n_extend_launches, n_shade_launches, n_batches;  /* FIREWORK_FUSED=1: no extend launches, the fused
                                                                 intersect+shade launches are counted and timed as shade */
    uint32_t tlas_nodes, blas_nodes;
    uint32_t reserved;                       /* bits 0-15 / 16-30: depth of the BLAS / TLAS walked; bit 31: this frame's launches were replayed as one hipGraph (option GRAPH) */
    /* HBM bytes THIS layout has to move, per kernel class, exact from the queue counters (DESIGN.md §5 gives the per-ray
       figures; SURVEY's generic 160 B/ray formula stays in algorithmic_bytes): what roofline fractions are computed from. */
    uint64_t bytes_raygen, bytes_extend, bytes_shade, bytes_accumulate;
    uint64_t deposits;                       /* radiance records written by k_shade (FW_FLAG_COUNT_DEPOSITS), else the terminated paths */
    uint64_t parked_rays;                    /* rays handed from the TLAS walk to k_blas (use_bvh with meshes) */
    double ms_wall;                          /* host wall time of the whole call (fw_render_scene: conversion + BVH + upload + render + D2H) */
    double ms_d2h;                           /* device -> host copies of the outputs (0 when outputs_on_device) */
} fw_stats;

typedef struct fw_scene fw_scene;  /* opaque: flattened SoA scene + BVHs resident in HBM */

/* ---- entry points -------------------------------------------------------- */
int fw_abi_version(void);
const char *fw_strerror(int status);
const char *fw_last_error(void);            /* thread-local detail for the last failing call */
int fw_device_count(void);                  /* number of visible HIP devices (0 if none) */

/* Initialisation of one device, explicit and idempotent (ABI v7): the HIP context, this library's code objects and kernel
   handles, its streams and pinned staging, and a path arena of `arena_bytes` (0 = the default: a third of the free HBM, at most
   64 GiB — every default-budget frame of the BASELINE configs fits; FW_INIT_NO_ARENA = none, the first render sizes its own).
   Loading the library makes NO HIP call and holds no memory; a host that never calls fw_init gets the same initialisation,
   without an arena, from its first fw_scene_create / fw_render* on the device (SURVEY §8(b) "Ownership": a lazily created
   per-device context is the only global state).  What it is for: the reference's timed region (main.rs:40-44) starts with
   the process already loaded; a host that wants that region free of one-off costs (0.2-1.1 s for context, code objects
   and, where the driver has pages to clear, the arena) calls fw_init first — `python -m firework_amd` and bench.py do.
   Calling it again with a larger arena_bytes grows the arena; a smaller one changes nothing. */
#define FW_INIT_NO_ARENA UINT64_MAX
int fw_init(int device, uint64_t arena_bytes);

/* `Scene -> SceneInternal` (scene.rs:111-135) + `build_bvh` (bvh.rs:79-85, mesh.rs:21-30):
   flatten, build TLAS/BLAS with the reference's median split, upload to `device`. */
int fw_scene_create(const fw_scene_desc *desc, int device, fw_scene **out);
void fw_scene_destroy(fw_scene *scene);

/* The hot path: render.rs:123-161 on an uploaded scene.
   Any output pointer may be NULL.  Sizes are N*3 with N = n_pixels (or width*height),
   index order = pixel_ids order, row 0 = image top (util.rs:31-33):
     rgb8       : Color quantisation `(c*255.99) as u8`   (util.rs:14-23)
     gamma_rgb  : post-gamma, clamped floats in [0,1]      (render.rs:185-187) — the parity metric's input
     linear_rgb : pre-gamma per-pixel sample mean          (render.rs:184) */
int fw_render(fw_scene *scene, const fw_render_params *params,
              uint8_t *rgb8, float *gamma_rgb, float *linear_rgb, fw_stats *stats);

/* One-shot render of a whole frame on SEVERAL GPUs from ONE process (what a single Rust binary calls; the Python hosts of
   this repo use one process per GPU and an RCCL gather instead): the frame is cut into 16x16 tiles dealt diagonally over the
   devices, one host thread per device creates the scene there and renders its pixels — with the keys a single GPU would
   use, so the image is bit-identical for any device list — copies its finished tiles peer-to-peer (hipMemcpyPeer: xGMI where
   the devices are linked) to the first listed device, which scatters them to their pixels and sends the frames to the
   caller's host buffers in one transfer.
   `devices` may name a device more than once (its calls are serialised).  render.rs:127-131 shards pixels over rayon workers
   the same way.  stats: counters summed, times = the slowest device. */
int fw_render_scene_tiled(const fw_scene_desc *desc, const fw_render_params *params, const int *devices, int n_devices,
                          uint8_t *rgb8, float *gamma_rgb, float *linear_rgb, fw_stats *stats);

/* Progressive / resumable rendering (progressive preview, checkpointable accumulation buffer).
   Renders the samples [first_sample, first_sample + params->samples) of every pixel, adds them to `accum`
   (n_pixels x 4 floats: r, g, b sums and the number of path segments of the samples that deposited a record — every sample
   unless the environment is black, where zero deposits are elided; all zeros before the first call; host memory, or device memory when
   params->outputs_on_device) and resolves accum / (first_sample + samples) into the output buffers (any may be NULL).
   Every random draw is keyed by (pixel, ABSOLUTE sample index) and a pixel's sums are taken in sample order, so k calls
   of n samples leave bit for bit the accum and the image of one call of k*n samples — whatever is done with `accum`
   between the calls: show a preview, write it to disk, resume in another process.
   Reference: render.rs:172-190 sums `samples` colours per pixel and divides once; there is no progressive mode there. */
int fw_render_progressive(fw_scene *scene, const fw_render_params *params, uint32_t first_sample, float *accum,
                          uint8_t *rgb8, float *gamma_rgb, float *linear_rgb, fw_stats *stats);

/* One-shot form with the reference's exact shape: `Renderer::render(&self, scene: Scene)`
   (render.rs:109): scene conversion + BVH build + render inside one call. */
int fw_render_scene(const fw_scene_desc *desc, const fw_render_params *params, int device,
                    uint8_t *rgb8, float *gamma_rgb, float *linear_rgb, fw_stats *stats);

/* The wavefront workspace (path pools, tens of GB for big frames) is cached per device between calls — the library's
   only global state.  This frees it (e.g. before handing the GPU to another library). */
void fw_release_workspace(int device);

/* Diagnostic: the kernels' division / square-root helpers against the compiler's IEEE expansion, bit for bit,
   on n hashed operand pairs.  mode 0 = magnitudes 2^-40..2^40 (must be 0 mismatches), mode 1 = all bit patterns. */
int fw_selftest_arith(int device, uint32_t n, uint32_t seed, int mode, uint64_t *div_mismatches, uint64_t *sqrt_mismatches);

/* Diagnostic: the libm-class functions of the path (firework_amd/csrc/fw_libm.h: glibc's logf, log10f, sinf, asinf, acosf,
   atanf, atan2f, powf restated for the device) evaluated on the device, element-wise, on host arrays: out[i] = fn(x[i] [, y[i]]).
   fn: 0 logf  1 log10f  2 sinf  3 asinf  4 acosf  5 atanf  6 atan2f(x[i], y[i]) = atan2(first, second)  7 powf(x[i], y[i]).
   y may be NULL for the one-argument functions.  tests/test_gpu_libm.py compares the results with the host's libm bit for bit. */
int fw_selftest_libm(int device, int fn, uint32_t n, const float *x, const float *y, float *out);

/* Runtime options.  The library reads the environment variables FIREWORK_<NAME> ONCE, when it is loaded; nothing on the render path
   looks at the environment.  fw_set_option changes one option afterwards (name with or without the FIREWORK_ prefix; value NULL =
   back to the default; name NULL = back to what the environment said at load time) and applies to the scenes created and the renders
   started after it returns.  Results never depend on an option — only which kernels produce them (every pair of settings is
   compared bit for bit in tests/) — except the diagnostics NO_EXACT / EXACT_ALL.  Names:
     BVH=median            walk the reference's own median-split topology instead of the SAH tree (parity / A-B mode)
     WIDE=0|f32|q8         no wide nodes (the pair-node kernels) / force an encoding of the wide nodes
     EXACT_ALL=1, NO_EXACT, EXACT_FORM=lane|wave      every ray / no ray through the literal reference walk; its form
     STREAMS=n, WAVES=n, PATHS_PER_BATCH=n            batches in flight, wave queues, pool size
     NO_DEFER NO_HIT4 NO_HOIST NO_LDS_TABLES NO_LDS_TREES NO_LDS_TRIS NO_SHORT_RAYS NO_TILE_ORDER NO_ZERO_SKIP
     DEP_PIXEL_MAJOR DEP_SLOT_MAJOR NO_CHAIN          the layout choices the tests force both ways
     EXACT_PRODUCT=1       scenes with a varying texture keep every scattering's attenuation (16 B per segment) and multiply back to front when
                           a path deposits — render.rs:23-28's own association, the pre-gamma means then equal the CPU oracle's bit for bit
                           (scenes of constant textures always do: their 8-byte chain state).  Default off: the running product, ~1 ulp away
     PHASE_LOCK=0|1        the two batches in flight tied in anti-phase by one event per segment (default: big box-list batches)
     GRAPH=0|1             a frame asked for twice in a row is captured into a hipGraph and replayed from then on (default: frames of
                           small batches, whose launches are short); fw_stats.reserved bit 31 reports a replay
     TRACE, DUMP_PATH=file                            host-side timing trace; one path's records (tools/diverge.py)
   Returns FW_ERR_BAD_ARG for a name this build does not know. */
int fw_set_option(const char *name, const char *value);

/* Diagnostic, CPU only: builds the WIDE nodes the LDS-resident walks step through (four children per node; format 1 = f32 planes,
   2 = planes quantised to 8 bits and rounded outward) over n item boxes (n x 6 floats: min.xyz max.xyz) and checks the finished
   tree: every item the leaf of exactly one slot, every child box as the device decodes it a superset of the exact one (f32: the
   item's own box bit for bit), free slots unhittable.  violations = 0 is the only acceptable answer; stats = nodes, leaves,
   free slots, depth. */
int fw_selftest_wide_bvh(const float *boxes, uint32_t n, int format, uint32_t *violations, uint32_t stats[4]);

/* Diagnostic, CPU only (ABI v7): the host-side tree builders over n item boxes (n x 6 floats) with `threads` host threads — the reference's
   median-split tree (bvh.rs:21-71: what fixes tie ranks and gate boxes) and the binned-SAH tree the device walks.  hashes = FNV-1a of the
   two node arrays, stats = nodes and depth of the median tree, nodes and depth of the SAH tree.  Scene creation builds a mesh's trees in
   parallel (the reference builds inside its timed region, main.rs:40-44, on one thread); any thread count must give the one-thread trees. */
int fw_selftest_bvh_build(const float *boxes, uint32_t n, int threads, uint64_t hashes[2], uint32_t stats[4]);

#ifdef __cplusplus
}
#endif
#endif /* FIREWORK_HIP_H */
