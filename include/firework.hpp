// firework.hpp — header-only C++ mirror of the reference's host API for the render path, over the C ABI of
// firework_hip.h.  The reference is compiled code (Rust); with no Rust toolchain in the image this is the
// compiled-language host side: the same builder names, argument order and defaults as
//
//   Scene / RenderObject                      reference src/scene.rs:19-91,270-334
//   Sphere, XYRect/XZRect/YZRect, Rect3d,     src/objects/*.rs
//     TriangleMesh, Cone, Cylinder, Disk
//   LambertianMat, MetalMat, DielectricMat,   src/material.rs
//     EmissiveMat, IsotropicMat
//   ConstantTexture, CheckerTexture, ...      src/texture.rs
//   ColorEnv, SkyEnv, HdrEnvironment          src/environment.rs, examples/hdri_test.rs:22-82
//   Rotor3, CameraSettings, Renderer          ultraviolet, src/camera.rs:18-71, src/render.rs:59-218
//
// so that `examples/cornell_box.rs` ports line by line (examples/cornell_box.cpp).  `Renderer::render` lowers
// the scene to fw_scene_desc and calls fw_render_scene — no CPU fallback; failures throw std::runtime_error
// carrying the reference's panic messages.
#pragma once
#include "firework_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace firework {

struct Vec3 { float x = 0, y = 0, z = 0; Vec3() = default; Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    static Vec3 broadcast(float v) { return {v, v, v}; } static Vec3 zero() { return {}; } static Vec3 one() { return {1, 1, 1}; } };
inline fw_vec3 lower(Vec3 v) { return fw_vec3{v.x, v.y, v.z}; }

struct Color { uint8_t r, g, b; };   // util.rs:6 `Color(pub u8, pub u8, pub u8)`
static_assert(sizeof(Color) == 3, "Color must be 3 packed bytes (same bytes as the ABI's rgb8 output)");

// ultraviolet::Rotor3 {s, bv:{xy,xz,yz}}; from_rotation_P(t) = {cos(t/2), unit_plane * -sin(t/2)} (pinned by scenes/*.yml)
struct Rotor3 {
    float s = 1, xy = 0, xz = 0, yz = 0;
    static Rotor3 identity() { return {}; }
    static Rotor3 plane(float angle, int which) {
        // sin/cos evaluated in double and rounded once (what firework_amd/api.py does; both reproduce the
        // rotors serialised in the reference's scenes/*.yml)
        float half = angle / 2.0f, ms = -(float)std::sin((double)half);
        Rotor3 r; r.s = (float)std::cos((double)half);
        r.xy = (which == 0 ? 1.0f : 0.0f) * ms; r.xz = (which == 1 ? 1.0f : 0.0f) * ms; r.yz = (which == 2 ? 1.0f : 0.0f) * ms;
        return r;
    }
    static Rotor3 from_rotation_xy(float a) { return plane(a, 0); }
    static Rotor3 from_rotation_xz(float a) { return plane(a, 1); }
    static Rotor3 from_rotation_yz(float a) { return plane(a, 2); }
    Rotor3 operator*(const Rotor3 &b) const {   // geometric product on (1, e12, e13, e23)
        // every sum is an fma chain (ultraviolet builds its products from mul_add): the association that reproduces the
        // rotor serialised in the reference's scenes/conics.yml bit for bit (same chain as firework_amd/api.py)
        const Rotor3 &a = *this; Rotor3 r;
        auto chain = [](float x0, float y0, float x1, float y1, float x2, float y2, float x3, float y3) {
            return std::fma(x0, y0, std::fma(x3, y3, std::fma(x2, y2, x1 * y1)));
        };
        r.s = chain(a.s, b.s, -a.xy, b.xy, -a.xz, b.xz, -a.yz, b.yz);
        r.xy = chain(a.xy, b.s, a.s, b.xy, a.yz, b.xz, -a.xz, b.yz);
        r.xz = chain(a.xz, b.s, a.s, b.xz, -a.yz, b.xy, a.xy, b.yz);
        r.yz = chain(a.yz, b.s, a.s, b.yz, a.xz, b.xy, -a.xy, b.xz);
        return r;
    }
    static Rotor3 from_euler_angles(float roll, float pitch, float yaw) { return from_rotation_xz(yaw) * from_rotation_yz(pitch) * from_rotation_xy(roll); }
};

using MaterialIdx = int32_t;
using RenderObjectIdx = size_t;

// ---- textures ---------------------------------------------------------------------------------------------
struct Texture {
    fw_texture t{}; std::shared_ptr<Texture> odd, even; std::shared_ptr<std::vector<uint8_t>> pixels;
};
using TexturePtr = std::shared_ptr<Texture>;
struct ConstantTexture {
    static TexturePtr new_(Vec3 c) { auto p = std::make_shared<Texture>(); p->t.kind = FW_TEX_CONSTANT; p->t.color = lower(c); return p; }
    static TexturePtr from_rgb(float r, float g, float b) { return new_({r, g, b}); }
};
struct CheckerTexture {
    static TexturePtr new_(TexturePtr odd, TexturePtr even, float scale) { auto p = std::make_shared<Texture>(); p->t.kind = FW_TEX_CHECKER; p->t.scale = scale; p->odd = std::move(odd); p->even = std::move(even); return p; }
    static TexturePtr with_colors(Vec3 odd, Vec3 even, float scale) { return new_(ConstantTexture::new_(odd), ConstantTexture::new_(even), scale); }
};
struct PerlinNoiseTexture { static TexturePtr new_(float scale) { auto p = std::make_shared<Texture>(); p->t.kind = FW_TEX_PERLIN; p->t.scale = scale; return p; } };
struct TurbulenceTexture { static TexturePtr new_(uint32_t depth, float scale) { auto p = std::make_shared<Texture>(); p->t.kind = FW_TEX_TURBULENCE; p->t.depth = depth; p->t.scale = scale; return p; } };
struct MarbleTexture { static TexturePtr new_(uint32_t depth, float scale) { auto p = std::make_shared<Texture>(); p->t.kind = FW_TEX_MARBLE; p->t.depth = depth; p->t.scale = scale; return p; } };
struct ImageTexture {   // RGB8, row 0 = top
    static TexturePtr new_(uint32_t w, uint32_t h, std::vector<uint8_t> rgb8) {
        if (rgb8.size() != (size_t)w * h * 3) throw std::runtime_error("ImageTexture: pixel buffer size mismatch");
        auto p = std::make_shared<Texture>(); p->t.kind = FW_TEX_IMAGE; p->t.img_w = w; p->t.img_h = h;
        p->pixels = std::make_shared<std::vector<uint8_t>>(std::move(rgb8)); p->t.img_rgb8 = p->pixels->data(); return p; }
};

// ---- materials ----------------------------------------------------------------------------------------------
struct Material { fw_material m{}; TexturePtr tex; };
struct LambertianMat { static Material new_(TexturePtr albedo) { Material r; r.m.kind = FW_MAT_LAMBERTIAN; r.tex = std::move(albedo); return r; }
                       static Material with_color(Vec3 c) { return new_(ConstantTexture::new_(c)); } };
struct MetalMat { static Material new_(Vec3 albedo, float roughness) { Material r; r.m.kind = FW_MAT_METAL; r.m.albedo = lower(albedo); r.m.roughness = roughness; return r; } };
struct DielectricMat { static Material new_(float ref_idx) { Material r; r.m.kind = FW_MAT_DIELECTRIC; r.m.ref_idx = ref_idx; return r; } };
struct EmissiveMat { static Material new_(TexturePtr albedo) { Material r; r.m.kind = FW_MAT_EMISSIVE; r.tex = std::move(albedo); return r; }
                     static Material with_color(Vec3 c) { return new_(ConstantTexture::new_(c)); } };
struct IsotropicMat { static Material new_(TexturePtr texture) { Material r; r.m.kind = FW_MAT_ISOTROPIC; r.tex = std::move(texture); return r; } };

// ---- shapes -------------------------------------------------------------------------------------------------
struct Shape {
    fw_shape s{}; std::shared_ptr<Shape> inner;
    std::shared_ptr<std::vector<float>> verts, normals, uvs; std::shared_ptr<std::vector<uint32_t>> indices;
};
inline Shape make_shape(int kind, MaterialIdx m) { Shape r; r.s.kind = kind; r.s.material = m; r.s.inner = -1; return r; }
struct Sphere { static Shape new_(float radius, MaterialIdx m) { Shape r = make_shape(FW_SHAPE_SPHERE, m); r.s.radius = radius; return r; } };
inline Shape make_rect(int kind, float a0, float a1, float b0, float b1, float k, MaterialIdx m) {
    Shape r = make_shape(kind, m); r.s.a_min = a0; r.s.a_max = a1; r.s.b_min = b0; r.s.b_max = b1; r.s.k = k; return r; }
struct XYRect { static Shape new_(float x0, float x1, float y0, float y1, float k, MaterialIdx m) { return make_rect(FW_SHAPE_XYRECT, x0, x1, y0, y1, k, m); } };
struct XZRect { static Shape new_(float x0, float x1, float z0, float z1, float k, MaterialIdx m) { return make_rect(FW_SHAPE_XZRECT, x0, x1, z0, z1, k, m); } };
struct YZRect { static Shape new_(float y0, float y1, float z0, float z1, float k, MaterialIdx m) { return make_rect(FW_SHAPE_YZRECT, y0, y1, z0, z1, k, m); } };
struct Rect3d { static Shape with_size(Vec3 size, MaterialIdx m) { Shape r = make_shape(FW_SHAPE_RECT3D, m); r.s.pos = {0, 0, 0}; r.s.size = lower(size); return r; } };
struct Cone { static Shape new_(float radius, float height, MaterialIdx m) { Shape r = make_shape(FW_SHAPE_CONE, m); r.s.radius = radius; r.s.height = height; return r; } };
constexpr float RADS_PER_DEG = 3.14159265358979323846f / 180.0f;   // f32::to_radians
struct Cylinder {
    static Shape new_(float radius, float height, MaterialIdx m) { return partial(radius, height, 360.0f, m); }
    static Shape partial(float radius, float height, float phi_deg, MaterialIdx m) { Shape r = make_shape(FW_SHAPE_CYLINDER, m); r.s.radius = radius; r.s.height = height; r.s.phi_max = phi_deg * RADS_PER_DEG; return r; } };
struct Disk {
    static Shape new_(float radius, MaterialIdx m) { Shape r = make_shape(FW_SHAPE_DISK, m); r.s.radius = radius; r.s.phi_max = 2.0f * 3.14159265358979323846f; return r; }
    static Shape partial(float radius, float phi_deg, float inner_radius, MaterialIdx m) { Shape r = make_shape(FW_SHAPE_DISK, m); r.s.radius = radius; r.s.phi_max = phi_deg * RADS_PER_DEG; r.s.inner_radius = inner_radius; return r; } };
struct TriangleMesh {   // mesh.rs:37-63: same two error strings
    static Shape new_(std::vector<float> verts /*3*n*/, std::vector<uint32_t> indicies, std::vector<float> normals /*empty or 3*n*/,
                      std::vector<float> uvs /*empty or 2*n*/, MaterialIdx m) {
        size_t n = verts.size() / 3;
        if (!normals.empty() && normals.size() != 3 * n) throw std::runtime_error("TriangleMesh::new() -- normals.len() must equal verts.len()");
        if (!uvs.empty() && uvs.size() != 2 * n) throw std::runtime_error("TriangleMesh::new() -- uvs.len() must equal verts.len()");
        Shape r = make_shape(FW_SHAPE_TRIANGLE_MESH, m);
        r.verts = std::make_shared<std::vector<float>>(std::move(verts)); r.indices = std::make_shared<std::vector<uint32_t>>(std::move(indicies));
        if (!normals.empty()) r.normals = std::make_shared<std::vector<float>>(std::move(normals));
        if (!uvs.empty()) r.uvs = std::make_shared<std::vector<float>>(std::move(uvs));
        return r; }
};

// ---- environments ---------------------------------------------------------------------------------------------
struct Environment { fw_environment e{}; std::shared_ptr<std::vector<float>> hdr; };
struct ColorEnv { static Environment new_(Vec3 c) { Environment r; r.e.kind = FW_ENV_COLOR; r.e.color = lower(c); return r; } };
struct SkyEnv { static Environment new_(Vec3 zenith, Vec3 horizon) { Environment r; r.e.kind = FW_ENV_SKY; r.e.zenith = lower(zenith); r.e.horizon = lower(horizon); return r; }
                static Environment default_() { return new_({0.5f, 0.7f, 1.0f}, {1, 1, 1}); } };
struct HdrEnvironment { static Environment new_(uint32_t w, uint32_t h, std::vector<float> rgb) { Environment r; r.e.kind = FW_ENV_HDR; r.e.hdr_w = w; r.e.hdr_h = h;
                        r.hdr = std::make_shared<std::vector<float>>(std::move(rgb)); r.e.hdr_rgb = r.hdr->data(); return r; } };

// ---- RenderObject / Scene -----------------------------------------------------------------------------------------
struct RenderObject {
    Shape obj; Vec3 pos; Rotor3 rotation; bool flip = false;
    static RenderObject new_(Shape s) { RenderObject r; r.obj = std::move(s); return r; }
    RenderObject position(float x, float y, float z) && { pos = {x, y, z}; return std::move(*this); }
    RenderObject position_vec(Vec3 p) && { pos = p; return std::move(*this); }
    RenderObject rotate(Rotor3 r) && { rotation = r; return std::move(*this); }
    RenderObject flip_normals() && { flip = !flip; return std::move(*this); }
};

struct Scene {
    std::vector<RenderObject> render_objects; std::vector<Material> materials; Environment environment = ColorEnv::new_({0, 0, 0});   // scene.rs:36
    static Scene new_() { return {}; }
    RenderObjectIdx add_object(RenderObject o) { render_objects.push_back(std::move(o)); return render_objects.size() - 1; }
    MaterialIdx add_material(Material m) { materials.push_back(std::move(m)); return (MaterialIdx)materials.size() - 1; }
    RenderObjectIdx add_volume(RenderObject o, float density, TexturePtr texture) {   // scene.rs:47-62
        MaterialIdx mat = add_material(IsotropicMat::new_(std::move(texture)));
        Shape medium = make_shape(FW_SHAPE_CONSTANT_MEDIUM, mat); medium.s.density = density;
        medium.inner = std::make_shared<Shape>(std::move(o.obj)); o.obj = std::move(medium);
        return add_object(std::move(o)); }
    void set_environment(Environment e) { environment = std::move(e); }
};

struct CameraSettings {   // camera.rs:26-36 defaults
    Vec3 cam_pos_{0, 0, -10}, look_at_{0, 0, 0}; float vfov = 30, aperture_ = 0, focus_dist_ = 10;
    static CameraSettings default_() { return {}; }
    CameraSettings cam_pos(Vec3 v) && { cam_pos_ = v; return std::move(*this); }
    CameraSettings look_at(Vec3 v) && { look_at_ = v; return std::move(*this); }
    CameraSettings field_of_view(float f) && { vfov = f; return std::move(*this); }
    CameraSettings aperture(float a) && { aperture_ = a; return std::move(*this); }
    CameraSettings focus_dist(float d) && { focus_dist_ = d; return std::move(*this); }
};

// Lowers a Scene to the flat arrays of fw_scene_desc (owning storage for the duration of a render call).
class Lowered {
  public:
    explicit Lowered(const Scene &sc) {
        for (const Material &m : sc.materials) { fw_material fm = m.m; fm.texture = m.tex ? add_texture(*m.tex) : -1; materials.push_back(fm); }
        for (const RenderObject &ro : sc.render_objects) {
            fw_object o{}; o.shape = add_shape(ro.obj); o.position = lower(ro.pos);
            o.rotation = fw_rotor3{ro.rotation.s, ro.rotation.xy, ro.rotation.xz, ro.rotation.yz}; o.flip_normals = ro.flip ? 1 : 0;
            objects.push_back(o); }
        desc.objects = objects.data(); desc.n_objects = (uint32_t)objects.size();
        desc.shapes = shapes.data(); desc.n_shapes = (uint32_t)shapes.size();
        desc.materials = materials.data(); desc.n_materials = (uint32_t)materials.size();
        desc.textures = textures.data(); desc.n_textures = (uint32_t)textures.size();
        desc.environment = sc.environment.e;
    }
    fw_scene_desc desc{};
  private:
    std::vector<fw_object> objects; std::vector<fw_shape> shapes; std::vector<fw_material> materials; std::vector<fw_texture> textures;
    int32_t add_texture(const Texture &t) {
        fw_texture ft = t.t;
        if (ft.kind == FW_TEX_CHECKER) { ft.odd = add_texture(*t.odd); ft.even = add_texture(*t.even); }
        textures.push_back(ft); return (int32_t)textures.size() - 1; }
    int32_t add_shape(const Shape &s) {
        fw_shape fs = s.s;
        if (fs.kind == FW_SHAPE_TRIANGLE_MESH) { fs.verts = s.verts->data(); fs.n_verts = (uint32_t)(s.verts->size() / 3);
            fs.indices = s.indices->data(); fs.n_indices = (uint32_t)s.indices->size();
            fs.normals = s.normals ? s.normals->data() : nullptr; fs.uvs = s.uvs ? s.uvs->data() : nullptr; }
        if (fs.kind == FW_SHAPE_CONSTANT_MEDIUM) fs.inner = add_shape(*s.inner);
        shapes.push_back(fs); return (int32_t)shapes.size() - 1; }
};

// fw_init (ABI v7): the device's one-off costs — HIP context, code objects, kernel handles, the path arena — where the host wants them, i.e.
// before the region it times (main.rs:40-44 starts with the reference's own binary loaded).  Optional: the first render does it otherwise.
inline void init(int device = 0, uint64_t arena_bytes = 0) {
    const int rc = fw_init(device, arena_bytes);
    if (rc != FW_OK) throw std::runtime_error(std::string(fw_strerror(rc)) + " | " + fw_last_error());
}

struct Renderer {   // render.rs:59-218; Default: 1920x1080, 128 spp, multithreaded, no BVH, gamma 2.2
    size_t width_ = 1920, height_ = 1080, samples_ = 128; bool multithreaded_ = true, use_bvh_ = false; float gamma_ = 2.2f;
    CameraSettings camera_; uint64_t seed_ = 0; int device_ = 0;
    static Renderer default_() { return {}; }
    Renderer width(size_t w) && { width_ = w; return std::move(*this); }
    Renderer height(size_t h) && { height_ = h; return std::move(*this); }
    Renderer samples(size_t s) && { samples_ = s; return std::move(*this); }
    Renderer multithreaded(bool m) && { multithreaded_ = m; return std::move(*this); }
    Renderer use_bvh(bool b) && { use_bvh_ = b; return std::move(*this); }
    Renderer gamma(float g) && { gamma_ = g; return std::move(*this); }
    Renderer camera(CameraSettings c) && { camera_ = c; return std::move(*this); }
    Renderer seed(uint64_t s) && { seed_ = s; return std::move(*this); }
    Renderer device(int d) && { device_ = d; return std::move(*this); }

    fw_render_params params() const {
        fw_render_params p{}; p.width = (uint32_t)width_; p.height = (uint32_t)height_; p.samples = (uint32_t)samples_; p.gamma = gamma_;
        p.use_bvh = use_bvh_; p.multithreaded = multithreaded_; p.seed = seed_; p.rng_mode = FW_RNG_CTR;
        p.camera = fw_camera_settings{lower(camera_.cam_pos_), lower(camera_.look_at_), camera_.vfov, camera_.aperture_, camera_.focus_dist_};
        return p; }

    // several GPUs from this one process (fw_render_scene_tiled): the image does not depend on the list
    Renderer devices(std::vector<int> d) && { devices_ = std::move(d); return std::move(*this); }
    std::vector<int> devices_;

    // `pub fn render(&self, scene: Scene) -> Vec<Color>` (render.rs:109): row 0 = image top
    std::vector<Color> render(const Scene &scene, fw_stats *stats = nullptr) const {
        Lowered low(scene);
        fw_render_params p = params();
        std::vector<Color> buffer(width_ * height_, Color{0, 0, 0});
        int rc = devices_.empty()
                     ? fw_render_scene(&low.desc, &p, device_, reinterpret_cast<uint8_t *>(buffer.data()), nullptr, nullptr, stats)
                     : fw_render_scene_tiled(&low.desc, &p, devices_.data(), (int)devices_.size(), reinterpret_cast<uint8_t *>(buffer.data()), nullptr, nullptr, stats);
        if (rc != FW_OK) throw std::runtime_error(std::string(fw_strerror(rc)) + " | " + fw_last_error());   // the reference panics here
        return buffer; }

    // Progressive preview (not in the reference): `passes` passes over the samples, `on_pass(k, image)` after each; `accum`
    // (width*height*4 floats) may be kept by the caller as a checkpoint.  The last image equals render()'s bit for bit.
    template <class F>
    std::vector<Color> render_progressive(const Scene &scene, size_t passes, F on_pass, std::vector<float> *accum_io = nullptr) const {
        Lowered low(scene);
        fw_scene *sc = nullptr;
        int rc = fw_scene_create(&low.desc, device_, &sc);
        if (rc != FW_OK) throw std::runtime_error(std::string(fw_strerror(rc)) + " | " + fw_last_error());
        std::vector<float> local; std::vector<float> &accum = accum_io ? *accum_io : local;
        accum.assign(width_ * height_ * 4, 0.f);
        std::vector<Color> buffer(width_ * height_, Color{0, 0, 0});
        passes = std::max<size_t>(1, std::min(passes, samples_));
        for (size_t k = 0; k < passes; k++) {
            const size_t lo = samples_ * k / passes, hi = samples_ * (k + 1) / passes;
            if (hi == lo) continue;
            fw_render_params p = params(); p.samples = (uint32_t)(hi - lo);
            rc = fw_render_progressive(sc, &p, (uint32_t)lo, accum.data(), reinterpret_cast<uint8_t *>(buffer.data()), nullptr, nullptr, nullptr);
            if (rc != FW_OK) { fw_scene_destroy(sc); throw std::runtime_error(std::string(fw_strerror(rc)) + " | " + fw_last_error()); }
            on_pass(k, buffer);
        }
        fw_scene_destroy(sc);
        return buffer; }
};

// window.rs:59-66 `save_image` (binary PPM: no PNG encoder is linked into this header)
inline void save_image_ppm(const std::vector<Color> &render, const std::string &path, size_t width, size_t height) {
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("Failed to save image to " + path);
    std::fprintf(f, "P6\n%zu %zu\n255\n", width, height);
    std::fwrite(render.data(), 3, width * height, f);
    std::fclose(f);
}

} // namespace firework
