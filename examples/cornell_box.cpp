// examples/cornell_box.cpp — the reference's examples/cornell_box.rs, line for line, on include/firework.hpp.
// usage: cornell_box [width height samples [out.ppm [passes]]]   (defaults 300 300 1000 as in the reference's main();
//        passes > 1: progressive render, the image is rewritten after every pass and ends up bit-identical)
#include "firework.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>

using namespace firework;

static float to_radians(float deg) { return deg * RADS_PER_DEG; }

Scene cornell_box() {   // cornell_box.rs:10-48
    Scene world = Scene::new_();

    MaterialIdx red = world.add_material(LambertianMat::with_color({0.65f, 0.05f, 0.05f}));
    MaterialIdx white = world.add_material(LambertianMat::with_color({0.73f, 0.73f, 0.73f}));
    MaterialIdx green = world.add_material(LambertianMat::with_color({0.12f, 0.45f, 0.15f}));
    MaterialIdx light = world.add_material(EmissiveMat::with_color({15.f, 15.f, 15.f}));

    world.add_object(RenderObject::new_(XZRect::new_(213.f, 343.f, 227.f, 332.f, 554.f, light)));
    world.add_object(RenderObject::new_(YZRect::new_(0.f, 555.f, 0.f, 555.f, 555.f, green)).flip_normals());
    world.add_object(RenderObject::new_(YZRect::new_(0.f, 555.f, 0.f, 555.f, 0.f, red)));
    world.add_object(RenderObject::new_(XZRect::new_(0.f, 555.f, 0.f, 555.f, 0.f, white)));
    world.add_object(RenderObject::new_(XZRect::new_(0.f, 555.f, 0.f, 555.f, 555.f, white)).flip_normals());
    world.add_object(RenderObject::new_(XYRect::new_(0.f, 555.f, 0.f, 555.f, 555.f, white)).flip_normals());
    world.add_object(RenderObject::new_(Rect3d::with_size({165.f, 165.f, 165.f}, white))
                         .rotate(Rotor3::from_rotation_xz(to_radians(18.f)))
                         .position(130.f, 0.f, 65.f));
    world.add_object(RenderObject::new_(Rect3d::with_size({165.f, 330.f, 165.f}, white))
                         .rotate(Rotor3::from_rotation_xz(to_radians(-15.f)))
                         .position(265.f, 0.f, 295.f));
    return world;
}

int main(int argc, char **argv) {
    size_t width = argc > 3 ? strtoul(argv[1], nullptr, 10) : 300, height = argc > 3 ? strtoul(argv[2], nullptr, 10) : 300,
           samples = argc > 3 ? strtoul(argv[3], nullptr, 10) : 1000;
    Scene scene = cornell_box();
    try { init(); } catch (const std::exception &e) { std::fprintf(stderr, "init failed: %s\n", e.what()); return 1; }   // fw_init: not part of the reference's timed region
    auto start = std::chrono::steady_clock::now();

    CameraSettings camera = CameraSettings::default_().cam_pos({278.f, 278.f, -800.f}).look_at({278.f, 278.f, 0.f}).field_of_view(40.f);
    Renderer renderer = Renderer::default_().width(width).height(height).samples(samples).camera(camera);

    fw_stats st{};
    std::vector<Color> render;
    const size_t passes = argc > 5 ? strtoul(argv[5], nullptr, 10) : 1;
    try {
        if (passes > 1) {
            render = renderer.render_progressive(scene, passes, [&](size_t k, const std::vector<Color> &img) {
                std::printf("pass %zu of %zu\n", k + 1, passes);
                if (argc > 4) save_image_ppm(img, argv[4], width, height);
            });
        } else render = renderer.render(scene, &st);
    }
    catch (const std::exception &e) { std::fprintf(stderr, "render failed: %s\n", e.what()); return 1; }

    auto end = std::chrono::steady_clock::now();
    std::printf("Finished Rendering in %lld s\n", (long long)std::chrono::duration_cast<std::chrono::seconds>(end - start).count());
    uint64_t h = 1469598103934665603ull;   // FNV-1a of the RGB8 buffer: lets a test compare with the Python host
    for (const Color &c : render) for (uint8_t b : {c.r, c.g, c.b}) { h ^= b; h *= 1099511628211ull; }
    std::printf("rays=%llu samples=%llu gpu_ms=%.3f fnv1a=%016llx\n", (unsigned long long)st.rays, (unsigned long long)st.samples, st.ms_render, (unsigned long long)h);
    if (argc > 4) save_image_ppm(render, argv[4], width, height);
    return 0;
}
