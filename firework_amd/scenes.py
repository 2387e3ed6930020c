"""The scenes of BASELINE.json's five configs, restated from the reference's examples/*.rs.

These are scene DEFINITIONS (data), written against this package's mirror of the reference API so
each builder reads like the example it restates.  Where an example depends on something that is
not in the reference tree the substitution is stated in the docstring.
"""
import math
import os

import numpy as np

from .api import (CameraSettings, CheckerTexture, Cone, ConstantTexture, Cylinder, DielectricMat, Disk, EmissiveMat,
                  HdrEnvironment, ImageTexture, LambertianMat, MetalMat, Rect3d, RenderObject, Renderer, Rotor3, Scene,
                  SkyEnv, Sphere, TriangleMesh, TurbulenceTexture, XYRect, XZRect, YZRect, to_radians)

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES_DIR = os.path.join(_ROOT, "scenes")


class LayoutRng:
    """Deterministic uniform [0,1) stream for scene LAYOUT only (where the reference seeds
    `tiny_rng::Rng::new(12345)`, whose algorithm is not in the tree: random_spheres.rs:70,
    part2_all.rs:83).  64-bit LCG (Knuth MMIX constants), top 24 bits."""

    def __init__(self, seed):
        self.state = seed & 0xFFFFFFFFFFFFFFFF

    def rand_f32(self):
        self.state = (self.state * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        return float(np.float32((self.state >> 40) / 16777216.0))


def cornell_box():
    """examples/cornell_box.rs:10-48 verbatim.  -> (Scene, Renderer as in main(): 300x300 @1000, no BVH)"""
    world = Scene.new()
    red = world.add_material(LambertianMat.with_color((0.65, 0.05, 0.05)))
    white = world.add_material(LambertianMat.with_color((0.73, 0.73, 0.73)))
    green = world.add_material(LambertianMat.with_color((0.12, 0.45, 0.15)))
    light = world.add_material(EmissiveMat.with_color((15.0, 15.0, 15.0)))
    world.add_object(RenderObject.new(XZRect.new(213.0, 343.0, 227.0, 332.0, 554.0, light)))
    world.add_object(RenderObject.new(YZRect.new(0.0, 555.0, 0.0, 555.0, 555.0, green)).flip_normals())
    world.add_object(RenderObject.new(YZRect.new(0.0, 555.0, 0.0, 555.0, 0.0, red)))
    world.add_object(RenderObject.new(XZRect.new(0.0, 555.0, 0.0, 555.0, 0.0, white)))
    world.add_object(RenderObject.new(XZRect.new(0.0, 555.0, 0.0, 555.0, 555.0, white)).flip_normals())
    world.add_object(RenderObject.new(XYRect.new(0.0, 555.0, 0.0, 555.0, 555.0, white)).flip_normals())
    world.add_object(RenderObject.new(Rect3d.with_size((165.0, 165.0, 165.0), white))
                     .rotate(Rotor3.from_rotation_xz(to_radians(18.0)))
                     .position(130.0, 0.0, 65.0))
    world.add_object(RenderObject.new(Rect3d.with_size((165.0, 330.0, 165.0), white))
                     .rotate(Rotor3.from_rotation_xz(to_radians(-15.0)))
                     .position(265.0, 0.0, 295.0))
    camera = CameraSettings.default().cam_pos((278.0, 278.0, -800.0)).look_at((278.0, 278.0, 0.0)).field_of_view(40.0)
    renderer = Renderer.default().width(300).height(300).samples(1000).camera(camera)
    return world, renderer


def random_spheres(seed=12345):
    """examples/random_spheres.rs:14-96.  Layout stream: LayoutRng(seed) instead of the
    unavailable tiny_rng::Rng — same construction order and draw order."""
    rand = LayoutRng(seed)
    scene = Scene.new()
    checker_mat = scene.add_material(LambertianMat.new(CheckerTexture.with_colors((0.2, 0.4, 0.1), (0.9, 0.9, 0.9), 10.0)))
    scene.add_object(RenderObject.new(Sphere.new(1000.0, checker_mat)).position(0.0, -1000.0, -1.0))
    for x in range(-11, 11):
        for y in range(-11, 11):
            center = np.array([x + 0.9 * rand.rand_f32(), 0.2, y + 0.9 * rand.rand_f32()], np.float32)
            if float(np.linalg.norm(center - np.array([4.0, 0.2, 0.9], np.float32))) > 0.9:
                r = rand.rand_f32()
                if r < 0.8:   # the reference's `x > 0.0 && x < 0.8` arm (0.0 exactly is unreachable!() there)
                    mat = scene.add_material(LambertianMat.with_color((rand.rand_f32() * rand.rand_f32(),
                                                                       rand.rand_f32() * rand.rand_f32(),
                                                                       rand.rand_f32() * rand.rand_f32())))
                elif r < 0.95:
                    mat = scene.add_material(MetalMat.new((0.5 * (1.0 + rand.rand_f32()), 0.5 * (1.0 + rand.rand_f32()),
                                                           0.5 * (1.0 + rand.rand_f32())), 0.5 * rand.rand_f32()))
                else:
                    mat = scene.add_material(DielectricMat.new(1.5))
                scene.add_object(RenderObject.new(Sphere.new(0.2, mat)).position_vec(center))
    glass = scene.add_material(DielectricMat.new(1.5))
    diffuse = scene.add_material(LambertianMat.with_color((0.4, 0.2, 0.1)))
    metal = scene.add_material(MetalMat.new((0.7, 0.6, 0.5), 0.0))
    scene.add_object(RenderObject.new(Sphere.new(1.0, glass)).position(0.0, 1.0, 0.0))
    scene.add_object(RenderObject.new(Sphere.new(1.0, diffuse)).position(-4.0, 1.0, 0.0))
    scene.add_object(RenderObject.new(Sphere.new(1.0, metal)).position(4.0, 1.0, 0.0))
    scene.set_environment(SkyEnv.default())
    camera = CameraSettings.default().cam_pos((13.0, 2.0, 3.0)).look_at((0.0, 0.0, 0.0)).aperture(0.1)
    renderer = Renderer.default().width(960).height(540).samples(32).use_bvh(True).camera(camera)
    return scene, renderer


def load_suzanne_mesh(material):
    """The TriangleMesh of scenes/suzanne.yml (= tobj 1.0 output for suzanne.obj: 1966 verts, 968 tris),
    kept as the data fixture scenes/suzanne_mesh.npz (scripts/make_fixtures.py)."""
    d = np.load(os.path.join(SCENES_DIR, "suzanne_mesh.npz"))
    return TriangleMesh.new(d["verts"], d["indicies"], None, None, material)


def suzanne():
    """examples/suzanne.rs:53-96"""
    scene = Scene.new()
    diffuse = scene.add_material(LambertianMat.new(ConstantTexture.new((0.8, 0.2, 0.3))))
    scene.add_object(RenderObject.new(load_suzanne_mesh(diffuse)))
    scene.set_environment(SkyEnv.default())
    blue = scene.add_material(LambertianMat.with_color((0.2, 0.2, 0.8)))
    scene.add_object(RenderObject.new(XZRect.new(-100.0, 100.0, -100.0, 100.0, 0.0, blue)).position(0.0, -1.0, 0.0))
    light = scene.add_material(EmissiveMat.with_color((8.0, 8.0, 8.0)))
    scene.add_object(RenderObject.new(YZRect.new(0.0, 20.0, 0.0, 20.0, -3.0, light))
                     .rotate(Rotor3.from_rotation_xz(-30.0)).position(0.0, 4.0, 10.0))
    camera = CameraSettings.default().cam_pos((1.0, 2.5, 5.0)).look_at((0.0, 0.0, 0.0)).field_of_view(40.0)
    renderer = Renderer.default().width(960).height(540).samples(512).use_bvh(True).camera(camera)
    return scene, renderer


def synthetic_hdr(width=4096, height=2048):
    """Stand-in for urban_street_04_4k.hdr (hdri_test.rs:52), which is not in the reference tree:
    a deterministic, seed-free equirect f32 RGB map — sky gradient over a dim ground plus one
    sun disc of radiance 1e4."""
    v = (np.arange(height, dtype=np.float32) + 0.5) / height          # 0 = top (zenith)
    u = (np.arange(width, dtype=np.float32) + 0.5) / width
    elev = (0.5 - v) * np.float32(math.pi)                            # +pi/2 at top
    t = np.clip(np.sin(elev) * 0.5 + 0.5, 0, 1).astype(np.float32)[:, None]
    sky = (1 - t) * np.array([1.0, 1.0, 1.0], np.float32) + t * np.array([0.3, 0.5, 1.0], np.float32)
    img = np.broadcast_to(sky[:, None, :], (height, width, 3)).copy()
    ground = (elev < 0)[:, None]
    img = np.where(ground[:, :, None], np.float32(0.25) * np.array([0.6, 0.55, 0.5], np.float32), img).astype(np.float32)
    az = (u * 2 * np.float32(math.pi))[None, :]
    sun_el, sun_az, sun_r = np.float32(0.9), np.float32(1.0), np.float32(0.03)
    cosd = (np.sin(elev)[:, None] * np.sin(sun_el) + np.cos(elev)[:, None] * np.cos(sun_el) * np.cos(az - sun_az))
    img[cosd > np.cos(sun_r)] = np.float32(1e4)
    return np.ascontiguousarray(img, dtype=np.float32)


def hdri_test(hdr=None):
    """examples/hdri_test.rs:85-118 with a synthetic environment map (see synthetic_hdr)."""
    scene = Scene.new()
    scene.set_environment(HdrEnvironment(synthetic_hdr() if hdr is None else hdr))
    glass = scene.add_material(DielectricMat.new(1.5))
    diffuse = scene.add_material(LambertianMat.with_color((0.8, 0.8, 0.8)))
    metal = scene.add_material(MetalMat.new((0.7, 0.7, 0.7), 0.0))
    scene.add_object(RenderObject.new(Sphere.new(1.0, glass)).position(0.0, 1.0, 0.0))
    scene.add_object(RenderObject.new(Sphere.new(1.0, diffuse)).position(-4.0, 1.0, 0.0))
    scene.add_object(RenderObject.new(Sphere.new(1.0, metal)).position(4.0, 1.0, 0.0))
    scene.add_object(RenderObject.new(XZRect.new(-100.0, 100.0, -100.0, 100.0, 0.0, diffuse)))
    camera = CameraSettings.default().cam_pos((0.0, 2.0, -10.0)).look_at((0.0, 0.0, 0.0))
    renderer = Renderer.default().width(500).height(250).samples(500).camera(camera)
    return scene, renderer


def volume_test(medium_radius=1.0, glass_radius=1.01, density=0.5, with_glass=True):
    """examples/volume_test.rs:11-67.  The defaults are the example as committed.  The reference's volume.png was rendered
    when the two radii were 1.5 and 1.51 (everything else as today: tests/test_reference_renders.py fits the silhouette and
    the inner circle of the glass sphere seen through the medium) — `volume_test(1.5, 1.51)` is that scene."""
    scene = Scene.new()
    glass = scene.add_material(DielectricMat.new(1.5))
    diffuse = scene.add_material(LambertianMat.with_color((0.8, 0.8, 0.8)))
    scene.add_material(MetalMat.new((0.7, 0.7, 0.7), 0.0))   # `metal` is added but unused in the example
    scene.add_volume(RenderObject.new(Sphere.new(medium_radius, diffuse)).position(0.0, 1.0, 0.0), density,
                     ConstantTexture.from_rgb(0.5, 0.0, 0.8))
    if with_glass:
        scene.add_object(RenderObject.new(Sphere.new(glass_radius, glass)).position(0.0, 1.0, 1.0))
    scene.add_object(RenderObject.new(XZRect.new(-100.0, 100.0, -100.0, 100.0, 0.0, diffuse)))
    light = scene.add_material(EmissiveMat.with_color((8.0, 8.0, 8.0)))
    scene.add_object(RenderObject.new(YZRect.new(0.0, 20.0, 0.0, 10.0, -3.0, light))
                     .rotate(Rotor3.from_rotation_xz(-30.0)).position(0.0, 0.0, -10.0))
    scene.set_environment(SkyEnv.default())
    camera = CameraSettings.default().cam_pos((0.0, 2.0, -10.0)).look_at((0.0, 0.0, 0.0))
    renderer = Renderer.default().width(960).height(540).samples(2048).camera(camera)
    return scene, renderer


def part2_all(seed=12345):
    """examples/part2_all.rs:13-98, restated with Scene::add_volume semantics (the example calls
    a `ConstantMedium::new(.., &mut scene)` that no longer exists in src/) and LayoutRng(seed)."""
    rand = LayoutRng(seed)
    scene = Scene.new()
    ground = scene.add_material(LambertianMat.with_color((0.48, 0.83, 0.53)))
    for x in range(20):
        for z in range(20):
            pos = (-10.0 + x, 0.0, -10.0 + z)
            size = (1.0, rand.rand_f32() + 0.01, 1.0)
            scene.add_object(RenderObject.new(Rect3d.with_size(size, ground)).position_vec(pos))
    light = scene.add_material(EmissiveMat.with_color((7.0, 7.0, 7.0)))
    scene.add_object(RenderObject.new(XZRect.new(1.23, 4.23, 1.47, 4.12, 5.54, light)))
    brown = scene.add_material(LambertianMat.with_color((0.7, 0.3, 0.1)))
    scene.add_object(RenderObject.new(Sphere.new(0.5, brown)).position(4.0, 4.0, 2.0))
    glass = scene.add_material(DielectricMat.new(1.5))
    scene.add_object(RenderObject.new(Sphere.new(0.5, glass)).position(2.6, 1.5, 0.45))
    metal = scene.add_material(MetalMat.new((0.8, 0.8, 0.9), 10.0))
    scene.add_object(RenderObject.new(Sphere.new(0.5, metal)).position(0.0, 1.5, 1.45))
    scene.add_object(RenderObject.new(Sphere.new(0.7, glass)).position(3.6, 1.5, 1.45))
    scene.add_volume(RenderObject.new(Sphere.new(0.7, glass)).position(3.6, 1.5, 1.45), 0.2,
                     ConstantTexture.new((0.2, 0.4, 0.9)))
    earth_mat = scene.add_material(LambertianMat.new(ImageTexture.from_path(os.path.join(SCENES_DIR, "earthmap.jpg"))))
    scene.add_object(RenderObject.new(Sphere.new(1.0, earth_mat)).position(4.0, 2.0, 4.0))
    noise = scene.add_material(LambertianMat.new(TurbulenceTexture.new(5, 10.0)))
    scene.add_object(RenderObject.new(Sphere.new(0.8, noise)).position(2.2, 2.8, 3.0))
    white = scene.add_material(LambertianMat.with_color((0.73, 0.73, 0.73)))
    for _ in range(1000):
        pos = 1.65 * np.array([rand.rand_f32(), rand.rand_f32(), rand.rand_f32()], np.float32) + np.array(
            [1.0, 2.7, 3.95], np.float32)
        scene.add_object(RenderObject.new(Sphere.new(0.1, white)).position_vec(pos))
    scene.add_volume(RenderObject.new(Sphere.new(5000.0, 0)), 0.0001, ConstantTexture.new((1.0, 1.0, 1.0)))
    camera = CameraSettings.default().cam_pos((-9.0, 3.0, -9.0)).look_at((1.0, 3.0, 2.0)).field_of_view(25.0)
    renderer = Renderer.default().width(600).height(800).samples(10000).use_bvh(True).camera(camera)
    return scene, renderer


def conics():
    """examples/conics.rs:11-93 (= scenes/conics.yml): Cylinder / Disk / Cone, the partial cylinder twice (the inner
    copy with flipped normals), an ImageTexture (uvmap.png), `from_euler_angles`.  960x540 @128, no BVH."""
    scene = Scene.new()
    uv_image_mat = scene.add_material(LambertianMat.new(ImageTexture.from_path(os.path.join(SCENES_DIR, "uvmap.png"))))
    scene.add_object(RenderObject.new(Cylinder.new(2.0, 3.0, uv_image_mat)).position(-4.2, 0.0, 0.0))
    blue = scene.add_material(LambertianMat.with_color((0.0, 0.2, 0.4)))
    scene.add_object(RenderObject.new(Disk.new(2.0, blue)).position(-4.2, 3.0, 0.0))
    scene.add_object(RenderObject.new(Cone.new(2.0, 3.0, uv_image_mat)).position(-1.0, 0.0, -4.0))
    euler = lambda: Rotor3.from_euler_angles(to_radians(90.0), to_radians(30.0), to_radians(-35.0))
    scene.add_object(RenderObject.new(Cylinder.partial(1.5, 3.0, 300.0, uv_image_mat)).position(3.0, 1.5, 1.0).rotate(euler()))
    scene.add_object(RenderObject.new(Cylinder.partial(1.49, 3.0, 300.0, uv_image_mat)).position(3.0, 1.5, 1.0).rotate(euler())
                     .flip_normals())
    scene.add_object(RenderObject.new(Disk.partial(1.5, 300.0, 0.8, uv_image_mat)).position(3.0, 1.5, 1.0).rotate(euler()))
    grey = scene.add_material(LambertianMat.with_color((0.5, 0.5, 0.5)))
    scene.add_object(RenderObject.new(XZRect.new(-100.0, 100.0, -100.0, 100.0, 0.0, grey)))
    light = scene.add_material(EmissiveMat.with_color((8.0, 8.0, 8.0)))
    scene.add_object(RenderObject.new(YZRect.new(0.0, 20.0, 0.0, 10.0, -3.0, light))
                     .rotate(Rotor3.from_rotation_xz(-30.0)).position(0.0, 0.0, -10.0))
    scene.set_environment(SkyEnv.default())
    camera = CameraSettings.default().cam_pos((6.0, 4.0, -7.0)).look_at((0.0, 1.5, 0.0)).field_of_view(60.0)
    renderer = Renderer.default().width(960).height(540).samples(128).camera(camera)
    return scene, renderer


def earth():
    """examples/earth.rs:12-53: four image-textured spheres (earthmap.jpg, uvmap.png).  800x800 @128, no BVH."""
    scene = Scene.new()
    earth_mat = scene.add_material(LambertianMat.new(ImageTexture.from_path(os.path.join(SCENES_DIR, "earthmap.jpg"))))
    uv_image_mat = scene.add_material(LambertianMat.new(ImageTexture.from_path(os.path.join(SCENES_DIR, "uvmap.png"))))
    scene.add_object(RenderObject.new(Sphere.new(0.25, earth_mat)))
    scene.add_object(RenderObject.new(Sphere.new(0.25, uv_image_mat)).position(1.0, 0.0, 0.0))
    scene.add_object(RenderObject.new(Sphere.new(0.25, earth_mat)).position(0.0, 1.0, 0.0))
    scene.add_object(RenderObject.new(Sphere.new(0.25, earth_mat)).position(0.0, 0.0, 1.0))
    grey = scene.add_material(LambertianMat.with_color((0.5, 0.5, 0.5)))
    scene.add_object(RenderObject.new(XZRect.new(-100.0, 100.0, -100.0, 100.0, 0.0, grey)))
    light = scene.add_material(EmissiveMat.with_color((8.0, 8.0, 8.0)))
    scene.add_object(RenderObject.new(YZRect.new(0.0, 20.0, 0.0, 10.0, -3.0, light))
                     .rotate(Rotor3.from_rotation_xz(-30.0)).position(0.0, 0.0, -10.0))
    scene.set_environment(SkyEnv.default())
    camera = CameraSettings.default().cam_pos((5.0, 5.0, 5.0)).look_at((0.0, 0.0, 0.0)).field_of_view(30.0)
    renderer = Renderer.default().width(800).height(800).samples(128).camera(camera)
    return scene, renderer


def load_teapot_meshes(material):
    """The four TriangleMeshes of scenes/teapot.yml (= tobj 1.0 output for teapot.obj's four `g` groups, with vertex
    normals: 6 320 triangles), kept as the data fixture scenes/teapot_mesh.npz (scripts/make_fixtures.py)."""
    d = np.load(os.path.join(SCENES_DIR, "teapot_mesh.npz"))
    return [TriangleMesh.new(d[f"verts{i}"], d[f"indicies{i}"], d[f"normals{i}"], None, material)
            for i in range(int(d["n_meshes"]))]


def teapot():
    """examples/teapot.rs:18-109 (= scenes/teapot.yml): smooth-normal meshes (mesh.rs:206-207), every mesh rotated by
    `from_rotation_xz(90.)` (radians, as written).  1920x1080 @512, BVH."""
    scene = Scene.new()
    diffuse = scene.add_material(LambertianMat.new(ConstantTexture.new((0.2, 0.8, 0.3))))
    for mesh in load_teapot_meshes(diffuse):
        scene.add_object(RenderObject.new(mesh).rotate(Rotor3.from_rotation_xz(90.0)))
    scene.set_environment(SkyEnv.default())
    grey = scene.add_material(LambertianMat.with_color((0.5, 0.5, 0.5)))
    scene.add_object(RenderObject.new(XZRect.new(-20.0, 20.0, -20.0, 20.0, 0.0, grey)))
    light = scene.add_material(EmissiveMat.with_color((20.0, 20.0, 20.0)))
    scene.add_object(RenderObject.new(YZRect.new(0.0, 4.0, 0.0, 4.0, -0.6, light))
                     .rotate(Rotor3.from_rotation_xz(-30.0)).position(0.0, 4.0, 10.0))
    camera = CameraSettings.default().cam_pos((1.0, 4.0, 8.0)).look_at((0.0, 1.0, 0.0)).field_of_view(40.0)
    renderer = Renderer.default().width(1920).height(1080).samples(512).use_bvh(True).camera(camera)
    return scene, renderer


# BASELINE.json configs: (scene builder, width, height, spp, use_bvh)
CONFIGS = {
    "C1_random_spheres": (random_spheres, 400, 225, 64, True),
    "C2_cornell_box": (cornell_box, 512, 512, 1024, False),
    "C3_suzanne": (suzanne, 1280, 720, 512, True),
    "C4a_hdri_test": (hdri_test, 1024, 1024, 512, False),
    "C4b_volume_test": (volume_test, 1024, 1024, 512, False),
    "C5_part2_all": (part2_all, 1920, 1080, 4096, True),
}


def config(name, width=None, height=None, samples=None):
    """(scene, renderer) for a BASELINE config, optionally at reduced size (parity tests)."""
    if name not in CONFIGS:          # an example of the reference by its builder's name here (teapot, conics, earth, ...): its own settings
        scene, renderer = globals()[name]()
        s = renderer.settings
        renderer.width(width or s["width"]).height(height or s["height"]).samples(samples or s["samples"])
        return scene, renderer
    build, w, h, spp, bvh = CONFIGS[name]
    scene, renderer = build()
    renderer.width(width or w).height(height or h).samples(samples or spp).use_bvh(bvh)
    return scene, renderer
