"""GPU parity tests (run with `-m gpu` on an MI355X): the HIP wavefront path, called through the C ABI,
against the CPU oracle on the same seeded inputs.

Tolerance (BASELINE.json north_star): per-pixel RGB within 1e-3 RMS of the CPU reference at the same
seed, measured on the post-gamma, clamped, pre-quantisation floats.  Because the device code
is compiled with -ffp-contract=off and IEEE division/sqrt, hit/miss decisions are bit-identical and the
observed RMS is far below the gate; the linear-image check below is much tighter than 1e-3.
"""
import os
import numpy as np
import pytest

from firework_amd import _lib, scenes
from firework_amd.api import (CameraSettings, CheckerTexture, ConstantTexture, DielectricMat, EmissiveMat,
                              HdrEnvironment, ImageTexture, LambertianMat, MarbleTexture, MetalMat,
                              PerlinNoiseTexture, Rect3d, RenderObject, Renderer, Rotor3, Scene, SkyEnv, Sphere,
                              TriangleMesh, TurbulenceTexture, XYRect, XZRect, YZRect)

pytestmark = pytest.mark.gpu

RMS_GATE = 1e-3


def rms(a, b):
    a = np.nan_to_num(np.asarray(a, np.float64), nan=0.0, posinf=1e30, neginf=-1e30)
    b = np.nan_to_num(np.asarray(b, np.float64), nan=0.0, posinf=1e30, neginf=-1e30)
    return float(np.sqrt(np.mean((a - b) ** 2)))


def check(oracle, scene, renderer, lin_rtol=2e-4):
    """Zero tolerance on the path and on the reference's output type: the device runs glibc's libm bits (fw_libm.h) and the reference's own
    walk for the rays whose result depends on traversal (k_extend_exact), so every path takes the oracle's segments — every ray count per depth
    must be the oracle's, and every u8.  What that rests on, by scene class (round 5):
      * constant textures only (cornell, suzanne, hdri, volume, teapot; the chain state): the pre-gamma means are the oracle's BIT FOR BIT — a
        path multiplies its attenuations back to front like the recursion of render.rs:23-28 — so the u8 are equal BY CONSTRUCTION;
      * a varying texture somewhere (C1, C5, earth, the random scenes): the default carries the running product ((a0 a1) a2) e, one rounding per
        path away from the reference's a0 (a1 (a2 e)): the means agree to ~1 ulp and a u8 can differ where a mean sits on a quantisation
        boundary — 1 of 6 000 random renders, one value (profiles/r04z_fuzz_seed40318.txt).  The u8 assertion below therefore holds for these
        scenes at the sizes and seeds of this suite, not by construction; option EXACT_PRODUCT=1 (fw_set_option) makes it by construction
        for them too, at the price of 16 more bytes per ray (test_exact_product_option_gives_the_oracles_means_bit_for_bit)."""
    gpu = renderer.render_full(scene)
    cpu = oracle.render(scene, renderer)
    r = rms(gpu.gamma, cpu.gamma)
    # pixels whose LINEAR value differs by more than float noise: a path took a different branch
    scale = np.maximum(np.abs(cpu.linear), 1e-3)
    bad = int((np.abs(gpu.linear - cpu.linear) > lin_rtol * scale + 1e-6).any(axis=1).sum())
    d8 = int((gpu.rgb8 != cpu.rgb8).sum())
    print(f"rms={r:.3e} bad_pixels={bad}/{gpu.linear.shape[0]} u8_diffs={d8} rays gpu={gpu.stats['rays']} cpu={cpu.stats['rays']}")
    assert r <= RMS_GATE
    assert bad == 0
    assert d8 == 0
    assert list(gpu.stats["rays_per_depth"]) == list(cpu.stats["rays_per_depth"])
    assert gpu.stats["rays"] == cpu.stats["rays"]
    assert gpu.stats["samples"] == cpu.stats["samples"]
    return gpu, cpu


def test_device_present():
    assert _lib.device_count() >= 1


def test_division_and_sqrt_helpers_are_ieee_exact():
    """fw_kernels.hip's fdiv/fsqrt (Newton/residual chain without v_div_scale) vs the compiler's `/` and sqrtf,
    bit for bit: exact on renderer-range operands; the full-range count is reported (scaling cases only)."""
    d, s = _lib.selftest_arith(1 << 26, seed=7, mode=0)
    assert (d, s) == (0, 0)
    d1, s1 = _lib.selftest_arith(1 << 24, seed=9, mode=1)
    print(f"full-range mismatches out of {1 << 24}: div={d1} sqrt={s1}")
    assert d1 < (1 << 24) * 0.2 and s1 < (1 << 24) * 0.2


def test_cornell_box(oracle):                      # C2: rects + rotated boxes + emissive, linear scene
    s, r = scenes.config("C2_cornell_box", 64, 64, 64)
    gpu, cpu = check(oracle, s, r)
    assert gpu.stats["rays"] == cpu.stats["rays"]
    assert gpu.stats["rays_per_depth"] == cpu.stats["rays_per_depth"]


def test_cornell_box_bvh(oracle):
    s, r = scenes.config("C2_cornell_box", 48, 48, 32)
    check(oracle, s, r.use_bvh(True))


def test_random_spheres(oracle):                   # C1: spheres, checker, metal, dielectric, sky, aperture, TLAS
    s, r = scenes.config("C1_random_spheres", 100, 56, 16)
    check(oracle, s, r)


def test_random_spheres_linear_scan(oracle):
    s, r = scenes.config("C1_random_spheres", 64, 36, 8)
    check(oracle, s, r.use_bvh(False))


def test_suzanne(oracle):                          # C3: triangle mesh BLAS + TLAS, rotated emissive rect
    s, r = scenes.config("C3_suzanne", 128, 72, 16)
    check(oracle, s, r)


def test_hdri(oracle):                             # C4a: HDR equirect environment (smaller map for the test)
    s, r = scenes.hdri_test(scenes.synthetic_hdr(512, 256))
    r.width(64).height(64).samples(16)
    check(oracle, s, r)


def test_volume(oracle):                           # C4b: ConstantMedium + Isotropic + glass
    s, r = scenes.config("C4b_volume_test", 64, 64, 32)
    check(oracle, s, r)


def test_part2(oracle):                            # C5: 1409 objects, image + turbulence textures, two media
    s, r = scenes.config("C5_part2_all", 96, 54, 8)
    check(oracle, s, r)


def test_textures_and_smooth_normals(oracle):
    """Checker-of-checker, Perlin, Marble, Image textures and a mesh WITH vertex normals and uvs
    (the smooth-normal branch, mesh.rs:206-207)."""
    rng = np.random.default_rng(3)
    sc = Scene.new()
    img = (rng.random((16, 32, 3)) * 255).astype(np.uint8)
    m_img = sc.add_material(LambertianMat.new(ImageTexture.new(img)))
    m_chk = sc.add_material(LambertianMat.new(CheckerTexture.new(CheckerTexture.with_colors((1, 0, 0), (0, 1, 0), 3.0),
                                                                  ConstantTexture.new((0.2, 0.2, 0.9)), 1.5)))
    m_per = sc.add_material(LambertianMat.new(PerlinNoiseTexture.new(2.0)))
    m_mar = sc.add_material(LambertianMat.new(MarbleTexture.new(4, 3.0)))
    m_tur = sc.add_material(EmissiveMat.new(TurbulenceTexture.new(3, 1.5)))
    sc.add_object(RenderObject.new(Sphere.new(1.0, m_img)).position(-2.2, 1.0, 0.0))
    sc.add_object(RenderObject.new(Sphere.new(1.0, m_per)).position(0.0, 1.0, 0.0))
    sc.add_object(RenderObject.new(Sphere.new(1.0, m_mar)).position(2.2, 1.0, 0.0)
                  .rotate(Rotor3.from_euler_angles(0.3, -0.4, 0.9)))
    sc.add_object(RenderObject.new(XZRect.new(-20.0, 20.0, -20.0, 20.0, 0.0, m_chk)))
    sc.add_object(RenderObject.new(XYRect.new(-3.0, 3.0, 0.0, 3.0, 2.5, m_tur)).flip_normals())
    # a small fan mesh with normals + uvs, image-textured
    verts = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0.2], [0, 1, 0], [-1, 0.5, 0.3]], np.float32)
    nrm = np.array([[0, 0, -1], [0.2, 0, -1], [0.2, 0.2, -1], [0, 0.2, -1], [-0.3, 0, -1]], np.float32)
    uvs = np.array([[0, 0], [1, 0], [1, 1], [0, 1], [0.5, 0.5]], np.float32)
    mesh = TriangleMesh.new(verts, [0, 1, 2, 0, 2, 3, 0, 3, 4], nrm, uvs, m_img)
    sc.add_object(RenderObject.new(mesh).position(-0.5, 2.2, -1.5).rotate(Rotor3.from_rotation_xz(0.4)))
    sc.set_environment(SkyEnv.default())
    cam = CameraSettings.default().cam_pos((0.5, 2.5, -9.0)).look_at((0.0, 1.0, 0.0)).field_of_view(35.0)
    for bvh in (False, True):
        r = Renderer.default().width(96).height(64).samples(16).use_bvh(bvh).camera(cam)
        check(oracle, sc, r)


def test_conics(oracle):
    """Cone / Cylinder (full and partial, inside + outside shells) / Disk (full and annular sector), image-textured
    so that their uv code runs — examples/conics.rs restated at a closer camera.  With use_bvh the disks vanish,
    exactly as in the reference: Disk::bounding_box is degenerate (disk.rs:85-90)."""
    import math
    from firework_amd.api import Cone, Cylinder, Disk
    rng = np.random.default_rng(5)
    sc = Scene.new()
    img = (rng.random((32, 32, 3)) * 255).astype(np.uint8)
    uvm = sc.add_material(LambertianMat.new(ImageTexture.new(img)))
    blue = sc.add_material(LambertianMat.with_color((0.0, 0.2, 0.4)))
    grey = sc.add_material(LambertianMat.with_color((0.5, 0.5, 0.5)))
    light = sc.add_material(EmissiveMat.with_color((8.0, 8.0, 8.0)))
    sc.add_object(RenderObject.new(Cylinder.new(2.0, 3.0, uvm)).position(-4.2, 0.0, 0.0))
    sc.add_object(RenderObject.new(Disk.new(2.0, blue)).position(-4.2, 3.0, 0.0))
    sc.add_object(RenderObject.new(Cone.new(2.0, 3.0, uvm)).position(-1.0, 0.0, -4.0))
    rot = Rotor3.from_euler_angles(math.radians(90.0), math.radians(30.0), math.radians(-35.0))
    sc.add_object(RenderObject.new(Cylinder.partial(1.5, 3.0, 300.0, uvm)).rotate(rot).position(3.0, 1.5, 1.0))
    sc.add_object(RenderObject.new(Cylinder.partial(1.49, 3.0, 300.0, uvm)).rotate(rot).position(3.0, 1.5, 1.0).flip_normals())
    sc.add_object(RenderObject.new(Disk.partial(1.5, 300.0, 0.8, uvm)).rotate(rot).position(3.0, 1.5, 1.0))
    sc.add_object(RenderObject.new(XZRect.new(-100.0, 100.0, -100.0, 100.0, 0.0, grey)))
    sc.add_object(RenderObject.new(YZRect.new(0.0, 20.0, 0.0, 10.0, -3.0, light)).rotate(Rotor3.from_rotation_xz(-30.0)).position(0.0, 0.0, -10.0))
    sc.set_environment(SkyEnv.default())
    cam = CameraSettings.default().cam_pos((0.0, 8.0, 14.0)).look_at((0.0, 1.0, 0.0)).field_of_view(40.0)
    for bvh in (False, True):
        r = Renderer.default().width(96).height(54).samples(16).use_bvh(bvh).camera(cam)
        check(oracle, sc, r)


def test_medium_around_a_box_and_seed(oracle):
    sc = Scene.new()
    white = sc.add_material(LambertianMat.with_color((0.7, 0.7, 0.7)))
    sc.add_volume(RenderObject.new(Rect3d.with_size((2.0, 2.0, 2.0), white)).position(-1.0, 0.0, -1.0)
                  .rotate(Rotor3.from_rotation_xz(0.5)), 0.8, ConstantTexture.from_rgb(0.9, 0.9, 0.9))
    sc.add_object(RenderObject.new(XZRect.new(-10.0, 10.0, -10.0, 10.0, 0.0, white)))
    sc.set_environment(SkyEnv.default())
    cam = CameraSettings.default().cam_pos((0.0, 3.0, -8.0)).look_at((0.0, 1.0, 0.0))
    r = Renderer.default().width(48).height(48).samples(32).camera(cam).seed(2 ** 33 + 5)
    check(oracle, sc, r)


def test_batching_and_pixel_subsets_are_bit_identical(oracle):
    """The image must not depend on the wavefront pool size, nor on which pixels a call renders:
    this is what makes an N-GPU tiled render identical to a 1-GPU render."""
    s, r = scenes.config("C2_cornell_box", 40, 40, 24)
    full = r.render_full(s)
    small = r.paths_per_batch(40 * 40 * 5).render_full(s)          # 5 spp per batch -> 5 batches, ragged last
    assert np.array_equal(full.linear, small.linear) and np.array_equal(full.rgb8, small.rgb8)
    assert full.stats["rays"] == small.stats["rays"]
    ids = np.arange(40 * 40, dtype=np.uint32)
    parts = [ids[k::3] for k in range(3)]                           # three interleaved "ranks"
    out = np.zeros_like(full.linear)
    for p in parts:
        out[p] = r.render_full(s, pixel_ids=p).linear
    assert np.array_equal(out, full.linear)


def test_image_does_not_depend_on_the_walked_tree(monkeypatch):
    """Hits are ordered by (t, in-order rank of the REFERENCE tree) and boxes are culled conservatively, so walking the
    binned-SAH tree (default) or the reference's median-split topology (FIREWORK_BVH=median) gives the same bits and the
    same ray counts — at sizes where exact ties and hits on box faces do occur (part2's floor of abutting boxes)."""
    for name, w, h, spp in (("C5_part2_all", 320, 180, 8), ("C3_suzanne", 160, 90, 16), ("C1_random_spheres", 200, 112, 16)):
        s, r = scenes.config(name, w, h, spp)
        monkeypatch.delenv("FIREWORK_BVH", raising=False)
        sah = r.render_full(s)
        monkeypatch.setenv("FIREWORK_BVH", "median")
        med = r.render_full(s)
        monkeypatch.delenv("FIREWORK_BVH", raising=False)
        assert sah.stats["rays"] == med.stats["rays"], name
        assert np.array_equal(sah.linear, med.linear), name


def test_degenerate_rect_bounds_admit_nothing_as_in_the_reference(oracle):
    """rect.rs:55-62 rejects on `x < min || x > max`, so an interval with min > max admits nothing: an XZRect given with
    swapped bounds is invisible, and a Rect3d with a negative size keeps only the faces whose own bounds are in order.
    (hit_rect's median form needs lo <= hi and hands these to the difference form.)"""
    sc = Scene.new()
    grey = sc.add_material(LambertianMat.with_color((0.6, 0.6, 0.6)))
    red = sc.add_material(LambertianMat.with_color((0.8, 0.1, 0.1)))
    green = sc.add_material(LambertianMat.with_color((0.1, 0.8, 0.1)))
    sc.add_object(RenderObject.new(XZRect.new(-20.0, 20.0, -20.0, 20.0, 0.0, grey)))
    sc.add_object(RenderObject.new(XYRect.new(2.0, -2.0, 0.0, 3.0, 1.0, red)))                      # a_min > a_max: never hit
    sc.add_object(RenderObject.new(Rect3d.with_size((1.5, 1.5, -1.5), green)).position(-3.0, 0.0, 0.0))
    sc.add_object(RenderObject.new(Rect3d.with_size((1.0, 2.0, 1.0), red)).rotate(Rotor3.from_rotation_xz(0.5)).position(1.5, 0.0, -1.0))
    sc.set_environment(SkyEnv.default())
    cam = CameraSettings.default().cam_pos((0.5, 3.0, -9.0)).look_at((0.0, 1.0, 0.0))
    for bvh in (False, True):
        r = Renderer.default().width(64).height(48).samples(16).use_bvh(bvh).camera(cam).seed(77)
        check(oracle, sc, r)


def test_progressive_passes_and_a_resumed_checkpoint_equal_one_render(tmp_path):
    """fw_render_progressive: draws are keyed by the absolute sample index and a pixel's sums are taken in sample order, so
    k passes — also interrupted and resumed from the saved accumulation buffer — give the bits of one render."""
    for name, bvh in (("C2_cornell_box", False), ("C3_suzanne", True)):
        s, r = scenes.config(name, 48, 36, 21)
        r.use_bvh(bvh)
        full = r.render_full(s)
        passes = list(r.render_progressive(s, 4))
        assert len(passes) == 4 and sum(p.stats["samples"] for p in passes) == full.stats["samples"]
        assert np.array_equal(passes[-1].linear, full.linear) and np.array_equal(passes[-1].rgb8, full.rgb8)
        assert not np.array_equal(passes[0].linear, full.linear)            # a preview, not the final image
        ck = str(tmp_path / f"{name}.npz")
        gen = r.render_progressive(s, 3, checkpoint=ck)
        next(gen); gen.close()                                                 # "killed" after the first pass
        resumed = list(r.render_progressive(s, 3, checkpoint=ck))
        assert len(resumed) == 2 and np.array_equal(resumed[-1].linear, full.linear)
    # a checkpoint of ANOTHER scene / camera / use_bvh is not blended in; a name without .npz is the file np.savez writes;
    # a truncated file restarts from zero instead of crashing
    s, r = scenes.config("C2_cornell_box", 48, 36, 12)
    full = r.render_full(s)
    ck = str(tmp_path / "ck")                                                  # no extension
    gen = r.render_progressive(s, 3, checkpoint=ck)
    next(gen); gen.close()
    import os
    assert os.path.exists(ck + ".npz") and not os.path.exists(ck)
    assert len(list(r.render_progressive(s, 3, checkpoint=ck))) == 2           # found and resumed
    s2, r2 = scenes.config("C3_suzanne", 48, 36, 12)
    other = list(r2.render_progressive(s2, 3, checkpoint=ck))                  # same size and seed, other scene
    assert len(other) == 3 and np.array_equal(other[-1].linear, r2.render_full(s2).linear)
    cam2 = CameraSettings.default().cam_pos((278.0, 278.0, -700.0)).look_at((278.0, 278.0, 0.0)).field_of_view(40.0)
    gen = r.render_progressive(s, 3, checkpoint=ck); next(gen); gen.close()
    moved = list(r.camera(cam2).render_progressive(s, 3, checkpoint=ck))       # same scene, other camera
    assert len(moved) == 3 and np.array_equal(moved[-1].linear, r.render_full(s).linear)
    with open(ck + ".npz", "r+b") as f:
        f.truncate(100)
    again = list(r.render_progressive(s, 2, checkpoint=ck))
    assert len(again) == 2 and np.array_equal(again[-1].linear, r.render_full(s).linear)


def test_single_process_tiled_render_is_bit_identical():
    """fw_render_scene_tiled (one host thread per device, 16x16 tiles dealt diagonally, host-side scatter): the same bits and
    counters as one device, here with device 0 listed three times (its calls are serialised); ragged frame sizes too."""
    for name, w, h, bvh in (("C2_cornell_box", 72, 50, False), ("C3_suzanne", 64, 36, True)):
        s, r = scenes.config(name, w, h, 9)
        r.use_bvh(bvh)
        one = r.render_full(s)
        tiled = _lib.render_scene_tiled(s.to_desc(), r, [0, 0, 0])
        assert np.array_equal(one.linear, tiled.linear) and np.array_equal(one.rgb8, tiled.rgb8) and np.array_equal(one.gamma, tiled.gamma)
        assert one.stats["rays"] == tiled.stats["rays"] and one.stats["samples"] == tiled.stats["samples"]
        assert one.stats["rays_per_depth"] == tiled.stats["rays_per_depth"]


def test_refilling_and_chunked_bvh_walks_are_bit_identical(monkeypatch):
    """use_bvh scenes walk with in-wave ray refill (k_extend_tlas; with meshes k_extend_tlas_park + k_blas); the chunked
    k_extend_bvh with its in-kernel BLAS runs stays selectable (FIREWORK_TLAS_REFILL=0): same bits, same ray counts."""
    for name, w, h, spp in (("C3_suzanne", 96, 54, 12), ("C5_part2_all", 96, 54, 6), ("C1_random_spheres", 100, 56, 12)):
        s, r = scenes.config(name, w, h, spp)
        monkeypatch.delenv("FIREWORK_TLAS_REFILL", raising=False)
        refill = r.render_full(s)
        monkeypatch.setenv("FIREWORK_TLAS_REFILL", "0")
        chunk = r.render_full(s)
        monkeypatch.delenv("FIREWORK_TLAS_REFILL", raising=False)
        assert refill.stats["rays_per_depth"] == chunk.stats["rays_per_depth"], name
        assert np.array_equal(refill.linear, chunk.linear), name


def test_wide_node_walks_are_bit_identical_to_the_pair_node_walks(monkeypatch):
    """The LDS-resident walks step through WIDE nodes (four children; f32 planes, or 8-bit planes rounded outward where the f32
    nodes do not fit a CU's LDS): same bits and ray counts as the pair-node kernels (WIDE=0), in either encoding."""
    for name, w, h, spp in (("C3_suzanne", 160, 90, 12), ("C5_part2_all", 160, 90, 6), ("C1_random_spheres", 100, 56, 12), ("teapot", 160, 90, 6)):
        s, r = scenes.config(name, w, h, spp)
        monkeypatch.setenv("FIREWORK_WIDE", "0")
        pair = r.render_full(s)
        for enc in ("f32", "q8", None):
            if enc is None:
                monkeypatch.delenv("FIREWORK_WIDE", raising=False)
            else:
                monkeypatch.setenv("FIREWORK_WIDE", enc)
            wide = r.render_full(s)
            assert wide.stats["rays_per_depth"] == pair.stats["rays_per_depth"], (name, enc)
            assert np.array_equal(wide.linear, pair.linear, equal_nan=True), (name, enc)
        monkeypatch.delenv("FIREWORK_WIDE", raising=False)


def test_chain_state_equals_the_running_product_in_u8_and_the_oracle_before_gamma(oracle, monkeypatch):
    """Where every attenuation is a constant of its material (cornell, suzanne, hdri, volume) a path carries its material ids (8 bytes
    of state instead of 16) and multiplies them out when it deposits, back to front — the association of the reference's recursion
    (render.rs:23-28).  Against the 16-byte running product (NO_CHAIN): same rays, same u8.  Against the oracle: the pre-gamma means
    BIT for bit (the running product is one rounding per path away from them)."""
    if os.environ.get("FIREWORK_FUSED") or os.environ.get("FIREWORK_SHADE_LIST"):
        pytest.skip("k_bounce and k_shade's list mode (A/B build) carry the running product: there is no chain state to compare under this switch")
    cases = [scenes.config("C2_cornell_box", 96, 96, 40), scenes.config("C3_suzanne", 160, 90, 16), scenes.config("C4b_volume_test", 96, 96, 24)]
    sh, rh = scenes.hdri_test(scenes.synthetic_hdr(512, 256))
    cases.append((sh, rh.width(96).height(96).samples(16)))
    for s, r in cases:
        monkeypatch.delenv("FIREWORK_NO_CHAIN", raising=False)
        chain = r.render_full(s)
        monkeypatch.setenv("FIREWORK_NO_CHAIN", "1")
        prod = r.render_full(s)
        monkeypatch.delenv("FIREWORK_NO_CHAIN", raising=False)
        cpu = oracle.render(s, r)
        assert chain.stats["bytes_shade"] < prod.stats["bytes_shade"]                     # the mode is on
        assert chain.stats["rays_per_depth"] == prod.stats["rays_per_depth"] == cpu.stats["rays_per_depth"]
        assert np.array_equal(chain.rgb8, prod.rgb8) and np.array_equal(chain.rgb8, cpu.rgb8)
        assert np.allclose(chain.linear, prod.linear, rtol=1e-5, atol=1e-7)
        assert np.array_equal(chain.linear, cpu.linear.astype(np.float32)), float(np.abs(chain.linear - cpu.linear).max())


def test_exact_product_option_gives_the_oracles_means_bit_for_bit(oracle, monkeypatch):
    """EXACT_PRODUCT=1: scenes with a varying texture (no chain state) keep the attenuation of every scattering (16 bytes per segment at the
    path's home slot) and a path that ends in light multiplies them back to front, a0 * (a1 * (... * e)) — the association of the
    reference's recursion (render.rs:23-28) instead of the running product's.  Pre-gamma means: the oracle's, bit for bit; without the
    option: within a few ulp.  Rays and u8: equal either way on these cases."""
    if os.environ.get("FIREWORK_FUSED") == "1":
        pytest.skip("the fused bounce kernel of the A/B build carries the running product: EXACT_PRODUCT does not apply to it")
    cases = [scenes.config("C1_random_spheres", 100, 56, 16), scenes.config("C5_part2_all", 96, 54, 4), scenes.config("earth", 64, 64, 8)]
    sc, cam = _random_scene(2)
    cases.append((sc, Renderer.default().width(72).height(48).samples(8).use_bvh(True).camera(cam).seed(2000006)))
    for s, r in cases:
        monkeypatch.delenv("FIREWORK_EXACT_PRODUCT", raising=False)
        prod = r.render_full(s)
        monkeypatch.setenv("FIREWORK_EXACT_PRODUCT", "1")
        exact = r.render_full(s)
        monkeypatch.delenv("FIREWORK_EXACT_PRODUCT", raising=False)
        cpu = oracle.render(s, r)
        assert exact.stats["bytes_shade"] > prod.stats["bytes_shade"]                     # the option is on
        assert exact.stats["rays_per_depth"] == prod.stats["rays_per_depth"] == cpu.stats["rays_per_depth"]
        assert np.array_equal(exact.rgb8, cpu.rgb8) and np.array_equal(prod.rgb8, cpu.rgb8)
        assert np.array_equal(exact.linear, cpu.linear.astype(np.float32)), float(np.abs(exact.linear - cpu.linear).max())
        assert np.allclose(prod.linear, cpu.linear, rtol=1e-5, atol=1e-7)


def test_fused_bounce_kernel_is_bit_identical_to_the_split_kernels(monkeypatch):
    """FIREWORK_FUSED=1 intersects and shades in one launch per segment (k_bounce) with the device functions of
    k_extend / k_shade: same bits, same ray counts, linear scan and TLAS."""
    for name, bvh in (("C2_cornell_box", False), ("C2_cornell_box", True), ("C4b_volume_test", False), ("C1_random_spheres", True)):
        s, r = scenes.config(name, 48, 40, 12)
        r.use_bvh(bvh)
        monkeypatch.delenv("FIREWORK_FUSED", raising=False)
        monkeypatch.setenv("FIREWORK_NO_CHAIN", "1")      # k_bounce carries the running product: compare with the split kernels doing the same
        split = r.render_full(s)
        monkeypatch.delenv("FIREWORK_NO_CHAIN", raising=False)
        monkeypatch.setenv("FIREWORK_FUSED", "1")
        fused = r.render_full(s)
        monkeypatch.delenv("FIREWORK_FUSED", raising=False)
        assert fused.stats["n_extend_launches"] == 0 and split.stats["n_extend_launches"] > 0
        assert np.array_equal(split.linear, fused.linear) and split.stats["rays_per_depth"] == fused.stats["rays_per_depth"]


def test_deposit_bitmap_layouts_and_hit_record_sizes_are_bit_identical(monkeypatch):
    """Over a black environment k_shade marks the paths that wrote a radiance record in a bitmap that is slot-major for whole
    frames and pixel-major for small pixel sets (DFrame.dep_pixel_major); scenes of spheres, rects and boxes get 4-byte hit
    records with t recomputed in k_shade (DFrame.hit4).  Each switch forced both ways: same bits, same deposit counts."""
    for w, h, spp in ((96, 96, 33), (50, 31, 7)):                    # spp not a multiple of 32: a pixel's bits straddle words
        s, r = scenes.config("C2_cornell_box", w, h, spp)
        r.count_deposits(True)
        frames = []
        for env in ({}, {"FIREWORK_DEP_SLOT_MAJOR": "1"}, {"FIREWORK_DEP_PIXEL_MAJOR": "1"}, {"FIREWORK_NO_HIT4": "1"}):
            for k in ("FIREWORK_DEP_SLOT_MAJOR", "FIREWORK_DEP_PIXEL_MAJOR", "FIREWORK_NO_HIT4"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            frames.append(r.render_full(s))
        for f in frames[1:]:
            assert np.array_equal(f.linear, frames[0].linear) and f.stats["deposits"] == frames[0].stats["deposits"]
            assert f.stats["rays_per_depth"] == frames[0].stats["rays_per_depth"]
        if not os.environ.get("FIREWORK_FUSED"):                                    # (k_bounce, A/B build: no hit records at all)
            assert frames[3].stats["bytes_shade"] > frames[0].stats["bytes_shade"]  # 8-byte hit records


def test_deferred_boxes_of_the_linear_scan_are_bit_identical(oracle, monkeypatch):
    """k_extend_linear_defer takes the trailing Rect3d objects of a linear scene out of the per-ray loop (per-box lists in LDS,
    64 rays per exact test).  Rooms with one, two and three trailing boxes — rotated, overlapping each other, standing on the
    floor (equal-t ties with the floor rect), one flipped, glass and metal among them, a sphere in between — rendered with the
    deferral, without it (FIREWORK_NO_DEFER) and by the oracle: same rays per depth, same bits between the two GPU paths."""
    rng = np.random.default_rng(77)
    u = lambda a, b: float(rng.uniform(a, b))
    for n_boxes in (1, 2, 3):
        sc = Scene.new()
        white = sc.add_material(LambertianMat.with_color((0.73, 0.73, 0.73)))
        red = sc.add_material(LambertianMat.with_color((0.65, 0.05, 0.05)))
        light = sc.add_material(EmissiveMat.with_color((9.0, 9.0, 9.0)))
        glass = sc.add_material(DielectricMat.new(1.5))
        metal = sc.add_material(MetalMat.new((0.8, 0.8, 0.9), 0.1))
        sc.add_object(RenderObject.new(XZRect.new(-4, 4, -4, 4, 0.0, white)))                    # floor
        sc.add_object(RenderObject.new(XZRect.new(-4, 4, -4, 4, 5.0, white)).flip_normals())     # ceiling
        sc.add_object(RenderObject.new(XZRect.new(-1, 1, -1, 1, 4.99, light)).flip_normals())
        sc.add_object(RenderObject.new(XYRect.new(-4, 4, 0, 5, 4.0, red)).flip_normals())
        sc.add_object(RenderObject.new(YZRect.new(0, 5, -4, 4, -4.0, white)))
        sc.add_object(RenderObject.new(Sphere.new(0.7, glass)).position(-2.0, 0.7, 1.0))
        mats = [white, metal, glass]
        for b in range(n_boxes):
            ro = RenderObject.new(Rect3d.with_size((u(1.0, 2.0), u(1.0, 3.0), u(1.0, 2.0)), mats[b % 3]))
            ro = ro.rotate(Rotor3.from_rotation_xz(u(-1.0, 1.0)) if b != 1 else Rotor3.from_euler_angles(u(-0.4, 0.4), u(-0.4, 0.4), u(-0.4, 0.4)))
            ro = ro.position(u(-1.5, 0.5), 0.0 if b != 1 else u(0.0, 0.8), u(-1.0, 1.0))         # boxes overlap; on the floor: ties
            if b == 2:
                ro = ro.flip_normals()
            sc.add_object(ro)
        cam = CameraSettings.default().cam_pos((0.5, 2.5, -11.0)).look_at((0.0, 2.2, 0.0)).field_of_view(38.0)
        r = Renderer.default().width(72).height(56).samples(24).use_bvh(False).camera(cam).seed(5 + n_boxes)
        monkeypatch.delenv("FIREWORK_NO_DEFER", raising=False)
        gpu, cpu = check(oracle, sc, r)
        monkeypatch.setenv("FIREWORK_NO_DEFER", "1")
        plain = r.render_full(sc)
        monkeypatch.delenv("FIREWORK_NO_DEFER", raising=False)
        assert np.array_equal(gpu.linear, plain.linear) and gpu.stats["rays_per_depth"] == plain.stats["rays_per_depth"]
        assert gpu.stats["rays_per_depth"] == cpu.stats["rays_per_depth"]


def test_a_repeated_frame_replayed_as_a_graph_is_the_same_frame(oracle, monkeypatch):
    """GRAPH: the second time in a row the same frame is asked for, its launches between fork and join are captured into a hipGraph; from
    the third time on they are replayed (fw_stats.reserved bit 31).  Same kernels, same dependencies: the frames and the ray counts are
    those of plain launches and of the oracle; a change of any argument (here: the seed) is a miss, never a stale replay."""
    monkeypatch.delenv("FIREWORK_PHASE_LOCK", raising=False)       # (the switch matrix forces the lock on; a locked frame is never captured)
    for name, w, h, spp in (("C1_random_spheres", 100, 56, 8), ("C3_suzanne", 96, 54, 4), ("C4b_volume_test", 64, 64, 4)):
        s, r = scenes.config(name, w, h, spp)
        ds = _lib.DeviceScene(s.to_desc(), 0)              # a scene that stays in HBM (a one-shot render uploads a new one every time: never the same frame twice)
        monkeypatch.setenv("FIREWORK_GRAPH", "0")
        plain = ds.render(r)
        assert not plain.stats["reserved"] >> 31
        monkeypatch.setenv("FIREWORK_GRAPH", "1")
        frames = [ds.render(r) for _ in range(5)]
        flags = [f.stats["reserved"] >> 31 for f in frames]
        assert flags[0] == 0 and flags[-2:] == [1, 1], (name, flags)       # as ever first; captured + launched once the same key has been seen twice, then replayed
        for f in frames:
            assert np.array_equal(f.rgb8, plain.rgb8) and np.array_equal(f.linear, plain.linear) and f.stats["rays_per_depth"] == plain.stats["rays_per_depth"]
        other = ds.render(r.seed(r.settings["seed"] + 1))                                  # another frame: not a replay, and not the old one
        assert not other.stats["reserved"] >> 31 and not np.array_equal(other.linear, plain.linear)
        cpu = oracle.render(s, r)
        assert np.array_equal(other.rgb8, cpu.rgb8) and other.stats["rays_per_depth"] == cpu.stats["rays_per_depth"]
        monkeypatch.delenv("FIREWORK_GRAPH", raising=False)


def test_camera_positions_with_zeros_of_either_sign(oracle):
    """A pinhole camera's rays all start at the camera position, so segment 0 stores 16 bytes per ray (the direction) instead of 24 — as long as
    `position + (+-0)` is the position, bit for bit: true for every coordinate except a NEGATIVE zero (-0 + +0 = +0; camera.rs:109-116 adds the
    lens offset even when the aperture is 0).  +0.0 coordinates (hdri_test, volume_test stand at x = 0.0) take the 16-byte form since round 5, a
    -0.0 coordinate keeps the 24-byte one; either way the frame is the oracle's."""
    from firework_amd.api import CameraSettings
    s, r = scenes.config("C4b_volume_test", 48, 48, 4)
    for pos, short in (((0.0, 2.0, -10.0), True), ((0.0, 0.0, -10.0), True), ((-0.0, 2.0, -10.0), False), ((3.0, -0.0, -10.0), False), ((3.0, 2.0, -10.0), True)):
        r.camera(CameraSettings.default().cam_pos(pos).look_at((0.0, 0.0, 0.0)))
        gpu, cpu = r.render_full(s), oracle.render(s, r)
        assert np.array_equal(gpu.rgb8, cpu.rgb8) and gpu.stats["rays_per_depth"] == cpu.stats["rays_per_depth"], pos
        assert np.array_equal(gpu.linear, cpu.linear), pos           # volume_test: constant textures only, the chain state: means bit for bit
        per_sample = gpu.stats["bytes_raygen"] / gpu.stats["samples"]
        assert per_sample in ((16, 20) if short else (24, 28)), (pos, per_sample)     # (+ 4 bytes where the frame is traced in the library's own tile order)


def test_device_error_word_turns_a_stack_overflow_into_a_status_code(monkeypatch):
    """A/B build only (make ab): the wide walks check every push against the LDS levels their launch reserved, and a wave that stops
    making progress leaves its loop; either sets the device's error word and the render returns FW_ERR_HIP instead of a frame
    (a fault inside a kernel cannot be turned into a code afterwards: the runtime aborts the process — round 4's r04t).
    Here: stacks of TWO levels for part2's TLAS (needs ~20) and suzanne's BLAS."""
    if not _lib.has_ab():
        pytest.skip("the error word exists in the A/B build only (make ab; FIREWORK_LIB=firework_amd/lib/variants/lib_ab.so)")
    from firework_amd import _abi as A
    for sw in ("FIREWORK_TLAS_REFILL", "FIREWORK_FUSED"):          # (the switch matrix runs this file under switches that take the wide walks out: TLAS_REFILL=0
        monkeypatch.delenv(sw, raising=False)                       #  walks suzanne's BLAS in place, FUSED=1 intersects in the bounce kernel — nothing to overflow)
    for name, w, h, spp in (("C5_part2_all", 96, 54, 4), ("C3_suzanne", 96, 54, 4)):
        s, r = scenes.config(name, w, h, spp)
        good = r.render_full(s)
        monkeypatch.setenv("FIREWORK_DEBUG_WIDE_LEVELS", "2")
        with pytest.raises(_lib.FireworkError) as e:
            r.render_full(s)
        monkeypatch.delenv("FIREWORK_DEBUG_WIDE_LEVELS", raising=False)
        assert e.value.status == A.FW_ERR_HIP and "stack overflow" in str(e.value)
        again = r.render_full(s)                    # the word was cleared: the next frame is a frame again
        assert np.array_equal(good.rgb8, again.rgb8)


def test_errors_cross_the_abi_as_codes():
    from firework_amd import _abi as A
    r = Renderer.default().width(8).height(8).samples(1)
    with pytest.raises(_lib.FireworkError) as e:
        r.render(Scene.new())
    assert e.value.status == A.FW_ERR_EMPTY_SCENE
    s, _ = scenes.cornell_box()
    with pytest.raises(_lib.FireworkError) as e:
        r.samples(0).render(s)
    assert e.value.status == A.FW_ERR_BAD_ARG


def test_error_codes_match_between_oracle_and_gpu(oracle):
    """Malformed inputs: same status from the HIP library and the oracle (the reference panics in these places)."""
    from firework_amd import _abi as A
    r = Renderer.default().width(8).height(8).samples(1)

    def status_gpu(scene, **kw):
        try:
            r.render_full(scene, **kw)
            return 0
        except _lib.FireworkError as e:
            return e.status

    def status_cpu(scene, **kw):
        try:
            oracle.render(scene, r, **kw)
            return 0
        except oracle.OracleError as e:
            return e.status

    # NaN position -> NaN bounding-box centre -> bvh.rs:34 "Float comparison failed in BVH constructor"
    sc = Scene.new()
    m = sc.add_material(LambertianMat.with_color((0.5, 0.5, 0.5)))
    sc.add_object(RenderObject.new(Sphere.new(1.0, m)).position(float("nan"), 0.0, 0.0))
    sc.add_object(RenderObject.new(Sphere.new(1.0, m)).position(2.0, 0.0, 0.0))
    sc.add_object(RenderObject.new(Sphere.new(1.0, m)).position(4.0, 0.0, 0.0))
    assert status_gpu(sc) == A.FW_ERR_NAN_BBOX
    r.use_bvh(True)
    assert status_cpu(sc) == A.FW_ERR_NAN_BBOX
    r.use_bvh(False)
    # material index out of range
    sc = Scene.new()
    sc.add_object(RenderObject.new(Sphere.new(1.0, 3)))
    assert status_gpu(sc) == status_cpu(sc) == A.FW_ERR_BAD_ARG
    # vertex index out of range
    sc = Scene.new()
    m = sc.add_material(LambertianMat.with_color((0.5, 0.5, 0.5)))
    sc.add_object(RenderObject.new(TriangleMesh.new(np.zeros((3, 3), np.float32), [0, 1, 7], None, None, m)))
    assert status_gpu(sc) == status_cpu(sc) == A.FW_ERR_BAD_ARG
    # pixel id outside the image
    s2, _ = scenes.cornell_box()
    assert status_gpu(s2, pixel_ids=np.array([0, 64], np.uint32)) == status_cpu(s2, pixel_ids=np.array([0, 64], np.uint32)) == A.FW_ERR_BAD_ARG


def test_ragged_sizes_and_single_pixel(oracle):
    """Non-multiple-of-64 path counts, 1x1 image, 1 spp, more pixels than pool slots."""
    s, _ = scenes.cornell_box()
    cam = CameraSettings.default().cam_pos((278.0, 278.0, -800.0)).look_at((278.0, 278.0, 0.0)).field_of_view(40.0)
    for (w, h, spp, ppb) in [(1, 1, 1, 0), (7, 3, 5, 0), (33, 17, 3, 100), (65, 1, 129, 4096)]:
        r = Renderer.default().width(w).height(h).samples(spp).camera(cam).paths_per_batch(ppb)
        check(oracle, s, r)


def test_full_size_properties_cornell():
    """At BASELINE's full size (512x512; spp reduced to keep the test short) check size-independent
    properties instead of the oracle: determinism, ray accounting, and the estimator's linearity in spp."""
    s, r = scenes.config("C2_cornell_box", 512, 512, 64)
    a = r.render_full(s)
    b = r.render_full(s)
    assert np.array_equal(a.linear, b.linear)                        # run-to-run determinism
    st = a.stats
    assert st["samples"] == 512 * 512 * 64
    assert st["rays_per_depth"][0] == st["samples"]                  # every sample traces a primary ray
    assert all(x >= y for x, y in zip(st["rays_per_depth"], st["rays_per_depth"][1:]))   # queues only shrink
    assert st["rays"] == sum(st["rays_per_depth"])
    assert 4.9 < st["rays"] / st["samples"] < 5.15                   # oracle: 5.02 segments/sample on this scene
    # 64 spp == mean of two disjoint 32-sample halves? not with per-sample keys starting at 0; instead:
    # the first 32 samples of the 64-spp render are exactly the 32-spp render.
    h = r.samples(32).render_full(s)
    assert h.stats["rays"] < st["rays"]
    m = float(np.mean(a.linear)), float(np.mean(h.linear))
    assert abs(m[0] - m[1]) / m[0] < 0.05


@pytest.mark.parametrize("name,spp", [("C3_suzanne", 4), ("C4a_hdri_test", 4), ("C4b_volume_test", 4), ("C5_part2_all", 2)])
def test_full_resolution_properties(name, spp):
    """BASELINE's full resolutions (spp reduced): size-independent properties instead of the oracle —
    run-to-run determinism, ray accounting, finite output, and independence from the pool size."""
    if name == "C4a_hdri_test":
        s, r = scenes.hdri_test(scenes.synthetic_hdr(1024, 512))
        r.width(1024).height(1024).samples(spp)
    else:
        s, r = scenes.config(name, samples=spp)
    a = r.render_full(s)
    b = r.paths_per_batch(r.settings["width"] * r.settings["height"] * max(1, spp // 2)).render_full(s)
    assert np.array_equal(a.rgb8, b.rgb8) and np.array_equal(a.linear, b.linear, equal_nan=True)
    st = a.stats
    assert st["samples"] == r.settings["width"] * r.settings["height"] * spp
    assert st["rays_per_depth"][0] == st["samples"] and st["rays"] == sum(st["rays_per_depth"])
    assert all(x >= y for x, y in zip(st["rays_per_depth"], st["rays_per_depth"][1:]))
    assert st["rays"] == b.stats["rays"]
    assert np.isfinite(a.gamma[~np.isnan(a.gamma)]).all() and a.rgb8.any()


def _random_scene(seed):
    """Every shape / material / texture / environment kind, random transforms (incl. sub-2.56-degree rotations that
    take the `cos_trace >= 0.999` branch of scene.rs:187,242), media around spheres and boxes, a small mesh."""
    import math
    from firework_amd.api import ColorEnv, Cone, Cylinder, Disk, IsotropicMat
    rng = np.random.default_rng(seed)
    u = lambda a, b: float(rng.uniform(a, b))
    sc = Scene.new()
    img = (rng.random((8, 8, 3)) * 255).astype(np.uint8)
    texs = [ConstantTexture.new((u(0.2, 0.9), u(0.2, 0.9), u(0.2, 0.9))), CheckerTexture.with_colors((0.1, 0.1, 0.1), (0.9, 0.9, 0.9), u(1, 6)),
            PerlinNoiseTexture.new(u(0.5, 4)), TurbulenceTexture.new(3, u(0.5, 3)), MarbleTexture.new(3, u(0.5, 3)), ImageTexture.new(img)]
    mats = [sc.add_material(LambertianMat.new(t)) for t in texs]
    mats += [sc.add_material(MetalMat.new((u(0.5, 1), u(0.5, 1), u(0.5, 1)), u(0, 0.6))), sc.add_material(DielectricMat.new(u(1.2, 1.8))),
             sc.add_material(EmissiveMat.with_color((u(2, 6), u(2, 6), u(2, 6))))]
    pick = lambda: int(rng.choice(mats))

    def rot():
        k = rng.integers(0, 4)
        if k == 0:
            return Rotor3.identity()
        if k == 1:
            return Rotor3.from_rotation_xz(u(-0.03, 0.03))          # below the 0.999 threshold
        if k == 2:
            return Rotor3.from_rotation_xz(u(-3, 3))
        return Rotor3.from_euler_angles(u(-1, 1), u(-1, 1), u(-1, 1))

    makers = [lambda: Sphere.new(u(0.3, 1.0), pick()), lambda: Rect3d.with_size((u(0.4, 1.5), u(0.4, 1.5), u(0.4, 1.5)), pick()),
              lambda: XYRect.new(-1, 1, -1, 1, u(-0.5, 0.5), pick()), lambda: XZRect.new(-1, 1, -1, 1, u(-0.5, 0.5), pick()),
              lambda: YZRect.new(-1, 1, -1, 1, u(-0.5, 0.5), pick()), lambda: Cone.new(u(0.4, 1), u(0.5, 1.5), pick()),
              lambda: Cylinder.partial(u(0.4, 1), u(0.5, 1.5), u(90, 360), pick()), lambda: Disk.partial(u(0.6, 1.2), u(90, 360), u(0, 0.4), pick())]
    for _ in range(int(rng.integers(6, 14))):
        ro = RenderObject.new(makers[int(rng.integers(0, len(makers)))]()).rotate(rot()).position(u(-4, 4), u(0, 3), u(-4, 4))
        if rng.random() < 0.3:
            ro = ro.flip_normals()
        sc.add_object(ro)
    # two media (sphere and box boundaries) and a little mesh with normals
    sc.add_volume(RenderObject.new(Sphere.new(u(0.8, 1.5), mats[0])).position(u(-3, 3), 1.5, u(-3, 3)), u(0.3, 1.5), texs[0])
    sc.add_volume(RenderObject.new(Rect3d.with_size((1.5, 1.5, 1.5), mats[0])).rotate(rot()).position(u(-3, 3), 0.2, u(-3, 3)), u(0.3, 1.5), texs[1])
    n = 6
    verts = np.array([[math.cos(2 * math.pi * k / n), math.sin(2 * math.pi * k / n), u(-0.2, 0.2)] for k in range(n)] + [[0, 0, 0.5]], np.float32)
    idx = [v for k in range(n) for v in (n, k, (k + 1) % n)]
    nrm = verts / np.maximum(np.linalg.norm(verts, axis=1, keepdims=True), 1e-3)
    mesh = RenderObject.new(TriangleMesh.new(verts, idx, nrm if seed % 2 else None, None, pick())).rotate(rot()).position(u(-2, 2), 1.5, u(-2, 2))
    if seed % 4 != 3:      # every fourth scene has no mesh: under use_bvh those take k_extend_tlas, the others k_extend_bvh
        sc.add_object(mesh)
    # checker floor NOT on the plane y = 0: there CheckerTexture's sign(sin(scale*y)) is the sign of a rounding
    # residue, i.e. noise in the reference itself, and any ulp upstream (log10f in a medium) flips the colour
    sc.add_object(RenderObject.new(XZRect.new(-30.0, 30.0, -30.0, 30.0, -0.37, mats[1])))
    sc.set_environment(SkyEnv.default() if seed % 3 else ColorEnv.new((0.3, 0.3, 0.4)))
    cam = CameraSettings.default().cam_pos((u(-2, 2), u(2, 5), -11.0)).look_at((0.0, 1.0, 0.0)).field_of_view(40.0).aperture(u(0, 0.2)).focus_dist(11.0)
    return sc, cam


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_scenes(oracle, seed):
    sc, cam = _random_scene(seed)
    r = Renderer.default().width(72).height(48).samples(8).use_bvh(bool(seed % 2)).camera(cam).seed(seed * 1000003)
    check(oracle, sc, r)


def test_cli_progressive_and_checkpoint_write_the_same_png(tmp_path):
    """`python -m firework_amd` (main.rs:6-62 flags) with --progressive / --checkpoint ends with the PNG of a plain run."""
    import subprocess, sys, os
    from PIL import Image
    from firework_amd.yaml_io import save_scene
    scene, _ = scenes.suzanne()
    yml = tmp_path / "s.yml"
    save_scene(scene, yml)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, "-m", "firework_amd", "--scene-file", str(yml), "-s", "12", "--width", "64", "--height", "36"]
    a, b, ck = tmp_path / "a.png", tmp_path / "b.png", tmp_path / "ck.npz"
    out = subprocess.run(base + ["-o", str(a)], cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "Finished Rendering in" in out.stdout, out.stderr
    out = subprocess.run(base + ["-o", str(b), "--progressive", "3", "--checkpoint", str(ck)], cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.count("pass ") == 3, out.stderr
    assert np.array_equal(np.asarray(Image.open(a)), np.asarray(Image.open(b))) and ck.exists()
