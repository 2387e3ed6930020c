"""GPU parity at every BASELINE config's FULL resolution (`-m gpu`): SURVEY §8(d) defines the 1e-3 RMS gate "at every
config's resolution"; spp is reduced so that the CPU oracle finishes in seconds (the images are compared sample for
sample — the counter RNG keys every draw by (pixel, sample) — so fewer samples test the same pixels, rays and branches).
Also: the reference's own scenes with big meshes (teapot.yml, 6 320 triangles with vertex normals) and a 44 k-triangle
mesh through the HIP path, and conics / earth at their example sizes."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from firework_amd import _lib, scenes
from firework_amd.api import CameraSettings, LambertianMat, RenderObject, Renderer, Scene, SkyEnv, TriangleMesh, XZRect

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RMS_GATE = 1e-3


def full_check(oracle, scene, renderer, name):
    """GPU vs oracle, same seed: every path takes the same segments (ray counts per depth equal), every u8 of the image is
    equal, and the float means agree to the rounding of a different product order (beta is multiplied left to right on the
    device, right to left by the reference's recursion).  No pixel is tolerated: the libm-class functions are glibc's
    (fw_libm.h) and the rays whose result depends on traversal order take the reference's own walk (k_extend_exact)."""
    gpu = renderer.render_full(scene)
    cpu = oracle.render(scene, renderer)
    g, c = np.nan_to_num(gpu.gamma.astype(np.float64)), np.nan_to_num(cpu.gamma.astype(np.float64))
    rms = float(np.sqrt(np.mean((g - c) ** 2)))
    scale = np.maximum(np.abs(np.nan_to_num(cpu.linear)), 1e-3)
    bad = int((np.abs(np.nan_to_num(gpu.linear) - np.nan_to_num(cpu.linear)) > 2e-5 * scale + 1e-7).any(axis=1).sum())
    d8 = int((gpu.rgb8 != cpu.rgb8).sum())
    s = renderer.settings
    print(f"{name} {s['width']}x{s['height']}@{s['samples']}: rms={rms:.3e} pixels beyond float noise={bad}/{g.shape[0]} u8_diffs={d8} "
          f"rays gpu={gpu.stats['rays']} cpu={cpu.stats['rays']}")
    assert rms <= RMS_GATE
    assert gpu.stats["rays_per_depth"] == cpu.stats["rays_per_depth"], name
    assert bad == 0 and d8 == 0, (name, bad, d8)
    return gpu, cpu


def test_c2_cornell_full_resolution(oracle):
    s, r = scenes.config("C2_cornell_box", samples=64)             # 512x512
    full_check(oracle, s, r, "C2")


def test_c1_random_spheres_full_config(oracle):
    s, r = scenes.config("C1_random_spheres")                       # 400x225 @64, the whole config
    full_check(oracle, s, r, "C1")


def test_c3_suzanne_full_resolution(oracle):
    s, r = scenes.config("C3_suzanne", samples=16)                  # 1280x720
    full_check(oracle, s, r, "C3")


def test_c4a_hdri_full_resolution_with_the_configs_4k_map(oracle):
    s, r = scenes.config("C4a_hdri_test", samples=16)               # 1024x1024, synthetic 4096x2048 f32 map (100 MB)
    assert s.environment.pixels.shape == (2048, 4096, 3)
    full_check(oracle, s, r, "C4a")


def test_c4b_volume_full_resolution(oracle):
    s, r = scenes.config("C4b_volume_test", samples=16)             # 1024x1024
    full_check(oracle, s, r, "C4b")


def test_c5_part2_full_resolution(oracle):
    s, r = scenes.config("C5_part2_all", samples=4)                 # 1920x1080
    full_check(oracle, s, r, "C5")


def test_teapot_yml_meshes_on_the_hip_path(oracle):
    """scenes/teapot.yml's four meshes (6 320 triangles with vertex normals: the smooth-normal branch mesh.rs:206-207)
    at a quarter of the example's resolution, with and without the TLAS."""
    for bvh in (True, False):
        s, r = scenes.teapot()
        r.width(480).height(270).samples(8).use_bvh(bvh)
        full_check(oracle, s, r, f"teapot bvh={bvh}")


def test_44k_triangle_mesh(oracle):
    n = 150
    xs = np.linspace(-4, 4, n, dtype=np.float32)
    X, Z = np.meshgrid(xs, xs, indexing="ij")
    Y = (0.4 * np.sin(2 * X) * np.cos(2 * Z)).astype(np.float32)
    verts = np.stack([X, Y, Z], -1).reshape(-1, 3)
    i, j = np.meshgrid(np.arange(n - 1), np.arange(n - 1), indexing="ij")
    a = (i * n + j).reshape(-1); b = a + 1; c = a + n; d = c + 1
    idx = np.stack([a, b, c, b, d, c], -1).reshape(-1).astype(np.uint32)
    sc = Scene.new()
    m = sc.add_material(LambertianMat.with_color((0.7, 0.6, 0.5)))
    mesh = TriangleMesh.new(verts, idx, None, None, m)
    assert mesh.num_tris() == 44402
    sc.add_object(RenderObject.new(mesh).position(0.0, 1.0, 0.0))
    sc.add_object(RenderObject.new(XZRect.new(-20.0, 20.0, -20.0, 20.0, -0.5, m)))
    sc.set_environment(SkyEnv.default())
    cam = CameraSettings.default().cam_pos((0.0, 6.0, -12.0)).look_at((0.0, 1.0, 0.0)).field_of_view(40.0)
    r = Renderer.default().width(160).height(90).samples(4).use_bvh(True).camera(cam)
    gpu, cpu = full_check(oracle, sc, r, "44k mesh")
    assert gpu.stats["parked_rays"] > 0                             # the mesh rays went through k_blas


def test_instanced_mesh_shares_one_blas(oracle):
    """Several objects that reference the same TriangleMesh shape: flattened and built once (fw_runtime.cpp caches the
    shape's parameters), rendered like separate copies."""
    s1, r = scenes.suzanne()
    mesh = s1.render_objects[0].obj
    sc = Scene.new()
    m = sc.add_material(LambertianMat.with_color((0.8, 0.3, 0.3)))
    mesh.material = m
    for k in range(4):
        sc.add_object(RenderObject.new(mesh).position(-4.5 + 3.0 * k, 0.0, 0.0))
    sc.add_object(RenderObject.new(XZRect.new(-50.0, 50.0, -50.0, 50.0, -1.0, m)))
    sc.set_environment(SkyEnv.default())
    cam = CameraSettings.default().cam_pos((0.0, 3.0, 12.0)).look_at((0.0, 0.0, 0.0)).field_of_view(40.0)
    r = Renderer.default().width(160).height(90).samples(4).use_bvh(True).camera(cam)
    gpu, cpu = full_check(oracle, sc, r, "instanced")
    assert gpu.stats["blas_nodes"] == 1023                          # one suzanne BLAS (968 triangles), not four


def test_conics_and_earth_examples(oracle):
    for name, build in (("conics", scenes.conics), ("earth", scenes.earth)):
        s, r = build()
        r.width(r.settings["width"] // 2).height(r.settings["height"] // 2).samples(8)
        full_check(oracle, s, r, name)


# ---- bench.py: the multi-rank launch and the collective ------------------------------------------------------------
def _bench(args, timeout=600):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


SMALL = ["--width", "128", "--height", "96", "--spp", "16", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-one-shot"]


def test_eight_way_tiled_frames_equal_the_one_device_frames():
    """The shares of an 8-rank cornell frame (32 Ki pixels each: pixel-major deposit bitmap, box lists, 4-byte hits) and of an
    8-rank part2 frame through fw_render_scene_tiled with device 0 listed eight times: the assembled full-size frames and
    their counters equal one render's."""
    for name, spp in (("C2_cornell_box", 24), ("C5_part2_all", 2)):      # C5: BASELINE's "tiled across 8", 1920x1080, TLAS in LDS
        s, r = scenes.config(name, samples=spp)
        one = r.render_full(s)
        tiled = _lib.render_scene_tiled(s.to_desc(), r, [0] * 8)
        assert np.array_equal(one.linear, tiled.linear) and np.array_equal(one.rgb8, tiled.rgb8), name
        assert one.stats["rays_per_depth"] == tiled.stats["rays_per_depth"] and one.stats["samples"] == tiled.stats["samples"], name


def test_bench_launches_its_own_ranks_and_the_frame_does_not_depend_on_them(tmp_path):
    """`python bench.py --gpus N` (no launcher): N ranks share this box's one GPU over gloo — the rehearsal mode of the
    N>1 path — and the assembled frame is bit-identical to the 1-rank frame."""
    f1, f3 = str(tmp_path / "f1.npy"), str(tmp_path / "f3.npy")
    a = _bench(SMALL + ["--dump-frame", f1])
    b = _bench(SMALL + ["--gpus", "3", "--backend", "gloo", "--dump-frame", f3])
    assert a["n_gpus"] == 1 and b["n_gpus"] == 3 and "world_size 3" in b["config"]["collective"]
    assert np.array_equal(np.load(f1), np.load(f3))
    assert b["value"] > 0 and b["ms_gather_per_step"] > 0


def test_bench_rccl_collective_runs_on_one_gpu(tmp_path):
    """backend nccl (= RCCL) with the gather forced at world size 1: the collective call itself executes on this GPU."""
    f1, f2 = str(tmp_path / "f1.npy"), str(tmp_path / "f2.npy")
    a = _bench(SMALL + ["--dump-frame", f1])
    b = _bench(SMALL + ["--force-collective", "--dump-frame", f2])
    assert "nccl, world_size 1" in b["config"]["collective"] and a["config"]["collective"] == "none"
    assert np.array_equal(np.load(f1), np.load(f2))


def test_bench_line_carries_roofline_parity_and_one_shot():
    d = _bench(["--width", "96", "--height", "96", "--spp", "32", "--steps", "2", "--warmup", "1", "--parity-seconds", "5"])
    assert d["unit"] == "Mrays/s" and d["scaling"] == "strong" and d["vs_baseline"] is None
    r = d["roofline"]
    # `roofline` names the kernel class with the larger exclusive time; the other class rides along, and no bound is ever null
    assert 0 < r["frac"] < 1 and r["peak"] == 8000.0
    if r["kernel"] == "k_shade":
        assert r["bound"] == "hbm" and r["other_kernel"]["kernel"] == "k_extend"
        assert r["other_kernel"]["bound"] in ("latency", "valu_issue", "hbm")
    else:
        assert r["kernel"].startswith("k_extend") and r["bound"] in ("latency", "valu_issue", "hbm")
        assert r["other_kernel"]["kernel"] == "k_shade" and r["other_kernel"]["bound"] == "hbm"
    assert d["parity"]["pass"] and d["parity"]["rays_equal"] and d["parity"]["u8_diffs"] == 0 and d["parity"]["timed_frame_equals_checked_frame"]
    o = d["one_shot"]
    assert o["ms_wall"] >= o["ms_scene"] + o["ms_render"] > 0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    lay, sur = d["roofline_frame"]["layout"], d["roofline_frame"]["survey"]
    assert 0 < lay["bytes"] < sur["bytes"]
    # round 3: the timed loop runs the library's own schedule, per-kernel times come from a labelled exclusive pass
    sch = d["schedule"]
    assert sch["timed_loop_ms_per_step"] == d["ms_per_step"] and sch["exclusive_pass_ms_per_step"] > 0 and "STREAMS=1" in sch["exclusive_pass"]
    assert d["device"]["copy_GBps"] > 100 and d["device"]["cus"] >= 1
    cold = d["one_shot_cold"]
    assert cold["ms_wall"] > o["ms_wall"] and cold["ms_scene"] >= 0 and cold["ms_render"] > 0       # a process's first call pays for pools and code objects


def test_stats_carry_the_layouts_own_bytes_and_the_deposits_really_written():
    """fw_stats (ABI v4): bytes per kernel class follow from the per-depth ray counts by the layout's record sizes
    (fw_device.h B_*), and FW_FLAG_COUNT_DEPOSITS counts the radiance records k_shade wrote — fewer than the samples over a
    black environment, where zeros are elided — independently of what an earlier, bigger render left in the workspace."""
    s, r = scenes.config("C2_cornell_box", 96, 96, 32)
    big = scenes.config("C2_cornell_box", 160, 160, 16)
    big[1].render_full(big[0])                                        # leaves non-zero records beyond this frame's slots
    a = r.count_deposits(True).render_full(s).stats
    R, S = a["rays_per_depth"], a["samples"]
    assert 0 < a["deposits"] < S                                      # most cornell paths end black
    later, ray0 = sum(R[1:]), 16                                      # pinhole camera: segment-0 rays are 16 B
    rd_ray = R[0] * ray0 + later * 24
    assert a["bytes_extend"] == rd_ray + 4 * a["rays"]                # 4-byte hit records: spheres, rects and boxes only (hit4)
    # chain state (round 4): cornell's attenuations are constants of its materials: 8 bytes of state per path, in and out
    assert a["bytes_shade"] == rd_ray + later * 8 + 4 * a["rays"] + later * (24 + 8) + a["deposits"] * 16
    assert a["bytes_raygen"] == S * ray0 + S * 4                      # rays + the tile-order ids (no zero records: dep_bits)
    assert a["n_batches"] == 2                                        # a box-list scene: two batches in flight (round 3)
    assert a["bytes_accumulate"] == a["deposits"] * 16 + S // 8 + a["n_batches"] * 96 * 96 * 2 * 16   # records with a set bit, the bits, the accumulator per batch
    b = r.render_full(s).stats                                        # again, same workspace: same count
    assert b["deposits"] == a["deposits"]
    c = r.count_deposits(False).render_full(s).stats
    assert c["deposits"] == S and c["bytes_shade"] > a["bytes_shade"]  # without the count every terminated path is assumed to write
    # a sky environment: nothing is elided
    s2, r2 = scenes.config("C4b_volume_test", 64, 64, 8)
    d = r2.count_deposits(True).render_full(s2).stats
    assert d["deposits"] == d["samples"]
    R2 = d["rays_per_depth"]                                          # a medium: 8-byte hit records (t is drawn, not recomputable)
    assert d["bytes_extend"] >= R2[0] * 16 + sum(R2[1:]) * 24 + 8 * d["rays"]
    # parked rays are counted where a mesh is walked by k_blas
    s3, r3 = scenes.config("C3_suzanne", 96, 54, 8)
    e = r3.render_full(s3).stats
    assert 0 < e["parked_rays"] < e["rays"] and e["bytes_extend"] > 32 * e["rays"] - 8 * e["samples"]
    assert e["ms_wall"] >= e["ms_render"] > 0 and e["ms_d2h"] >= 0
