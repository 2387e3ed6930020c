"""The paths on which the HIP renderer and the CPU oracle used to part ways (found by tools/diverge.py at the start of round 3,
DESIGN.md §6), pinned one by one (-m gpu): the device traces each (pixel, sample) alone and must take the oracle's segments —
same length, same colour — and the frames they come from must have equal ray counts per depth.

  C2 cornell 512x512, pixel 112819, sample 170   a bounce ray with d.y = 0 starting ON the ceiling: t = 0/0 = NaN, which rect.rs:47-62
                                                  accepts, after which every later hit is accepted too (`t > NaN` is false); the
                                                  median form of the rectangle test rejected them (hit_rect: literal t interval now)
  C3 suzanne 1280x720, five pixels                rays whose SIGNED largest direction component is tiny (util.rs:104-118 picks it as
                                                  the shear axis of mesh.rs:147-162): shear factors of 10^4, "hits" outside the
                                                  triangle's box, found or not depending on the walk (k_extend_exact walks them)
  C5 part2 1920x1080, two pixels                  rays from ~1000 units away in the fog against spheres of radius 0.1: the discriminant is
                                                  rounding noise and the reference tests the sphere behind its DoubleLeaf's box (k_extend_exact)
  (the other eight C5 paths of that hunt started with the fog medium's log10f, ocml vs glibc by one ulp: fw_libm.h)
  Round 4 (tools/full_parity.py: every config at its FULL sample count; one path in ~4e8 still differed):
  C5 part2, pixels 1704230 / 1872136             camera rays that touch a box of the 20 x 20 grid exactly at an edge: the box's own slab interval is empty
                                                  (`tmax > tmin` fails), but the reference tests the box behind the UNION of its DoubleLeaf, which passes
  C5 part2, pixel 828574 (three samples), 845940  rays grazing a sphere of the cluster within the rounding of its discriminant: outside the sphere's own box by
                                                  1e-4, inside its DoubleLeaf's; and a grazing root that precedes the entry of its own box by 1.2e-5 (culled)
                                                  -> relaxed exit planes + obj_gate_ok / tri_gate_ok (the reference's leaf-node rule itself), cull slack 2^-10
  C3 suzanne, pixels 515109 / 528016; teapot      shear ratios 2^-7.2 and 2^-9.97, just above the exact list's 2^-10: triangle hits whose t precedes the entry
                                                  of their own box -> the SOFT class (no culling below 2^-5)
  C3 suzanne 1280x720, pixel 415420, sample 43    found later by its COST (oracle.find_nan_paths): a bounce ray with d.z = 0 exactly whose
                                                  other components are negative: util.rs:104-118 picks z, mesh.rs:160-162 divides by it,
                                                  the triangle "hit" has t = NaN, the point is NaN and the path goes on for its remaining
                                                  five segments with NaN rays that pass every box (k_extend_exact; the ordinary walks skip them)"""
import copy

import numpy as np
import pytest

from firework_amd import _lib, scenes

pytestmark = pytest.mark.gpu

CASES = [
    ("C2_cornell_box", 512, 512, [(112819, 170)]),
    ("C3_suzanne", 1280, 720, [(355384, 10), (360504, 9), (478225, 6), (525350, 14), (767410, 10), (415420, 43),
                               (515109, 499), (528016, 231)]),                        # round 4, found at 512 spp: the SOFT class
    ("C5_part2_all", 1920, 1080, [(148623, 0), (400990, 0), (139590, 3), (238497, 1), (442459, 3), (579166, 1), (628372, 0), (655626, 1), (900664, 1),
                                  (828574, 122), (828574, 172), (828574, 232), (845940, 176), (1704230, 212), (1872136, 81)]),   # round 4, found at 256 spp: leaf-node gating, cull slack
    ("teapot", 1920, 1080, [(1491249, 33)]),                                          # round 4, found at 64 spp
]


@pytest.mark.parametrize("name,w,h,paths", CASES)
def test_formerly_diverging_paths_follow_the_oracle(oracle, monkeypatch, name, w, h, paths):
    monkeypatch.setenv("FIREWORK_NO_ZERO_SKIP", "1")              # every path deposits: accum.w is its length in segments
    scene, renderer = scenes.config(name, w, h, 1)
    sd = scene.to_desc()
    ds = _lib.DeviceScene(sd)
    try:
        for pixel, sample in paths:
            segs, colour = oracle.trace_path(sd, renderer, pixel, sample)
            length = int(segs[:, 15].sum())
            accum = np.zeros((1, 4), np.float32)
            ds.render_progressive(renderer, sample, accum, np.array([pixel], np.uint32))
            assert int(accum[0, 3]) == length, (name, pixel, sample, int(accum[0, 3]), length)
            c = np.nan_to_num(colour.astype(np.float64))
            assert np.allclose(np.nan_to_num(accum[0, :3].astype(np.float64)), c, rtol=1e-5, atol=1e-7), (name, pixel, sample)
    finally:
        ds.close()


def test_ray_counts_per_depth_equal_the_oracles_on_the_frames_they_came_from(oracle):
    """the whole frames of the hunt (C3 at 16 spp, C5 at 4 spp; C2's 1024-spp frame is bench.py's own parity leg)"""
    for name, spp in (("C3_suzanne", 16), ("C5_part2_all", 4)):
        scene, renderer = scenes.config(name, samples=spp)
        gpu, cpu = renderer.render_full(scene), oracle.render(scene, renderer)
        assert gpu.stats["rays_per_depth"] == cpu.stats["rays_per_depth"], name
        assert np.array_equal(gpu.rgb8, cpu.rgb8), name


def test_every_ray_through_the_exact_walk_gives_the_same_frames(oracle, monkeypatch):
    """FIREWORK_EXACT_ALL=1: every ray is traced by the literal reference walk (k_extend_exact) — the renderer then IS bvh.rs:115-151 —
    and the frame equals the default one (fast walks + the flagged few) and the oracle's.  In both forms of the walk: one ray per
    lane (what a list of that length takes by itself) and one ray per wave with the shared walk (FIREWORK_EXACT_FORM=wave)."""
    for name, w, h, spp in (("C3_suzanne", 320, 180, 8), ("C5_part2_all", 240, 135, 4), ("C1_random_spheres", 200, 112, 8), ("teapot", 160, 90, 4)):
        scene, renderer = scenes.config(name, w, h, spp)
        monkeypatch.delenv("FIREWORK_EXACT_ALL", raising=False)
        monkeypatch.delenv("FIREWORK_EXACT_FORM", raising=False)
        fast = renderer.render_full(scene)
        cpu = oracle.render(scene, renderer)
        monkeypatch.setenv("FIREWORK_EXACT_ALL", "1")
        for form in ("lane", "wave"):
            monkeypatch.setenv("FIREWORK_EXACT_FORM", form)
            exact = renderer.render_full(scene)
            assert exact.stats["rays_per_depth"] == cpu.stats["rays_per_depth"] == fast.stats["rays_per_depth"], (name, form)
            assert np.array_equal(exact.linear, fast.linear, equal_nan=True) and np.array_equal(exact.rgb8, cpu.rgb8), (name, form)
        monkeypatch.delenv("FIREWORK_EXACT_ALL", raising=False)
        monkeypatch.delenv("FIREWORK_EXACT_FORM", raising=False)
