#!/usr/bin/env python3
"""Regenerates the DATA fixtures that come from the reference's data files (not its source code).

  scenes/suzanne_mesh.npz   <- /root/reference/scenes/suzanne.yml  (TriangleMesh verts/indices =
                               tobj 1.0's output for suzanne.obj; 1966 verts, 968 triangles)
  scenes/earthmap.jpg       <- /root/reference/earthmap.jpg        (ImageTexture of part2_all)
  tests/golden/reference_png_stats.json <- block means of the reference's committed renders
                               (cornell_box.png, suzanne.png, volume.png, ...): the only
                               reference-produced outputs that exist
  tests/golden/reference_yaml_pins.json <- rotor values found in scenes/*.yml
  tests/golden/reference_png_lattice.npz <- every 2nd/4th/8th pixel of the reference's committed renders whose
                               example still matches src/ (cornell_box, suzanne, volume, conics, Earth, teapot):
                               the pixels tests/test_reference_renders.py renders with the oracle at the example's own
                               resolution and spp (pixel_ids = the same lattice)
                               + random_spheres.png: where it leaves the sky (silhouettes of the three big spheres, the horizon) and
                               every 2nd pixel of its lower rows (the checker ground)
  scenes/uvmap.png          <- /root/reference/uvmap.png           (ImageTexture of conics.rs / earth.rs)
  scenes/teapot_mesh.npz    <- /root/reference/scenes/teapot.yml   (4 TriangleMeshes with vertex normals = tobj 1.0's
                               output for teapot.obj; 6 320 triangles)

Run in the build container only (/root/reference does not exist on the GPU box)."""
import json
import os
import shutil
import sys

import numpy as np
import yaml
from PIL import Image

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    with open(f"{REF}/scenes/suzanne.yml") as f:
        y = yaml.safe_load(f)
    mesh = y["render_objects"][0]["obj"]
    assert mesh["object_type"] == "TriangleMesh"
    verts = np.array([[v["x"], v["y"], v["z"]] for v in mesh["verts"]], np.float32)
    idx = np.array(mesh["indicies"], np.uint32)
    assert mesh["normals"] is None and mesh["uvs"] is None
    np.savez_compressed(f"{ROOT}/scenes/suzanne_mesh.npz", verts=verts, indicies=idx)
    print("suzanne:", verts.shape, idx.shape)
    shutil.copyfile(f"{REF}/earthmap.jpg", f"{ROOT}/scenes/earthmap.jpg")

    def blocks(name, rows, cols):
        im = np.asarray(Image.open(f"{REF}/{name}").convert("RGB"), np.float64)
        h, w, _ = im.shape
        bh, bw = h // rows, w // cols
        out = [[[round(float(x), 2) for x in im[r * bh:(r + 1) * bh, c * bw:(c + 1) * bw].reshape(-1, 3).mean(0)]
                for c in range(cols)] for r in range(rows)]
        return dict(width=w, height=h, rows=rows, cols=cols, block_means=out,
                    mean=[round(float(x), 2) for x in im.reshape(-1, 3).mean(0)])

    stats = {
        "cornell_box.png": blocks("cornell_box.png", 6, 6),
        "suzanne.png": blocks("suzanne.png", 4, 6),
        "volume.png": blocks("volume.png", 4, 6),
        "random_spheres.png": blocks("random_spheres.png", 4, 6),
        "part2_final.png": blocks("part2_final.png", 4, 3),
    }
    with open(f"{ROOT}/tests/golden/reference_png_stats.json", "w") as f:
        json.dump(stats, f, indent=1)

    # lattice subsamples: (file, stride); pixel (y, x) with y % stride == x % stride == stride // 2
    lat = {}
    for name, stride in (("cornell_box.png", 2), ("suzanne.png", 4), ("volume.png", 4), ("conics.png", 4), ("Earth.png", 4),
                         ("teapot.png", 8)):
        im = np.asarray(Image.open(f"{REF}/{name}").convert("RGB"), np.uint8)
        key = name.split(".")[0]
        lat[key] = np.ascontiguousarray(im[stride // 2::stride, stride // 2::stride])
        lat[key + "_meta"] = np.array([im.shape[1], im.shape[0], stride], np.int32)      # width, height, stride
    # part2_final.png (600x800, examples/part2_all.rs): the 190x190 window around the TurbulenceTexture sphere, every pixel —
    # box heights and the 1000 small spheres come from tiny_rng (not in the tree), the noise pattern on the sphere does not
    im = np.asarray(Image.open(f"{REF}/part2_final.png").convert("RGB"), np.uint8)
    assert im.shape == (800, 600, 3)
    lat["part2_turbulence_window"] = np.ascontiguousarray(im[325:515, 180:370])
    lat["part2_turbulence_window_meta"] = np.array([600, 800, 180, 325, 370, 515], np.int32)     # width, height, x0, y0, x1, y1
    # ... and the 200x200 window around the MetalMat sphere (albedo .8 .8 .9, roughness 10) at (0, 1.5, 1.45): nothing random on it either
    lat["part2_metal_window"] = np.ascontiguousarray(im[496:696, 248:448])
    lat["part2_metal_window_meta"] = np.array([600, 800, 248, 496, 448, 696], np.int32)
    # random_spheres.png (960x540, examples/random_spheres.rs): the small spheres come from tiny_rng (not in the tree), the three big spheres, the
    # ground sphere and its CheckerTexture do not.  Round 5: (a) where the image leaves the sky — the silhouettes of the metal, the diffuse and
    # the glass sphere and the horizon of the ground sphere, measured as the first pixel along a scanline that differs from the sky colour of
    # its row (taken at column 940) by more than 40 grey levels in sum — and (b) every 2nd pixel of the rows 300..539, where the checker's
    # cells are several pixels wide.  tests/test_reference_renders.py fits nothing to them: it renders the oracle with a camera and compares.
    im = np.asarray(Image.open(f"{REF}/random_spheres.png").convert("RGB"), np.uint8)
    assert im.shape == (540, 960, 3)
    f = im.astype(np.float32)

    def edge_row(y, xs, thr=40):
        for x in xs:
            if np.abs(f[y, x] - f[y, 940]).sum() > thr:
                return x

    def edge_col(x, ys, thr=40):
        for y in ys:
            if np.abs(f[y, x] - f[y, 940]).sum() > thr:
                return y
    lat["random_spheres_metal_edge"] = np.array([(edge_row(y, range(930, 500, -1)), y) for y in range(110, 158, 3)] +
                                                [(x, edge_col(x, range(40, 200))) for x in range(560, 700, 8)], np.int32)      # (x, y) of the first non-sky pixel
    lat["random_spheres_diffuse_edge"] = np.array([(edge_row(y, range(5, 500)), y) for y in range(112, 158, 3)] +
                                                  [(x, edge_col(x, range(40, 200))) for x in range(362, 402, 5)], np.int32)
    lat["random_spheres_glass_edge"] = np.array([(x, edge_col(x, range(40, 200))) for x in range(455, 497, 3)], np.int32)
    lat["random_spheres_horizon"] = np.array([(x, edge_col(x, range(100, 300), 25)) for x in (5, 60, 100, 150, 200, 250, 300, 800, 850, 900, 950)], np.int32)
    lat["random_spheres_ground"] = np.ascontiguousarray(im[300:540:2, 0:960:2])
    lat["random_spheres_ground_meta"] = np.array([960, 540, 0, 300, 2], np.int32)      # width, height, x0, y0, stride
    np.savez_compressed(f"{ROOT}/tests/golden/reference_png_lattice.npz", **lat)
    shutil.copyfile(f"{REF}/uvmap.png", f"{ROOT}/scenes/uvmap.png")

    with open(f"{REF}/scenes/teapot.yml") as f:
        y = yaml.load(f, Loader=getattr(yaml, "CSafeLoader", yaml.SafeLoader))
    tp = {}
    n = 0
    for ro in y["render_objects"]:
        m = ro["obj"]
        if m["object_type"] != "TriangleMesh":
            continue
        tp[f"verts{n}"] = np.array([[v["x"], v["y"], v["z"]] for v in m["verts"]], np.float32)
        tp[f"normals{n}"] = np.array([[v["x"], v["y"], v["z"]] for v in m["normals"]], np.float32)
        tp[f"indicies{n}"] = np.array(m["indicies"], np.uint32)
        assert m["uvs"] is None
        n += 1
    tp["n_meshes"] = np.array(n, np.int32)
    np.savez_compressed(f"{ROOT}/scenes/teapot_mesh.npz", **tp)
    print("teapot:", n, "meshes,", sum(tp[f"indicies{i}"].shape[0] // 3 for i in range(n)), "triangles")

    pins = []
    for sc in ("conics.yml", "suzanne.yml", "teapot.yml"):
        with open(f"{REF}/scenes/{sc}") as f:
            yy = yaml.safe_load(f)
        for i, ro in enumerate(yy["render_objects"]):
            r = ro["rotation"]
            if r["s"] != 1.0:
                pins.append(dict(scene=sc, object=i, s=r["s"], xy=r["bv"]["xy"], xz=r["bv"]["xz"], yz=r["bv"]["yz"]))
    with open(f"{ROOT}/tests/golden/reference_yaml_pins.json", "w") as f:
        json.dump(pins, f, indent=1)
    print(len(pins), "rotor pins")


if __name__ == "__main__":
    sys.exit(main())
