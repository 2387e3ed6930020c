#!/usr/bin/env python3
"""Golden vectors from the build's own CPU oracle (the reference cannot run, so goldens come from
the restatement): per BASELINE scene a small counter-RNG render — linear float image, u8 image, rays per depth.
tests/test_goldens.py pins the oracle to them (regression) and the GPU tests compare the HIP path to the same files."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from firework_amd import scenes  # noqa: E402
from oracle import oracle_binding as ob  # noqa: E402

CASES = {  # name: (config, width, height, spp)
    "C1_random_spheres": (40, 22, 8), "C2_cornell_box": (32, 32, 16), "C3_suzanne": (40, 22, 8),
    "C4b_volume_test": (32, 32, 8), "C5_part2_all": (40, 22, 4),
}


def hdri_small():
    s, r = scenes.hdri_test(scenes.synthetic_hdr(256, 128))
    return s, r.width(32).height(32).samples(8)


def build(name):
    if name == "C4a_hdri_test":
        return hdri_small()
    w, h, spp = CASES[name]
    return scenes.config(name, w, h, spp)


def main():
    out = os.path.join(ROOT, "tests", "golden")
    for name in list(CASES) + ["C4a_hdri_test"]:
        s, r = build(name)
        res = ob.render(s, r, n_threads=0)
        np.savez_compressed(os.path.join(out, f"oracle_{name}.npz"), linear=res.linear, rgb8=res.rgb8,
                            rays_per_depth=np.array(res.stats["rays_per_depth"], np.uint64),
                            width=r.settings["width"], height=r.settings["height"], samples=r.settings["samples"],
                            use_bvh=int(r.settings["use_bvh"]))
        print(name, res.linear.shape, res.stats["rays"])


if __name__ == "__main__":
    main()
