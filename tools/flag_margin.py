"""The margin of the rules that decide which rays leave the fast walks' ordinary regime:

    python tools/flag_margin.py [--out profiles/r04_flag_margin.json]

For C3 (suzanne 1280x720 @16) and C5 (part2 1920x1080 @4) the number of pixels whose ray count differs from the oracle's, with
  * the exact list's shear threshold at 2^-8 .. 2^-12 (EXACT_SHEAR_LOG2), with and without the SOFT class (SOFT_SHEAR_LOG2 = 5 / 0),
  * the far rule at 256 .. 4096 x the smallest object (EXACT_FAR_X),
  * no exact walk at all (NO_EXACT), and every ray through it (EXACT_ALL=1, C3 at its full resolution: the renderer then IS bvh.rs:115-151).
Options are set through fw_set_option (the library reads the environment only when it is loaded).  The oracle is the checker."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["FIREWORK_NO_ZERO_SKIP"] = "1"        # every path deposits its length: accum.w = rays of the pixel

import numpy as np  # noqa: E402

from firework_amd import _lib, scenes  # noqa: E402
from oracle import oracle_binding as ob  # noqa: E402


def gpu_counts(sd, renderer):
    ds = _lib.DeviceScene(sd)                    # the flag rule is fixed when the scene is created
    n = renderer.settings["width"] * renderer.settings["height"]
    accum = np.zeros((n, 4), np.float32)
    t0 = time.perf_counter()
    res = ds.render_progressive(renderer, 0, accum, None)
    dt = time.perf_counter() - t0
    ds.close()
    return accum[:, 3].astype(np.int64), res, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    report = {}
    for name, spp in (("C3_suzanne", 16), ("C5_part2_all", 4)):
        scene, renderer = scenes.config(name, samples=spp)
        sd = scene.to_desc()
        ocnt = ob.render_counts(sd, renderer).astype(np.int64)
        rows = []

        def row(label, **opts):
            for k, v in opts.items():
                _lib.set_option(k, v)
            try:
                g, res, dt = gpu_counts(sd, renderer)
            finally:
                for k in opts:
                    _lib.set_option(k, None)
            r = dict(setting=label, pixels_with_other_ray_count=int((g != ocnt).sum()), rays_gpu=int(g.sum()), rays_oracle=int(ocnt.sum()), ms=round(dt * 1e3, 1))
            rows.append(r)
            print(name, json.dumps(r), flush=True)

        row("default (exact list 2^-10, far 1024 x, SOFT 2^-5)")
        for e in (8, 9, 10, 11, 12):
            row(f"exact shear 2^-{e}, SOFT on", EXACT_SHEAR_LOG2=str(e))
            row(f"exact shear 2^-{e}, SOFT off", EXACT_SHEAR_LOG2=str(e), SOFT_SHEAR_LOG2="0")
        for fx in (256, 512, 1024, 2048, 4096):
            row(f"far rule {fx} x", EXACT_FAR_X=str(fx))
        row("no exact walk (NO_EXACT), SOFT on", NO_EXACT="1")
        row("no exact walk, SOFT off", NO_EXACT="1", SOFT_SHEAR_LOG2="0")
        report[name] = dict(width=renderer.settings["width"], height=renderer.settings["height"], spp=spp, rows=rows)
    # every ray through the literal walk against the default, C3 at its full resolution
    scene, renderer = scenes.config("C3_suzanne", samples=4)
    fast = renderer.render_full(scene)
    _lib.set_option("EXACT_ALL", "1")
    try:
        t0 = time.perf_counter()
        exact = renderer.render_full(scene)
        dt = time.perf_counter() - t0
    finally:
        _lib.set_option("EXACT_ALL", None)
    cpu = ob.render(scene, renderer)
    report["exact_all_vs_default_C3_1280x720@4"] = dict(
        rays_per_depth_equal=bool(list(fast.stats["rays_per_depth"]) == list(exact.stats["rays_per_depth"]) == list(cpu.stats["rays_per_depth"])),
        linear_identical=bool(np.array_equal(fast.linear, exact.linear, equal_nan=True)), u8_equal_to_oracle=bool(np.array_equal(exact.rgb8, cpu.rgb8)),
        exact_all_ms=round(float(exact.stats["ms_render"]), 1), default_ms=round(float(fast.stats["ms_render"]), 1), exact_all_call_s=round(dt, 2))
    print(json.dumps(report["exact_all_vs_default_C3_1280x720@4"]), flush=True)
    if a.out:
        json.dump(report, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
