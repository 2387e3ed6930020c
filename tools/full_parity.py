"""Every BASELINE config at its FULL sample count, GPU against the CPU oracle at the same seed: one JSON line per config with the
parity object (rays per depth on both sides, differing u8 values, RMS) and the oracle's seconds.

    python tools/full_parity.py [--out gpurun_out/full_parity.jsonl] [C3_suzanne:512 C4a_hdri_test:512 ...]

Default list: the configs of BASELINE.json at their own sample counts, C5 at 256 spp (1/16 of its 4096: ~1.8e9 rays, the
oracle needs ~20 s of a 256-core host per 1e9 rays).  A config whose frames differ also lists the differing pixels (first 64),
which is what tools/diverge.py takes from there.  The oracle is the checker (test infrastructure); the GPU frame comes through
the C ABI like every other frame.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from firework_amd import scenes  # noqa: E402
from oracle import oracle_binding as ob  # noqa: E402

DEFAULT = ["C1_random_spheres:64", "C2_cornell_box:1024", "C3_suzanne:512", "C4a_hdri_test:512", "C4b_volume_test:512", "C5_part2_all:256", "teapot:64"]


def one(name, spp, cores):
    scene, renderer = scenes.config(name, samples=spp)
    s = renderer.settings
    t0 = time.perf_counter()
    gpu = renderer.render_full(scene)
    t_gpu = time.perf_counter() - t0
    t0 = time.perf_counter()
    cpu = ob.render(scene, renderer, n_threads=cores)
    t_cpu = time.perf_counter() - t0
    g, c = np.nan_to_num(gpu.gamma.astype(np.float64)), np.nan_to_num(cpu.gamma.astype(np.float64))
    rms = float(np.sqrt(np.mean((g - c) ** 2)))
    scale = np.maximum(np.abs(np.nan_to_num(cpu.linear)), 1e-3)
    badmask = (np.abs(np.nan_to_num(gpu.linear) - np.nan_to_num(cpu.linear)) > 2e-5 * scale + 1e-7).any(axis=1)
    d8mask = (gpu.rgb8 != cpu.rgb8).any(axis=1)
    rg, rc = [int(x) for x in gpu.stats["rays_per_depth"]], [int(x) for x in cpu.stats["rays_per_depth"]]
    rec = {"config": name, "width": s["width"], "height": s["height"], "spp": spp, "use_bvh": bool(s["use_bvh"]),
           "rays_equal": rg == rc, "rays_gpu": int(gpu.stats["rays"]), "rays_oracle": int(cpu.stats["rays"]),
           "rays_per_depth_gpu": rg, "rays_per_depth_oracle": rc,
           "u8_diffs": int((gpu.rgb8 != cpu.rgb8).sum()), "u8_values": int(cpu.rgb8.size),
           "pixels_beyond_float_noise": int(badmask.sum()), "rms_gamma": rms, "gate": 1e-3, "pass": bool(rms <= 1e-3),
           "oracle_seconds": round(t_cpu, 1), "oracle_cores": cores, "gpu_call_seconds": round(t_gpu, 2), "gpu_ms_render": round(float(gpu.stats["ms_render"]), 2)}
    if rec["u8_diffs"] or rec["pixels_beyond_float_noise"] or not rec["rays_equal"]:
        rec["differing_pixels"] = [int(x) for x in np.nonzero(badmask | d8mask)[0][:64]]
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="*", default=DEFAULT)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    cores = os.cpu_count() or 1
    ok = True
    for case in a.cases:
        name, spp = case.split(":")
        rec = one(name, int(spp), cores)
        line = json.dumps(rec)
        print(line, flush=True)
        if a.out:
            with open(a.out, "a") as f:
                f.write(line + "\n")
        ok = ok and rec["rays_equal"] and rec["u8_diffs"] == 0
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
