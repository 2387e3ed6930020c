"""Where the walks and k_shade lose their lanes, section by section (round 5).  Needs a -DFW_PHASE_STATS build:
    tools/build_variant.sh phase -DFW_PHASE_STATS
    FIREWORK_LIB=firework_amd/lib/variants/lib_phase.so FIREWORK_STREAMS=1 python tools/phase_stats.py C3_suzanne:64 C5_part2_all:16 ...
The wide walks (k_blas_wide / k_extend_tlas_wide) and k_shade keep separate tables.
wave cycles = time the wave spent in the section; lanes = lane-cycles / (64 x wave cycles)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from firework_amd import scenes, _lib

WALK_BLAS = {0: "refill: start_walk", 1: "node step 1", 2: "node step 2", 3: "triangle put aside", 4: "triangle held", 5: "gate check",
             6: "finish / next mesh", 7: "[busy lanes per round]", 18: "kernel span"}
WALK_TLAS = {0: "prep_block", 1: "node step", 3: "object test", 6: "finish", 7: "[busy lanes per round]", 8: "[sphere tests]", 9: "[Rect3d tests]",
             10: "[medium tests]", 11: "[other tests]", 18: "kernel span"}
SHADE = {0: "chunk body (shade_path)", 1: "miss", 2: "object + material + rebuild_hit", 3: "texture_sample", 4: "emissive: carried()",
         5: "scatter (incl. rejection loop)", 6: "deposit", 7: "compact", 8: "[Lambertian]", 9: "[Metal]", 10: "[Dielectric]", 12: "[Isotropic]",
         18: "kernel span", 19: "[rejection attempts]"}


def table(vals, names, rays, title):
    print(f"  -- {title}")
    span = vals[3 * 18 + 1] or 1
    for k in sorted(names):
        lane, wave, cnt = vals[3 * k], vals[3 * k + 1], vals[3 * k + 2]
        if not (lane or wave or cnt):
            continue
        if names[k].startswith("["):
            print(f"     {names[k]:34s} lanes {lane / rays:8.3f} per ray, {lane / max(cnt, 1):6.2f} per entry ({lane / max(cnt, 1) / 64:5.1%}), entries per 64 rays {64 * cnt / rays:7.2f}")
        else:
            print(f"     {names[k]:34s} {wave / span:6.1%} of the span, lanes {lane / (64 * max(wave, 1)):5.1%}, entries per 64 rays {64 * cnt / rays:7.2f}, cycles per entry {wave / max(cnt, 1):7.0f}")


def main():
    lib = C.CDLL(os.environ["FIREWORK_LIB"])
    n = 120
    out = (C.c_ulonglong * n)()
    for spec in sys.argv[1:]:
        cfg, spp = spec.split(":")
        scene, renderer = scenes.config(cfg, None, None, int(spp))
        ds = _lib.DeviceScene(scene.to_desc(), 0)
        renderer.time_kernels(True)
        ds.render(renderer)                       # warm
        lib.fw_debug_phase_stats(out, n)
        st = ds.render(renderer).stats
        assert lib.fw_debug_phase_stats(out, n) == n
        vals = list(out)
        rays = st["rays"]
        print(f"{cfg} @{spp}: {rays} rays, {st['parked_rays']} parked, extend {st['ms_extend']:.2f} ms, shade {st['ms_shade']:.2f} ms (instrumented build)")
        if any(vals[:60]):
            table(vals[:60], WALK_BLAS if st["parked_rays"] else WALK_TLAS, st["parked_rays"] or rays, "k_blas_wide (per parked ray)" if st["parked_rays"] else "k_extend_tlas_wide")
        table(vals[60:], SHADE, rays, "k_shade")
        ds.close()


if __name__ == "__main__":
    main()
