#!/bin/bash
# how many rays are flagged for the exact walk, per config and per rule
for cfg in "C3_suzanne 1280 720 16" "C5_part2_all 1920 1080 4" "C1_random_spheres 400 225 64" "teapot 1920 1080 4"; do
  set -- $cfg
  FIREWORK_TRACE=1 python3 - "$@" <<'PY' 2>&1 | grep -E "left the wavefront|rays" | head -4
import sys; sys.path.insert(0, '.')
from firework_amd import scenes
name, w, h, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
s, r = scenes.config(name, w, h, spp)
res = r.render_full(s)
print(name, "rays", res.stats["rays"], "ms", round(res.stats["ms_render"],2))
PY
done
