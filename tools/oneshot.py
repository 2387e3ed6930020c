"""`Renderer::render(scene)` as the reference times it (main.rs:40-44: scene conversion + BVH build + render), host buffers out.
FIREWORK_TRACE=1 prints where each scene creation spends its time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from firework_amd import scenes
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for name in ("C2_cornell_box", "C3_suzanne", "C4b_volume_test"):
    s, r = scenes.config(name)
    sd = s.to_desc()
    for rep in range(reps):
        t0 = time.perf_counter(); res = r.render_full(sd); dt = time.perf_counter() - t0
        st = res.stats
        print(f"{name} rep{rep}: wall {dt*1e3:.1f} ms (library {st['ms_wall']:.1f}) = scene {st['ms_scene']:.1f} + device {st['ms_render']:.1f} + d2h {st['ms_d2h']:.2f} + rest {st['ms_wall'] - st['ms_scene'] - st['ms_render'] - st['ms_d2h']:.1f}; {st['rays']/dt/1e6:.0f} Mrays/s wall-inclusive", flush=True)
