"""`Renderer::render(scene)` as the reference times it (main.rs:40-44: scene conversion + BVH build + render), host buffers out."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from firework_amd import scenes
for name in ("C2_cornell_box", "C3_suzanne", "C4b_volume_test"):
    s, r = scenes.config(name)
    sd = s.to_desc()
    for rep in range(3):
        t0 = time.perf_counter(); res = r.render_full(sd); dt = time.perf_counter() - t0
        st = res.stats
        print(f"{name} rep{rep}: wall {dt*1e3:.1f} ms = scene {st['ms_scene']:.1f} + device {st['ms_render']:.1f} + alloc/copies/host {dt*1e3 - st['ms_scene'] - st['ms_render']:.1f}; {st['rays']/dt/1e6:.0f} Mrays/s wall-inclusive", flush=True)
