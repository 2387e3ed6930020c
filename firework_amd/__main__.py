"""`python -m firework_amd` — the reference's command line (src/main.rs:6-62) on the HIP path.

    firework --scene-file <yml> -s <samples> [-n <name>] [-o <png>]

Same fixed view as main.rs: camera (0,30,50) -> (0,0,0), fov 40, 960x540, use_bvh(true); prints
`Finished Rendering in {} s`.  Without `-o` the reference opens a minifb window; a GPU node has no display,
so `-o` is required here.  Extra flags (not in the reference): --width/--height/--seed/--device, and
--progressive N (write the image after each of N passes) / --checkpoint FILE (save the accumulation buffer after every
pass and resume from it: the finished image is bit-identical to an uninterrupted render)."""
import argparse
import sys
import time


def main(argv=None):
    ap = argparse.ArgumentParser(prog="firework")
    ap.add_argument("--scene-file", required=True)
    ap.add_argument("-n", "--name", default=None)
    ap.add_argument("-s", "--samples", type=int, required=True)
    ap.add_argument("-o", "--output", default=None)
    ap.add_argument("--width", type=int, default=960)
    ap.add_argument("--height", type=int, default=540)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--progressive", type=int, default=0, metavar="N", help="render in N passes, saving the image after each")
    ap.add_argument("--checkpoint", default=None, metavar="FILE", help=".npz accumulation checkpoint: written after every pass, resumed from if present")
    opt = ap.parse_args(argv)

    from . import _lib
    from .api import CameraSettings, Renderer, save_image
    from .yaml_io import load_scene

    _lib.init(opt.device)      # fw_init: context, code objects and the path arena before the timed region, like the loading of the reference's binary (main.rs:40)

    scene = load_scene(opt.scene_file)
    camera = CameraSettings.default().cam_pos((0.0, 30.0, 50.0)).look_at((0.0, 0.0, 0.0)).field_of_view(40.0)
    renderer = (Renderer.default().width(opt.width).height(opt.height).samples(opt.samples).use_bvh(True)
                .camera(camera).seed(opt.seed))
    start = time.time()
    if opt.progressive > 0 or opt.checkpoint:
        render = None
        for k, res in enumerate(renderer.render_progressive(scene, max(1, opt.progressive), device=opt.device, checkpoint=opt.checkpoint)):
            render = res.rgb8
            if opt.output:
                save_image(render, opt.output, opt.width, opt.height)
            print(f"pass {k + 1}: {int(time.time() - start)} s")
        if render is None:          # the checkpoint already held every sample
            render = renderer.render(scene, device=opt.device)
    else:
        render = renderer.render(scene, device=opt.device)
    print(f"Finished Rendering in {int(time.time() - start)} s")
    if opt.output:
        print(f'Saving image to "{opt.output}"')
        save_image(render, opt.output, opt.width, opt.height)
    else:
        name = opt.name or "Firework Render"
        print(f"{name}: no display on this node; pass -o/--output to save the image", file=sys.stderr)
        return 2
    return 0


if __name__ == "__main__":
    sys.exit(main())
