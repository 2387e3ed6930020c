"""Host logic that needs no GPU: bench.py's launcher, the profile/source matching, scene content hashes, shape sharing."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from firework_amd import scenes  # noqa: E402
from firework_amd.api import LambertianMat, RenderObject, Scene, SkyEnv  # noqa: E402


def test_bench_self_launch_starts_one_rank_per_gpu_without_touching_the_gpu(monkeypatch):
    """`python bench.py --gpus 4` with no launcher: the parent only spawns `torch.distributed.run` (it never imports
    torch itself, so it cannot have initialised HIP before the children start) and returns their exit code."""
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1"])
    had_torch = "torch" in sys.modules
    rc = bench.self_launch(bench.parse_args(["--gpus", "4", "--steps", "2", "--warmup", "1"]))
    assert rc == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert ("torch" in sys.modules) == had_torch


def test_kernel_source_sha_tracks_the_kernel_sources():
    a = bench.kernel_source_sha()
    assert len(a) == 16 and a == bench.kernel_source_sha()
    import hashlib
    h = hashlib.sha256()
    for rel in bench.KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    assert a == h.hexdigest()[:16]


def test_scene_content_hash():
    a = scenes.cornell_box()[0].to_desc().content_hash()
    assert a == scenes.cornell_box()[0].to_desc().content_hash()          # stable across rebuilds (no pointers inside)
    assert a != scenes.volume_test()[0].to_desc().content_hash()
    s, _ = scenes.suzanne()
    h0 = s.to_desc().content_hash()
    s.render_objects[0].obj.verts[0, 0] += 1e-3                            # one vertex moves
    assert s.to_desc().content_hash() != h0
    s2, _ = scenes.cornell_box()
    s2.render_objects[3].position(0.0, 0.5, 0.0)
    assert s2.to_desc().content_hash() != a


def test_objects_that_share_a_shape_share_its_record():
    s, _ = scenes.suzanne()
    mesh = s.render_objects[0].obj
    sc = Scene.new()
    m = sc.add_material(LambertianMat.with_color((0.5, 0.5, 0.5)))
    mesh.material = m
    for k in range(3):
        sc.add_object(RenderObject.new(mesh).position(float(k), 0.0, 0.0))
    sc.set_environment(SkyEnv.default())
    d = sc.to_desc()
    assert d.desc.n_objects == 3 and d.desc.n_shapes == 1
    assert [d.objects[i].shape for i in range(3)] == [0, 0, 0]


def test_python_and_cpp_rotor_products_agree_bit_for_bit(tmp_path):
    """firework_amd/api.py emulates the f32 fma chain of the rotor product (round-to-odd in f64, then one rounding to f32);
    include/firework.hpp uses std::fma.  10 000 random rotor pairs must give the same bits in both hosts."""
    import subprocess
    import numpy as np
    from firework_amd.api import Rotor3
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "rotor.cpp"
    src.write_text('#include "firework.hpp"\n#include <cstdio>\nint main(int c, char **v) { FILE *f = fopen(v[1], "rb"), *o = fopen(v[2], "wb"); float a[8];\n'
                   '  while (fread(a, 4, 8, f) == 8) { firework::Rotor3 x{a[0], a[1], a[2], a[3]}, y{a[4], a[5], a[6], a[7]}; firework::Rotor3 r = x * y; float q[4] = {r.s, r.xy, r.xz, r.yz}; fwrite(q, 4, 4, o); }\n'
                   '  fclose(f); fclose(o); return 0; }\n')
    exe = tmp_path / "rotor"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(root, "include"), "-o", str(exe), str(src)])
    rng = np.random.default_rng(11)
    ab = (rng.standard_normal((10000, 8)) * np.exp(rng.uniform(-3, 3, (10000, 8)))).astype(np.float32)
    (tmp_path / "in.bin").write_bytes(ab.tobytes())
    subprocess.check_call([str(exe), str(tmp_path / "in.bin"), str(tmp_path / "out.bin")])
    cpp = np.frombuffer((tmp_path / "out.bin").read_bytes(), np.float32).reshape(-1, 4)
    py = np.array([[r.s, r.xy, r.xz, r.yz] for r in (Rotor3(*map(float, p[:4])) * Rotor3(*map(float, p[4:])) for p in ab)], np.float32)
    assert np.array_equal(py.view(np.uint32), cpp.view(np.uint32))
