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
