"""OBJ loader vs the reference's own artefacts: tobj 1.0's output for suzanne.obj is serialised in the
reference's scenes/suzanne.yml (kept here as scenes/suzanne_mesh.npz), and teapot.obj -> scenes/teapot.yml."""
import os

import numpy as np
import pytest

from firework_amd.obj import load_obj

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"


def test_suzanne_obj_reproduces_the_reference_yaml_mesh():
    models = load_obj(os.path.join(HERE, "golden", "suzanne.obj"))
    assert len(models) == 1
    m = models[0]
    d = np.load(os.path.join(ROOT, "scenes", "suzanne_mesh.npz"))
    assert m["positions"].shape == (1966, 3) and m["indices"].shape == (2904,)
    assert np.array_equal(m["positions"], d["verts"])          # f32-equal
    assert np.array_equal(m["indices"], d["indicies"])


@pytest.mark.skipif(not os.path.exists(f"{REF}/teapot.obj"), reason="reference tree not present")
def test_teapot_obj_matches_reference_teapot_yml():
    from firework_amd.yaml_io import load_scene
    models = load_obj(f"{REF}/teapot.obj")                      # CRLF file, 4 groups, vn normals
    sc = load_scene(f"{REF}/scenes/teapot.yml")
    meshes = [ro.obj for ro in sc.render_objects if type(ro.obj).__name__ == "TriangleMesh"]
    assert len(models) == len(meshes) == 4
    for m, ref in zip(models, meshes):
        assert np.array_equal(m["indices"], ref.indicies)
        assert np.array_equal(m["positions"], ref.verts)
        assert m["normals"] is not None and np.array_equal(m["normals"], ref.normals)
