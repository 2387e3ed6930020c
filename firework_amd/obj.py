"""Wavefront OBJ loader with tobj 1.0's default behaviour (the loader the reference's examples use:
examples/suzanne.rs:19, teapot.rs).  Verified against the reference's own serialised mesh: loading
suzanne.obj reproduces scenes/suzanne.yml exactly (1966 vertices, 2904 indices) — SURVEY §8f.3.

Rules: one model per `o`/`g` group; a new vertex is emitted the first time each (v, vt, vn) index triple is
seen, in file order; polygons are fan-triangulated (c0, ck, ck+1); normals / texcoords are kept only if the
file has them for the model."""
import numpy as np


class ObjModel:
    def __init__(self, name):
        self.name = name
        self.positions, self.normals, self.texcoords, self.indices = [], [], [], []


def load_obj(path):
    v, vt, vn = [], [], []
    models, cur, seen = [], None, {}

    def start(name):
        nonlocal cur, seen
        cur, seen = ObjModel(name), {}
        models.append(cur)

    with open(path, "r", errors="replace") as f:
        for raw in f:
            line = raw.strip()
            if not line or line[0] == "#":
                continue
            parts = line.split()
            tag = parts[0]
            if tag == "v":
                v.append(tuple(float(x) for x in parts[1:4]))
            elif tag == "vt":
                vt.append(tuple(float(x) for x in parts[1:3]))
            elif tag == "vn":
                vn.append(tuple(float(x) for x in parts[1:4]))
            elif tag in ("o", "g"):
                name = parts[1] if len(parts) > 1 else "unnamed_object"
                if cur is not None and not cur.indices:
                    models.pop()              # an empty group does not produce a model
                start(name)
            elif tag == "f":
                if cur is None:
                    start("unnamed_object")
                corners = []
                for tok in parts[1:]:
                    fields = tok.split("/")
                    iv = int(fields[0])
                    it = int(fields[1]) if len(fields) > 1 and fields[1] else 0
                    inn = int(fields[2]) if len(fields) > 2 and fields[2] else 0
                    # OBJ indices are 1-based; negative = relative to the end
                    iv = iv - 1 if iv > 0 else len(v) + iv
                    it = (it - 1 if it > 0 else len(vt) + it) if it else -1
                    inn = (inn - 1 if inn > 0 else len(vn) + inn) if inn else -1
                    key = (iv, it, inn)
                    idx = seen.get(key)
                    if idx is None:
                        idx = len(cur.positions)
                        seen[key] = idx
                        cur.positions.append(v[iv])
                        if it >= 0:
                            cur.texcoords.append(vt[it])
                        if inn >= 0:
                            cur.normals.append(vn[inn])
                    corners.append(idx)
                for k in range(1, len(corners) - 1):
                    cur.indices += [corners[0], corners[k], corners[k + 1]]
    out = []
    for m in models:
        if not m.indices:
            continue
        pos = np.asarray(m.positions, np.float32).reshape(-1, 3)
        nrm = np.asarray(m.normals, np.float32).reshape(-1, 3) if len(m.normals) == len(m.positions) and m.normals else None
        tex = np.asarray(m.texcoords, np.float32).reshape(-1, 2) if len(m.texcoords) == len(m.positions) and m.texcoords else None
        out.append(dict(name=m.name, positions=pos, indices=np.asarray(m.indices, np.uint32), normals=nrm, texcoords=tex))
    return out


def add_obj(scene, file_name, material, with_normals=False):
    """examples/suzanne.rs:15-51 `add_obj` (normals dropped, as there) / examples/teapot.rs (normals kept)."""
    from .api import RenderObject, TriangleMesh
    ids = []
    for m in load_obj(file_name):
        mesh = TriangleMesh.new(m["positions"], m["indices"], m["normals"] if with_normals else None, None, material)
        ids.append(scene.add_object(RenderObject.new(mesh)))
    return ids
