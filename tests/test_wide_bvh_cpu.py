"""The WIDE-node builder (fw_runtime.cpp: wide_convert; four children per node, what the LDS-resident walks step through) checked
on the CPU through the diagnostic entry point fw_selftest_wide_bvh: every item the leaf of exactly one slot, every child box as the
DEVICE decodes it (f32 planes as they are; 8-bit planes through one fma) a superset of the exact one, free slots unhittable."""
import os

import numpy as np
import pytest

from firework_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F32, Q8 = 1, 2


def tri_boxes(verts, idx):
    """mesh.rs:221-242: box of the three vertices, a flat axis padded by 0.001 on both sides"""
    p = verts[idx.reshape(-1, 3)]
    lo, hi = p.min(axis=1), p.max(axis=1)
    flat = np.abs(hi - lo) < np.float32(0.001)
    lo = np.where(flat, lo - np.float32(0.001), lo).astype(np.float32)
    hi = np.where(flat, hi + np.float32(0.001), hi).astype(np.float32)
    return np.concatenate([lo, hi], axis=1)


def mesh_boxes(name):
    d = np.load(os.path.join(ROOT, "scenes", name))
    out = []
    if "verts" in d.files:
        out.append(tri_boxes(d["verts"].astype(np.float32), d["indicies"].astype(np.int64)))
    else:
        k = 0
        while f"verts{k}" in d.files:
            out.append(tri_boxes(d[f"verts{k}"].astype(np.float32), d[f"indicies{k}"].astype(np.int64)))
            k += 1
    return out


@pytest.mark.parametrize("fmt", [F32, Q8])
def test_wide_nodes_of_the_references_meshes(fmt):
    for name in ("suzanne_mesh.npz", "teapot_mesh.npz"):
        for boxes in mesh_boxes(name):
            bad, st = _lib.selftest_wide_bvh(boxes, fmt)
            assert bad == 0, (name, fmt, st)
            assert st["leaves"] == boxes.shape[0]
            assert st["nodes"] * 4 == st["leaves"] + (st["nodes"] - 1) + st["free_slots"]      # every slot is a leaf, a child node or free
            assert st["nodes"] <= (boxes.shape[0] + 1) // 2                                       # at least two children per node


@pytest.mark.parametrize("fmt", [F32, Q8])
def test_wide_nodes_of_random_and_degenerate_boxes(fmt):
    rng = np.random.default_rng(11)
    for n in (1, 2, 3, 4, 5, 9, 64, 1000, 5000):
        c = (rng.random((n, 3)) * 40 - 20).astype(np.float32)
        e = (rng.random((n, 3)) ** 4 * 3).astype(np.float32)
        e[rng.random((n, 3)) < 0.1] = 0                              # flat boxes
        boxes = np.concatenate([c - e, c + e], axis=1).astype(np.float32)
        if n >= 9:
            boxes[0] = [-5000, -5000, -5000, 5000, 5000, 5000]      # one box around everything (part2's fog)
            boxes[1, 3:] = boxes[1, :3]                              # a point
            boxes[2] = boxes[3]                                      # two equal boxes
        bad, st = _lib.selftest_wide_bvh(boxes, fmt)
        assert bad == 0, (n, fmt, st)
        assert st["leaves"] == n


def test_wide_nodes_of_a_grid_of_touching_boxes():
    """part2's 20 x 20 boxes share faces: extents that are exact multiples of the quantum"""
    xs, zs = np.meshgrid(np.arange(-10, 10), np.arange(-10, 10))
    lo = np.stack([xs.ravel(), np.zeros(400), zs.ravel()], axis=1).astype(np.float32)
    hi = lo + np.array([1.0, 0.37, 1.0], np.float32)
    for fmt in (F32, Q8):
        bad, st = _lib.selftest_wide_bvh(np.concatenate([lo, hi], axis=1), fmt)
        assert bad == 0 and st["leaves"] == 400, (fmt, st)
