"""Find the paths on which the HIP renderer and the CPU oracle part ways, and name the step at which they do.

    python tools/diverge.py C2_cornell_box 512 512 1024 [--max-pixels 6] [--out gpurun_out/diverge_C2.json]

1. Full frame on both sides with per-pixel RAY COUNTS (GPU: accum.w of a progressive render with every path depositing,
   FIREWORK_NO_ZERO_SKIP=1; oracle: fwo_render_counts).  Pixels whose counts differ hold a diverging path even when both
   ends are black.  Pixels whose linear colour differs by more than float noise are added.
2. For such a pixel, every sample alone on both sides (path length per sample) -> the diverging (pixel, sample) pairs.
3. For such a path: the oracle's segments (fwo_trace_path) next to the GPU's (FIREWORK_DUMP_PATH) -> the first segment
   whose ray or hit differs, under each runtime switch in turn (FIREWORK_NO_DEFER, FIREWORK_NO_HIT4, FIREWORK_NO_SHORT_RAYS,
   FIREWORK_BVH=median, FIREWORK_NO_LDS_TREES, FIREWORK_TLAS_REFILL=0, FIREWORK_NO_HOIST) and, when variants are built
   (firework_amd/lib/variants/lib_slowdiv.so = -DFW_FAST_DIV=0), in a child process with FIREWORK_LIB set.

The oracle is the checker here (test infrastructure); nothing of it runs on the product path.
"""
import argparse
import json
import os
import struct
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["FIREWORK_NO_ZERO_SKIP"] = "1"        # every path deposits its length: accum.w = rays of the pixel (read when the library is loaded, below)

import numpy as np  # noqa: E402

from firework_amd import _lib, scenes  # noqa: E402
from oracle import oracle_binding as ob  # noqa: E402

SWITCHES = ["FIREWORK_NO_DEFER", "FIREWORK_NO_HIT4", "FIREWORK_NO_SHORT_RAYS", "FIREWORK_NO_LDS_TREES", "FIREWORK_TLAS_REFILL=0",
            "FIREWORK_BVH=median", "FIREWORK_NO_HOIST", "FIREWORK_NO_LDS_TABLES", "FIREWORK_NO_TILE_ORDER"]


def clone(renderer, samples):
    import copy
    r = copy.copy(renderer)
    r.settings = dict(renderer.settings)
    r.settings["samples"] = samples
    return r


def gpu_counts(ds, renderer, pixel_ids=None):
    n = len(pixel_ids) if pixel_ids is not None else renderer.settings["width"] * renderer.settings["height"]
    accum = np.zeros((n, 4), np.float32)
    ids = None if pixel_ids is None else np.ascontiguousarray(np.asarray(pixel_ids, np.uint32))
    res = ds.render_progressive(renderer, 0, accum, ids)
    return accum[:, 3].astype(np.int64), res


def gpu_lengths(ds, renderer, pixel, spp):
    """path length of every sample of one pixel: spp one-sample renders"""
    r1 = clone(renderer, 1)
    ids = np.array([pixel], np.uint32)
    out = np.zeros(spp, np.int64)
    for s in range(spp):
        accum = np.zeros((1, 4), np.float32)
        ds.render_progressive(r1, s, accum, ids)
        out[s] = int(accum[0, 3])
    return out


def gpu_dump(ds, renderer, pixel, sample):
    """(11,16) float32 records of FIREWORK_DUMP_PATH + header dict + path length"""
    r1 = clone(renderer, 1)
    ids = np.array([pixel], np.uint32)
    fd, path = tempfile.mkstemp(suffix=".bin")
    os.close(fd)
    _lib.set_option("DUMP_PATH", path)
    try:
        accum = np.zeros((1, 4), np.float32)
        ds.render_progressive(r1, sample, accum, ids)
        raw = open(path, "rb").read()
    finally:
        _lib.set_option("DUMP_PATH", None)
        os.unlink(path)
    hdr = struct.unpack("8I", raw[:32])
    cam = struct.unpack("3f", raw[16:28])
    rec = np.frombuffer(raw[32:], np.float32).reshape(11, 16).copy()
    return rec, dict(pinhole0=hdr[1], hit4=hdr[2], prim_bits=hdr[3], cam_pos=cam, n_defer=hdr[7]), int(accum[0, 3])


def decode(rec, hdr, length):
    """GPU records -> list of dicts per segment (ray o/d, t, object, prim)"""
    segs = []
    for s in range(min(length, 11)):
        r = rec[s]
        if s == 0 and hdr["pinhole0"]:
            o, d = list(hdr["cam_pos"]), [float(r[0]), float(r[1]), float(r[2])]
        else:
            o, d = [float(r[0]), float(r[1]), float(r[2])], [float(r[3]), float(r[4]), float(r[5])]
        code = int(np.frombuffer(np.float32(r[10] if hdr["hit4"] else r[11]).tobytes(), np.uint32)[0])
        t = None if hdr["hit4"] else float(r[10])
        miss = code == 0xFFFFFFFF
        segs.append(dict(o=o, d=d, t=t, miss=miss, obj=None if miss else code >> hdr["prim_bits"],
                         prim=None if miss else code & ((1 << hdr["prim_bits"]) - 1)))
    return segs


def first_difference(gsegs, otr):
    """index of the first segment whose ray differs bitwise, or whose hit/miss or t differs"""
    for s, g in enumerate(gsegs):
        o = otr[s]
        if o[15] == 0:
            return s, "oracle path ended before this segment"
        ray = np.array(g["o"] + g["d"], np.float32)
        if ray.tobytes() != o[:6].astype(np.float32).tobytes():
            return s, "ray differs"
        if g["miss"] != (o[6] == 0):
            return s, "hit/miss differs"
        if not g["miss"] and g["t"] is not None and np.float32(g["t"]).tobytes() != np.float32(o[7]).tobytes():
            return s, "t differs"
    if len(gsegs) < 11 and otr[len(gsegs)][15] != 0:
        return len(gsegs), "GPU path ended before this segment"
    return None, "identical"


def with_env(setting):
    k, _, v = setting.partition("=")
    return k, (v or "1")


def hunt(name, w, h, spp, max_pixels, out_path, tol, pixels=None):
    t0 = time.time()
    scene, renderer = scenes.config(name, w, h, spp)
    sd = scene.to_desc()
    ds = _lib.DeviceScene(sd)
    if pixels:       # --pixels: the frames were compared elsewhere (tools/full_parity.py); only these pixels are searched
        ids = np.array(pixels, np.uint32)
        gcnt_p, gres = gpu_counts(ds, renderer, ids)
        ocnt_p = ob.render_counts(sd, renderer, ids).astype(np.int64)
        gcnt = np.zeros(w * h, np.int64); ocnt = np.zeros(w * h, np.int64)
        gcnt[ids] = gcnt_p; ocnt[ids] = ocnt_p
        t1 = time.time()
    else:
        gcnt, gres = gpu_counts(ds, renderer)
        assert int(gcnt.sum()) == int(gres.stats["rays"]), (int(gcnt.sum()), gres.stats["rays"])
        t1 = time.time()
        ocnt = ob.render_counts(sd, renderer).astype(np.int64)
    ores = None
    report = dict(config=name, width=w, height=h, spp=spp, use_bvh=bool(renderer.settings["use_bvh"]),
                  rays_gpu=int(gcnt.sum()), rays_oracle=int(ocnt.sum()), s_gpu=round(t1 - t0, 2), s_oracle=round(time.time() - t1, 2))
    bad = np.array(pixels, np.int64) if pixels else np.nonzero(gcnt != ocnt)[0]
    report["pixels_with_other_ray_count"] = [int(x) for x in bad]
    print(json.dumps(report), flush=True)
    paths = []
    for pix in [int(x) for x in bad[:max_pixels]]:
        ol = ob.path_lengths(sd, renderer, pix, 0, spp).astype(np.int64)
        gl = gpu_lengths(ds, renderer, pix, spp)
        assert int(gl.sum()) == int(gcnt[pix]), "one-sample renders disagree with the frame"
        cand = [int(x) for x in np.nonzero(ol != gl)[0]]
        if not cand and pixels:     # equal lengths but another colour: compare every sample's colour
            r1 = clone(renderer, 1)
            for s in range(spp):
                accum = np.zeros((1, 4), np.float32)
                ds.render_progressive(r1, s, accum, np.array([pix], np.uint32))
                _, ocol = ob.trace_path(sd, renderer, pix, s)
                if not np.allclose(np.nan_to_num(accum[0, :3].astype(np.float64)), np.nan_to_num(ocol.astype(np.float64)), rtol=1e-4, atol=1e-7):
                    cand.append(s)
        for s in cand:
            otr, ocol = ob.trace_path(sd, renderer, pix, s)
            entry = dict(pixel=pix, x=pix % w, row=pix // w, sample=s, len_gpu=int(gl[s]), len_oracle=int(ol[s]), switches={})
            rec, hdr, length = gpu_dump(ds, renderer, pix, s)
            gsegs = decode(rec, hdr, length)
            seg, why = first_difference(gsegs, otr)
            entry["first_difference"] = dict(segment=seg, what=why)
            k = seg if seg is not None and seg < len(gsegs) else (len(gsegs) - 1)
            kk = max(0, k - 1) if why == "ray differs" else k          # a ray that differs was made by the segment before
            entry["gpu_segments"] = gsegs[max(0, kk - 1):kk + 2]
            entry["oracle_segments"] = [dict(o=[float(x) for x in otr[j][:3]], d=[float(x) for x in otr[j][3:6]], hit=bool(otr[j][6]), t=float(otr[j][7]),
                                             material=int(otr[j][8]), point=[float(x) for x in otr[j][9:12]], normal=[float(x) for x in otr[j][12:15]])
                                        for j in range(max(0, kk - 1), min(11, kk + 2)) if otr[j][15]]
            entry["hdr"] = {k2: (list(v) if isinstance(v, tuple) else v) for k2, v in hdr.items()}
            # the same path under every runtime switch (scene-creation switches need a new device scene)
            for sw in SWITCHES:
                key, val = with_env(sw)
                try:
                    _lib.set_option(key, val)
                except _lib.FireworkError:
                    continue                     # a switch of the A/B build only (make ab, FIREWORK_LIB=...lib_ab.so)
                try:
                    ds2 = _lib.DeviceScene(sd)
                    rec2, hdr2, len2 = gpu_dump(ds2, renderer, pix, s)
                    seg2, why2 = first_difference(decode(rec2, hdr2, len2), otr)
                    entry["switches"][sw] = dict(len_gpu=len2, first_difference=seg2, what=why2)
                    ds2.close()
                finally:
                    _lib.set_option(key, None)
            # library variants (compile-time switches), each in a child process
            vdir = os.path.join(ROOT, "firework_amd", "lib", "variants")
            for vname in ("slowdiv", "nocull"):
                lib = os.path.join(vdir, f"lib_{vname}.so")
                if os.path.exists(lib):
                    env = dict(os.environ, FIREWORK_LIB=lib)
                    cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", name, str(w), str(h), str(spp), str(pix), str(s)],
                                        env=env, capture_output=True, text=True, timeout=600)
                    try:
                        entry["switches"]["lib_" + vname] = json.loads(cp.stdout.strip().splitlines()[-1])
                    except Exception:
                        entry["switches"]["lib_" + vname] = dict(error=cp.stderr[-400:])
            paths.append(entry)
            print(json.dumps(entry), flush=True)
    report["paths"] = paths
    # colour differences beyond float noise that the counts do not explain
    if tol > 0:
        ores = ob.render(sd, renderer)
        lin_g = gres.linear.astype(np.float64)
        d = np.abs(lin_g - ores.linear).max(axis=1)
        scale = np.maximum(np.abs(ores.linear).max(axis=1), 1e-3)
        odd = np.nonzero(d > tol * scale)[0]
        report["pixels_linear_beyond_tol"] = [int(x) for x in odd[:64]]
        report["n_pixels_linear_beyond_tol"] = int(odd.size)
        report["u8_differ"] = int((gres.rgb8 != ores.rgb8).sum())
    report["s_total"] = round(time.time() - t0, 2)
    if out_path:
        os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
        json.dump(report, open(out_path, "w"), indent=1)
    print(json.dumps({k: v for k, v in report.items() if k != "paths"}), flush=True)
    ds.close()
    return report


def one(name, w, h, spp, pix, s):
    scene, renderer = scenes.config(name, w, h, spp)
    sd = scene.to_desc()
    ds = _lib.DeviceScene(sd)
    otr, _ = ob.trace_path(sd, renderer, pix, s)
    rec, hdr, length = gpu_dump(ds, renderer, pix, s)
    seg, why = first_difference(decode(rec, hdr, length), otr)
    print(json.dumps(dict(len_gpu=length, first_difference=seg, what=why)))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        a = sys.argv[2:]
        one(a[0], int(a[1]), int(a[2]), int(a[3]), int(a[4]), int(a[5]))
        sys.exit(0)
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("width", type=int)
    ap.add_argument("height", type=int)
    ap.add_argument("spp", type=int)
    ap.add_argument("--max-pixels", type=int, default=6)
    ap.add_argument("--tol", type=float, default=0.0, help="also list pixels whose linear colour differs by more than tol x its magnitude")
    ap.add_argument("--out", default=None)
    ap.add_argument("--pixels", default=None, help="comma-separated pixel ids to search instead of comparing whole frames first")
    a = ap.parse_args()
    px = [int(x) for x in a.pixels.split(",")] if a.pixels else None
    hunt(a.config, a.width, a.height, a.spp, max(a.max_pixels, len(px) if px else 0), a.out, a.tol, px)
