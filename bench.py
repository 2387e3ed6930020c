#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on BASELINE.json's config.

metric : Mrays/s (+ Msamples/s and frame time) of cornell_box 512x512 @1024spp  (configs[1])
step   : one full frame through the HIP wavefront path, tiles sharded over the ranks, + the single gather of the
         finished tiles to rank 0.  TIMED REGION: the scene is already resident in HBM and the frame stays in HBM
         (device-resident steady state); the region the reference itself times (main.rs:40-44: scene conversion + BVH
         build + render, frame back on the host) is reported beside it as `one_shot`, never as `value`.
N>1    : one rank per GPU over RCCL (backend "nccl").  `python bench.py --gpus N` starts its own N ranks (children of a
         parent that never touches the GPU); under torch.distributed.run (WORLD_SIZE set) it is a rank itself.

Prints ONE JSON line on rank 0 (contract in the task statement) with these extra objects:
  schedule       — the timed loop runs the library's own schedule (two batches in flight where that is faster); per-kernel times come
                   from a labelled exclusive pass (one batch in flight) run right after it
  roofline       — k_shade (the kernel that moves most HBM bytes): THIS layout's algorithmic bytes (fw_stats.bytes_shade,
                   exact from the queue counters, elided zero deposits not counted) / its HIP-event time vs 8 TB/s;
                   `traffic` = PMC bytes per launch from the committed rocprofv3 passes, only when they were taken from
                   the same kernel sources as this build (else null)
  roofline_frame — the whole frame: the layout's own bytes and SURVEY §8(d)'s generic 160 B/ray formula, both over device time
  one_shot       — fw_render_scene: conversion + BVH build + upload + render + D2H (main.rs:40-44's region), median of 3 warm calls
  one_shot_cold  — the same call as the FIRST call of a fresh process (what the reference's binary times)
  device         — clocks and a measured copy rate of the box the line was taken on
  parity         — the same frame (or a pixel lattice of it, full spp) against the CPU oracle at the same seed
  cpu_baseline   — the CPU oracle in the reference's sequential-LCG mode timed on this host's cores (bounded sample)
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec
KERNEL_SOURCES = ("firework_amd/csrc/fw_kernels.hip", "firework_amd/csrc/fw_runtime.cpp", "firework_amd/csrc/fw_device.h", "firework_amd/csrc/fw_libm.h", "Makefile")


def kernel_source_sha():
    """Identifies the build a profile under profiles/ was taken from (scripts/summarize_prof.py stores the same hash)."""
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C2_cornell_box")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--paths-per-batch", type=int, default=0)
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip per-launch HIP events (no roofline object)")
    ap.add_argument("--cpu-spp", type=int, default=0, help="spp of the bounded CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU legs (cpu_baseline and parity)")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-one-shot", action="store_true")
    ap.add_argument("--parity-seconds", type=float, default=45.0, help="CPU budget of the parity leg (pixel lattice chosen to fit)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (=RCCL over xGMI) for real multi-GPU runs; gloo stages the gather through host memory and "
                         "lets several ranks share one GPU — a functional rehearsal of the N>1 path on a 1-GPU box")
    ap.add_argument("--force-collective", action="store_true",
                    help="N=1 only: initialise the process group and send the tiles through the gather anyway (exercises the RCCL call on one GPU)")
    ap.add_argument("--dump-frame", default=None, help="rank 0 writes the assembled u8 frame (H*W,3) to this .npy")
    return ap.parse_args(argv)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks as children BEFORE anything here touches the GPU
    (this parent never imports torch), relay their output, return their exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))

    # ONE line on stdout: gloo and RCCL print banners to the C-level stdout of every rank ("[Gloo] Rank 0 is connected ...",
    # "RCCL version : ..."), so everything written to fd 1 from here on goes to stderr and the JSON line is written to the
    # saved descriptor at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist
    from firework_amd import scenes
    from firework_amd.tiles import TiledRenderer

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    n_dev = torch.cuda.device_count()
    if args.backend == "nccl" and world > n_dev:
        raise SystemExit(f"{world} ranks but {n_dev} GPU(s) visible: RCCL needs one GPU per rank "
                         f"(use --backend gloo to rehearse the N>1 path on fewer GPUs)")
    device_index = local_rank if args.backend == "nccl" else local_rank % max(1, n_dev)
    use_dist = world > 1 or args.force_collective
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(s.getsockname()[1]))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(device_index)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group("gloo")

    from firework_amd import _lib
    _lib.init(device_index)        # fw_init: context, code objects, kernel handles and the default path arena — explicit since ABI v7, before any timed region
    scene, renderer = scenes.config(args.config, args.width, args.height, args.spp)
    if args.paths_per_batch:
        renderer.paths_per_batch(args.paths_per_batch)
    # (per-launch HIP events — FW_FLAG_TIME_KERNELS — are recorded in the labelled exclusive pass only: 70 event records per frame are a
    #  cost of their own in a 1.5 ms frame, and a frame that carries them is not replayed as a graph)
    s = renderer.settings
    tr = TiledRenderer(scene, renderer, rank, world, device_index, dist=dist if use_dist else None,
                       host_staged_gather=(args.backend == "gloo"), force_collective=args.force_collective)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if use_dist:      # communicator set-up (seconds for RCCL) must not land in a timed step when --warmup is 0
        dist.barrier()
        tr.tg.assemble()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        tr.render_frame()
    barrier()
    tr.take_gather_ms()
    t0 = time.perf_counter()
    keys = ("rays", "samples", "ms_render", "ms_extend", "ms_shade", "ms_raygen", "ms_accumulate", "n_extend_launches",
            "n_shade_launches", "bytes_raygen", "bytes_extend", "bytes_shade", "bytes_accumulate", "algorithmic_bytes")
    acc = {k: 0 for k in keys}
    ms_gather = 0.0
    for _ in range(args.steps):
        tr.render_frame()
        for k in acc:
            acc[k] += tr.last_stats[k]
    barrier()                     # device-wide: the last frame's gather (its own stream when there is a collective) included
    elapsed = time.perf_counter() - t0
    ms_gather = tr.take_gather_ms()

    # max over ranks of the elapsed time; sums over ranks of the work counters
    vec = torch.tensor([elapsed, acc["rays"], acc["samples"], acc["ms_extend"], acc["ms_shade"], acc["ms_render"], ms_gather],
                       dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
    if world > 1:
        mx = vec.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0])
        ms_gather = float(mx[6])
    total_rays, total_samples = float(vec[1]), float(vec[2])
    rank_spread = None
    if world > 1:      # what an imbalance would look like: per-rank rays and device time (max / mean / min over the ranks)
        mine = torch.tensor([acc["rays"], acc["ms_render"]], dtype=torch.float64, device=vec.device)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rr = [float(t[0]) for t in allr]; mm = [float(t[1]) / args.steps for t in allr]
        rank_spread = {"rays_max_over_mean": max(rr) / (sum(rr) / world), "rays_per_rank": rr,
                       "ms_render_max": max(mm), "ms_render_min": min(mm), "ms_render_per_rank": mm}

    if rank == 0:
        frame = tr.frame.cpu().numpy()
        assert frame.shape == (s["width"] * s["height"], 3)
        if args.dump_frame:
            np.save(args.dump_frame, frame)
        ms_step = elapsed * 1e3 / args.steps
        headline = args.config == "C2_cornell_box" and not (args.width or args.height or args.spp)
        out = {
            "metric": "Mrays/s, cornell_box 512x512 @1024spp (scene resident in HBM; whole frame incl. tile gather)" if headline else f"Mrays/s, {args.config}",
            "value": total_rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.config} {s['width']}x{s['height']} @{s['samples']}spp", "use_bvh": bool(s["use_bvh"]),
                       "tiles": "16x16, dealt diagonally over ranks",
                       "collective": (f"one gather of u8 tiles to rank 0 ({args.backend}, world_size {dist.get_world_size()})" if use_dist else "none"),
                       "rng": "counter (pcg4d) keyed by (seed,pixel,sample,dimension)", "seed": s["seed"]},
            "timed_region": "device-resident steady state: scene already in HBM, frame assembled in HBM on rank 0 "
                            "(the reference's own region, main.rs:40-44, is `one_shot`)",
            "msamples_per_s": total_samples / elapsed / 1e6,
            "frame_time_ms": ms_step,
            "rays_per_sample": total_rays / max(1.0, total_samples),
            "ms_gather_per_step": ms_gather / args.steps,
        }
        if not args.no_kernel_timing:
            # The timed loop runs the library's own schedule (two batches in flight where that is faster: their kernels overlap, so
            # their HIP-event times do too).  The roofline divides by a kernel's OWN time: a labelled exclusive pass, one batch in flight.
            ex_acc, ex_ms = exclusive_pass(tr, keys, frames=max(2, min(5, args.steps)))
            # the library keeps two batches in flight whenever a frame has two (fw_runtime.cpp: n_lanes) unless STREAMS says otherwise
            overlapped = int(tr.last_stats.get("n_batches", 1)) >= 2 and os.environ.get("FIREWORK_STREAMS", "") != "1"
            out["schedule"] = {"timed_loop": "two batches in flight on two streams (kernel times overlap)" if overlapped else "one batch in flight",
                               "timed_loop_ms_per_step": ms_step, "exclusive_pass_ms_per_step": ex_ms,
                               "exclusive_pass": f"FIREWORK_STREAMS=1, {ex_acc['frames']} frames after the timed loop: every kernel's HIP-event time is its own; "
                                                 "roofline, roofline_frame and kernel_ms_per_step are taken from it"}
            out.update(roofline_objects(ex_acc, args, tr, renderer, steps=ex_acc["frames"]))
        out["device"] = device_info(device_index)
        if world > 1:
            out["ranks"] = rank_spread
        if world == 1 and not args.no_one_shot:
            out["one_shot"] = one_shot(scene, renderer, device_index)
            out["one_shot_cold"] = one_shot_cold(args)
        if world == 1 and not args.no_cpu_baseline:
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            out["cpu_baseline"] = cpu_baseline(args, s, cores)
            if not args.no_parity:
                out["parity"] = parity(args, tr, renderer, scene, frame, out["cpu_baseline"], cores)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    tr.close()
    if use_dist:
        if world > 1:
            dist.barrier()          # rank 0 finishes its extra (untimed) work before any rank tears the group down
        dist.destroy_process_group()


def exclusive_pass(tr, keys, frames):
    """`frames` frames with one batch in flight (option STREAMS=1): per-kernel HIP-event times that belong to one kernel each."""
    from firework_amd import _lib
    _lib.set_option("STREAMS", "1")
    tr.renderer.time_kernels(True)
    def frame():          # this rank's share, no collective (only rank 0 runs the exclusive pass)
        return tr.scene.render(tr.renderer, pixel_ids=tr.tg.ids, out_device_ptrs=(tr.tg.local.data_ptr(), None, None))
    try:
        frame()                                             # the lane's pools may have to grow: not timed
        acc = {k: 0 for k in keys}
        import torch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(frames):
            st = frame()
            for k in acc:
                acc[k] += st[k]
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / frames
    finally:
        _lib.set_option("STREAMS", os.environ.get("FIREWORK_STREAMS"))      # back to what the library was loaded with
        tr.renderer.time_kernels(False)
    acc["frames"] = frames
    return acc, ms


def device_info(device_index):
    """What the ±10 % box-to-box spread of the HBM-bound kernels should be read against: clocks as the driver reports them and a
    measured device-to-device copy rate (1 GiB, best of 5)."""
    import torch
    info = {}
    try:
        p = torch.cuda.get_device_properties(device_index)
        info.update(name=p.name, cus=p.multi_processor_count, hbm_gib=round(p.total_memory / 2 ** 30, 1),
                    sclk_mhz=getattr(p, "clock_rate", 0) / 1e3, mclk_mhz=getattr(p, "memory_clock_rate", 0) / 1e3)
        n = 1 << 28
        a = torch.empty(n, dtype=torch.float32, device=f"cuda:{device_index}").fill_(1.0)
        b = torch.empty_like(a)
        best = 0.0
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); b.copy_(a); e1.record(); torch.cuda.synchronize()
            best = max(best, 2 * 4 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        info["copy_GBps"] = round(best, 1)
        info["copy_probe"] = "torch d2d copy of 1 GiB (read + write counted), best of 5"
        del a, b
    except Exception as e:
        info["error"] = repr(e)
    try:
        # under rocprofv3 the profiler's preloaded library initialises the GPU in every child process, and rocm-smi is a `#!/usr/bin/env python3`
        # script: its env -> python3 hop would be an exec after GPU initialisation, which the GPU boxes refuse.  No clocks in profiled runs.
        if "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ):
            raise RuntimeError("profiled run: rocm-smi not called")
        cp = subprocess.run(["rocm-smi", "--showclocks", "--json"], capture_output=True, text=True, timeout=20)
        if cp.returncode == 0 and cp.stdout.strip().startswith("{"):
            d = json.loads(cp.stdout)
            card = d.get(f"card{device_index}") or next(iter(d.values()))
            info["rocm_smi_clocks"] = {k: v for k, v in card.items() if "clk" in k.lower()}
            import re
            for name in ("sclk", "mclk"):       # "(2362Mhz)": what the card runs at now; torch's properties report 0 on this image
                m = re.search(r"(\d+)\s*mhz", str(card.get(f"{name} clock speed:", "")), re.I)
                if m and not info.get(f"{name}_mhz"):
                    info[f"{name}_mhz"] = float(m.group(1))
    except Exception:
        pass
    return info


def one_shot_cold(args):
    """The region main.rs:40-44 times, in the state the reference's binary is in when it times it: a process's FIRST call.  Two fresh
    child processes render the frame once each: one the way the CLI and this bench do (fw_init first, then the call), one the way a host
    that never heard of fw_init does (the first call initialises the device itself).  Every part is reported; `ms_cold_total` = library
    load + fw_init + first call is what a fresh `python -m firework_amd` user waits for beyond the interpreter's own imports."""
    def child(with_init):
        code = ("import sys, time, json; sys.path.insert(0, %r)\n"
                "from firework_amd import scenes, _lib\n"
                "s, r = scenes.config(%r, %r, %r, %r); sd = s.to_desc()\n"
                "t0 = time.perf_counter(); import torch; torch_ms = (time.perf_counter() - t0) * 1e3\n"      # the Python host's plumbing, not the library
                "t0 = time.perf_counter(); _lib.load(); load_ms = (time.perf_counter() - t0) * 1e3\n"
                "init_ms = 0.0\n"
                "if %r:\n"
                "    t0 = time.perf_counter(); _lib.init(0); init_ms = (time.perf_counter() - t0) * 1e3\n"
                "t0 = time.perf_counter(); res = _lib.render_scene(sd, r); dt = (time.perf_counter() - t0) * 1e3\n"
                "st = res.stats\n"
                "out = dict(ms_wall=dt, ms_library=st['ms_wall'], ms_scene=st['ms_scene'], ms_render=st['ms_render'], ms_d2h=st['ms_d2h'], mrays_per_s=st['rays'] / dt / 1e3,\n"
                "           ms_import_torch=torch_ms, ms_library_load=load_ms, ms_fw_init=init_ms, ms_cold_total=load_ms + init_ms + dt)\n"
                "seq = []\n"
                "for name in (%r):\n"
                "    s2, r2 = scenes.config(name); sd2 = s2.to_desc(); w = []\n"
                "    for rep in range(3):\n"
                "        t0 = time.perf_counter(); _lib.render_scene(sd2, r2); w.append((time.perf_counter() - t0) * 1e3)\n"
                "    seq.append(dict(config=name, first_ms=w[0], warm_ms=min(w[1:]), first_over_warm=w[0] / min(w[1:])))\n"
                "out['then_first_calls_in_the_same_process'] = seq\n"
                "print(json.dumps(out))\n"
                % (ROOT, args.config, args.width, args.height, args.spp, with_init,
                   ("C3_suzanne", "C4b_volume_test") if with_init and args.config == "C2_cornell_box" and not args.spp else ()))
        cp = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, FIREWORK_TRACE="1"))
        d = json.loads([l for l in cp.stdout.splitlines() if l.startswith("{")][-1])
        d["trace"] = [l for l in cp.stderr.splitlines() if l.startswith("[firework]")][:12]     # where the first call's host time went (FIREWORK_TRACE)
        return d
    try:
        d = child(True)
        d["region"] = ("first fw_render_scene call of a fresh process after fw_init: ms_wall = scene + render + D2H; before it ms_import_torch (the Python host's own plumbing), ms_library_load (dlopen: no HIP call since "
                       "ABI v7) and ms_fw_init (HIP context, code objects, kernel handles, staging, the default path arena: 0.4 ms to 1.4 s by the state of the "
                       "device's memory); ms_cold_total = all three; then the first calls of two other configs in the same process")
        try:
            lazy = child(False)
            d["without_fw_init"] = {k: lazy[k] for k in ("ms_wall", "ms_import_torch", "ms_library_load", "ms_cold_total", "ms_render", "ms_scene", "trace")}
            d["without_fw_init"]["region"] = "the same first call in a process that never calls fw_init: the call initialises the device itself and sizes the arena for this frame"
        except Exception as e:
            d["without_fw_init"] = {"error": repr(e)}
        return d
    except Exception as e:
        return {"error": repr(e)}


def roofline_objects(acc, args, tr, renderer, steps=None):
    """Rank-0 kernel classes (every rank runs the same kernels on 1/N of the pixels).  Bytes: fw_stats.bytes_* = this
    layout's own algorithmic HBM bytes per kernel class, exact from the queue counters (fw_runtime.cpp);
    the deposits that k_shade elides over a black environment are counted by one extra untimed frame."""
    import glob
    res = {}
    steps = steps or args.steps
    exact = tr.scene.render(renderer.count_deposits(True), pixel_ids=tr.tg.ids, out_device_ptrs=(tr.tg.local.data_ptr(), None, None))
    renderer.count_deposits(False)
    shd_bytes, ext_bytes = float(exact["bytes_shade"]), float(exact["bytes_extend"])       # per frame
    shd_s, ext_s = acc["ms_shade"] / 1e3 / steps, acc["ms_extend"] / 1e3 / steps
    n_sh = max(1, acc["n_shade_launches"] // steps)
    n_ex = max(1, acc["n_extend_launches"] // steps)
    sha = kernel_source_sha()
    traffic = traffic_src = frac_traffic = other_traffic = None
    single_extend_kernel = False
    other_bound = other_src = None
    workload = f"{args.config} {renderer.settings['width']}x{renderer.settings['height']} @{renderer.settings['samples']}spp"
    # a counter summary of this build: of this very workload, else of the same scene and size at another sample count (a launch is one
    # segment of one BATCH, and batches have the same size whatever the frame's sample count: part2 is profiled at 256 of its 4096 spp)
    def same_launches(d):
        w = d.get("workload") or ""
        if w == workload:
            return 2
        return 1 if w.split(" @")[0] == workload.split(" @")[0] and w.split(" @")[0] else 0

    def candidates(pattern):
        found = []
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
            try:
                d = json.load(open(path))
            except Exception:
                continue
            if d.get("source_sha") == sha and same_launches(d):
                found.append((same_launches(d), path, d))
        return [(path, d) for _, path, d in sorted(found, key=lambda x: -x[0])]

    for path, d in candidates("*_pmc.json"):
        try:
            if tr.world != 1:
                continue                                  # the profiles are whole-frame runs: a rank's share moves 1/N of those bytes per launch
            k = [k for k in d["kernels"] if "k_shade" in k][0]
            traffic, traffic_src = d["kernels"][k]["hbm_bytes_per_launch"], os.path.relpath(path, ROOT) + ("" if d.get("workload") == workload else f" (taken at {d.get('workload')}: same batch per launch)")
            ke = max((k for k in d["kernels"] if "k_extend" in k or "k_blas" in k), key=lambda k: d["kernels"][k]["us_total_in_these_passes"])
            other_traffic = d["kernels"][ke]["hbm_bytes_per_launch"]
            single_extend_kernel = sum(1 for k in d["kernels"] if "k_extend" in k or "k_blas" in k) == 1
            break
        except Exception:
            continue
    shade_bound = None
    for path, d in candidates("*_sq.json"):
        try:
            k = max((k for k in d["kernels"] if "k_extend" in k or "k_blas" in k), key=lambda k: d["kernels"][k]["us_total"])   # the longest of the extend-class kernels
            other_bound = d["kernels"][k]["bound"]
            other_src = os.path.relpath(path, ROOT) + f": {k}" + ("" if d.get("workload") == workload else f" (taken at {d.get('workload')})")
            ks = [k for k in d["kernels"] if "k_shade" in k]
            if ks:
                e = d["kernels"][ks[0]]
                shade_bound = {"counters": e["bound"], "valu_issue_util": e.get("valu_issue_util"), "hbm_util_of_streaming": e.get("hbm_util_of_streaming"),
                               "lane_utilisation": e.get("lane_utilisation"), "source": os.path.relpath(path, ROOT)}
            break
        except Exception:
            continue
    ach = shd_bytes / shd_s / 1e9 if shd_s > 0 else 0.0
    if traffic is not None and shd_s > 0:
        frac_traffic = traffic * n_sh / shd_s / 1e9 / HBM_PEAK_GBS
    # what bounds k_shade: the SQ counters of THIS build where a summary exists (scripts/summarize_prof.py: the larger of the measured
    # utilisations when it reaches 0.7, else "latency"); "hbm" by kernel class only when there is none, and the line says so
    res["roofline"] = {
        "bound": shade_bound["counters"] if shade_bound else "hbm",
        "bound_source": shade_bound["source"] if shade_bound else "default by kernel class (k_shade streams the queue arrays); no counter summary of this build and workload under profiles/",
        "kernel": "k_shade", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
        "traffic": traffic, "traffic_source": traffic_src, "frac_traffic": frac_traffic,
        "algorithmic_bytes_per_launch": shd_bytes / n_sh, "avg_launch_us": shd_s * 1e6 / n_sh,
        "bytes_per_ray": shd_bytes / max(1.0, float(exact["rays"])),
        "counters": shade_bound,      # what the SQ counters of this build say k_shade waits for (round 4: with the 8-byte state its instruction issue is as busy as its HBM streams)
        # ... and what the kernel answered when asked directly (round 5, cornell): the label above is the counter rule's, the ablations do not bear it out
        "ablations": {"source": "profiles/r05d_shade_ablations.txt, r05b_shade_pipe_ab.txt, r05l_shade_two_ahead.txt; profiles/NOTES_r05.md section 2",
                      "minus_40pct_vector_instructions": "no change (19.3-19.4 vs 19.1-19.3 ms)", "static_vmcnt_waits": "no change",
                      "six_waves_per_simd": "no change (19.6-19.7)", "prefetch_one_and_a_half_chunks_ahead": "no change",
                      "streams_and_compaction_alone": "5.5 TB/s = the box's copy rate (the kernel itself: 4.1 TB/s)",
                      "same_binary_two_processes_one_box": "17.6 and 19.4 ms: follows where the arena was placed"},
        "bytes": "this layout's streams, exact from the queue counters; zero deposits elided over a black environment are not counted "
                 f"({exact['deposits']} of {exact['samples']} radiance records written)",
        "other_kernel": {"kernel": "k_extend", "bound": other_bound, "bound_source": other_src,
                         "ms_per_step": ext_s * 1e3, "achieved": ext_bytes / ext_s / 1e9 if ext_s > 0 else 0.0,
                         "frac_hbm": ext_bytes / ext_s / 1e9 / HBM_PEAK_GBS if ext_s > 0 else 0.0,
                         "algorithmic_bytes_per_launch": ext_bytes / n_ex, "avg_launch_us": ext_s * 1e6 / n_ex,
                         "traffic": other_traffic,     # PMC bytes per launch of the longest extend-class kernel (same source as above)
                         "traffic_over_algorithmic": other_traffic / (ext_bytes / n_ex) if other_traffic and single_extend_kernel else None},
    }
    # `roofline` names the kernel class with the larger exclusive time (round 4).  For cornell that is k_shade (above); where the tree
    # walks dominate (suzanne, part2, teapot, random_spheres) it is the extend class, whose bound is not HBM: its algorithmic bytes over its
    # time are quoted against the HBM peak all the same (that is what the fraction means), next to the bound the counter summary names.
    ok = res["roofline"]["other_kernel"]
    if ok["bound"] is None:
        ok["bound"] = "valu_issue"
        ok["bound_source"] = "default by kernel class (profiles/r04z_*_sq.json: the wide-node walks and the scans issue vector instructions in 0.86-0.99 of their SIMD cycles); no counter summary of this build and workload under profiles/"
    if ext_s > shd_s:
        sh = {k: v for k, v in res["roofline"].items() if k != "other_kernel"}
        res["roofline"] = {"bound": ok["bound"], "kernel": "k_extend (scan + walks of one segment)", "achieved": ok["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": ok["frac_hbm"], "traffic": ok["traffic"], "bound_source": ok["bound_source"],
                           "algorithmic_bytes_per_launch": ok["algorithmic_bytes_per_launch"], "avg_launch_us": ok["avg_launch_us"],
                           "bytes_per_ray": ext_bytes / max(1.0, float(exact["rays"])), "ms_per_step": ok["ms_per_step"],
                           "note": "the dominant kernel class of this workload is not bound by HBM: frac is its layout bytes over its time against the 8 TB/s peak, bound is what the SQ counters say it waits for",
                           "other_kernel": dict(sh, ms_per_step=shd_s * 1e3)}
    dev_s = acc["ms_render"] / 1e3 / steps
    layout = float(exact["bytes_raygen"] + exact["bytes_extend"] + exact["bytes_shade"] + exact["bytes_accumulate"])
    survey = float(exact["algorithmic_bytes"])
    res["roofline_frame"] = {"unit": "GB/s", "device_ms": dev_s * 1e3,
                             "layout": {"bytes": layout, "achieved": layout / dev_s / 1e9, "frac": layout / dev_s / 1e9 / HBM_PEAK_GBS,
                                        "formula": "bytes_raygen + bytes_extend + bytes_shade + bytes_accumulate (fw_stats) over device time"},
                             "survey": {"bytes": survey, "achieved": survey / dev_s / 1e9, "frac": survey / dev_s / 1e9 / HBM_PEAK_GBS,
                                        "formula": "160*rays + 24*samples (SURVEY §8d, a generic split-kernel layout) over device time"}}
    res["kernel_ms_per_step"] = {k: acc[k] / steps for k in ("ms_render", "ms_raygen", "ms_extend", "ms_shade", "ms_accumulate")}
    res["roofline"]["times"] = "exclusive pass (one batch in flight), see `schedule`"
    return res


def one_shot(scene, renderer, device_index):
    """The region the reference times (main.rs:40-44): Scene -> SceneInternal + BVH build (+ upload) + render + the frame
    back in host memory, through fw_render_scene.  Median of three calls after one warm call (the path pools stay cached
    per device between calls, like any allocator would keep them)."""
    from firework_amd import _lib
    sd = scene.to_desc()
    runs = []
    for rep in range(4):
        t0 = time.perf_counter()
        res = _lib.render_scene(sd, renderer, device=device_index)
        wall = (time.perf_counter() - t0) * 1e3
        st = res.stats
        runs.append({"ms_wall": st["ms_wall"], "ms_wall_incl_binding": wall, "ms_scene": st["ms_scene"], "ms_render": st["ms_render"],
                     "ms_d2h": st["ms_d2h"], "mrays_per_s": st["rays"] / st["ms_wall"] / 1e3})
    runs = sorted(runs[1:], key=lambda r: r["ms_wall"])
    med = dict(runs[1])
    med["ms_wall_all"] = [round(r["ms_wall"], 2) for r in runs]
    med["region"] = "fw_render_scene: flatten + BVH build + upload + render + D2H of rgb8/gamma/linear frames (PCIe-inclusive)"
    return med


def cpu_baseline(args, s, cores):
    from firework_amd import scenes
    from firework_amd._abi import FW_RNG_LCG
    from oracle import oracle_binding as ob       # CPU oracle: the checker, timed as the reported baseline
    native = ob.native_timing_build()             # the same source at -O3 -march=native, built on this host: timed only, never the checker
    # bounded sample: ~16 s of CPU work split between the two builds below, sized by a one-sample pilot (part2 costs 30 x cornell per sample)
    if args.cpu_spp:
        cpu_spp = args.cpu_spp
    else:
        pscene, pr = scenes.config(args.config, args.width, args.height, 1)
        p0 = time.perf_counter(); ob.render(pscene, pr, rng_mode=FW_RNG_LCG, n_threads=cores); pdt = max(1e-3, time.perf_counter() - p0)
        cpu_spp = max(2, min(s["samples"], int(16.0 / pdt)))
    def timed(spp, timing_lib=None):
        cscene, cr = scenes.config(args.config, args.width, args.height, spp)
        c0 = time.perf_counter()
        cres = ob.render(cscene, cr, rng_mode=FW_RNG_LCG, n_threads=cores, timing_lib=timing_lib)   # reference semantics: per-pixel sequential LCG
        cdt = time.perf_counter() - c0
        return cres, cr, cdt
    half = max(1, cpu_spp // 2) if native else cpu_spp
    cres, cr, cdt = timed(half)
    out = {"value": cres.stats["rays"] / cdt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
           "msamples_per_s": cres.stats["samples"] / cdt / 1e6,
           "build": "oracle/liboracle.so: -O2 -ffp-contract=off -mfma (the checker's build)",
           "timed_region": "scene conversion + BVH build + render (main.rs:40-44), C++ restatement of the reference's rayon loop",
           "sample": f"{args.config} {cr.settings['width']}x{cr.settings['height']} @{half}spp "
                     f"({cres.stats['samples']} samples, {cdt:.1f} s; cost is linear in spp)"}
    if native:
        nres, ncr, ndt = timed(half, native)
        checker = dict(value=out["value"], msamples_per_s=out["msamples_per_s"], build=out["build"], sample=out["sample"])
        out.update(value=nres.stats["rays"] / ndt / 1e6, msamples_per_s=nres.stats["samples"] / ndt / 1e6,
                   build="the same source at -O3 -march=native -ffp-contract=off, built on this host for this leg only (never the checker)",
                   sample=f"{args.config} {ncr.settings['width']}x{ncr.settings['height']} @{half}spp ({nres.stats['samples']} samples, {ndt:.1f} s; cost is linear in spp)",
                   checker_build=checker)
    return out


def parity(args, tr, renderer, scene, frame_u8, cpu, cores):
    """north_star's gate where it is defined: per-pixel RGB of the bench config, GPU vs the CPU oracle at the same seed
    (counter RNG), RMS on the post-gamma clamped floats (render.rs:184-190).  All pixels when the CPU budget allows,
    otherwise every k-th pixel in x and y at the full sample count."""
    import numpy as np
    from oracle import oracle_binding as ob
    s = renderer.settings
    w, h, spp = s["width"], s["height"], s["samples"]
    est = w * h * spp / max(1e-9, (cpu.get("checker_build") or cpu)["msamples_per_s"] * 1e6)          # seconds for every pixel at the checker build's measured rate
    stride = 1
    while est / (stride * stride) > args.parity_seconds:
        stride += 1
    ys, xs = np.meshgrid(np.arange(stride // 2, h, stride), np.arange(stride // 2, w, stride), indexing="ij")
    ids = (ys * w + xs).reshape(-1).astype(np.uint32)
    g = tr.scene.render(renderer, pixel_ids=ids)                          # HIP path, host buffers, same pixels
    t0 = time.perf_counter()
    c = ob.render(scene, renderer, pixel_ids=ids, n_threads=cores)
    dt = time.perf_counter() - t0
    def rms_of(a, b):
        # a value that is not a number on BOTH sides counts as equal (part2's turbulence texture has negative albedos: powf of a negative mean is
        # NaN in the reference, the oracle and on the device alike); on one side only it counts as a difference of 1
        a = a.astype(np.float64); b = b.astype(np.float64)
        both = np.isnan(a) & np.isnan(b)
        one = np.isnan(a) ^ np.isnan(b)
        d = np.where(both, 0.0, np.where(one, 1.0, np.nan_to_num(a) - np.nan_to_num(b)))
        return float(np.sqrt(np.mean(d ** 2))), int(both.sum()), int(one.sum())
    rms, nan_both, nan_one = rms_of(g.gamma, c.gamma)
    rms_lin = rms_of(g.linear, c.linear)[0]
    return {"rms_gamma": rms, "rms_linear": rms_lin, "gate": 1e-3, "pass": bool(rms <= 1e-3 and nan_one == 0),
            "nan_values_on_both_sides": nan_both, "nan_values_on_one_side": nan_one,
            "u8_diffs": int((g.rgb8 != c.rgb8).sum()), "u8_values": int(c.rgb8.size),
            "rays_equal": bool(g.stats["rays"] == c.stats["rays"]), "rays_gpu": int(g.stats["rays"]), "rays_oracle": int(c.stats["rays"]),
            "timed_frame_equals_checked_frame": bool(np.array_equal(frame_u8[ids], g.rgb8)),
            "pixels": int(ids.shape[0]), "pixel_stride": stride, "spp": spp, "oracle_seconds": round(dt, 1),
            "what": f"{args.config} {w}x{h} @{spp}spp, " + ("every pixel" if stride == 1 else f"every {stride}{'nd' if stride == 2 else 'rd' if stride == 3 else 'th'} pixel in x and y")
                    + ", GPU vs CPU oracle (FW_RNG_CTR, same seed)"}


if __name__ == "__main__":
    main()
