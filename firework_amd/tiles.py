"""Multi-GPU framebuffer tiling.

Pixels are independent in the reference (render.rs:127-131: one rayon task per pixel, its own RNG),
so the frame shards with NO data-path collective: rank r renders an interleaved set of tiles with the
same per-pixel keys a single GPU would use, and the only communication is one final gather of the
finished tiles to rank 0 (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).
An N-rank render is therefore bit-identical to a 1-rank render.
"""
import numpy as np

import os
TILE = int(os.environ.get("FW_TILE", "16"))


def tile_owner(t, tx, world, scheme="diagonal"):
    """Rank that renders tile t (tiles numbered row-major, tx per row).
    "roundrobin": t % world — degenerates into vertical stripes whenever world divides tx (16 tiles per row of a
    512-wide frame and 2/4/8 ranks), which loads ranks unevenly on scenes with vertical structure (cornell's walls).
    "diagonal": each tile row is shifted by one more rank, so every rank's tiles spread over both axes."""
    if scheme == "roundrobin":
        return t % world
    if scheme == "diagonal":
        return (t % tx + t // tx) % world
    if scheme == "hash":
        h = (t * 0x9E3779B1) & 0xFFFFFFFF
        h ^= h >> 15; h = (h * 0x85EBCA6B) & 0xFFFFFFFF; h ^= h >> 13
        return h % world
    raise ValueError(scheme)


def tile_pixel_ids(width, height, rank, world, tile=TILE, scheme="diagonal"):
    """Linear pixel indices (render.rs:127 `idx`) of the tiles owned by `rank` (see tile_owner)."""
    tx, ty = (width + tile - 1) // tile, (height + tile - 1) // tile
    ids = []
    for t in range(tx * ty):
        if tile_owner(t, tx, world, scheme) != rank:
            continue
        y0, x0 = (t // tx) * tile, (t % tx) * tile
        ys = np.arange(y0, min(y0 + tile, height), dtype=np.uint32)
        xs = np.arange(x0, min(x0 + tile, width), dtype=np.uint32)
        ids.append((ys[:, None] * np.uint32(width) + xs[None, :]).reshape(-1))
    return np.concatenate(ids) if ids else np.zeros(0, np.uint32)


class TileGather:
    """Rank bookkeeping + the one collective.  Device-agnostic (cuda tensors with nccl, cpu tensors with gloo)."""

    def __init__(self, width, height, rank, world, device, dist=None, tile=TILE, host_staged=False, force_collective=False):
        import torch
        self.torch, self.dist = torch, dist
        # force_collective: a 1-rank group still sends its tiles through dist.gather (the RCCL call on one GPU)
        self.collective = world > 1 or (force_collective and dist is not None)
        self.host_staged = host_staged and self.collective
        self.width, self.height, self.rank, self.world, self.device = width, height, rank, world, device
        all_ids = [tile_pixel_ids(width, height, r, world, tile) for r in range(world)]
        self.ids = np.ascontiguousarray(all_ids[rank])
        self.counts = [int(a.shape[0]) for a in all_ids]
        self.n_local, self.n_max = self.counts[rank], max(self.counts)
        # padded to equal size for gather; two of them, so that frame k's tiles can be gathered while frame k+1 is rendered (TiledRenderer)
        self.locals = [torch.zeros((self.n_max, 3), dtype=torch.uint8, device=device) for _ in range(2)]
        self.local = self.locals[0]
        self.frame = None
        if rank == 0:
            self.frame = torch.zeros((width * height, 3), dtype=torch.uint8, device=device)
            self.all_ids_dev = [torch.from_numpy(a.astype(np.int64)).to(device) for a in all_ids]
            if self.collective:
                self.gather_list = [torch.zeros((self.n_max, 3), dtype=torch.uint8, device=device) for _ in range(world)]

    def assemble(self, local=None):
        """`local[:n_local]` (default `self.local`) holds this rank's finished pixels -> full frame on rank 0 (None elsewhere)."""
        local = self.local if local is None else local
        if not self.collective:
            self.frame.index_copy_(0, self.all_ids_dev[0], local[: self.n_local])
            return self.frame
        # the only collective of the whole path: <= W*H*3 bytes in total
        if self.host_staged:     # rehearsal mode (gloo): same call, tensors staged through host memory
            got = [self.torch.empty((self.n_max, 3), dtype=self.torch.uint8) for _ in range(self.world)] if self.rank == 0 else None
            self.dist.gather(local.cpu(), got, dst=0)
            if self.rank == 0:
                for r in range(self.world):
                    self.gather_list[r].copy_(got[r])
        else:
            self.dist.gather(local, self.gather_list if self.rank == 0 else None, dst=0)
        if self.rank != 0:
            return None
        for r in range(self.world):
            self.frame.index_copy_(0, self.all_ids_dev[r], self.gather_list[r][: self.counts[r]])
        return self.frame


class TiledRenderer:
    """One process per GPU: uploaded scene + this rank's tiles; `render_frame()` = local fw_render into HBM
    + TileGather.assemble()."""

    def __init__(self, scene, renderer, rank=0, world=1, device=0, tile=TILE, dist=None, host_staged_gather=False,
                 force_collective=False):
        import torch
        from . import _lib
        self.torch, self.renderer = torch, renderer
        s = renderer.settings
        self.dev = torch.device("cuda", device)
        torch.cuda.set_device(self.dev)
        self.tg = TileGather(s["width"], s["height"], rank, world, self.dev, dist, tile, host_staged=host_staged_gather,
                             force_collective=force_collective)
        self.last_ms_gather = 0.0
        _lib.init(device)          # fw_init on this rank's own device (idempotent): nothing of it lands in a rank's first frame
        self.scene = _lib.DeviceScene(scene if hasattr(scene, "ptr") else scene.to_desc(), device)
        self.world = world
        self.last_stats = None
        self.frame = None
        # With a collective, the gather (and rank 0's scatter of the shares into the frame) of frame k runs on its own stream
        # while frame k+1 is rendered into the other tile buffer: at 8 ranks a cornell frame is ~5 ms per rank, and rank 0's
        # serial gather + eight scatters would otherwise sit between every two frames.  The frames are the same either way.
        self.side = torch.cuda.Stream(self.dev) if self.tg.collective and not host_staged_gather else None
        self.n_frames = 0
        self.gather_done = [None, None]       # per tile buffer: the event after its last gather
        self.gather_events = []               # (start, stop) pairs of the gathers not yet summed by take_gather_ms()

    def render_frame(self):
        """Returns the (H*W,3) uint8 device tensor on rank 0 (None elsewhere).  With a collective the tensor is complete once
        `wait()` (or any device-wide synchronisation) has returned: the gather of this frame overlaps the next render."""
        torch = self.torch
        main = torch.cuda.current_stream(self.dev)
        k = self.n_frames & 1
        local = self.tg.locals[k] if self.side is not None else self.tg.local
        self.n_frames += 1
        if self.side is not None and self.gather_done[k] is not None:
            main.wait_event(self.gather_done[k])            # the buffer's previous tiles have been sent
        self.last_stats = self.scene.render(self.renderer, pixel_ids=self.tg.ids,
                                            out_device_ptrs=(local.data_ptr(), None, None), stream=main.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if self.side is None:
            # the gather (and the scatter of the tiles into the frame) on torch's current stream = the stream fw_render ran on
            e0.record()
            self.frame = self.tg.assemble(local)
            e1.record()
            e1.synchronize()
            self.last_ms_gather = e0.elapsed_time(e1)
            self.gather_events.append((e0, e1))
            return self.frame
        rendered = torch.cuda.Event()
        rendered.record(main)
        with torch.cuda.stream(self.side):
            self.side.wait_event(rendered)
            e0.record()
            self.frame = self.tg.assemble(local)
            e1.record()
        self.gather_done[k] = e1
        self.gather_events.append((e0, e1))
        return self.frame

    def wait(self):
        """Until every gather issued so far has finished (the frame returned by the last render_frame() is complete)."""
        if self.side is not None:
            self.side.synchronize()

    def take_gather_ms(self):
        """Device time of the gathers since the last call, in ms (waits for them)."""
        self.wait()
        ms = 0.0
        for e0, e1 in self.gather_events:
            e1.synchronize()
            ms += e0.elapsed_time(e1)
        self.gather_events = []
        return ms

    def close(self):
        self.scene.close()
