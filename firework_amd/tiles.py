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
        self.local = torch.zeros((self.n_max, 3), dtype=torch.uint8, device=device)   # padded to equal size for gather
        self.frame = None
        if rank == 0:
            self.frame = torch.zeros((width * height, 3), dtype=torch.uint8, device=device)
            self.all_ids_dev = [torch.from_numpy(a.astype(np.int64)).to(device) for a in all_ids]
            if self.collective:
                self.gather_list = [torch.zeros((self.n_max, 3), dtype=torch.uint8, device=device) for _ in range(world)]

    def assemble(self):
        """`self.local[:n_local]` holds this rank's finished pixels -> full frame on rank 0 (None elsewhere)."""
        if not self.collective:
            self.frame.index_copy_(0, self.all_ids_dev[0], self.local[: self.n_local])
            return self.frame
        # the only collective of the whole path: <= W*H*3 bytes in total
        if self.host_staged:     # rehearsal mode (gloo): same call, tensors staged through host memory
            got = [self.torch.empty((self.n_max, 3), dtype=self.torch.uint8) for _ in range(self.world)] if self.rank == 0 else None
            self.dist.gather(self.local.cpu(), got, dst=0)
            if self.rank == 0:
                for r in range(self.world):
                    self.gather_list[r].copy_(got[r])
        else:
            self.dist.gather(self.local, self.gather_list if self.rank == 0 else None, dst=0)
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
        self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        self.last_ms_gather = 0.0
        self.scene = _lib.DeviceScene(scene if hasattr(scene, "ptr") else scene.to_desc(), device)
        self.world = world
        self.last_stats = None
        self.frame = None

    def render_frame(self):
        """Returns the (H*W,3) uint8 device tensor on rank 0 (None elsewhere)."""
        stream = self.torch.cuda.current_stream(self.dev).cuda_stream
        self.last_stats = self.scene.render(self.renderer, pixel_ids=self.tg.ids,
                                            out_device_ptrs=(self.tg.local.data_ptr(), None, None), stream=stream)
        # the gather (and the scatter of the tiles into the frame) on torch's current stream = the stream fw_render ran on
        self.ev[0].record()
        self.frame = self.tg.assemble()
        self.ev[1].record()
        self.ev[1].synchronize()
        self.last_ms_gather = self.ev[0].elapsed_time(self.ev[1])
        return self.frame

    def close(self):
        self.scene.close()
