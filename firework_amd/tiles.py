"""Multi-GPU framebuffer tiling.

Pixels are independent in the reference (render.rs:127-131: one rayon task per pixel, its own RNG),
so the frame shards with NO data-path collective: rank r renders an interleaved set of tiles with the
same per-pixel keys a single GPU would use, and the only communication is one final gather of the
finished tiles to rank 0 (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).
An N-rank render is therefore bit-identical to a 1-rank render.
"""
import numpy as np

TILE = 32


def tile_pixel_ids(width, height, rank, world, tile=TILE):
    """Linear pixel indices (render.rs:127 `idx`) of the tiles owned by `rank`: tiles are numbered
    row-major and dealt round-robin, which balances cheap (sky/black) and expensive rows."""
    tx, ty = (width + tile - 1) // tile, (height + tile - 1) // tile
    ids = []
    for t in range(rank, tx * ty, world):
        y0, x0 = (t // tx) * tile, (t % tx) * tile
        ys = np.arange(y0, min(y0 + tile, height), dtype=np.uint32)
        xs = np.arange(x0, min(x0 + tile, width), dtype=np.uint32)
        ids.append((ys[:, None] * np.uint32(width) + xs[None, :]).reshape(-1))
    return np.concatenate(ids) if ids else np.zeros(0, np.uint32)


def max_tile_pixels(width, height, world, tile=TILE):
    return max(int(tile_pixel_ids(width, height, r, world, tile).shape[0]) for r in range(world))


class TiledRenderer:
    """One process per GPU.  Holds the uploaded scene, this rank's pixel list and the device buffers;
    `render_frame()` = local fw_render into HBM + one gather to rank 0."""

    def __init__(self, scene, renderer, rank=0, world=1, device=0, tile=TILE, dist=None):
        import torch
        from . import _lib
        self.torch, self.dist = torch, dist
        self.rank, self.world, self.renderer = rank, world, renderer
        s = renderer.settings
        self.width, self.height = s["width"], s["height"]
        self.ids = np.ascontiguousarray(tile_pixel_ids(self.width, self.height, rank, world, tile))
        self.n_local = int(self.ids.shape[0])
        self.n_max = max_tile_pixels(self.width, self.height, world, tile) if world > 1 else self.n_local
        self.dev = torch.device("cuda", device)
        torch.cuda.set_device(self.dev)
        self.scene = _lib.DeviceScene(scene if hasattr(scene, "ptr") else scene.to_desc(), device)
        self.rgb8 = torch.zeros((self.n_max, 3), dtype=torch.uint8, device=self.dev)
        self.ids_dev = torch.from_numpy(self.ids.astype(np.int64)).to(self.dev)
        if world > 1:
            all_ids = [tile_pixel_ids(self.width, self.height, r, world, tile) for r in range(world)]
            self.all_counts = [int(a.shape[0]) for a in all_ids]
            if rank == 0:
                self.all_ids_dev = [torch.from_numpy(a.astype(np.int64)).to(self.dev) for a in all_ids]
                self.gather_list = [torch.zeros((self.n_max, 3), dtype=torch.uint8, device=self.dev) for _ in range(world)]
        self.frame = torch.zeros((self.width * self.height, 3), dtype=torch.uint8, device=self.dev) if rank == 0 else None
        self.last_stats = None

    def render_frame(self):
        """Returns the (H*W,3) uint8 device tensor on rank 0 (None elsewhere)."""
        torch = self.torch
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        pix = None if (self.world == 1) else self.ids
        self.last_stats = self.scene.render(self.renderer, pixel_ids=pix,
                                            out_device_ptrs=(self.rgb8.data_ptr(), None, None), stream=stream)
        if self.world == 1:
            self.frame = self.rgb8[: self.n_local]
            return self.frame
        # the only collective: finished tiles -> rank 0 (payload <= W*H*3 bytes in total)
        self.dist.gather(self.rgb8, self.gather_list if self.rank == 0 else None, dst=0)
        if self.rank == 0:
            for r in range(self.world):
                self.frame.index_copy_(0, self.all_ids_dev[r], self.gather_list[r][: self.all_counts[r]])
            return self.frame
        return None

    def close(self):
        self.scene.close()
