"""N>1 path on CPU: 2 ranks over gloo.  The tile partition + the single gather are the product code
(firework_amd/tiles.py); the per-rank renderer is swapped for the CPU oracle because there is no GPU here.
Checks: the tiled frame assembled on rank 0 is bit-identical to a single-rank render."""
import os
import sys

import numpy as np
import pytest

from firework_amd.tiles import tile_pixel_ids

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("w,h,world,tile", [(512, 512, 8, 32), (100, 56, 3, 32), (40, 40, 2, 16), (33, 17, 4, 8)])
def test_tiles_partition_the_frame(w, h, world, tile):
    parts = [tile_pixel_ids(w, h, r, world, tile) for r in range(world)]
    allp = np.concatenate(parts)
    assert allp.shape[0] == w * h and np.array_equal(np.sort(allp), np.arange(w * h))
    if (w, h, world, tile) == (512, 512, 8, 32):
        assert len({p.shape[0] for p in parts}) == 1      # BASELINE config: perfectly even


@pytest.mark.parametrize("scheme", ["diagonal", "roundrobin", "hash"])
def test_every_dealing_scheme_partitions_and_diagonal_spreads_over_both_axes(scheme):
    w = h = 512
    for world in (2, 4, 8):
        parts = [tile_pixel_ids(w, h, r, world, 16, scheme) for r in range(world)]
        assert np.array_equal(np.sort(np.concatenate(parts)), np.arange(w * h))
        if scheme == "diagonal":     # the default: equal shares, and every rank owns tiles in every tile row AND column
            assert len({p.shape[0] for p in parts}) == 1
            for p in parts:
                assert len(np.unique((p % w) // 16)) == w // 16 and len(np.unique((p // w) // 16)) == h // 16
        if scheme == "roundrobin":   # why it is not the default: vertical stripes when world divides the tiles per row
            assert len(np.unique((parts[0] % w) // 16)) == (w // 16) // world


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from firework_amd import scenes
    from firework_amd.tiles import TileGather
    from oracle import oracle_binding as ob
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene, renderer = scenes.config("C2_cornell_box", 40, 24, 4)
    tg = TileGather(40, 24, rank, world, torch.device("cpu"), dist, tile=8)
    res = ob.render(scene, renderer, pixel_ids=tg.ids, n_threads=1)
    tg.local[: tg.n_local] = torch.from_numpy(res.rgb8)
    frame = tg.assemble()
    if rank == 0:
        q.put(frame.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_equals_single_rank(oracle):
    import torch.multiprocessing as mp
    from firework_amd import scenes
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    frame = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    scene, renderer = scenes.config("C2_cornell_box", 40, 24, 4)
    single = oracle.render(scene, renderer, n_threads=1)
    assert np.array_equal(frame, single.rgb8)
