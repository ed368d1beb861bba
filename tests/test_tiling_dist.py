"""Tile sharding + the single gather collective, world_size 2 over gloo on CPU (the GPU path swaps in RCCL and
lv_render_tiles_device; the tile logic is identical)."""
import os
import socket
import sys

import numpy as np
import pytest

from common import ROOT
from linevis_amd import tiling


def test_tiles_cover_the_frame_once():
    for (w, h, t) in [(1920, 1080, 64), (100, 70, 32), (64, 64, 64), (130, 65, 16)]:
        tiles = tiling.make_tiles(w, h, t)
        cover = np.zeros((h, w), dtype=np.int32)
        for x0, y0 in tiles:
            cover[y0:y0 + t, x0:x0 + t] += 1
        assert cover.min() == 1 and cover.max() == 1
        assert len({(int(a), int(b)) for a, b in tiles}) == len(tiles)


def test_round_robin_assignment_partitions_tiles():
    tiles = tiling.make_tiles(1920, 1080, 64)
    for n in (1, 2, 4, 8):
        parts = [tiling.assign_tiles(tiles, r, n) for r in range(n)]
        assert sum(len(p) for p in parts) == len(tiles)
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
        merged = {(int(a), int(b)) for p in parts for a, b in p}
        assert len(merged) == len(tiles)
        assert max(len(p) for p in parts) == tiling.tiles_per_rank(len(tiles), n)


def _pattern(x, y):
    return np.stack([x % 256, y % 256, (x * 7 + y * 13) % 256, np.full_like(x, 255)], axis=-1).astype(np.uint8)


def _worker(rank, world, port, width, height, tile, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sf = tiling.ShardedFrame(width, height, tile, rank, world, torch.device("cpu"))

        def fake_render(out, tiles_xy, tw, th):
            for i, (x0, y0) in enumerate(tiles_xy):
                yy, xx = np.meshgrid(np.arange(th) + int(y0), np.arange(tw) + int(x0), indexing="ij")
                out[i] = torch.from_numpy(_pattern(xx, yy))
        sf.render_local(fake_render)
        dist.barrier()
        sf.gather()
        frame = sf.assemble()
        if rank == 0:
            yy, xx = np.meshgrid(np.arange(height), np.arange(width), indexing="ij")
            q.put(bool(np.array_equal(frame, _pattern(xx, yy))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("size", [(200, 120, 32), (96, 64, 64)])
def test_gather_assembles_frame_world2(size):
    import torch.multiprocessing as mp
    width, height, tile = size
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, width, height, tile, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


def test_detile_helper():
    tiles = tiling.make_tiles(50, 30, 16)
    px = np.zeros((len(tiles), 16, 16, 4), dtype=np.uint8)
    for i, (x0, y0) in enumerate(tiles):
        yy, xx = np.meshgrid(np.arange(16) + int(y0), np.arange(16) + int(x0), indexing="ij")
        px[i] = _pattern(xx, yy)
    frame = tiling.detile(px, tiles, 50, 30, 16)
    yy, xx = np.meshgrid(np.arange(30), np.arange(50), indexing="ij")
    assert np.array_equal(frame, _pattern(xx, yy))
