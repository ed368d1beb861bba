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
        yy, xx = np.meshgrid(np.arange(height), np.arange(width), indexing="ij")
        want = _pattern(xx, yy)

        def fake_render(out, tiles_xy, tw, th):
            for i, (x0, y0) in enumerate(tiles_xy):
                ty, tx = np.meshgrid(np.arange(th) + int(y0), np.arange(tw) + int(x0), indexing="ij")
                out[i] = torch.from_numpy(_pattern(tx, ty))

        def frame_ok():
            sf.render_local(fake_render)
            dist.barrier()
            sf.gather()
            a, b = sf.assemble(), sf.assemble_device()   # bench.py uses assemble_device()
            if rank != 0:
                return a is None and b is None
            return bool(np.array_equal(a, want)) and bool(np.array_equal(b.numpy(), want))
        ok = frame_ok()                                   # round robin along the Morton order (ragged last slot)
        # re-deal by cost: a heavy blob in the middle of the picture, every rank reports the cost of its OWN tiles only
        cx, cy = width / 2.0, height / 2.0
        local = [1000.0 * np.exp(-(((x0 + tile / 2 - cx) / (0.2 * width)) ** 2 + ((y0 + tile / 2 - cy) / (0.2 * height)) ** 2))
                 for x0, y0 in sf.local_tiles.astype(np.float64)]
        costs = sf.rebalance(local, base_cost=1.0)
        loads = [costs[ix].sum() + len(ix) for ix in sf.assignment]
        balanced = max(loads) <= sum(loads) / world + costs.max() + 1.0     # the bound longest-processing-time-first guarantees
        ok = ok and frame_ok() and balanced               # gather pieces are padded to the largest rank's tile count
        if rank == 0:
            q.put(bool(ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,size", [(2, (200, 120, 32)), (2, (96, 64, 64)), (4, (200, 120, 32)), (4, (130, 70, 64))])
def test_gather_assembles_frame(world, size):
    """assemble() and assemble_device() (the path bench.py uses) for world 2 and 4 under gloo, round robin and re-dealt by
    cost, including tile counts that do not divide by the world size and ranks that own fewer tiles than others."""
    import torch.multiprocessing as mp
    width, height, tile = size
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, width, height, tile, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


def test_cost_weighted_deal_balances_a_peaked_cost_map():
    rng = np.random.default_rng(3)
    costs = rng.exponential(1.0, 510) ** 3 + 0.05      # a few tiles carry most of the work, like the centre of config 3
    for world in (2, 4, 8):
        parts = tiling.assign_tiles_by_cost(costs, world)
        assert sorted(np.concatenate(parts).tolist()) == list(range(510))
        loads = np.array([costs[p].sum() for p in parts])
        rr = np.array([costs[r::world].sum() for r in range(world)])
        assert loads.max() <= max(rr.max(), costs.max()) + 1e-9
        assert loads.max() <= 1.02 * max(loads.mean(), costs.max())
        assert all(np.all(np.diff(p) > 0) for p in parts)                      # each rank's list stays in Morton order
        assert [p.tolist() for p in tiling.assign_tiles_by_cost(costs, world)] == [p.tolist() for p in parts]   # deterministic


def test_bench_dry_run_collective_sequence_world2():
    """`bench.py --gpus 2 --dry-run` under torch.distributed.run: the exact collective sequence of a multi-GPU run (counter
    all-reduce, cost all-reduce + re-deal, K x gather, max-over-ranks all-reduce, barriers) on gloo / CPU tensors."""
    import json
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
                        "--warmup", "1", "--dry-run"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["scaling"] == "strong" and j["cpu_baseline"] is None


def test_bench_gpus2_dry_run_launches_itself():
    """The driver's command form (`python bench.py --gpus N ...`, no torchrun, WORLD_SIZE unset) must not exit: bench.py starts the N
    ranks itself under torch.distributed.run and forwards rank 0's line, which names the path that ran."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly ONE JSON line on stdout"
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1
    assert j["launch"].startswith("self-launched") and j["launch_attempts"] == []
    mg = j["multi_gpu"]
    assert mg["ranks_observed"] == 2 and len(mg["tiles_per_rank"]) == 2 and sum(mg["tiles_per_rank"]) == 30 * 17
    assert len(mg["render_ms_per_rank"]) == 2 and len(mg["gather_ms_per_rank"]) == 2


def test_bench_self_launch_reports_every_failed_path(tmp_path):
    """When the torch.distributed.run launch fails, the dry run (which has no one-process form) must fail LOUDLY with the attempt's
    stderr, not silently; here the launch is broken on purpose with an impossible rendezvous timeout of the child."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["LV_BENCH_LAUNCH_TIMEOUT"] = "0.01"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--dry-run"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "torch.distributed.run launch failed" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_detile_helper():
    tiles = tiling.make_tiles(50, 30, 16)
    px = np.zeros((len(tiles), 16, 16, 4), dtype=np.uint8)
    for i, (x0, y0) in enumerate(tiles):
        yy, xx = np.meshgrid(np.arange(16) + int(y0), np.arange(16) + int(x0), indexing="ij")
        px[i] = _pattern(xx, yy)
    frame = tiling.detile(px, tiles, 50, 30, 16)
    yy, xx = np.meshgrid(np.arange(30), np.arange(50), indexing="ij")
    assert np.array_equal(frame, _pattern(xx, yy))


_RCCL_WORLD1 = r"""
import os, sys, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
try:
    dist.barrier()
    c = torch.arange(6, dtype=torch.float64, device=dev)
    dist.all_reduce(c)                                   # counters of the instrumented frame
    t = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)             # max-over-ranks time
    out = (torch.arange(3 * 64 * 64 * 4, device=dev) % 251).to(torch.uint8).view(3, 64, 64, 4)
    got = [torch.empty_like(out)]
    dist.gather(out, gather_list=got, dst=0)             # the one data-path collective: RGBA8 tiles to rank 0
    torch.cuda.synchronize()
    ok = bool(torch.equal(got[0], out)) and float(t.item()) == 1.25 and c.tolist() == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0]
    print("RCCL_WORLD1_OK" if ok else "RCCL_WORLD1_MISMATCH", dist.get_backend(), dist.get_world_size())
finally:
    dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_torch_rccl_process_group_on_one_gpu():
    """The collectives bench.py / tiling.ShardedFrame issue with N > 1 (barrier, sum and max all-reduce of float64 scalars, gather of
    uint8 tiles to rank 0) through torch.distributed's nccl (= RCCL) backend on a 1-rank communicator: everything of the N > 1 path a
    single MI355X can exercise -- the RCCL library torch ships, communicator set-up under this image's HSA settings, uint8 gather."""
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _RCCL_WORLD1], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
