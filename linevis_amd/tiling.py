"""Screen-tile sharding of a frame over the GPUs of one node.

Every pixel depends only on the read-only scene and its own RNG seed tea(x + y * W, ...) with GLOBAL pixel
coordinates (TubeRayTracing.glsl:215-217, VulkanRayTracedAmbientOcclusion.glsl:188,289-291), so tiles are
independent units: each rank holds a full scene replica + LBVH, renders its tiles (dealt round-robin along a
Morton order at first, then re-dealt by measured cost: ShardedFrame.rebalance) with one `lv_render_tiles_device` call, and ONE gather of RGBA8 tiles over
RCCL/xGMI assembles the frame on rank 0.  There is no other data-path collective.

torch is plumbing here: device buffers, the current HIP stream and torch.distributed (backend "nccl" = RCCL).
"""
import numpy as np


def _morton2(x, y):
    def part(v):
        v = np.asarray(v, dtype=np.uint64)
        v = (v | (v << 16)) & np.uint64(0x0000FFFF0000FFFF)
        v = (v | (v << 8)) & np.uint64(0x00FF00FF00FF00FF)
        v = (v | (v << 4)) & np.uint64(0x0F0F0F0F0F0F0F0F)
        v = (v | (v << 2)) & np.uint64(0x3333333333333333)
        v = (v | (v << 1)) & np.uint64(0x5555555555555555)
        return v
    return part(x) | (part(y) << np.uint64(1))


def make_tiles(width, height, tile=64):
    """Origins (x0, y0) of the tile x tile rectangles covering width x height, in Morton order."""
    nx, ny = -(-width // tile), -(-height // tile)
    gx, gy = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
    gx, gy = gx.reshape(-1), gy.reshape(-1)
    order = np.argsort(_morton2(gx, gy), kind="stable")
    return np.stack([gx[order] * tile, gy[order] * tile], axis=1).astype(np.uint32)


def assign_tiles(tiles, rank, world_size):
    """Round-robin deal: rank r gets tiles r, r + N, r + 2N, ..."""
    return np.ascontiguousarray(tiles[rank::world_size])


def tiles_per_rank(num_tiles, world_size):
    return -(-num_tiles // world_size)


def assign_tiles_by_cost(costs, world_size):
    """Cost-weighted deal (longest processing time first): tiles in order of falling cost, each to the rank with the least
    cost so far (ties: fewer tiles, then lower rank).  Deterministic, so every rank computes the same answer from the same cost
    vector.  Returns a list of index arrays into the tile list, each in ascending (= Morton) order."""
    costs = np.asarray(costs, dtype=np.float64)
    order = np.lexsort((np.arange(len(costs)), -costs))
    load = np.zeros(world_size, dtype=np.float64)
    count = np.zeros(world_size, dtype=np.int64)
    owner = np.zeros(len(costs), dtype=np.int64)
    for t in order:
        r = int(np.lexsort((np.arange(world_size), count, load))[0])
        owner[t] = r
        load[r] += costs[t]
        count[r] += 1
    return [np.flatnonzero(owner == r) for r in range(world_size)]


def detile(tile_pixels, tiles_xy, width, height, tile):
    """tile_pixels: [n, tile, tile, 4] uint8 (numpy) -> frame [height, width, 4]."""
    frame = np.zeros((height, width, 4), dtype=np.uint8)
    for i, (x0, y0) in enumerate(np.asarray(tiles_xy)):
        x0, y0 = int(x0), int(y0)
        w, h = min(tile, width - x0), min(tile, height - y0)
        if w > 0 and h > 0:
            frame[y0:y0 + h, x0:x0 + w] = tile_pixels[i, :h, :w]
    return frame


class ShardedFrame:
    """Per-rank state of a tile-sharded frame: tile list, device output buffer, gather buffers.

    The deal starts as round robin along the Morton order; `rebalance()` re-deals the tiles by measured cost (the RTAO hit
    pixels per tile of the previous frame): the heavy tiles sit in the middle of the picture, and a frame is as slow as its
    slowest rank."""

    def __init__(self, width, height, tile, rank, world_size, device):
        self.width, self.height, self.tile = int(width), int(height), int(tile)
        self.rank, self.world = int(rank), int(world_size)
        self.all_tiles = make_tiles(width, height, tile)
        self.device = device
        self._set_assignment([np.arange(r, len(self.all_tiles), self.world) for r in range(self.world)])

    def _set_assignment(self, index_lists):
        import torch
        t = self.tile
        self.assignment = [np.asarray(ix, dtype=np.int64) for ix in index_lists]
        assert sorted(np.concatenate(self.assignment).tolist()) == list(range(len(self.all_tiles)))
        self.local_tiles = np.ascontiguousarray(self.all_tiles[self.assignment[self.rank]])
        self.slots = max(len(ix) for ix in self.assignment)  # equal-sized gather pieces (padded for ranks with fewer tiles)
        self.out = torch.zeros((self.slots, t, t, 4), dtype=torch.uint8, device=self.device)
        self.gathered = None
        if self.rank == 0 and self.world > 1:
            self.gathered = [torch.zeros_like(self.out) for _ in range(self.world)]
        # position of tile (gy, gx) inside the rank-major concatenation of the gathered pieces
        nx = -(-self.width // t)
        perm = np.zeros(len(self.all_tiles), dtype=np.int64)
        for r, ix in enumerate(self.assignment):
            for i, ti in enumerate(ix):
                x0, y0 = self.all_tiles[ti]
                perm[(int(y0) // t) * nx + int(x0) // t] = r * self.slots + i
        self._perm = torch.from_numpy(perm).to(self.device)

    def rebalance(self, local_costs, base_cost=1.0):
        """Re-deal the tiles by cost.  local_costs[i] = measured cost of this rank's i-th tile (e.g. capi.Context.ao_tile_costs());
        the per-tile vector is completed across ranks with ONE all-reduce (control plane, outside any timed region), then
        every rank computes the same longest-processing-time-first assignment.  base_cost: fixed cost of a tile that traces
        nothing but its primary rays, in the same unit."""
        import torch
        costs = np.zeros(len(self.all_tiles), dtype=np.float64)
        lc = np.asarray(local_costs, dtype=np.float64)
        assert len(lc) == len(self.assignment[self.rank])
        costs[self.assignment[self.rank]] = lc
        if self.world > 1:
            import torch.distributed as dist
            t = torch.from_numpy(costs).to(self.device)
            dist.all_reduce(t)
            costs = t.cpu().numpy()
        self._set_assignment(assign_tiles_by_cost(costs + base_cost, self.world))
        return costs

    def redeal(self, costs, base_cost=1.0):
        """Apply a complete per-tile cost vector (as returned by rebalance() of another ShardedFrame of the same geometry)."""
        self._set_assignment(assign_tiles_by_cost(np.asarray(costs, dtype=np.float64) + base_cost, self.world))

    def render_local(self, render_tiles_fn):
        """render_tiles_fn(out_tensor, tiles_xy[n,2], tile_w, tile_h) fills out_tensor[:n]."""
        if len(self.local_tiles):
            render_tiles_fn(self.out, self.local_tiles, self.tile, self.tile)

    def gather(self):
        """One gather of RGBA8 tiles to rank 0 (RCCL over xGMI on GPUs, gloo on CPU)."""
        if self.world == 1:
            return
        import torch.distributed as dist
        dist.gather(self.out, gather_list=self.gathered if self.rank == 0 else None, dst=0)

    def assemble_device(self):
        """Rank 0: de-tile on the device with three tensor ops -> uint8 tensor [H, W, 4] (other ranks: None)."""
        if self.rank != 0:
            return None
        import torch
        t = self.tile
        nx, ny = -(-self.width // t), -(-self.height // t)
        pieces = self.out if self.world == 1 else torch.cat(self.gathered, dim=0)
        grid = pieces.index_select(0, self._perm).view(ny, nx, t, t, 4)
        frame = grid.permute(0, 2, 1, 3, 4).reshape(ny * t, nx * t, 4)
        return frame[:self.height, :self.width]

    def assemble(self):
        """Rank 0: de-tile the gathered pieces into the frame (numpy [H, W, 4]); other ranks: None."""
        if self.rank != 0:
            return None
        pieces = [self.out] if self.world == 1 else self.gathered
        frame = np.zeros((self.height, self.width, 4), dtype=np.uint8)
        for r, piece in enumerate(pieces):
            tiles = self.all_tiles[self.assignment[r]]
            px = piece[:len(tiles)].cpu().numpy()
            for i, (x0, y0) in enumerate(tiles):
                x0, y0 = int(x0), int(y0)
                w, h = min(self.tile, self.width - x0), min(self.tile, self.height - y0)
                frame[y0:y0 + h, x0:x0 + w] = px[i, :h, :w]
        return frame


def hip_render_tiles_fn(ctx, mode, wait_for_consumer=True):
    """Adapter: renders tiles with a capi.Context into a torch uint8 tensor on the context's device.

    The context gets a torch stream of its own (a real handle: lv_set_stream(NULL) would mean "the context's private stream",
    which is what torch's default stream -- handle 0 -- would select, leaving the kernels unordered against the gather).
    Every call is bracketed by stream waits: the render stream waits for what the caller's current stream has queued so far
    (the consumers of the previous frame's output), and the caller's stream then waits for the render -- so the gather and the
    de-tiling that follow are ordered after the kernels whatever stream the caller runs on, without a host synchronisation.
    wait_for_consumer=False drops the first wait (frames in flight on several contexts with separate output buffers)."""
    import torch
    stream = torch.cuda.Stream(device=torch.device("cuda", ctx.device))
    ctx.set_stream(stream.cuda_stream)

    def fn(out_tensor, tiles_xy, tile_w, tile_h):
        cur = torch.cuda.current_stream()
        if wait_for_consumer:
            stream.wait_stream(cur)
        ctx.render_tiles_device(out_tensor.data_ptr(), tiles_xy, tile_w, tile_h, mode=mode)
        cur.wait_stream(stream)
    fn.stream = stream
    return fn


class FramePipeline:
    """Frames in flight: F slots, each a (ShardedFrame, render function of its own context / scene replica / HIP stream), used
    round robin.  Frame k + 1 is queued on the other slot's stream while frame k's AO sample kernel still runs, so the
    latency-bound tile kernels of one frame (primary rays: ~0.13 + 0.23 ms whatever the tile count) overlap the throughput-
    bound kernel of the other.  Worth it when a rank owns a fraction of the tiles (8 ranks: 1.03 -> 0.81 ms per frame per rank
    on config 3); with all tiles on one GPU the AO kernel already fills the chip (no gain), so F = 1 there.
    Every frame is still a complete frame; only the order in which the GPU works through the kernels of consecutive frames
    changes.  A slot's buffers are reused only after the consumer of their previous frame (gather + de-tiling, queued on the
    caller's stream) has finished: the slot's stream waits for an event recorded behind that consumer."""

    def __init__(self, slots):
        self.slots = list(slots)      # [(ShardedFrame, render_tiles_fn)]
        self.k = 0
        self._consumed = [None] * len(self.slots)

    def submit(self):
        """Queue one complete frame (render -> gather -> de-tile on rank 0); returns rank 0's frame tensor (else None)."""
        import torch
        i = self.k % len(self.slots)
        sf, fn = self.slots[i]
        stream = getattr(fn, "stream", None)
        if self._consumed[i] is not None and stream is not None:
            stream.wait_event(self._consumed[i])
        sf.render_local(fn)
        sf.gather()
        frame = sf.assemble_device()
        if stream is not None:
            ev = torch.cuda.Event()
            ev.record()
            self._consumed[i] = ev
        self.k += 1
        return frame
