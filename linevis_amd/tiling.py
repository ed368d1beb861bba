"""Screen-tile sharding of a frame over the GPUs of one node.

Every pixel depends only on the read-only scene and its own RNG seed tea(x + y * W, ...) with GLOBAL pixel
coordinates (TubeRayTracing.glsl:215-217, VulkanRayTracedAmbientOcclusion.glsl:188,289-291), so tiles are
independent units: each rank holds a full scene replica + LBVH, renders its tiles (dealt round-robin along a
Morton order for load balance) with one `lv_render_tiles_device` call, and ONE gather of RGBA8 tiles over
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
    """Per-rank state of a tile-sharded frame: tile list, device output buffer, gather buffers."""

    def __init__(self, width, height, tile, rank, world_size, device):
        import torch
        self.width, self.height, self.tile = int(width), int(height), int(tile)
        self.rank, self.world = int(rank), int(world_size)
        self.all_tiles = make_tiles(width, height, tile)
        self.local_tiles = assign_tiles(self.all_tiles, rank, world_size)
        self.slots = tiles_per_rank(len(self.all_tiles), world_size)  # equal-sized gather pieces
        self.device = device
        self.out = torch.zeros((self.slots, tile, tile, 4), dtype=torch.uint8, device=device)
        self.gathered = None
        if rank == 0 and world_size > 1:
            self.gathered = [torch.zeros_like(self.out) for _ in range(world_size)]

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
        if not hasattr(self, "_perm"):
            # position of tile (gy, gx) inside the rank-major concatenation of the gathered pieces
            perm = np.zeros(nx * ny, dtype=np.int64)
            for r in range(self.world):
                for i, (x0, y0) in enumerate(assign_tiles(self.all_tiles, r, self.world)):
                    perm[(int(y0) // t) * nx + int(x0) // t] = r * self.slots + i
            self._perm = torch.from_numpy(perm).to(self.device)
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
            tiles = assign_tiles(self.all_tiles, r, self.world)
            px = piece[:len(tiles)].cpu().numpy()
            for i, (x0, y0) in enumerate(tiles):
                x0, y0 = int(x0), int(y0)
                w, h = min(self.tile, self.width - x0), min(self.tile, self.height - y0)
                frame[y0:y0 + h, x0:x0 + w] = px[i, :h, :w]
        return frame


def hip_render_tiles_fn(ctx, mode):
    """Adapter: renders tiles with a capi.Context into a torch uint8 tensor on the context's device.  The context
    is switched to torch's current stream once, so the gather that follows is ordered after the kernels."""
    import torch
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    def fn(out_tensor, tiles_xy, tile_w, tile_h):
        ctx.render_tiles_device(out_tensor.data_ptr(), tiles_xy, tile_w, tile_h, mode=mode)
    return fn
