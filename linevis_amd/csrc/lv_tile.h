// lv_tile.h -- what the tile kernels (lv_render.hip, lv_mlat.hip) share: counter flushes and the workgroup -> pixel map.
#ifndef LV_TILE_H
#define LV_TILE_H

#include "lv_device.h"
#include "lv_trace.h"

namespace {

__device__ __forceinline__ void lv_flush_max_nodes(const LvCounters& c, LvDevCounters* dc) {
    uint32_t m = uint32_t(c.nodes);
#pragma unroll
    for (int ofs = 32; ofs > 0; ofs >>= 1) m = max(m, (uint32_t)__shfl_xor(m, ofs, 64));
    if (lv_lane() == 0) atomicMax(&dc->maxNodesPerPixel, m);
}
__device__ __forceinline__ void lv_flush_counters(const LvCounters& c, LvDevCounters* dc, bool aoRays = false) {
    unsigned long long r = lv_wave_sum_u64(c.rays), n = lv_wave_sum_u64(c.nodes), p = lv_wave_sum_u64(c.prims),
                       h = lv_wave_sum_u64(c.hits);
    if (lv_lane() == 0) {
        atomicAdd(&dc->rays, r);
        atomicAdd(&dc->nodes, n);
        atomicAdd(&dc->prims, p);
        atomicAdd(&dc->hits, h);
        if (aoRays) {
            atomicAdd(&dc->aoRays, r);
            atomicAdd(&dc->aoNodes, n);
            atomicAdd(&dc->aoPrims, p);
        }
    }
}

// Pixel of this thread (see lv_block_pixel for the pixel <-> wave map).  The dispatcher deals workgroups to the 8 XCDs
// round-robin (block b -> XCD b % 8, each with a private L2).  Blocks are regrouped so that runs of LV_XCD_GROUP consecutive logical blocks (one
// 64x64 tile = 16 blocks) land on the same XCD, and consecutive groups go to consecutive XCDs: spatial neighbours share
// an L2 while every XCD still gets an even share of the dense and the empty parts of the picture (a contiguous 1/8
// of the frame per XCD left 7 XCDs idle behind the one that owned the centre).  Speed only; never correctness.
#define LV_XCD_GROUP 16u
struct LvPixel {
    uint32_t x, y;       // viewport pixel
    uint32_t outIndex;   // index into the tile-major output
    uint32_t group;      // 64x64-pixel group of the launch: tile index * groups per tile + group inside the tile
    bool inTile, inView;
};

__device__ __forceinline__ bool lv_block_pixel(const LvUniforms& U, const LvTiles& T, LvPixel& p, uint32_t blockId) {
    const uint32_t blocksPerTile = T.blocksX * T.blocksY;
    const uint32_t nb = T.numTiles * blocksPerTile;
    const uint32_t xcd = blockId % 8u, j = blockId / 8u;   // j-th block this XCD receives
    const uint32_t b = ((j / LV_XCD_GROUP) * 8u + xcd) * LV_XCD_GROUP + (j % LV_XCD_GROUP);
    if (b >= nb) { p.inTile = false; p.inView = false; p.group = 0u; return false; }
    // the 16 blocks of slot b / 16 render the group the dispatch order names (heaviest first, LvTiles::groupOrder)
    const uint32_t bg = T.groupOrder ? T.groupOrder[b / LV_XCD_GROUP] * LV_XCD_GROUP + (b % LV_XCD_GROUP) : b;
    const uint32_t tile = bg / blocksPerTile, rem = bg % blocksPerTile;
    // A tile is cut into 64x64-pixel groups of 16 blocks (T.blocksX / T.blocksY are multiples of 4).  Inside a group the
    // 64 waves do NOT own 8x8 patches: wave W takes pixel (W & 7, W >> 3) of each of the group's 8x8 cells, i.e. 64 pixels
    // spread over the whole group at stride 8.  The cost of a pixel varies by orders of magnitude over a few dozen
    // pixels (a ray through the core of a bundle vs. one that misses it), and a wave runs as long as the SUM of its
    // pixels' work / 64 -- with contiguous patches the waves over the core set the duration of the whole kernel however
    // few tiles a GPU owns; interleaved, every wave of a group carries the same mix (measured: k_ao_primary 0.34 -> 0.27,
    // k_render_rt 0.45 -> 0.37, k_ppll_gather 3.0 -> 2.0 ms; with 1/8 of the tiles 0.30 -> 0.13 and 0.45 -> 0.23 ms).
    const uint32_t groupsX = T.blocksX / 4u;
    const uint32_t group = rem / 16u, W = (rem % 16u) * 4u + (threadIdx.x >> 6), l = threadIdx.x & 63u;
#ifdef LV_PIXEL_PATCH // experiment: contiguous 8x8 patch per wave (coherent rays) instead of the stride-8 interleave (balanced waves)
    const uint32_t lx = (group % groupsX) * 64u + (W & 7u) * 8u + (l & 7u);
    const uint32_t ly = (group / groupsX) * 64u + (W >> 3) * 8u + (l >> 3);
#else
    const uint32_t lx = (group % groupsX) * 64u + (l & 7u) * 8u + (W & 7u);
    const uint32_t ly = (group / groupsX) * 64u + (l >> 3) * 8u + (W >> 3);
#endif
    p.inTile = lx < T.tileW && ly < T.tileH;
    p.x = T.tilesXY[2 * tile] + lx;
    p.y = T.tilesXY[2 * tile + 1] + ly;
    p.inView = p.inTile && p.x < U.width && p.y < U.height;
    p.outIndex = (tile * T.tileH + ly) * T.tileW + lx;
    p.group = tile * (blocksPerTile / 16u) + group;
    return true;
}

__device__ __forceinline__ bool lv_block_pixel(const LvUniforms& U, const LvTiles& T, LvPixel& p) {
    return lv_block_pixel(U, T, p, blockIdx.x);
}

// Cost of a group for the next frame's dispatch order: every wave adds the time it was resident.
__device__ __forceinline__ unsigned long long lv_group_clock() { return wall_clock64(); }
__device__ __forceinline__ void lv_group_cost_add(const LvTiles& T, const LvPixel& p, unsigned long long t0) {
    if (T.groupCost && (threadIdx.x & 63u) == 0u) {
        const unsigned long long dt = wall_clock64() - t0;
        atomicAdd(&T.groupCost[p.group], uint32_t(dt > 0x00FFFFFFull ? 0x00FFFFFFull : dt));
    }
}

} // namespace

#endif
