// lv_bvh.hip -- GPU LBVH build over line-segment capsules.
//
// Replaces the driver-side acceleration-structure build of the reference:
//   LineData::getTubeAabbBottomLevelAS      src/LineData/LineData.cpp:879-907
//   LineData::getRayTracingTubeAabbTopLevelAS  src/LineData/LineData.cpp:1057-1075
// over the per-segment AABBs min(p0,p1)-r .. max(p0,p1)+r of LineDataFlow.cpp:2223-2234.
//
// Pipeline (all on the context's stream):
//   k_seg_boxes   segment AABBs (+ conservative pad) and scene bounds (wave reduce + 6 atomics per wave)
//   k_morton      63-bit Morton keys of box centroids
//   radix sort    rocprim::radix_sort_pairs (key, segment)
//   k_leaves      32-byte segment records + leaf boxes written in Morton order
//   k_karras      Karras 2012 topology: one thread per internal node
//   k_refit_pass  bottom-up AABB + height, one pass per tree level (kernel boundaries are the synchronisation)
//   k_collapse_*  greedy area-guided collapse into 64-byte compressed 4-wide nodes (8-bit child boxes + references), one
//                 BFS level of the wide tree per pass, rocPRIM exclusive scan for the deterministic node numbering
#include <algorithm>
#include <cstring>
#include <utility>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "lv_internal.h"

namespace {

// Scene bounds: the box kernels run grid-stride and merge per WORKGROUP (wave reduce -> LDS -> 6 atomics on the six
// encoded words); one set of atomics per wave of a one-primitive-per-thread launch was 94 k contended atomics = half of
// k_seg_boxes at 1 M segments.  Every thread of the block must call it.
__device__ __forceinline__ void lv_block_bounds(const float mn[3], const float mx[3], uint32_t* boundsOrd) {
    __shared__ float s_red[6][LV_BLOCK / LV_WAVE];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float a = lv_wave_min(mn[k]), b = lv_wave_max(mx[k]);
        if (lv_lane() == 0) { s_red[k][threadIdx.x >> 6] = a; s_red[3 + k][threadIdx.x >> 6] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = s_red[threadIdx.x][0];
        for (int w = 1; w < LV_BLOCK / LV_WAVE; w++)
            v = threadIdx.x < 3 ? fminf(v, s_red[threadIdx.x][w]) : fmaxf(v, s_red[threadIdx.x][w]);
        if (threadIdx.x < 3) atomicMin(&boundsOrd[threadIdx.x], lv_f2ord(v));
        else atomicMax(&boundsOrd[threadIdx.x], lv_f2ord(v));
    }
}

__global__ __launch_bounds__(LV_BLOCK) void k_seg_boxes(const lv_line_point* __restrict__ points,
                                                        const uint32_t* __restrict__ segIdx, uint32_t nSeg, float radius,
                                                        float pad, float* __restrict__ boxOrig,
                                                        uint32_t* __restrict__ boundsOrd) {
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (uint32_t s = blockIdx.x * LV_BLOCK + threadIdx.x; s < nSeg; s += gridDim.x * LV_BLOCK) {
        const float* p0 = points[segIdx[2 * s]].linePosition;
        const float* p1 = points[segIdx[2 * s + 1]].linePosition;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float lo = (fminf(p0[k], p1[k]) - radius) - pad, hi = (fmaxf(p0[k], p1[k]) + radius) + pad;
            boxOrig[6 * size_t(s) + k] = lo;
            boxOrig[6 * size_t(s) + 3 + k] = hi;
            mn[k] = fminf(mn[k], lo);
            mx[k] = fmaxf(mx[k], hi);
        }
    }
    lv_block_bounds(mn, mx, boundsOrd);
}

__device__ __forceinline__ uint64_t expandBits21(uint64_t v) {
    v &= 0x1fffffull;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

__global__ __launch_bounds__(LV_BLOCK) void k_morton(const float* __restrict__ boxOrig, uint32_t nSeg,
                                                     const uint32_t* __restrict__ boundsOrd,
                                                     uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    uint32_t s = blockIdx.x * LV_BLOCK + threadIdx.x;
    if (s >= nSeg) return;
    uint64_t q[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float smin = lv_ord2f(boundsOrd[k]), smax = lv_ord2f(boundsOrd[3 + k]);
        float c = 0.5f * (boxOrig[6 * size_t(s) + k] + boxOrig[6 * size_t(s) + 3 + k]);
        float u = (c - smin) / fmaxf(smax - smin, 1e-30f);
        u = fminf(fmaxf(u, 0.0f), 1.0f);
        q[k] = uint64_t(fminf(2097151.0f, u * 2097152.0f));
    }
    keys[s] = (expandBits21(q[0]) << 2) | (expandBits21(q[1]) << 1) | expandBits21(q[2]);
    vals[s] = s;
}

__global__ __launch_bounds__(LV_BLOCK) void k_leaves(const lv_line_point* __restrict__ points,
                                                     const uint32_t* __restrict__ segIdx, const float* __restrict__ boxOrig,
                                                     const uint32_t* __restrict__ sortedVals, uint32_t nSeg,
                                                     float4* __restrict__ segs, float4* __restrict__ segAxis,
                                                     uint32_t* __restrict__ leafSeg, uint32_t* __restrict__ segToLeaf,
                                                     float* __restrict__ leafBox, float4* __restrict__ prismFrames) {
    uint32_t i = blockIdx.x * LV_BLOCK + threadIdx.x;
    if (i >= nSeg) return;
    uint32_t s = sortedVals[i];
    const lv_line_point& a = points[segIdx[2 * s]];
    const lv_line_point& b = points[segIdx[2 * s + 1]];
    segs[2 * size_t(i)] = make_float4(a.linePosition[0], a.linePosition[1], a.linePosition[2], a.lineAttribute);
    segs[2 * size_t(i) + 1] = make_float4(b.linePosition[0], b.linePosition[1], b.linePosition[2], b.lineAttribute);
    // normalize(p1 - p0) of rayTubeIntersection, computed once per segment instead of once per leaf test: the same norm3 on the
    // same operands gives the same bits (3 subtractions, 1 square root and 3 IEEE divisions = ~50 of the capsule test's ~350 instructions)
    const f3 td = norm3(mk3(b.linePosition[0], b.linePosition[1], b.linePosition[2]) - mk3(a.linePosition[0], a.linePosition[1], a.linePosition[2]));
    segAxis[i] = make_float4(td.x, td.y, td.z, 0.0f);
    leafSeg[i] = s;
    segToLeaf[s] = i;
    // frames of the two line points in leaf order, for the vertex stage of the rasterised prism (lv_prism.h): {tangent, point index}
    // {normal, lineStartIndex} per point -- with `segs` the whole 48-B records, 96 contiguous bytes per (ray, segment) test
    const uint32_t ia = segIdx[2 * s], ib = segIdx[2 * s + 1];
    prismFrames[4 * size_t(i) + 0] = make_float4(a.lineTangent[0], a.lineTangent[1], a.lineTangent[2], __uint_as_float(ia));
    prismFrames[4 * size_t(i) + 1] = make_float4(a.lineNormal[0], a.lineNormal[1], a.lineNormal[2], __uint_as_float(a.lineStartIndex));
    prismFrames[4 * size_t(i) + 2] = make_float4(b.lineTangent[0], b.lineTangent[1], b.lineTangent[2], __uint_as_float(ib));
    prismFrames[4 * size_t(i) + 3] = make_float4(b.lineNormal[0], b.lineNormal[1], b.lineNormal[2], __uint_as_float(b.lineStartIndex));
#pragma unroll
    for (int k = 0; k < 6; k++) leafBox[6 * size_t(i) + k] = boxOrig[6 * size_t(s) + k];
}

// triangle tubes: padded AABB of every triangle (the same box the ray-triangle test clips t against)
// Leaves of the triangle LBVH are GROUPS of `group` consecutive triangles of the input order (the tessellator emits the two
// triangles of a tube face, then the next face, then the next segment: consecutive triangles are neighbours): box of group g =
// union of the padded boxes of its triangles.  A tree over N / 4 leaves is one wide level lower and a quarter the size of a tree
// over N (config 3: 12 M triangles, 0.26 GB of nodes that miss the caches); the closest hit does not depend on the topology.
__global__ __launch_bounds__(LV_BLOCK) void k_tri_boxes(const lv_tube_vertex* __restrict__ verts,
                                                        const uint32_t* __restrict__ triIdx, uint32_t nTri, uint32_t group,
                                                        uint32_t nLeaves, float pad, float* __restrict__ boxOrig,
                                                        uint32_t* __restrict__ boundsOrd) {
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (uint32_t g = blockIdx.x * LV_BLOCK + threadIdx.x; g < nLeaves; g += gridDim.x * LV_BLOCK) {
        float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        for (uint32_t s = g * group; s < min(nTri, (g + 1u) * group); s++) {
            const float* a = verts[triIdx[3 * size_t(s)]].vertexPosition;
            const float* b = verts[triIdx[3 * size_t(s) + 1]].vertexPosition;
            const float* c = verts[triIdx[3 * size_t(s) + 2]].vertexPosition;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                lo[k] = fminf(lo[k], fminf(fminf(a[k], b[k]), c[k]) - pad);
                hi[k] = fmaxf(hi[k], fmaxf(fmaxf(a[k], b[k]), c[k]) + pad);
            }
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            boxOrig[6 * size_t(g) + k] = lo[k];
            boxOrig[6 * size_t(g) + 3 + k] = hi[k];
            mn[k] = fminf(mn[k], lo[k]);
            mx[k] = fmaxf(mx[k], hi[k]);
        }
    }
    lv_block_bounds(mn, mx, boundsOrd);
}

// PAIR RECORDS (64 B per leaf, round 6): the two triangles of a leaf of `triangle_leaf_size` 2 almost always share two vertices
// -- the tessellator emits a tube face as (a, b, c)(a, c, d), a cap quad as (a, b, c)(b, d, c), two fan triangles as (p, x1, x0)
// (p, x2, x1) -- so the leaf stores FOUR vertices instead of 2 x 3: {q0.xyz, index of the first triangle}{q1.xyz, code}{q2.xyz, 0}
// {q3.xyz, 0}; triangle 0 = (q0, q1, q2) in its own vertex order, triangle 1 = (q[code & 3], q[code >> 2 & 3], q[code >> 4 & 3]) in ITS
// own order (the ray-triangle test is evaluated on exactly the operands of the 48-B records: same bits).  A pair is encodable iff the
// second triangle has at most one vertex INDEX the first one lacks; k_tri_pairs_check says whether every pair of the mesh is (a
// caller's arbitrary mesh may not be: the build then keeps the 48-B records).  1 M segments: 579 -> 386 MB of leaf data.
__device__ __forceinline__ bool lv_tri_pair_code(const uint32_t* __restrict__ triIdx, uint32_t s, uint32_t nTri, uint32_t& code,
                                                 uint32_t& extra) {
    const uint32_t a0 = triIdx[3 * size_t(s)], a1 = triIdx[3 * size_t(s) + 1], a2 = triIdx[3 * size_t(s) + 2];
    code = 0u; extra = a0;
    if (s + 1u >= nTri) { code = 0xFFFFFFFFu; return true; } // odd triangle count: the last leaf's second slot is empty
    bool haveExtra = false;
    for (int k = 0; k < 3; k++) {
        const uint32_t b = triIdx[3 * size_t(s + 1u) + k];
        uint32_t sel;
        if (b == a0) sel = 0u;
        else if (b == a1) sel = 1u;
        else if (b == a2) sel = 2u;
        else {
            if (haveExtra && b != extra) return false;
            haveExtra = true; extra = b; sel = 3u;
        }
        code |= sel << (2 * k);
    }
    return true;
}
__global__ __launch_bounds__(LV_BLOCK) void k_tri_pairs_check(const uint32_t* __restrict__ triIdx, uint32_t nTri, uint32_t nLeaves,
                                                              uint32_t* __restrict__ notEncodable) {
    bool bad = false;
    for (uint32_t g = blockIdx.x * LV_BLOCK + threadIdx.x; g < nLeaves; g += gridDim.x * LV_BLOCK) {
        uint32_t code, extra;
        bad |= !lv_tri_pair_code(triIdx, 2u * g, nTri, code, extra);
    }
    if (__ballot(bad) && (threadIdx.x & 63u) == 0u) atomicOr(notEncodable, 1u);
}

// 48-byte triangle records, `group` per leaf, leaves in Morton order: {v0.xyz, original triangle index}{v1.xyz, 0}{v2.xyz, 0};
// the slots an incomplete last group leaves empty hold NaN vertices (the test rejects them: every comparison with NaN fails).
// PAIRS: the 64-byte pair records above (group == 2, every pair encodable).
template <bool PAIRS>
__global__ __launch_bounds__(LV_BLOCK) void k_tri_leaves(const lv_tube_vertex* __restrict__ verts,
                                                         const uint32_t* __restrict__ triIdx,
                                                         const float* __restrict__ boxOrig,
                                                         const uint32_t* __restrict__ sortedVals, uint32_t nTri, uint32_t group,
                                                         uint32_t nLeaves, float4* __restrict__ tris, float* __restrict__ leafBox) {
    uint32_t i = blockIdx.x * LV_BLOCK + threadIdx.x;
    if (i >= nLeaves) return;
    const uint32_t g = sortedVals[i];
    const float nan = __uint_as_float(0x7FC00000u);
    if (PAIRS) {
        const uint32_t s = 2u * g;
        uint32_t code, extra;
        lv_tri_pair_code(triIdx, s, nTri, code, extra);
        const float* a = verts[triIdx[3 * size_t(s)]].vertexPosition;
        const float* b = verts[triIdx[3 * size_t(s) + 1]].vertexPosition;
        const float* c = verts[triIdx[3 * size_t(s) + 2]].vertexPosition;
        float4* rec = tris + 4 * size_t(i);
        rec[0] = make_float4(a[0], a[1], a[2], __uint_as_float(s));
        rec[2] = make_float4(c[0], c[1], c[2], 0.0f);
        if (code == 0xFFFFFFFFu) { // no second triangle: (q3, q3, q3) of NaN vertices is rejected by every comparison of the test
            rec[1] = make_float4(b[0], b[1], b[2], __uint_as_float(0x3Fu));
            rec[3] = make_float4(nan, nan, nan, 0.0f);
        } else {
            const float* d = verts[extra].vertexPosition;
            rec[1] = make_float4(b[0], b[1], b[2], __uint_as_float(code));
            rec[3] = make_float4(d[0], d[1], d[2], 0.0f);
        }
    } else {
        for (uint32_t j = 0; j < group; j++) {
            const uint32_t s = g * group + j;
            float4* rec = tris + 3 * (size_t(i) * group + j);
            if (s < nTri) {
                const float* a = verts[triIdx[3 * size_t(s)]].vertexPosition;
                const float* b = verts[triIdx[3 * size_t(s) + 1]].vertexPosition;
                const float* c = verts[triIdx[3 * size_t(s) + 2]].vertexPosition;
                rec[0] = make_float4(a[0], a[1], a[2], __uint_as_float(s));
                rec[1] = make_float4(b[0], b[1], b[2], 0.0f);
                rec[2] = make_float4(c[0], c[1], c[2], 0.0f);
            } else {
                rec[0] = make_float4(nan, nan, nan, __uint_as_float(0xFFFFFFFFu));
                rec[1] = make_float4(nan, nan, nan, 0.0f);
                rec[2] = make_float4(nan, nan, nan, 0.0f);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 6; k++) leafBox[6 * size_t(i) + k] = boxOrig[6 * size_t(g) + k];
}

__device__ __forceinline__ int lv_delta(const uint64_t* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    uint64_t a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz(uint32_t(i) ^ uint32_t(j));
    return __clzll((long long)(a ^ b));
}

// Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees", HPG 2012.
// Internal nodes 0..n-2 (root = 0); child reference = index | LEAF_BIT for leaves.
__global__ __launch_bounds__(LV_BLOCK) void k_karras(const uint64_t* __restrict__ keys, int n, uint32_t* __restrict__ childL,
                                                     uint32_t* __restrict__ childR, uint32_t* __restrict__ rangeLo,
                                                     uint32_t* __restrict__ rangeHi) {
    int i = blockIdx.x * LV_BLOCK + threadIdx.x;
    if (i >= n - 1) return;
    int d = (lv_delta(keys, n, i, i + 1) - lv_delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    int dmin = lv_delta(keys, n, i, i - d);
    int lmax = 2;
    while (lv_delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (lv_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = lv_delta(keys, n, i, j);
    int s = 0;
    int t = l;
    do {
        t = (t + 1) / 2;
        if (lv_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    int gamma = i + s * d + min(d, 0);
    int lo = min(i, j), hi = max(i, j);
    uint32_t left, right;
    if (lo == gamma) left = uint32_t(gamma) | LV_LEAF_BIT;
    else left = uint32_t(gamma);
    if (hi == gamma + 1) right = uint32_t(gamma + 1) | LV_LEAF_BIT;
    else right = uint32_t(gamma + 1);
    childL[i] = left;
    childR[i] = right;
    rangeLo[i] = uint32_t(lo); // leaves lo ... hi (sorted order) = the subtree of node i (i is lo or hi)
    rangeHi[i] = uint32_t(hi);
}

// ---------------------------------------------------------------- treelet rebuild (accel_build = fast_trace)
// The reference asks its driver for VK_BUILD_ACCELERATION_STRUCTURE_PREFER_FAST_TRACE (LineData.cpp:740-741,903,942,980): build time
// is spent on trace speed.  Here: Morton order decides which leaves belong together down to subtrees of at most `treelet_leaves`
// (default 512, option treelet_leaves 3 ... 4096) leaves, and every such subtree is rebuilt by ONE WAVE with a binned surface-area heuristic (16 bins per axis on the box centres,
// boxes and counts in LDS, the best of the 45 planes, stable partition, smaller half first).  The Karras numbering makes it an in-place
// operation: a subtree over the sorted leaves lo ... hi owns the internal nodes lo + 1 ... hi - 1 and its root (lo or hi), so the new
// topology is written into the old slots (the subtree's root keeps its index -- its parent points there) and refit / collapse run unchanged.
// Measured on the CPU model first (tools/bvhlab, hyb256 with 16 bins): - 5.7 % node steps per AO ray on config 3's capsules,
// - 7.2 % on its triangle tubes; closest hits do not depend on the topology.
// (the treelet size is a run-time value: 32 bytes of dynamic LDS per leaf, i.e. 128 KB for 4096 leaves -- clamped at launch time to
// what hipDeviceAttributeMaxSharedMemoryPerBlock of the device allows)
#define LV_TREELET_BINS 16u

__global__ __launch_bounds__(LV_BLOCK) void k_treelet_roots(uint32_t nInternal, const uint32_t* __restrict__ childL,
                                                            const uint32_t* __restrict__ childR, const uint32_t* __restrict__ rangeLo,
                                                            const uint32_t* __restrict__ rangeHi, uint32_t maxLeaves,
                                                            uint32_t* __restrict__ roots, uint32_t* __restrict__ count) {
    const uint32_t i = blockIdx.x * LV_BLOCK + threadIdx.x;
    if (i >= nInternal) return;
    const uint32_t size = rangeHi[i] - rangeLo[i] + 1u;
    if (i == 0u && size <= maxLeaves) { if (size >= 3u) roots[atomicAdd(count, 1u)] = 0u; return; }
    if (size <= maxLeaves) return;
    const uint32_t c[2] = {childL[i], childR[i]};
    for (int k = 0; k < 2; k++) {
        if (c[k] & LV_LEAF_BIT) continue;
        const uint32_t cs = rangeHi[c[k]] - rangeLo[c[k]] + 1u;
        if (cs <= maxLeaves && cs >= 3u) roots[atomicAdd(count, 1u)] = c[k]; // (two leaves have one topology)
    }
}

__device__ __forceinline__ float lv_wave_min_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float lv_wave_max_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Ranges of at most `treelet_lane_leaves` (default 6) leaves are not split by the wave: 480 of the 511 splits of a 512-leaf treelet
// have fewer leaves than the wave has lanes, and each of them costs the whole wave the same ~11 k cycles.  They are
// queued instead and built 64 at a time, ONE LANE PER RANGE, by the same algorithm evaluated serially (same bins, same cost expression,
// same tie rule: lowest axis, lowest plane; a plane whose last left bin is empty repeats its predecessor's cost and is skipped) -- the
// tree is the same tree, node for node.  Slot numbers no longer follow from the order of the work: the serial order hands a range with
// n leaves the n - 2 slots that follow its children's, so every queued range carries its own first slot (`vbase`).
#define LV_TREELET_SYNC() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")
#define LV_TREELET_LANE_STACK 6u      // smaller half first: depth <= log2(64)
#define LV_TREELET_LANE_MAX 64u
__device__ __forceinline__ uint32_t lv_treelet_bin(const float* b, int ax, float cmn, float scale) {
    const int bi = int(((b[ax] + b[3 + ax]) - cmn) * scale);
    return uint32_t(bi < 0 ? 0 : (bi > int(LV_TREELET_BINS) - 1 ? int(LV_TREELET_BINS) - 1 : bi));
}
__device__ void lv_treelet_lane_build(const float (*s_box)[6], uint32_t* s_idx, uint32_t* s_tmp, uint32_t* stk, uint32_t a, uint32_t lo,
                                      uint32_t hi, uint32_t slot, uint32_t v, uint32_t* __restrict__ childL,
                                      uint32_t* __restrict__ childR) {
    uint32_t sp = 0;
    while (true) {
        const uint32_t n = hi - lo;
        uint32_t nl;
        if (n == 2u) {
            nl = 1u;
        } else {
            float cmn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, cmx[3] = {-3.0e38f, -3.0e38f, -3.0e38f}, scale[3];
            for (uint32_t j = lo; j < hi; j++) {
                const float* b = s_box[s_idx[j]];
#pragma unroll
                for (int k = 0; k < 3; k++) { const float c = b[k] + b[3 + k]; cmn[k] = fminf(cmn[k], c); cmx[k] = fmaxf(cmx[k], c); }
            }
#pragma unroll
            for (int k = 0; k < 3; k++) { const float ext = cmx[k] - cmn[k]; scale[k] = ext > 0.0f ? float(LV_TREELET_BINS) / ext : 0.0f; }
            float bestCost = 3.0e38f;
            uint32_t bestAx = 0u, bestPlane = 0u;
#pragma unroll
            for (int ax = 0; ax < 3; ax++) {
                uint32_t occ = 0u;
                for (uint32_t j = lo; j < hi; j++) occ |= 1u << lv_treelet_bin(s_box[s_idx[j]], ax, cmn[ax], scale[ax]);
                for (uint32_t plane = 1u; plane < LV_TREELET_BINS; plane++) {
                    if (!((occ >> (plane - 1u)) & 1u)) continue;   // same left set as the plane before (or none): cannot win the tie
                    if ((occ >> plane) == 0u) break;               // nothing on the right from here on
                    float mn[2][3], mx[2][3];
                    uint32_t cn[2] = {0u, 0u};
#pragma unroll
                    for (int h = 0; h < 2; h++)
#pragma unroll
                        for (int k = 0; k < 3; k++) { mn[h][k] = 3.0e38f; mx[h][k] = -3.0e38f; }
                    for (uint32_t j = lo; j < hi; j++) {
                        const float* b = s_box[s_idx[j]];
                        const bool right = lv_treelet_bin(b, ax, cmn[ax], scale[ax]) >= plane;
                        cn[0] += right ? 0u : 1u;
                        cn[1] += right ? 1u : 0u;
#pragma unroll
                        for (int k = 0; k < 3; k++) {
                            mn[0][k] = right ? mn[0][k] : fminf(mn[0][k], b[k]);
                            mx[0][k] = right ? mx[0][k] : fmaxf(mx[0][k], b[3 + k]);
                            mn[1][k] = right ? fminf(mn[1][k], b[k]) : mn[1][k];
                            mx[1][k] = right ? fmaxf(mx[1][k], b[3 + k]) : mx[1][k];
                        }
                    }
                    float area[2];
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const float dx = mx[h][0] - mn[h][0], dy = mx[h][1] - mn[h][1], dz = mx[h][2] - mn[h][2];
                        area[h] = dx * dy + dy * dz + dz * dx;
                    }
                    const float cost = area[0] * float(cn[0]) + area[1] * float(cn[1]);
                    if (cost < bestCost) { bestCost = cost; bestAx = uint32_t(ax); bestPlane = plane; }
                }
            }
            if (bestCost >= 3.0e38f) {
                nl = n / 2u;
            } else {
                const float bc = bestAx == 0u ? cmn[0] : (bestAx == 1u ? cmn[1] : cmn[2]);
                const float bs = bestAx == 0u ? scale[0] : (bestAx == 1u ? scale[1] : scale[2]);
                uint32_t cntL = 0u;
                for (uint32_t j = lo; j < hi; j++) {
                    const float* b = s_box[s_idx[j]];
                    const float c2 = bestAx == 0u ? b[0] + b[3] : (bestAx == 1u ? b[1] + b[4] : b[2] + b[5]);
                    const int bi = int((c2 - bc) * bs);
                    const uint32_t bn = uint32_t(bi < 0 ? 0 : (bi > int(LV_TREELET_BINS) - 1 ? int(LV_TREELET_BINS) - 1 : bi));
                    cntL += bn < bestPlane ? 1u : 0u;
                }
                nl = cntL;
                uint32_t runL = 0u, runR = 0u;
                for (uint32_t j = lo; j < hi; j++) {
                    const uint32_t e = s_idx[j];
                    const float* b = s_box[e];
                    const float c2 = bestAx == 0u ? b[0] + b[3] : (bestAx == 1u ? b[1] + b[4] : b[2] + b[5]);
                    const int bi = int((c2 - bc) * bs);
                    const uint32_t bn = uint32_t(bi < 0 ? 0 : (bi > int(LV_TREELET_BINS) - 1 ? int(LV_TREELET_BINS) - 1 : bi));
                    if (bn < bestPlane) s_tmp[lo + runL++] = e; else s_tmp[lo + nl + runR++] = e;
                }
                for (uint32_t j = lo; j < hi; j++) s_idx[j] = s_tmp[j];
                if (nl == 0u || nl == n) nl = n / 2u;
            }
        }
        const uint32_t nr = n - nl;
        const uint32_t sL = nl > 1u ? v++ : 0u, sR = nr > 1u ? v++ : 0u;
        childL[slot] = nl > 1u ? sL : ((a + s_idx[lo]) | LV_LEAF_BIT);
        childR[slot] = nr > 1u ? sR : ((a + s_idx[lo + nl]) | LV_LEAF_BIT);
        if (nl == 2u) { childL[sL] = (a + s_idx[lo]) | LV_LEAF_BIT; childR[sL] = (a + s_idx[lo + 1u]) | LV_LEAF_BIT; }
        if (nr == 2u) { childL[sR] = (a + s_idx[lo + nl]) | LV_LEAF_BIT; childR[sR] = (a + s_idx[lo + nl + 1u]) | LV_LEAF_BIT; }
        const bool goL = nl > 2u, goR = nr > 2u;
        if (goL && goR) {
            const bool leftFirst = nl <= nr;
            const uint32_t nF = leftFirst ? nl : nr;
            stk[3u * sp] = (leftFirst ? lo + nl : lo) | ((leftFirst ? hi : lo + nl) << 16);
            stk[3u * sp + 1u] = leftFirst ? sR : sL;
            stk[3u * sp + 2u] = v + nF - 2u;
            sp++;
            if (leftFirst) { hi = lo + nl; slot = sL; } else { lo = lo + nl; slot = sR; }
        } else if (goL) {
            hi = lo + nl; slot = sL;
        } else if (goR) {
            lo = lo + nl; slot = sR;
        } else {
            if (sp == 0u) break;
            sp--;
            lo = stk[3u * sp] & 0xFFFFu; hi = stk[3u * sp] >> 16; slot = stk[3u * sp + 1u]; v = stk[3u * sp + 2u];
        }
    }
}

// GROUP builder (treelet_group_leaves, the default): the queued ranges of at most G leaves (G = 8 or 16) are built 64 / G at a time, G lanes
// per range, one leaf per lane.  1 M segments: the whole build 4.4 -> 2.4 ms (wave only -> groups of 16 and 8; one lane per range: 3.7).
// With so few leaves the bins are not materialised: a leaf's own bin on an axis names the lowest plane that has this leaf and
// every leaf of a lower or equal bin on its left -- these are all distinct non-trivial partitions of the axis, each under the lowest
// plane number that produces it, i.e. exactly the candidates that can win under the tie rule (lowest cost, lowest axis, lowest plane).
// Lane i evaluates its three candidates against the boxes of the group's other leaves (fetched lane to lane), the group's best key
// (cost bits, 15 axis + plane - 1) decides, the partition is a ballot.  Same bins, same unions, same cost expression: the same tree.
// DOWN != 0: a group hands the child ranges of 3 ... DOWN leaves to the list of the next smaller group size (s_out) instead of building them.
template <uint32_t G, uint32_t DOWN>
__device__ void lv_treelet_group_build(const float (*s_box)[6], uint32_t* s_idx, const uint32_t* s_items, uint32_t numItems,
                                       uint32_t* s_gstack, uint32_t a, uint32_t* __restrict__ childL, uint32_t* __restrict__ childR,
                                       uint32_t* s_out, uint32_t& outCount) {
    constexpr uint32_t NG = 64u / G;
    const uint32_t lane = threadIdx.x, gi = lane % G, g = lane / G, gbase = g * G;
    uint32_t* stk = s_gstack + g * 3u * LV_TREELET_LANE_STACK;
    uint32_t nextItem = 0u;                 // wave-uniform
    bool has = false;                       // group-uniform state, replicated in the group's lanes
    uint32_t lo = 0u, hi = 0u, slot = 0u, v = 0u, sp = 0u;
    while (true) {
        // groups without a range take the next queued items, in group order
        {
            const unsigned long long idle = __ballot(!has && gi == 0u);
            const uint32_t rank = uint32_t(__popcll(idle & ((1ull << gbase) - 1ull)));
            if (!has && nextItem + rank < numItems) {
                const uint32_t it = nextItem + rank;
                lo = s_items[3u * it] & 0xFFFFu; hi = s_items[3u * it] >> 16; slot = s_items[3u * it + 1u]; v = s_items[3u * it + 2u];
                sp = 0u;
                has = true;
            }
            nextItem = min(numItems, nextItem + uint32_t(__popcll(idle)));
        }
        if (__ballot(has) == 0ull) break;
        const uint32_t n = has ? hi - lo : 0u;
        const bool valid = gi < n;
        uint32_t e = 0u;
        float b[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        if (valid) {
            e = s_idx[lo + gi];
#pragma unroll
            for (int k = 0; k < 6; k++) b[k] = s_box[e][k];
        }
        // bounds of the box centres over the group
        float cmn[3], cmx[3], scale[3];
#pragma unroll
        for (int k = 0; k < 3; k++) { const float c = b[k] + b[3 + k]; cmn[k] = valid ? c : 3.0e38f; cmx[k] = valid ? c : -3.0e38f; }
#pragma unroll
        for (uint32_t o = 1u; o < G; o <<= 1) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                cmn[k] = fminf(cmn[k], __shfl_xor(cmn[k], int(o), 64));
                cmx[k] = fmaxf(cmx[k], __shfl_xor(cmx[k], int(o), 64));
            }
        }
        uint32_t bn[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float ext = cmx[k] - cmn[k];
            scale[k] = ext > 0.0f ? float(LV_TREELET_BINS) / ext : 0.0f;
            bn[k] = lv_treelet_bin(b, k, cmn[k], scale[k]);
        }
        const uint32_t packed = valid ? (bn[0] | (bn[1] << 4) | (bn[2] << 8) | 0x1000u) : 0u;
        // the three candidates of this lane: unions of the leaves on either side of "bin <= own bin"
        float mn[3][2][3], mx[3][2][3];
        uint32_t cn[3][2];
#pragma unroll
        for (int ax = 0; ax < 3; ax++)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                cn[ax][h] = 0u;
#pragma unroll
                for (int k = 0; k < 3; k++) { mn[ax][h][k] = 3.0e38f; mx[ax][h][k] = -3.0e38f; }
            }
#pragma unroll
        for (uint32_t j = 0u; j < G; j++) {
            const int src = int(gbase + j);
            const uint32_t pj = uint32_t(__shfl(int(packed), src, 64));
            float bj[6];
#pragma unroll
            for (int k = 0; k < 6; k++) bj[k] = __shfl(b[k], src, 64);
            const bool vj = (pj & 0x1000u) != 0u;
#pragma unroll
            for (int ax = 0; ax < 3; ax++) {
                const uint32_t bjx = (pj >> (4 * ax)) & 15u;
                const bool left = vj && bjx <= bn[ax], right = vj && bjx > bn[ax];
                cn[ax][0] += left ? 1u : 0u;
                cn[ax][1] += right ? 1u : 0u;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    mn[ax][0][k] = left ? fminf(mn[ax][0][k], bj[k]) : mn[ax][0][k];
                    mx[ax][0][k] = left ? fmaxf(mx[ax][0][k], bj[3 + k]) : mx[ax][0][k];
                    mn[ax][1][k] = right ? fminf(mn[ax][1][k], bj[k]) : mn[ax][1][k];
                    mx[ax][1][k] = right ? fmaxf(mx[ax][1][k], bj[3 + k]) : mx[ax][1][k];
                }
            }
        }
        unsigned long long key = ~0ull;
#pragma unroll
        for (int ax = 0; ax < 3; ax++) {
            if (valid && cn[ax][1] != 0u) {   // (the left side holds this leaf; plane = own bin + 1 <= 15 because a higher bin exists)
                float area[2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const float dx = mx[ax][h][0] - mn[ax][h][0], dy = mx[ax][h][1] - mn[ax][h][1], dz = mx[ax][h][2] - mn[ax][h][2];
                    area[h] = dx * dy + dy * dz + dz * dx;
                }
                const float cost = area[0] * float(cn[ax][0]) + area[1] * float(cn[ax][1]);
                if (cost < 3.0e38f) {
                    const unsigned long long kk = ((unsigned long long)__float_as_uint(cost) << 32) |
                                                  (uint32_t(ax) * (LV_TREELET_BINS - 1u) + bn[ax]);   // plane - 1 = own bin
                    key = kk < key ? kk : key;
                }
            }
        }
#pragma unroll
        for (uint32_t o = 1u; o < G; o <<= 1) {
            const unsigned long long ok = (unsigned long long)__shfl_xor((long long)key, int(o), 64);
            key = ok < key ? ok : key;
        }
        uint32_t nl;
        if (n == 2u) {
            nl = 1u;
        } else if (key == ~0ull) {
            nl = n / 2u;   // every centre in one bin on every axis: split the range in the middle
        } else {
            const uint32_t bl = uint32_t(key & 0xFFFFFFFFull), ax = bl / (LV_TREELET_BINS - 1u), plane = bl % (LV_TREELET_BINS - 1u) + 1u;
            const uint32_t mine = ax == 0u ? bn[0] : (ax == 1u ? bn[1] : bn[2]);
            const bool left = valid && mine < plane;
            const uint32_t gm = uint32_t((__ballot(left) >> gbase) & ((1ull << G) - 1ull));
            const uint32_t gv = uint32_t((__ballot(valid) >> gbase) & ((1ull << G) - 1ull));
            const uint32_t below = (1u << gi) - 1u;
            nl = uint32_t(__popc(gm));
            if (has && valid && n != 2u)
                s_idx[lo + (left ? uint32_t(__popc(gm & below)) : nl + uint32_t(__popc(gv & ~gm & below)))] = e;   // stable partition, in place
        }
        LV_TREELET_SYNC();
        // children: slots in the serial order (both children's, then the subtree of the half the serial build takes first: the smaller)
        uint32_t nr = 0u, sL = 0u, sR = 0u, vL = 0u, vR = 0u;
        bool goL = false, goR = false;
        if (has) {
            nr = n - nl;
            sL = nl > 1u ? v++ : 0u;
            sR = nr > 1u ? v++ : 0u;
            if (gi == 0u) {
                childL[slot] = nl > 1u ? sL : ((a + s_idx[lo]) | LV_LEAF_BIT);
                childR[slot] = nr > 1u ? sR : ((a + s_idx[lo + nl]) | LV_LEAF_BIT);
                // a child of two leaves has one topology: written here, it never becomes a range of its own
                if (nl == 2u) { childL[sL] = (a + s_idx[lo]) | LV_LEAF_BIT; childR[sL] = (a + s_idx[lo + 1u]) | LV_LEAF_BIT; }
                if (nr == 2u) { childL[sR] = (a + s_idx[lo + nl]) | LV_LEAF_BIT; childR[sR] = (a + s_idx[lo + nl + 1u]) | LV_LEAF_BIT; }
            }
            const bool leftFirst = nl <= nr;
            vL = (nr > 1u && !leftFirst) ? v + nr - 2u : v;
            vR = (nl > 1u && leftFirst) ? v + nl - 2u : v;
            goL = nl > 2u;
            goR = nr > 2u;
        }
        if (DOWN != 0u) {
            const bool dL = goL && nl <= DOWN, dR = goR && nr <= DOWN;
            const unsigned long long mL = __ballot(dL && gi == 0u), mR = __ballot(dR && gi == 0u), belowLane = (1ull << lane) - 1ull;
            if (dL && gi == 0u) {
                const uint32_t ix = outCount + uint32_t(__popcll(mL & belowLane));
                s_out[3u * ix] = lo | ((lo + nl) << 16); s_out[3u * ix + 1u] = sL; s_out[3u * ix + 2u] = vL;
            }
            if (dR && gi == 0u) {
                const uint32_t ix = outCount + uint32_t(__popcll(mL)) + uint32_t(__popcll(mR & belowLane));
                s_out[3u * ix] = (lo + nl) | (hi << 16); s_out[3u * ix + 1u] = sR; s_out[3u * ix + 2u] = vR;
            }
            outCount += uint32_t(__popcll(mL)) + uint32_t(__popcll(mR));
            goL = goL && !dL;
            goR = goR && !dR;
        }
        if (has) {
            if (goL && goR) {
                const bool leftFirst = nl <= nr;
                if (gi == 0u) {
                    stk[3u * sp] = (leftFirst ? lo + nl : lo) | ((leftFirst ? hi : lo + nl) << 16);
                    stk[3u * sp + 1u] = leftFirst ? sR : sL;
                    stk[3u * sp + 2u] = leftFirst ? vR : vL;
                }
                sp++;
                if (leftFirst) { hi = lo + nl; slot = sL; v = vL; } else { lo = lo + nl; slot = sR; v = vR; }
            } else if (goL) {
                hi = lo + nl; slot = sL; v = vL;
            } else if (goR) {
                lo = lo + nl; slot = sR; v = vR;
            } else if (sp == 0u) {
                has = false;
            } else {
                sp--;
                LV_TREELET_SYNC();
                lo = stk[3u * sp] & 0xFFFFu; hi = stk[3u * sp] >> 16; slot = stk[3u * sp + 1u]; v = stk[3u * sp + 2u];
            }
        }
        LV_TREELET_SYNC();
    }
}

// SCAN = false keeps the round-3 form of the plane evaluation (one lane per plane looping over the bins, shuffle reduction): the
// reference form the tests compare the scan form against (option treelet_plane_eval = loop).
// The workgroup is ONE wave: its LDS operations execute in order, so the phases are separated by wave-scope fences (compiler ordering)
// instead of __syncthreads().  Measured with cycle counters per phase (EXPERIMENTS.md 11.5): a split costs ~11 k cycles whatever its
// size up to 64 leaves -- neither the barriers nor the LDS round trips (scan and loop form of the plane evaluation: 4.6 vs 4.4 ms for
// the whole build) but the ~1 500 instructions a lone wave issues for it; 70 % of a treelet's splits have <= 8 leaves.
template <bool SCAN>
__global__ __launch_bounds__(64) void k_treelet_rebuild(const uint32_t* __restrict__ roots, const uint32_t* __restrict__ rangeLo,
                                                        const uint32_t* __restrict__ rangeHi, const float* __restrict__ leafBox,
                                                        uint32_t maxLeaves, uint32_t laneLeaves, uint32_t groupLeaves,
                                                        uint32_t* __restrict__ childL, uint32_t* __restrict__ childR) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[]; // maxLeaves x (24-byte box + two index arrays)
    float (*s_box)[6] = reinterpret_cast<float (*)[6]>(s_dyn);
    uint32_t* s_idx = reinterpret_cast<uint32_t*>(s_dyn + size_t(maxLeaves) * 24);
    uint32_t* s_tmp = s_idx + maxLeaves;
    __shared__ uint32_t s_bmin[3][LV_TREELET_BINS][3], s_bmax[3][LV_TREELET_BINS][3], s_bcnt[3][LV_TREELET_BINS];
    __shared__ uint32_t s_stack[4 * 16];
    __shared__ unsigned long long s_best;
    __shared__ uint32_t s_small[3 * 160];                                 // queued small ranges: {lo | hi << 16, slot, vbase} (64 + what 16 ranges of 9 ... 16 leaves hand down)
    __shared__ uint32_t s_mid[3 * 16];                                    // queued ranges of 9 ... 16 leaves (treelet_group_leaves = 16)
    __shared__ uint32_t s_lstack[64 * 3 * LV_TREELET_LANE_STACK];         // their builders' stacks, one per lane
    const uint32_t lane = threadIdx.x;
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint32_t root = roots[blockIdx.x];
    const uint32_t a = rangeLo[root], m = rangeHi[root] - a + 1u; // leaves a ... a + m - 1, internal nodes a ... a + m - 2
    for (uint32_t j = lane; j < m; j += 64u) {
#pragma unroll
        for (int k = 0; k < 6; k++) s_box[j][k] = leafBox[6 * size_t(a + j) + k];
        s_idx[j] = j;
    }
    LV_TREELET_SYNC();
    // A Karras node is the first or the last leaf index of its range: the subtree's internal nodes are a ... a + m - 2 when the root
    // is a, a + 1 ... a + m - 1 when it is a + m - 1 -- either way the m - 2 nodes below the root are a + 1 ... a + m - 2.
    uint32_t sp = 0, lo = 0, hi = m, slot = root, v = a + 1u, nSmall = 0u;
    uint32_t nMid = 0u;
    auto runSmall = [&]() {   // groups of 16 lanes for the ranges of 9 ... 16 leaves, of 8 lanes for the smaller ones; or one lane per range
        LV_TREELET_SYNC();
        if (groupLeaves != 0u) {
            uint32_t none = 0u;
            if (nMid) lv_treelet_group_build<16u, 8u>(s_box, s_idx, s_mid, nMid, s_lstack, a, childL, childR, s_small, nSmall);
            nMid = 0u;
            LV_TREELET_SYNC();
            if (nSmall) lv_treelet_group_build<8u, 0u>(s_box, s_idx, s_small, nSmall, s_lstack, a, childL, childR, nullptr, none);
        } else if (lane < nSmall) {
            lv_treelet_lane_build(s_box, s_idx, s_tmp, &s_lstack[lane * 3u * LV_TREELET_LANE_STACK], a, s_small[3u * lane] & 0xFFFFu,
                                  s_small[3u * lane] >> 16, s_small[3u * lane + 1u], s_small[3u * lane + 2u], childL, childR);
        }
        nSmall = 0u;
        LV_TREELET_SYNC();
    };
    if (groupLeaves != 0u) laneLeaves = groupLeaves;   // the queue threshold
    while (true) {
        const uint32_t n = hi - lo;
        if (laneLeaves != 0u && n <= laneLeaves) {
            if (groupLeaves == 16u && n > 8u) {
                if (lane == 0u) { s_mid[3u * nMid] = lo | (hi << 16); s_mid[3u * nMid + 1u] = slot; s_mid[3u * nMid + 2u] = v; }
                nMid++;
            } else {
                if (lane == 0u) { s_small[3u * nSmall] = lo | (hi << 16); s_small[3u * nSmall + 1u] = slot; s_small[3u * nSmall + 2u] = v; }
                nSmall++;
            }
            if (nSmall == 64u || nMid == 16u) runSmall();
            if (sp == 0u) break;
            sp--;
            LV_TREELET_SYNC();
            lo = s_stack[4 * sp]; hi = s_stack[4 * sp + 1]; slot = s_stack[4 * sp + 2]; v = s_stack[4 * sp + 3];
            LV_TREELET_SYNC();
            continue;
        }
        uint32_t nl;
        if (n == 2u) {
            nl = 1u;
        } else {
            // bounds of the box centres (twice the centre: no need to halve)
            float cmn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, cmx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
            for (uint32_t j = lo + lane; j < hi; j += 64u) {
                const float* b = s_box[s_idx[j]];
#pragma unroll
                for (int k = 0; k < 3; k++) { const float c = b[k] + b[3 + k]; cmn[k] = fminf(cmn[k], c); cmx[k] = fmaxf(cmx[k], c); }
            }
            float scale[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                cmn[k] = lv_wave_min_f(cmn[k]);
                cmx[k] = lv_wave_max_f(cmx[k]);
                const float ext = cmx[k] - cmn[k];
                scale[k] = ext > 0.0f ? float(LV_TREELET_BINS) / ext : 0.0f;
            }
            for (uint32_t j = lane; j < 3u * LV_TREELET_BINS; j += 64u) {
                const uint32_t ax = j / LV_TREELET_BINS, bn = j % LV_TREELET_BINS;
#pragma unroll
                for (int k = 0; k < 3; k++) { s_bmin[ax][bn][k] = 0xFFFFFFFFu; s_bmax[ax][bn][k] = 0u; }
                s_bcnt[ax][bn] = 0u;
            }
            if (lane == 63u) s_best = ~0ull;
            LV_TREELET_SYNC();
            for (uint32_t j = lo + lane; j < hi; j += 64u) {
                const float* b = s_box[s_idx[j]];
#pragma unroll
                for (int ax = 0; ax < 3; ax++) {
                    const int bi = int(((b[ax] + b[3 + ax]) - cmn[ax]) * scale[ax]);
                    const uint32_t bn = uint32_t(bi < 0 ? 0 : (bi > int(LV_TREELET_BINS) - 1 ? int(LV_TREELET_BINS) - 1 : bi));
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        atomicMin(&s_bmin[ax][bn][k], lv_f2ord(b[k]));
                        atomicMax(&s_bmax[ax][bn][k], lv_f2ord(b[3 + k]));
                    }
                    atomicAdd(&s_bcnt[ax][bn], 1u);
                }
            }
            LV_TREELET_SYNC();
            // the 3 x 15 split planes: cost = area(left) * count(left) + area(right) * count(right).  Lane (axis, bin) = row of 16 lanes per
            // axis loads its bin, an inclusive prefix and an inclusive suffix scan over the row (DPP row shifts: no LDS round trips)
            // give every lane the union of the bins up to / from its own; plane p of an axis = prefix of bin p - 1 | suffix of bin p.
            // (the round-3 form -- one lane per plane looping over the 16 bins, treelet_plane_eval = loop -- measures the same)
            float cost = 3.0e38f;
            float bc;
            uint32_t bl;
            if constexpr (SCAN) {
            const uint32_t rowAx = lane / LV_TREELET_BINS, rowBin = lane % LV_TREELET_BINS;
            {
                float pmn[3], pmx[3], smn[3], smx[3];
                uint32_t pc = 0u;
                if (rowAx < 3u) {
                    pc = s_bcnt[rowAx][rowBin];
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        pmn[k] = pc ? lv_ord2f(s_bmin[rowAx][rowBin][k]) : 3.0e38f;
                        pmx[k] = pc ? lv_ord2f(s_bmax[rowAx][rowBin][k]) : -3.0e38f;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 3; k++) { pmn[k] = 3.0e38f; pmx[k] = -3.0e38f; }
                }
                uint32_t sc = pc;
#pragma unroll
                for (int k = 0; k < 3; k++) { smn[k] = pmn[k]; smx[k] = pmx[k]; }
#define LV_DPP_F(old, v, ctrl) __uint_as_float(uint32_t(__builtin_amdgcn_update_dpp(int(__float_as_uint(old)), int(__float_as_uint(v)), ctrl, 0xF, 0xF, false)))
#define LV_DPP_U(old, v, ctrl) uint32_t(__builtin_amdgcn_update_dpp(int(old), int(v), ctrl, 0xF, 0xF, false))
#define LV_SCAN_STEP(d)                                                                                              \
                {                                                                                                    \
                    pc += LV_DPP_U(0u, pc, 0x110 + d);                                                               \
                    sc += LV_DPP_U(0u, sc, 0x100 + d);                                                               \
                    _Pragma("unroll") for (int k = 0; k < 3; k++) {                                                  \
                        pmn[k] = fminf(pmn[k], LV_DPP_F(3.0e38f, pmn[k], 0x110 + d));   /* row_shr: from lane - d */  \
                        pmx[k] = fmaxf(pmx[k], LV_DPP_F(-3.0e38f, pmx[k], 0x110 + d));                               \
                        smn[k] = fminf(smn[k], LV_DPP_F(3.0e38f, smn[k], 0x100 + d));   /* row_shl: from lane + d */  \
                        smx[k] = fmaxf(smx[k], LV_DPP_F(-3.0e38f, smx[k], 0x100 + d));                               \
                    }                                                                                                \
                }
                LV_SCAN_STEP(1) LV_SCAN_STEP(2) LV_SCAN_STEP(4) LV_SCAN_STEP(8)
                // left side of plane rowBin = the prefix of the bin before
                const uint32_t lc = LV_DPP_U(0u, pc, 0x111);
                float lmn[3], lmx[3];
#pragma unroll
                for (int k = 0; k < 3; k++) { lmn[k] = LV_DPP_F(3.0e38f, pmn[k], 0x111); lmx[k] = LV_DPP_F(-3.0e38f, pmx[k], 0x111); }
#undef LV_SCAN_STEP
#undef LV_DPP_U
#undef LV_DPP_F
                if (rowAx < 3u && rowBin >= 1u && lc != 0u && sc != 0u) {
                    const float dx0 = lmx[0] - lmn[0], dy0 = lmx[1] - lmn[1], dz0 = lmx[2] - lmn[2];
                    const float dx1 = smx[0] - smn[0], dy1 = smx[1] - smn[1], dz1 = smx[2] - smn[2];
                    const float a0 = dx0 * dy0 + dy0 * dz0 + dz0 * dx0, a1 = dx1 * dy1 + dy1 * dz1 + dz1 * dx1;
                    cost = a0 * float(lc) + a1 * float(sc);
                }
            }
            // best plane of the wave (lowest cost, ties: lowest axis, then lowest plane): one 64-bit LDS atomicMin on (cost bits, plane
            // number) -- costs are non-negative, their bit patterns order like the values
            if (cost < 3.0e38f)
                atomicMin(&s_best, ((unsigned long long)__float_as_uint(cost) << 32) | (rowAx * (LV_TREELET_BINS - 1u) + rowBin - 1u));
            LV_TREELET_SYNC();
            const unsigned long long bestKey = s_best;
            bc = bestKey == ~0ull ? 3.0e38f : __uint_as_float(uint32_t(bestKey >> 32));
            bl = uint32_t(bestKey & 0xFFFFFFFFull);
            } else {
            // round-3 form: one lane per plane, loop over the bins; best plane by a shuffle reduction (lowest cost, ties: lowest lane)
            if (lane < 3u * (LV_TREELET_BINS - 1u)) {
                const uint32_t ax = lane / (LV_TREELET_BINS - 1u), plane = lane % (LV_TREELET_BINS - 1u) + 1u;
                float mn[2][3], mx[2][3];
                uint32_t cn[2] = {0u, 0u};
#pragma unroll
                for (int h = 0; h < 2; h++)
#pragma unroll
                    for (int k = 0; k < 3; k++) { mn[h][k] = 3.0e38f; mx[h][k] = -3.0e38f; }
                for (uint32_t bn = 0; bn < LV_TREELET_BINS; bn++) {
                    const uint32_t c = s_bcnt[ax][bn];
                    if (c == 0u) continue;
                    const int h = bn < plane ? 0 : 1;
                    cn[h] += c;
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        mn[h][k] = fminf(mn[h][k], lv_ord2f(s_bmin[ax][bn][k]));
                        mx[h][k] = fmaxf(mx[h][k], lv_ord2f(s_bmax[ax][bn][k]));
                    }
                }
                if (cn[0] != 0u && cn[1] != 0u) {
                    float area[2];
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const float dx = mx[h][0] - mn[h][0], dy = mx[h][1] - mn[h][1], dz = mx[h][2] - mn[h][2];
                        area[h] = dx * dy + dy * dz + dz * dx;
                    }
                    cost = area[0] * float(cn[0]) + area[1] * float(cn[1]);
                }
            }
            bc = cost;
            bl = lane;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float oc = __shfl_xor(bc, o, 64);
                const uint32_t ol = uint32_t(__shfl_xor(int(bl), o, 64));
                if (oc < bc || (oc == bc && ol < bl)) { bc = oc; bl = ol; }
            }
            }
            if (bc >= 3.0e38f) {
                nl = n / 2u; // every centre in one bin on every axis: split the range in the middle
            } else {
                const uint32_t ax = bl / (LV_TREELET_BINS - 1u), plane = bl % (LV_TREELET_BINS - 1u) + 1u;
                // stable partition of s_idx[lo, hi) by "bin < plane" through s_tmp
                uint32_t cntL = 0;
                for (uint32_t base = lo; base < hi; base += 64u) {
                    const uint32_t j = base + lane;
                    bool left = false;
                    if (j < hi) {
                        const float* b = s_box[s_idx[j]];
                        const int bi = int(((b[ax] + b[3 + ax]) - cmn[ax]) * scale[ax]);
                        const uint32_t bn = uint32_t(bi < 0 ? 0 : (bi > int(LV_TREELET_BINS) - 1 ? int(LV_TREELET_BINS) - 1 : bi));
                        left = bn < plane;
                    }
                    cntL += uint32_t(__popcll(__ballot(left)));
                }
                nl = cntL;
                uint32_t runL = 0, runR = 0;
                for (uint32_t base = lo; base < hi; base += 64u) {
                    const uint32_t j = base + lane;
                    const bool valid = j < hi;
                    bool left = false;
                    uint32_t e = 0;
                    if (valid) {
                        e = s_idx[j];
                        const float* b = s_box[e];
                        const int bi = int(((b[ax] + b[3 + ax]) - cmn[ax]) * scale[ax]);
                        const uint32_t bn = uint32_t(bi < 0 ? 0 : (bi > int(LV_TREELET_BINS) - 1 ? int(LV_TREELET_BINS) - 1 : bi));
                        left = bn < plane;
                    }
                    const unsigned long long mL = __ballot(left), mR = __ballot(valid && !left);
                    if (valid) {
                        if (left) s_tmp[lo + runL + uint32_t(__popcll(mL & below))] = e;
                        else s_tmp[lo + nl + runR + uint32_t(__popcll(mR & below))] = e;
                    }
                    runL += uint32_t(__popcll(mL));
                    runR += uint32_t(__popcll(mR));
                }
                LV_TREELET_SYNC();
                for (uint32_t j = lo + lane; j < hi; j += 64u) s_idx[j] = s_tmp[j];
                LV_TREELET_SYNC();
                if (nl == 0u || nl == n) nl = n / 2u; // cannot happen with finite boxes (both sides of the chosen plane hold leaves); NaN input
            }
        }
        // children of `slot`: a single leaf becomes a leaf reference, a longer range gets a slot of its own
        const uint32_t nr = n - nl;
        const uint32_t sL = nl > 1u ? v++ : 0u, sR = nr > 1u ? v++ : 0u;   // the serial order: both children's slots, then the first half's subtree
        if (lane == 0u) {
            childL[slot] = nl > 1u ? sL : ((a + s_idx[lo]) | LV_LEAF_BIT);
            childR[slot] = nr > 1u ? sR : ((a + s_idx[lo + nl]) | LV_LEAF_BIT);
            // a child of two leaves has one topology: written here, it never becomes a range of its own
            if (nl == 2u) { childL[sL] = (a + s_idx[lo]) | LV_LEAF_BIT; childR[sL] = (a + s_idx[lo + 1u]) | LV_LEAF_BIT; }
            if (nr == 2u) { childL[sR] = (a + s_idx[lo + nl]) | LV_LEAF_BIT; childR[sR] = (a + s_idx[lo + nl + 1u]) | LV_LEAF_BIT; }
        }
        const bool goL = nl > 2u, goR = nr > 2u;
        // continue with the smaller half that still needs a split, stack the other (depth <= log2 of the treelet size); the stacked
        // half's slots follow the nF - 2 slots of the first half's subtree
        if (goL && goR) {
            const bool leftFirst = nl <= nr;
            const uint32_t plo = leftFirst ? lo + nl : lo, phi = leftFirst ? hi : lo + nl, pslot = leftFirst ? sR : sL;
            const uint32_t nF = leftFirst ? nl : nr;
            if (lane == 0u) { s_stack[4 * sp] = plo; s_stack[4 * sp + 1] = phi; s_stack[4 * sp + 2] = pslot; s_stack[4 * sp + 3] = v + nF - 2u; }
            sp++;
            if (leftFirst) { hi = lo + nl; slot = sL; } else { lo = lo + nl; slot = sR; }
        } else if (goL) {
            hi = lo + nl; slot = sL;
        } else if (goR) {
            lo = lo + nl; slot = sR;
        } else {
            if (sp == 0u) break;
            sp--;
            LV_TREELET_SYNC();
            lo = s_stack[4 * sp]; hi = s_stack[4 * sp + 1]; slot = s_stack[4 * sp + 2]; v = s_stack[4 * sp + 3];
        }
        LV_TREELET_SYNC();
    }
    if (nSmall || nMid) runSmall();
}

// Bottom-up boxes + heights in PASSES: in pass k every internal node whose two children were finished in an EARLIER pass
// (leaves always are) computes its box and height and stamps itself with k.  The kernel boundary between passes is the only
// synchronisation -- per-CU L1s and per-XCD L2s are not coherent, and the classic "second thread to arrive continues"
// refit needs an agent-scope release/acquire pair (an L2 write-back / invalidate) per node and thread: 3.8 ms for 1 M
// segments against 0.5 ms for the ~30 passes a Morton-ordered tree of that size needs (one pass per tree level).
__global__ __launch_bounds__(LV_BLOCK) void k_refit_pass(uint32_t nInternal, uint32_t pass, const uint32_t* __restrict__ childL,
                                                         const uint32_t* __restrict__ childR,
                                                         const float* __restrict__ leafBox, float* nodeBox,
                                                         uint32_t* height, uint32_t* done) {
    const uint32_t i = blockIdx.x * LV_BLOCK + threadIdx.x;
    if (i >= nInternal || done[i] != 0u) return;
    const uint32_t cl = childL[i], cr = childR[i];
    uint32_t hl = 0, hr = 0;
    const float *bl, *br;
    if (cl & LV_LEAF_BIT) bl = leafBox + 6 * size_t(cl & ~LV_LEAF_BIT);
    else {
        const uint32_t d = done[cl];
        if (d == 0u || d >= pass) return; // not finished, or finished in THIS pass (not visible yet)
        bl = nodeBox + 6 * size_t(cl);
        hl = height[cl];
    }
    if (cr & LV_LEAF_BIT) br = leafBox + 6 * size_t(cr & ~LV_LEAF_BIT);
    else {
        const uint32_t d = done[cr];
        if (d == 0u || d >= pass) return;
        br = nodeBox + 6 * size_t(cr);
        hr = height[cr];
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        nodeBox[6 * size_t(i) + k] = fminf(bl[k], br[k]);
        nodeBox[6 * size_t(i) + 3 + k] = fmaxf(bl[k + 3], br[k + 3]);
    }
    height[i] = max(hl, hr) + 1u;
    done[i] = pass;
}

// 64-byte COMPRESSED 4-wide node = 4 x float4 (one dwordx4 load each):
//   q0 = {origin.x, origin.y, origin.z, scale.x}
//   q1 = {scale.y, scale.z, qmin.x[4 bytes], qmin.y[4 bytes]}
//   q2 = {qmin.z[4], qmax.x[4], qmax.y[4], qmax.z[4]}            byte k of a word = child slot k
//   q3 = {child0, child1, child2, child3}   index | LV_LEAF_BIT for leaves, LV_INVALID for an empty slot
// Child boxes are stored as 8-bit offsets on the grid origin + q * scale spanned by the union of the children
// (Ylitie et al., "Efficient Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs", HPG 2017): decoded minima
// never exceed and maxima never fall below the exact child box, so culling stays conservative and hits stay exact.
// Why: with divergent rays every lane of a wave reads a different node, and the vector L1 serves a divergent dwordx4
// load at one lane per cycle -- the 7 loads of an uncompressed 4-wide node made k_ao_rays L1-bound (4.4 G cache
// accesses = 7.1 of its 7.6 ms).  4 loads per node cut that by 43 % and halve the node footprint (L2 / MALL hit rate).
__device__ __forceinline__ float lv_dec(float origin, float scale, uint32_t q) { return __builtin_fmaf(float(q), scale, origin); }

__device__ __forceinline__ void lv_write_wide_node(float4* out, int ns, const uint32_t slotRef[4], const float b[4][6]) {
    // quantisation grid: origin = min over children, scale = extent / 255 rounded up
    float origin[3], scale[3];
    uint32_t qmin[3] = {0, 0, 0}, qmax[3] = {0, 0, 0}; // byte k = slot k
#pragma unroll
    for (int a = 0; a < 3; a++) {
        float lo = 3.0e38f, hi = -3.0e38f;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (k < ns) { lo = fminf(lo, b[k][a]); hi = fmaxf(hi, b[k][3 + a]); }
        origin[a] = lo;
        float sc = (hi - lo) / 255.0f;
        sc = sc * 1.000002f + 1e-30f;
        scale[a] = sc;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (k >= ns) continue;
            int q0 = int(floorf((b[k][a] - lo) / sc));
            q0 = min(max(q0, 0), 255);
            while (q0 > 0 && lv_dec(lo, sc, uint32_t(q0)) > b[k][a]) q0--;       // decoded min <= exact min
            int q1 = int(ceilf((b[k][3 + a] - lo) / sc));
            q1 = min(max(q1, 0), 255);
            while (q1 < 255 && lv_dec(lo, sc, uint32_t(q1)) < b[k][3 + a]) q1++; // decoded max >= exact max
            qmin[a] |= uint32_t(q0) << (8 * k);
            qmax[a] |= uint32_t(q1) << (8 * k);
        }
        // q = 255 must still cover hi: widen the scale in the (rounding) case it does not
        while (lv_dec(lo, scale[a], 255u) < hi) scale[a] = scale[a] * 1.00001f + 1e-30f;
        // empty slots: inverted box (min plane 255, max plane 0 on every axis) -> the slab test of lv_node_step rejects them by
        // itself (near plane behind far plane by the node's whole extent) and needs no explicit "slot occupied" test; their
        // reference stays LV_INVALID, so even a false accept could only change the visiting order, never the result
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (k >= ns) qmin[a] |= 255u << (8 * k);
    }
    out[0] = make_float4(origin[0], origin[1], origin[2], scale[0]);
    out[1] = make_float4(scale[1], scale[2], __uint_as_float(qmin[0]), __uint_as_float(qmin[1]));
    out[2] = make_float4(__uint_as_float(qmin[2]), __uint_as_float(qmax[0]), __uint_as_float(qmax[1]),
                         __uint_as_float(qmax[2]));
    out[3] = make_float4(__uint_as_float(slotRef[0]), __uint_as_float(slotRef[1]), __uint_as_float(slotRef[2]),
                         __uint_as_float(slotRef[3]));
}

// Collapse of the binary LBVH into 4-wide nodes, one BFS level of the WIDE tree per pass.  A wide node starts from the
// two children of its binary root and greedily replaces the internal slot with the largest surface area by that node's
// two children until four slots are filled (the standard SAH-guided collapse for wide BVHs): slots that are cheap to hit
// are opened first, and fewer slots stay empty than with "two binary levels per wide node" (1 M segments: 0.48 M nodes,
// 3.1 children per node; an optimal dynamic-programming collapse reaches 3.5 but traces no faster, EXPERIMENTS.md 3.1).
//   k_collapse_select  frontier item i (a binary node) -> its <= 4 slots as binary references + the number of internal ones
//   exclusive scan     -> position of each item's internal slots in the next frontier (deterministic BFS numbering)
//   k_collapse_emit    writes the compressed node (index base + i) and the next frontier
__device__ __forceinline__ float lv_box_half_area(const float* b) {
    const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return (dx * dy + dy * dz) + dz * dx;
}

__device__ __forceinline__ uint32_t lv_collapse_select_item(uint32_t r, const uint32_t* __restrict__ childL,
                                                            const uint32_t* __restrict__ childR, const float* __restrict__ nodeBox,
                                                            uint32_t* slotsOut) {
    uint32_t s0 = childL[r], s1 = childR[r], s2 = LV_INVALID, s3 = LV_INVALID;
    int ns = 2;
#pragma unroll
    for (int e = 0; e < 2; e++) {
        // internal slot with the largest area (first one on ties)
        float best = -1.0f;
        int bk = -1;
        const uint32_t cur[4] = {s0, s1, s2, s3};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (k < ns && !(cur[k] & LV_LEAF_BIT)) {
                const float a = lv_box_half_area(nodeBox + 6 * size_t(cur[k]));
                if (a > best) { best = a; bk = k; }
            }
        }
        if (bk < 0) break;
        const uint32_t c = bk == 0 ? s0 : (bk == 1 ? s1 : s2); // bk < ns <= 3 here
        const uint32_t cl = childL[c], cr = childR[c];
        if (bk == 0) s0 = cl; else if (bk == 1) s1 = cl; else s2 = cl;
        if (ns == 2) s2 = cr; else s3 = cr;
        ns++;
    }
    slotsOut[0] = s0; slotsOut[1] = s1; slotsOut[2] = s2; slotsOut[3] = s3;
    uint32_t ni = 0;
    if (!(s0 & LV_LEAF_BIT)) ni++;
    if (!(s1 & LV_LEAF_BIT)) ni++;
    if (s2 != LV_INVALID && !(s2 & LV_LEAF_BIT)) ni++;
    if (s3 != LV_INVALID && !(s3 & LV_LEAF_BIT)) ni++;
    return ni;
}

__global__ __launch_bounds__(LV_BLOCK) void k_collapse_select(const uint32_t* __restrict__ frontier, uint32_t count,
                                                              const uint32_t* __restrict__ childL,
                                                              const uint32_t* __restrict__ childR,
                                                              const float* __restrict__ nodeBox,
                                                              uint32_t* __restrict__ slots,
                                                              uint32_t* __restrict__ internalCount) {
    const uint32_t i = blockIdx.x * LV_BLOCK + threadIdx.x;
    if (i >= count) return;
    internalCount[i] = lv_collapse_select_item(frontier[i], childL, childR, nodeBox, slots + 4 * size_t(i));
}

// node `index` of the wide tree from its four slots; its internal slots become entries firstNext ... of the next frontier
__device__ __forceinline__ void lv_collapse_emit_item(uint32_t index, uint32_t nextBase, const uint32_t* slots4, uint32_t firstNext,
                                                      const float* __restrict__ leafBox, const float* __restrict__ nodeBox,
                                                      uint32_t* nextFrontier, float4* __restrict__ nodes) {
    uint32_t slotRef[4];
    float b[4][6];
    int ns = 0;
    uint32_t j = firstNext;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t c = slots4[k];
        slotRef[k] = LV_INVALID;
#pragma unroll
        for (int a = 0; a < 6; a++) b[k][a] = 0.0f;
        if (c == LV_INVALID) continue; // slots are filled from the front: ns == k here
        const float* src;
        if (c & LV_LEAF_BIT) {
            src = leafBox + 6 * size_t(c & ~LV_LEAF_BIT);
            slotRef[k] = c;
        } else {
            src = nodeBox + 6 * size_t(c);
            nextFrontier[j] = c;
            slotRef[k] = nextBase + j;
            j++;
        }
#pragma unroll
        for (int a = 0; a < 6; a++) b[k][a] = src[a];
        ns = k + 1;
    }
    lv_write_wide_node(nodes + 4 * size_t(index), ns, slotRef, b);
}

__global__ __launch_bounds__(LV_BLOCK) void k_collapse_emit(uint32_t count, uint32_t base, uint32_t nextBase,
                                                            const uint32_t* __restrict__ slots,
                                                            const uint32_t* __restrict__ offsets,
                                                            const float* __restrict__ leafBox,
                                                            const float* __restrict__ nodeBox,
                                                            uint32_t* __restrict__ nextFrontier, float4* __restrict__ nodes) {
    const uint32_t i = blockIdx.x * LV_BLOCK + threadIdx.x;
    if (i >= count) return;
    lv_collapse_emit_item(base + i, nextBase, slots + 4 * size_t(i), offsets[i], leafBox, nodeBox, nextFrontier, nodes);
}

// The first levels of the wide tree hold 1, 3, 10, 30, 94, ... nodes: ONE workgroup builds them (frontier, slots and offsets in LDS, one
// block-wide scan per level) up to the first level with more than LV_COLLAPSE_TOP nodes -- five levels of a million-leaf tree that
// cost five launches and a host round trip each.  Same numbering: the tree is the same, byte for byte (accel_collapse_top = false keeps
// the pass per level from the root).  Measured, 1 M segments: 2.39 -> 2.31 ms with 256; with 1024 (eight levels, 1 400 nodes quantised by
// one CU) 2.50 -- the lone workgroup is slower than the launches it saves.
#define LV_COLLAPSE_TOP 256u
__global__ __launch_bounds__(LV_BLOCK) void k_collapse_top(const uint32_t* __restrict__ childL, const uint32_t* __restrict__ childR,
                                                           const float* __restrict__ leafBox, const float* __restrict__ nodeBox,
                                                           uint32_t* __restrict__ frontierOut, float4* __restrict__ nodes,
                                                           uint32_t* __restrict__ state /* {count, base, levels} */) {
    constexpr uint32_t ITEMS = LV_COLLAPSE_TOP / LV_BLOCK;
    __shared__ uint32_t s_front[2][4 * LV_COLLAPSE_TOP];
    __shared__ uint32_t s_slots[4 * LV_COLLAPSE_TOP];
    __shared__ uint32_t s_scan[LV_BLOCK];
    const uint32_t t = threadIdx.x;
    uint32_t count = 1u, base = 0u, levels = 0u, cur = 0u;
    if (t == 0u) s_front[0][0] = 0u;   // the root of the binary tree
    __syncthreads();
    while (count > 0u && count <= LV_COLLAPSE_TOP) {
        uint32_t ni[ITEMS], sum = 0u;
#pragma unroll
        for (uint32_t k = 0; k < ITEMS; k++) {
            const uint32_t i = t * ITEMS + k;
            ni[k] = i < count ? lv_collapse_select_item(s_front[cur][i], childL, childR, nodeBox, &s_slots[4u * i]) : 0u;
            sum += ni[k];
        }
        // block-wide exclusive scan of the per-thread sums (Hillis-Steele over LDS)
        s_scan[t] = sum;
        __syncthreads();
        for (uint32_t o = 1u; o < LV_BLOCK; o <<= 1) {
            const uint32_t v = t >= o ? s_scan[t - o] : 0u;
            __syncthreads();
            s_scan[t] += v;
            __syncthreads();
        }
        const uint32_t total = s_scan[LV_BLOCK - 1u];
        uint32_t off = s_scan[t] - sum;
#pragma unroll
        for (uint32_t k = 0; k < ITEMS; k++) {
            const uint32_t i = t * ITEMS + k;
            if (i < count) lv_collapse_emit_item(base + i, base + count, &s_slots[4u * i], off, leafBox, nodeBox, s_front[cur ^ 1u], nodes);
            off += ni[k];
        }
        base += count;
        count = total;
        levels++;
        cur ^= 1u;
        __syncthreads();
    }
    for (uint32_t i = t; i < count; i += LV_BLOCK) frontierOut[i] = s_front[cur][i];
    if (t == 0u) { state[0] = count; state[1] = base; state[2] = levels; }
}

// single-segment scene: one node with one occupied slot
__global__ void k_single_node(const float* __restrict__ leafBox, float4* __restrict__ nodes) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float sc[3];
    for (int a = 0; a < 3; a++) {
        sc[a] = ((leafBox[3 + a] - leafBox[a]) / 255.0f) * 1.000002f + 1e-30f;
        while (lv_dec(leafBox[a], sc[a], 255u) < leafBox[3 + a]) sc[a] = sc[a] * 1.00001f + 1e-30f;
    }
    nodes[0] = make_float4(leafBox[0], leafBox[1], leafBox[2], sc[0]);
    // slot 0 = the leaf (planes 0 .. 255), slots 1-3 empty = inverted boxes (min 255, max 0)
    nodes[1] = make_float4(sc[1], sc[2], __uint_as_float(0xFFFFFF00u), __uint_as_float(0xFFFFFF00u));
    nodes[2] = make_float4(__uint_as_float(0xFFFFFF00u), __uint_as_float(255u), __uint_as_float(255u), __uint_as_float(255u));
    nodes[3] = make_float4(__uint_as_float(0u | LV_LEAF_BIT), __uint_as_float(LV_INVALID), __uint_as_float(LV_INVALID),
                           __uint_as_float(LV_INVALID));
}

inline uint32_t nblocks(uint64_t n) { return uint32_t((n + LV_BLOCK - 1) / LV_BLOCK); }

} // namespace

// Shared build: `boxes(boxOrig, bounds)` fills the per-primitive AABBs + scene bounds, `leaves(sortedVals, boxOrig,
// leafBox)` writes the primitive records and leaf boxes in Morton order; the topology / refit / collapse stages are the
// same for capsules and triangles.
template <class BOXES, class LEAVES>
static int lv_bvh_build_core(lv_ctx* ctx, uint32_t n, LvDeviceBuffer& nodesOut, uint32_t& numNodesOut, uint32_t& depthOut,
                             uint32_t& wideDepthOut, int evBase /* ctx->ev[evBase], [evBase + 1] bracket the build; -1: untimed */, BOXES&& boxes, LEAVES&& leaves) {
    const bool timed = evBase >= 0;
    hipStream_t st = ctx->stream;
    const uint32_t nInternal = n > 1 ? n - 1 : 1;
    uint32_t numWide = 1, wideLevels = 0;
    int rc;
    // 4-wide nodes: one per even-depth binary node; at most all of them (a degenerate chain has ~n/2)
    if ((rc = lv_buf_reserve(ctx, nodesOut, size_t(nInternal) * 64))) return rc;

    // temporaries: carved out of ONE arena that the context keeps between builds (≈ 200 B per primitive) -- seventeen
    // hipMalloc / hipFree pairs per build cost more host time than the kernels of a 1 M-segment build take
    struct Tmp { void* ptr = nullptr; };
    Tmp boxOrig, leafBox, nodeBox, keysA, keysB, valsA, valsB, childL, childR, height, flags, bounds, sortTmp, depth, evenFlag,
            wideIndex, slots;
    auto freeAll = [&]() {};
#define LV_TRY(expr)                 \
    do {                             \
        int _rc = (expr);            \
        if (_rc) { freeAll(); return _rc; } \
    } while (0)
#define LV_HIPF(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess) {                                                                           \
            freeAll();                                                                                    \
            return lv_fail(ctx, LV_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
        }                                                                                                 \
    } while (0)
    // small read-backs (refit / collapse progress) land in pinned memory: a 4-byte hipMemcpy into pageable memory costs
    // 100+ us per round trip, and a build makes ~20 of them
    if (!ctx->pinned) LV_HIPF(hipHostMalloc((void**)&ctx->pinned, 64, hipHostMallocDefault));
    volatile uint32_t* pin = ctx->pinned;
    size_t sortBytes = 0, scanBytes = 0;
    LV_HIPF(rocprim::radix_sort_pairs(nullptr, sortBytes, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr,
                                      (uint32_t*)nullptr, n, 0, 63, st));
    LV_HIPF(rocprim::exclusive_scan(nullptr, scanBytes, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, nInternal,
                                    rocprim::plus<uint32_t>(), st));
    {
        struct Req { Tmp* t; size_t bytes; };
        const Req reqs[] = {{&boxOrig, size_t(n) * 24}, {&leafBox, size_t(n) * 24}, {&nodeBox, size_t(nInternal) * 24},
                            {&keysA, size_t(n) * 8}, {&keysB, size_t(n) * 8}, {&valsA, size_t(n) * 4}, {&valsB, size_t(n) * 4},
                            {&childL, size_t(nInternal) * 4}, {&childR, size_t(nInternal) * 4},
                            {&height, size_t(nInternal) * 4}, {&flags, size_t(nInternal) * 4}, {&bounds, 6 * 4},
                            {&sortTmp, std::max<size_t>(std::max(sortBytes, scanBytes), 16)},
                            {&depth, size_t(nInternal) * 4}, {&evenFlag, size_t(nInternal) * 4},
                            {&wideIndex, size_t(nInternal) * 4}, {&slots, size_t(nInternal) * 16}};
        size_t total = 0;
        for (const Req& r : reqs) total += (r.bytes + 255) & ~size_t(255);
        LV_TRY(lv_buf_reserve(ctx, ctx->buildArena, total));
        size_t off = 0;
        for (const Req& r : reqs) { r.t->ptr = (char*)ctx->buildArena.ptr + off; off += (r.bytes + 255) & ~size_t(255); }
    }

    if (timed) LV_HIPF(hipEventRecord(ctx->ev[evBase], st));
    // bounds: min slots start at ord(+big) = 0xFFFFFFFF-ish, max slots at 0
    {
        uint32_t* init = ctx->pinned + 8; // words 8..13 of the pinned block (stays valid while the copy is in flight)
        init[0] = init[1] = init[2] = 0xFFFFFFFFu;
        init[3] = init[4] = init[5] = 0u;
        LV_HIPF(hipMemcpyAsync(bounds.ptr, init, 24, hipMemcpyHostToDevice, st));
    }
    boxes((float*)boxOrig.ptr, (uint32_t*)bounds.ptr);
    k_morton<<<nblocks(n), LV_BLOCK, 0, st>>>((const float*)boxOrig.ptr, n, (const uint32_t*)bounds.ptr,
                                              (uint64_t*)keysA.ptr, (uint32_t*)valsA.ptr);
    {
        size_t tmpBytes = sortBytes;
        LV_HIPF(rocprim::radix_sort_pairs(sortTmp.ptr, tmpBytes, (uint64_t*)keysA.ptr, (uint64_t*)keysB.ptr,
                                          (uint32_t*)valsA.ptr, (uint32_t*)valsB.ptr, n, 0, 63, st));
    }
    leaves((const uint32_t*)valsB.ptr, (const float*)boxOrig.ptr, (float*)leafBox.ptr);
    if (n == 1) {
        k_single_node<<<1, 64, 0, st>>>((const float*)leafBox.ptr, (float4*)nodesOut.ptr);
    } else {
        LV_HIPF(hipMemsetAsync(flags.ptr, 0, size_t(nInternal) * 4, st));
        // (depth / evenFlag are free until the collapse: the leaf ranges of the Karras nodes; wideIndex / height: the treelet list)
        k_karras<<<nblocks(nInternal), LV_BLOCK, 0, st>>>((const uint64_t*)keysB.ptr, int(n), (uint32_t*)childL.ptr,
                                                          (uint32_t*)childR.ptr, (uint32_t*)depth.ptr, (uint32_t*)evenFlag.ptr);
        if (ctx->opt.accelFastTrace && n >= 3u) {
            LV_HIPF(hipMemsetAsync(height.ptr, 0, 4, st));
            k_treelet_roots<<<nblocks(nInternal), LV_BLOCK, 0, st>>>(nInternal, (const uint32_t*)childL.ptr, (const uint32_t*)childR.ptr,
                                                                     (const uint32_t*)depth.ptr, (const uint32_t*)evenFlag.ptr,
                                                                     ctx->opt.treeletLeaves, (uint32_t*)wideIndex.ptr,
                                                                     (uint32_t*)height.ptr);
            LV_HIPF(hipMemcpyAsync((void*)ctx->pinned, height.ptr, 4, hipMemcpyDeviceToHost, st));
            LV_HIPF(hipStreamSynchronize(st));
            const uint32_t numTreelets = pin[0];
            if (numTreelets) {
                const size_t ldsBytes = size_t(ctx->opt.treeletLeaves) * 32;
                int maxLds = 0;
                LV_HIPF(hipDeviceGetAttribute(&maxLds, hipDeviceAttributeMaxSharedMemoryPerBlock, ctx->device));
                if (ldsBytes > size_t(maxLds)) {
                    freeAll();
                    return lv_fail(ctx, LV_E_INVALID, "treelet_leaves = %u needs %zu bytes of LDS per workgroup, the device has %d",
                                   ctx->opt.treeletLeaves, ldsBytes, maxLds);
                }
                auto* kernel = ctx->opt.treeletPlaneScan ? k_treelet_rebuild<true> : k_treelet_rebuild<false>;
                if (ldsBytes > 48 * 1024) // beyond the default limit of dynamic LDS (gfx950: 160 KB per workgroup)
                    LV_HIPF(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(ldsBytes)));
                kernel<<<numTreelets, 64, ldsBytes, st>>>(
                        (const uint32_t*)wideIndex.ptr, (const uint32_t*)depth.ptr, (const uint32_t*)evenFlag.ptr,
                        (const float*)leafBox.ptr, ctx->opt.treeletLeaves, ctx->opt.treeletLaneLeaves, ctx->opt.treeletGroupLeaves,
                        (uint32_t*)childL.ptr, (uint32_t*)childR.ptr);
            }
        }
        // refit: one pass per level of the binary tree; the root's stamp is polled every 8 passes
        for (uint32_t pass = 1;; pass++) {
            k_refit_pass<<<nblocks(nInternal), LV_BLOCK, 0, st>>>(nInternal, pass, (const uint32_t*)childL.ptr,
                                                                  (const uint32_t*)childR.ptr, (const float*)leafBox.ptr,
                                                                  (float*)nodeBox.ptr, (uint32_t*)height.ptr,
                                                                  (uint32_t*)flags.ptr);
            if (pass % 8u == 0u) {
                LV_HIPF(hipMemcpyAsync((void*)ctx->pinned, flags.ptr, 4, hipMemcpyDeviceToHost, st));
                LV_HIPF(hipStreamSynchronize(st));
                if (pin[0]) break;
                if (pass > 4096u) { freeAll(); return lv_fail(ctx, LV_E_HIP, "LBVH refit did not converge"); }
            }
        }
            // collapse: one pass per BFS level of the wide tree (see k_collapse_select)
        {
            uint32_t* frontier = (uint32_t*)depth.ptr;
            uint32_t* nextFrontier = (uint32_t*)evenFlag.ptr;
            // the small levels at the top in one launch (k_collapse_top), then one pass per level
            uint32_t count = 1, base = 0;
            if (ctx->opt.collapseTop) {
                k_collapse_top<<<1, LV_BLOCK, 0, st>>>((const uint32_t*)childL.ptr, (const uint32_t*)childR.ptr, (const float*)leafBox.ptr,
                                                       (const float*)nodeBox.ptr, frontier, (float4*)nodesOut.ptr,
                                                       (uint32_t*)bounds.ptr /* free since k_morton */);
                LV_HIPF(hipMemcpyAsync((void*)ctx->pinned, bounds.ptr, 12, hipMemcpyDeviceToHost, st));
                LV_HIPF(hipStreamSynchronize(st));
                count = pin[0]; base = pin[1]; wideLevels = pin[2];
            } else {
                LV_HIPF(hipMemsetAsync(depth.ptr, 0, 4, st)); // frontier A = {root = binary node 0}
            }
            while (count > 0) {
                k_collapse_select<<<nblocks(count), LV_BLOCK, 0, st>>>(frontier, count, (const uint32_t*)childL.ptr,
                                                                       (const uint32_t*)childR.ptr, (const float*)nodeBox.ptr,
                                                                       (uint32_t*)slots.ptr, (uint32_t*)wideIndex.ptr);
                size_t tb = scanBytes;
                LV_HIPF(rocprim::exclusive_scan(sortTmp.ptr, tb, (uint32_t*)wideIndex.ptr, (uint32_t*)flags.ptr, 0u, count,
                                                rocprim::plus<uint32_t>(), st));
                k_collapse_emit<<<nblocks(count), LV_BLOCK, 0, st>>>(count, base, base + count, (const uint32_t*)slots.ptr,
                                                                     (const uint32_t*)flags.ptr, (const float*)leafBox.ptr,
                                                                     (const float*)nodeBox.ptr, nextFrontier,
                                                                     (float4*)nodesOut.ptr);
                LV_HIPF(hipMemcpyAsync((void*)ctx->pinned, (const uint32_t*)flags.ptr + (count - 1), 4, hipMemcpyDeviceToHost, st));
                LV_HIPF(hipMemcpyAsync((void*)(ctx->pinned + 1), (const uint32_t*)wideIndex.ptr + (count - 1), 4,
                                       hipMemcpyDeviceToHost, st));
                LV_HIPF(hipStreamSynchronize(st));
                base += count;
                count = pin[0] + pin[1];
                std::swap(frontier, nextFrontier);
                wideLevels++;
            }
            numWide = base;
        }
    }
    LV_HIPF(hipGetLastError());
    if (timed) LV_HIPF(hipEventRecord(ctx->ev[evBase + 1], st));
    depthOut = 1;
    wideDepthOut = wideLevels ? wideLevels : 1u;
    if (n > 1) {
        LV_HIPF(hipMemcpyAsync((void*)ctx->pinned, height.ptr, 4, hipMemcpyDeviceToHost, st));
        LV_HIPF(hipStreamSynchronize(st));
        depthOut = pin[0];                 // height of the binary LBVH (reported); the traversal stack is sized by wideDepthOut
    } else {
        LV_HIPF(hipStreamSynchronize(st));
    }
    freeAll();
#undef LV_TRY
#undef LV_HIPF
    numNodesOut = numWide;
    return LV_OK;
}

int lv_bvh_build(lv_ctx* ctx) {
    const uint32_t n = ctx->numSegs;
    hipStream_t st = ctx->stream;
    ctx->accelValid = false;
    ctx->bvhDepth = 0;
    ctx->numNodes = 0;
    if (n == 0) {
        ctx->accelValid = true;
        ctx->accelLineWidth = lv_accel_width(ctx);
        return LV_OK;
    }
    const float radius = lv_accel_width(ctx) * 0.5f;
    const float pad = radius * 1e-3f + 1e-6f;
    int rc;
    if ((rc = lv_buf_reserve(ctx, ctx->segs, size_t(n) * 32))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->segAxis, size_t(n) * 16))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->leafSeg, size_t(n) * 4))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->segToLeaf, size_t(n) * 4))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->prismFrames, size_t(n) * 64))) return rc;
    const lv_line_point* points = (const lv_line_point*)ctx->points.ptr;
    const uint32_t* segIdx = (const uint32_t*)ctx->segIdx.ptr;
    rc = lv_bvh_build_core(
            ctx, n, ctx->nodes, ctx->numNodes, ctx->bvhDepth, ctx->wideDepth, 0,
            [&](float* boxOrig, uint32_t* bounds) {
                k_seg_boxes<<<std::min(nblocks(n), 2048u), LV_BLOCK, 0, st>>>(points, segIdx, n, radius, pad, boxOrig, bounds);
            },
            [&](const uint32_t* sortedVals, const float* boxOrig, float* leafBox) {
                k_leaves<<<nblocks(n), LV_BLOCK, 0, st>>>(points, segIdx, boxOrig, sortedVals, n, (float4*)ctx->segs.ptr,
                                                          (float4*)ctx->segAxis.ptr, (uint32_t*)ctx->leafSeg.ptr, (uint32_t*)ctx->segToLeaf.ptr,
                                                          leafBox, (float4*)ctx->prismFrames.ptr);
            });
    if (rc) return rc;
    ctx->accelValid = true;
    ctx->accelLineWidth = lv_accel_width(ctx);
    ctx->evBuildValid = true;
    return LV_OK;
}

// LBVH over the triangle tubes (VulkanRayTracedAmbientOcclusion.cpp:444-456 builds a triangle BLAS/TLAS from the same
// buffers).  The pad depends on the line width like the capsule pad, so the tree is rebuilt when it changes.
int lv_bvh_build_triangles(lv_ctx* ctx) {
    const uint32_t n = ctx->numTris;
    hipStream_t st = ctx->stream;
    ctx->triAccelValid = false;
    ctx->triBvhDepth = 0;
    ctx->numTriNodes = 0;
    const float radius = ctx->opt.lineWidth * 0.5f;
    ctx->triPad = radius * 1e-3f + 1e-6f;
    if (n == 0) {
        ctx->triAccelValid = true;
        ctx->triAccelLineWidth = ctx->opt.lineWidth;
        return LV_OK;
    }
    int rc;
    const uint32_t group = ctx->opt.triLeafSize, nLeaves = (n + group - 1u) / group;
    ctx->triLeafSize = group;
    const lv_tube_vertex* verts = (const lv_tube_vertex*)ctx->triVerts.ptr;
    const uint32_t* triIdx = (const uint32_t*)ctx->triIdx.ptr;
    // 64-B pair records (k_tri_leaves<true>) when the leaves hold two triangles and every pair of the mesh shares two vertex indices
    bool pairs = group == 2u && ctx->opt.triLeafPairs;
    if (pairs) {
        LvDeviceBuffer& flag = ctx->triPairFlag;
        if ((rc = lv_buf_reserve(ctx, flag, 4))) return rc;
        if (!ctx->pinned) LV_HIP(ctx, hipHostMalloc((void**)&ctx->pinned, 64, hipHostMallocDefault));
        LV_HIP(ctx, hipMemsetAsync(flag.ptr, 0, 4, st));
        k_tri_pairs_check<<<std::min(nblocks(nLeaves), 4096u), LV_BLOCK, 0, st>>>(triIdx, n, nLeaves, (uint32_t*)flag.ptr);
        LV_HIP(ctx, hipMemcpyAsync((void*)ctx->pinned, flag.ptr, 4, hipMemcpyDeviceToHost, st));
        LV_HIP(ctx, hipStreamSynchronize(st));
        pairs = ctx->pinned[0] == 0u;
    }
    ctx->triLeafPairs = pairs;
    if ((rc = lv_buf_reserve(ctx, ctx->tris, pairs ? size_t(nLeaves) * 64 : size_t(nLeaves) * group * 48))) return rc;
    const float pad = ctx->triPad;
    rc = lv_bvh_build_core(
            ctx, nLeaves, ctx->triNodes, ctx->numTriNodes, ctx->triBvhDepth, ctx->triWideDepth, 14,
            [&](float* boxOrig, uint32_t* bounds) {
                k_tri_boxes<<<std::min(nblocks(nLeaves), 2048u), LV_BLOCK, 0, st>>>(verts, triIdx, n, group, nLeaves, pad, boxOrig, bounds);
            },
            [&](const uint32_t* sortedVals, const float* boxOrig, float* leafBox) {
                if (pairs)
                    k_tri_leaves<true><<<nblocks(nLeaves), LV_BLOCK, 0, st>>>(verts, triIdx, boxOrig, sortedVals, n, group, nLeaves,
                                                                              (float4*)ctx->tris.ptr, leafBox);
                else
                    k_tri_leaves<false><<<nblocks(nLeaves), LV_BLOCK, 0, st>>>(verts, triIdx, boxOrig, sortedVals, n, group, nLeaves,
                                                                               (float4*)ctx->tris.ptr, leafBox);
            });
    if (rc) return rc;
    ctx->triAccelValid = true;
    ctx->triAccelLineWidth = ctx->opt.lineWidth;
    ctx->evTriBuildValid = true;
    return LV_OK;
}
