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
//   k_refit       bottom-up AABB + height, second arriver continues (agent-scope release/acquire)
//   k_collapse_*  greedy area-guided collapse into 64-byte compressed 4-wide nodes (8-bit child boxes + references), one
//                 BFS level of the wide tree per pass, rocPRIM exclusive scan for the deterministic node numbering
#include <cstring>
#include <utility>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "lv_internal.h"

namespace {

__global__ __launch_bounds__(LV_BLOCK) void k_seg_boxes(const lv_line_point* __restrict__ points,
                                                        const uint32_t* __restrict__ segIdx, uint32_t nSeg, float radius,
                                                        float pad, float* __restrict__ boxOrig,
                                                        uint32_t* __restrict__ boundsOrd) {
    uint32_t s = blockIdx.x * LV_BLOCK + threadIdx.x;
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    if (s < nSeg) {
        const float* p0 = points[segIdx[2 * s]].linePosition;
        const float* p1 = points[segIdx[2 * s + 1]].linePosition;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            mn[k] = (fminf(p0[k], p1[k]) - radius) - pad;
            mx[k] = (fmaxf(p0[k], p1[k]) + radius) + pad;
            boxOrig[6 * size_t(s) + k] = mn[k];
            boxOrig[6 * size_t(s) + 3 + k] = mx[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float a = lv_wave_min(mn[k]), b = lv_wave_max(mx[k]);
        if (lv_lane() == 0) {
            atomicMin(&boundsOrd[k], lv_f2ord(a));
            atomicMax(&boundsOrd[3 + k], lv_f2ord(b));
        }
    }
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
                                                     float4* __restrict__ segs, uint32_t* __restrict__ leafSeg,
                                                     uint32_t* __restrict__ segToLeaf, float* __restrict__ leafBox) {
    uint32_t i = blockIdx.x * LV_BLOCK + threadIdx.x;
    if (i >= nSeg) return;
    uint32_t s = sortedVals[i];
    const lv_line_point& a = points[segIdx[2 * s]];
    const lv_line_point& b = points[segIdx[2 * s + 1]];
    segs[2 * size_t(i)] = make_float4(a.linePosition[0], a.linePosition[1], a.linePosition[2], a.lineAttribute);
    segs[2 * size_t(i) + 1] = make_float4(b.linePosition[0], b.linePosition[1], b.linePosition[2], b.lineAttribute);
    leafSeg[i] = s;
    segToLeaf[s] = i;
#pragma unroll
    for (int k = 0; k < 6; k++) leafBox[6 * size_t(i) + k] = boxOrig[6 * size_t(s) + k];
}

// triangle tubes: padded AABB of every triangle (the same box the ray-triangle test clips t against)
__global__ __launch_bounds__(LV_BLOCK) void k_tri_boxes(const lv_tube_vertex* __restrict__ verts,
                                                        const uint32_t* __restrict__ triIdx, uint32_t nTri, float pad,
                                                        float* __restrict__ boxOrig, uint32_t* __restrict__ boundsOrd) {
    uint32_t s = blockIdx.x * LV_BLOCK + threadIdx.x;
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    if (s < nTri) {
        const float* a = verts[triIdx[3 * size_t(s)]].vertexPosition;
        const float* b = verts[triIdx[3 * size_t(s) + 1]].vertexPosition;
        const float* c = verts[triIdx[3 * size_t(s) + 2]].vertexPosition;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            mn[k] = fminf(fminf(a[k], b[k]), c[k]) - pad;
            mx[k] = fmaxf(fmaxf(a[k], b[k]), c[k]) + pad;
            boxOrig[6 * size_t(s) + k] = mn[k];
            boxOrig[6 * size_t(s) + 3 + k] = mx[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float a = lv_wave_min(mn[k]), b = lv_wave_max(mx[k]);
        if (lv_lane() == 0) {
            atomicMin(&boundsOrd[k], lv_f2ord(a));
            atomicMax(&boundsOrd[3 + k], lv_f2ord(b));
        }
    }
}

// 48-byte triangle records in Morton order: {v0.xyz, original triangle index}{v1.xyz, 0}{v2.xyz, 0}
__global__ __launch_bounds__(LV_BLOCK) void k_tri_leaves(const lv_tube_vertex* __restrict__ verts,
                                                         const uint32_t* __restrict__ triIdx,
                                                         const float* __restrict__ boxOrig,
                                                         const uint32_t* __restrict__ sortedVals, uint32_t nTri,
                                                         float4* __restrict__ tris, float* __restrict__ leafBox) {
    uint32_t i = blockIdx.x * LV_BLOCK + threadIdx.x;
    if (i >= nTri) return;
    uint32_t s = sortedVals[i];
    const float* a = verts[triIdx[3 * size_t(s)]].vertexPosition;
    const float* b = verts[triIdx[3 * size_t(s) + 1]].vertexPosition;
    const float* c = verts[triIdx[3 * size_t(s) + 2]].vertexPosition;
    tris[3 * size_t(i)] = make_float4(a[0], a[1], a[2], __uint_as_float(s));
    tris[3 * size_t(i) + 1] = make_float4(b[0], b[1], b[2], 0.0f);
    tris[3 * size_t(i) + 2] = make_float4(c[0], c[1], c[2], 0.0f);
#pragma unroll
    for (int k = 0; k < 6; k++) leafBox[6 * size_t(i) + k] = boxOrig[6 * size_t(s) + k];
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
                                                     uint32_t* __restrict__ childR, uint32_t* __restrict__ parentInternal,
                                                     uint32_t* __restrict__ parentLeaf) {
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
    if (lo == gamma) { left = uint32_t(gamma) | LV_LEAF_BIT; parentLeaf[gamma] = uint32_t(i); }
    else { left = uint32_t(gamma); parentInternal[gamma] = uint32_t(i); }
    if (hi == gamma + 1) { right = uint32_t(gamma + 1) | LV_LEAF_BIT; parentLeaf[gamma + 1] = uint32_t(i); }
    else { right = uint32_t(gamma + 1); parentInternal[gamma + 1] = uint32_t(i); }
    childL[i] = left;
    childR[i] = right;
    if (i == 0) parentInternal[0] = LV_INVALID;
}

__device__ __forceinline__ void lv_child_box(uint32_t c, const float* __restrict__ leafBox, const float* nodeBox,
                                             const uint32_t* height, float b[6], uint32_t& hgt) {
    // plain loads: callers issue an agent-scope acquire first
    if (c & LV_LEAF_BIT) {
        const float* p = leafBox + 6 * size_t(c & ~LV_LEAF_BIT);
#pragma unroll
        for (int k = 0; k < 6; k++) b[k] = p[k];
        hgt = 0;
    } else {
        const float* p = nodeBox + 6 * size_t(c);
#pragma unroll
        for (int k = 0; k < 6; k++) b[k] = p[k];
        hgt = height[c];
    }
}

// One thread per leaf walks towards the root; the first thread to reach a node stops, the second one (which
// finds flag == 1) owns it.  Cross-CU visibility: per-CU L1s are never refreshed by other CUs' stores and the
// per-XCD L2s are not coherent, so the producer publishes with an agent-scope release before the counter RMW and
// the consumer invalidates with an agent-scope acquire after it (cdna_hip_programming.md §6 Guideline 16).
__global__ __launch_bounds__(LV_BLOCK) void k_refit(int n, const uint32_t* __restrict__ childL,
                                                    const uint32_t* __restrict__ childR,
                                                    const uint32_t* __restrict__ parentInternal,
                                                    const uint32_t* __restrict__ parentLeaf,
                                                    const float* __restrict__ leafBox, float* nodeBox, uint32_t* height,
                                                    uint32_t* flags) {
    int i = blockIdx.x * LV_BLOCK + threadIdx.x;
    if (i >= n) return;
    uint32_t p = parentLeaf[i];
    while (p != LV_INVALID) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t old = __hip_atomic_fetch_add(&flags[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == 0) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        float a[6], b[6];
        uint32_t ha, hb;
        lv_child_box(childL[p], leafBox, nodeBox, height, a, ha);
        lv_child_box(childR[p], leafBox, nodeBox, height, b, hb);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            nodeBox[6 * size_t(p) + k] = fminf(a[k], b[k]);
            nodeBox[6 * size_t(p) + 3 + k] = fmaxf(a[k + 3], b[k + 3]);
        }
        height[p] = max(ha, hb) + 1;
        p = parentInternal[p];
    }
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
// 3.1 children per node; an optimal dynamic-programming collapse reaches 3.5 but traces no faster).
//   k_collapse_select  frontier item i (a binary node) -> its <= 4 slots as binary references + the number of internal ones
//   exclusive scan     -> position of each item's internal slots in the next frontier (deterministic BFS numbering)
//   k_collapse_emit    writes the compressed node (index base + i) and the next frontier
__device__ __forceinline__ float lv_box_half_area(const float* b) {
    const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return (dx * dy + dy * dz) + dz * dx;
}

__global__ __launch_bounds__(LV_BLOCK) void k_collapse_select(const uint32_t* __restrict__ frontier, uint32_t count,
                                                              const uint32_t* __restrict__ childL,
                                                              const uint32_t* __restrict__ childR,
                                                              const float* __restrict__ nodeBox,
                                                              uint32_t* __restrict__ slots,
                                                              uint32_t* __restrict__ internalCount) {
    const uint32_t i = blockIdx.x * LV_BLOCK + threadIdx.x;
    if (i >= count) return;
    const uint32_t r = frontier[i];
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
    slots[4 * size_t(i) + 0] = s0; slots[4 * size_t(i) + 1] = s1; slots[4 * size_t(i) + 2] = s2; slots[4 * size_t(i) + 3] = s3;
    uint32_t ni = 0;
    if (!(s0 & LV_LEAF_BIT)) ni++;
    if (!(s1 & LV_LEAF_BIT)) ni++;
    if (s2 != LV_INVALID && !(s2 & LV_LEAF_BIT)) ni++;
    if (s3 != LV_INVALID && !(s3 & LV_LEAF_BIT)) ni++;
    internalCount[i] = ni;
}

__global__ __launch_bounds__(LV_BLOCK) void k_collapse_emit(uint32_t count, uint32_t base, uint32_t nextBase,
                                                            const uint32_t* __restrict__ slots,
                                                            const uint32_t* __restrict__ offsets,
                                                            const float* __restrict__ leafBox,
                                                            const float* __restrict__ nodeBox,
                                                            uint32_t* __restrict__ nextFrontier, float4* __restrict__ nodes) {
    const uint32_t i = blockIdx.x * LV_BLOCK + threadIdx.x;
    if (i >= count) return;
    uint32_t slotRef[4];
    float b[4][6];
    int ns = 0;
    uint32_t j = offsets[i];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t c = slots[4 * size_t(i) + k];
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
    lv_write_wide_node(nodes + 4 * size_t(base + i), ns, slotRef, b);
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
    nodes[1] = make_float4(sc[1], sc[2], __uint_as_float(0u), __uint_as_float(0u));
    nodes[2] = make_float4(__uint_as_float(0u), __uint_as_float(255u), __uint_as_float(255u), __uint_as_float(255u));
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
                             uint32_t& wideDepthOut, bool timed, BOXES&& boxes, LEAVES&& leaves) {
    hipStream_t st = ctx->stream;
    const uint32_t nInternal = n > 1 ? n - 1 : 1;
    uint32_t numWide = 1, wideLevels = 0;
    int rc;
    // 4-wide nodes: one per even-depth binary node; at most all of them (a degenerate chain has ~n/2)
    if ((rc = lv_buf_reserve(ctx, nodesOut, size_t(nInternal) * 64))) return rc;

    // temporaries
    LvDeviceBuffer boxOrig, leafBox, nodeBox, keysA, keysB, valsA, valsB, childL, childR, parI, parL, height, flags, bounds,
            sortTmp, depth, evenFlag, wideIndex, slots;
    auto freeAll = [&]() {
        for (LvDeviceBuffer* b : {&boxOrig, &leafBox, &nodeBox, &keysA, &keysB, &valsA, &valsB, &childL, &childR, &parI,
                                  &parL, &height, &flags, &bounds, &sortTmp, &depth, &evenFlag, &wideIndex, &slots})
            lv_buf_free(*b);
    };
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

    LV_TRY(lv_buf_reserve(ctx, boxOrig, size_t(n) * 24));
    LV_TRY(lv_buf_reserve(ctx, leafBox, size_t(n) * 24));
    LV_TRY(lv_buf_reserve(ctx, nodeBox, size_t(nInternal) * 24));
    LV_TRY(lv_buf_reserve(ctx, keysA, size_t(n) * 8));
    LV_TRY(lv_buf_reserve(ctx, keysB, size_t(n) * 8));
    LV_TRY(lv_buf_reserve(ctx, valsA, size_t(n) * 4));
    LV_TRY(lv_buf_reserve(ctx, valsB, size_t(n) * 4));
    LV_TRY(lv_buf_reserve(ctx, childL, size_t(nInternal) * 4));
    LV_TRY(lv_buf_reserve(ctx, childR, size_t(nInternal) * 4));
    LV_TRY(lv_buf_reserve(ctx, parI, size_t(nInternal) * 4));
    LV_TRY(lv_buf_reserve(ctx, parL, size_t(n) * 4));
    LV_TRY(lv_buf_reserve(ctx, height, size_t(nInternal) * 4));
    LV_TRY(lv_buf_reserve(ctx, flags, size_t(nInternal) * 4));
    LV_TRY(lv_buf_reserve(ctx, bounds, 6 * 4));
    LV_TRY(lv_buf_reserve(ctx, depth, size_t(nInternal) * 4));
    LV_TRY(lv_buf_reserve(ctx, evenFlag, size_t(nInternal) * 4));
    LV_TRY(lv_buf_reserve(ctx, wideIndex, size_t(nInternal) * 4));
    LV_TRY(lv_buf_reserve(ctx, slots, size_t(nInternal) * 16));

    if (timed) LV_HIPF(hipEventRecord(ctx->ev[0], st));
    // bounds: min slots start at ord(+big) = 0xFFFFFFFF-ish, max slots at 0
    {
        uint32_t init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
        LV_HIPF(hipMemcpyAsync(bounds.ptr, init, sizeof(init), hipMemcpyHostToDevice, st));
        LV_HIPF(hipStreamSynchronize(st)); // init[] is a stack array
    }
    boxes((float*)boxOrig.ptr, (uint32_t*)bounds.ptr);
    k_morton<<<nblocks(n), LV_BLOCK, 0, st>>>((const float*)boxOrig.ptr, n, (const uint32_t*)bounds.ptr,
                                              (uint64_t*)keysA.ptr, (uint32_t*)valsA.ptr);
    {
        size_t tmpBytes = 0;
        LV_HIPF(rocprim::radix_sort_pairs(nullptr, tmpBytes, (uint64_t*)keysA.ptr, (uint64_t*)keysB.ptr,
                                          (uint32_t*)valsA.ptr, (uint32_t*)valsB.ptr, n, 0, 63, st));
        LV_TRY(lv_buf_reserve(ctx, sortTmp, tmpBytes ? tmpBytes : 16));
        LV_HIPF(rocprim::radix_sort_pairs(sortTmp.ptr, tmpBytes, (uint64_t*)keysA.ptr, (uint64_t*)keysB.ptr,
                                          (uint32_t*)valsA.ptr, (uint32_t*)valsB.ptr, n, 0, 63, st));
    }
    leaves((const uint32_t*)valsB.ptr, (const float*)boxOrig.ptr, (float*)leafBox.ptr);
    if (n == 1) {
        k_single_node<<<1, 64, 0, st>>>((const float*)leafBox.ptr, (float4*)nodesOut.ptr);
    } else {
        LV_HIPF(hipMemsetAsync(flags.ptr, 0, size_t(nInternal) * 4, st));
        k_karras<<<nblocks(nInternal), LV_BLOCK, 0, st>>>((const uint64_t*)keysB.ptr, int(n), (uint32_t*)childL.ptr,
                                                          (uint32_t*)childR.ptr, (uint32_t*)parI.ptr,
                                                          (uint32_t*)parL.ptr);
        k_refit<<<nblocks(n), LV_BLOCK, 0, st>>>(int(n), (const uint32_t*)childL.ptr, (const uint32_t*)childR.ptr,
                                                 (const uint32_t*)parI.ptr, (const uint32_t*)parL.ptr,
                                                 (const float*)leafBox.ptr, (float*)nodeBox.ptr, (uint32_t*)height.ptr,
                                                 (uint32_t*)flags.ptr);
        // collapse: one pass per BFS level of the wide tree (see k_collapse_select)
        {
            uint32_t zero = 0u;
            LV_HIPF(hipMemcpyAsync(depth.ptr, &zero, 4, hipMemcpyHostToDevice, st)); // frontier A = {root}
            LV_HIPF(hipStreamSynchronize(st));
            uint32_t* frontier = (uint32_t*)depth.ptr;
            uint32_t* nextFrontier = (uint32_t*)evenFlag.ptr;
            uint32_t count = 1, base = 0;
            size_t scanBytes = 0;
            LV_HIPF(rocprim::exclusive_scan(nullptr, scanBytes, (uint32_t*)wideIndex.ptr, (uint32_t*)flags.ptr, 0u, nInternal,
                                            rocprim::plus<uint32_t>(), st));
            LV_TRY(lv_buf_reserve(ctx, sortTmp, scanBytes ? scanBytes : 16));
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
                uint32_t lastOff = 0, lastCnt = 0;
                LV_HIPF(hipMemcpyAsync(&lastOff, (const uint32_t*)flags.ptr + (count - 1), 4, hipMemcpyDeviceToHost, st));
                LV_HIPF(hipMemcpyAsync(&lastCnt, (const uint32_t*)wideIndex.ptr + (count - 1), 4, hipMemcpyDeviceToHost, st));
                LV_HIPF(hipStreamSynchronize(st));
                base += count;
                count = lastOff + lastCnt;
                std::swap(frontier, nextFrontier);
                wideLevels++;
            }
            numWide = base;
        }
    }
    LV_HIPF(hipGetLastError());
    if (timed) LV_HIPF(hipEventRecord(ctx->ev[1], st));
    depthOut = 1;
    wideDepthOut = wideLevels ? wideLevels : 1u;
    if (n > 1) {
        uint32_t h = 0;
        LV_HIPF(hipMemcpyAsync(&h, height.ptr, 4, hipMemcpyDeviceToHost, st));
        LV_HIPF(hipStreamSynchronize(st));
        depthOut = h;                 // height of the binary LBVH (reported); the traversal stack is sized by wideDepthOut
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
        ctx->accelLineWidth = ctx->opt.lineWidth;
        return LV_OK;
    }
    const float radius = ctx->opt.lineWidth * 0.5f;
    const float pad = radius * 1e-3f + 1e-6f;
    int rc;
    if ((rc = lv_buf_reserve(ctx, ctx->segs, size_t(n) * 32))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->leafSeg, size_t(n) * 4))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->segToLeaf, size_t(n) * 4))) return rc;
    const lv_line_point* points = (const lv_line_point*)ctx->points.ptr;
    const uint32_t* segIdx = (const uint32_t*)ctx->segIdx.ptr;
    rc = lv_bvh_build_core(
            ctx, n, ctx->nodes, ctx->numNodes, ctx->bvhDepth, ctx->wideDepth, true,
            [&](float* boxOrig, uint32_t* bounds) {
                k_seg_boxes<<<nblocks(n), LV_BLOCK, 0, st>>>(points, segIdx, n, radius, pad, boxOrig, bounds);
            },
            [&](const uint32_t* sortedVals, const float* boxOrig, float* leafBox) {
                k_leaves<<<nblocks(n), LV_BLOCK, 0, st>>>(points, segIdx, boxOrig, sortedVals, n, (float4*)ctx->segs.ptr,
                                                          (uint32_t*)ctx->leafSeg.ptr, (uint32_t*)ctx->segToLeaf.ptr,
                                                          leafBox);
            });
    if (rc) return rc;
    ctx->accelValid = true;
    ctx->accelLineWidth = ctx->opt.lineWidth;
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
    if ((rc = lv_buf_reserve(ctx, ctx->tris, size_t(n) * 48))) return rc;
    const lv_tube_vertex* verts = (const lv_tube_vertex*)ctx->triVerts.ptr;
    const uint32_t* triIdx = (const uint32_t*)ctx->triIdx.ptr;
    const float pad = ctx->triPad;
    rc = lv_bvh_build_core(
            ctx, n, ctx->triNodes, ctx->numTriNodes, ctx->triBvhDepth, ctx->triWideDepth, false,
            [&](float* boxOrig, uint32_t* bounds) {
                k_tri_boxes<<<nblocks(n), LV_BLOCK, 0, st>>>(verts, triIdx, n, pad, boxOrig, bounds);
            },
            [&](const uint32_t* sortedVals, const float* boxOrig, float* leafBox) {
                k_tri_leaves<<<nblocks(n), LV_BLOCK, 0, st>>>(verts, triIdx, boxOrig, sortedVals, n,
                                                              (float4*)ctx->tris.ptr, leafBox);
            });
    if (rc) return rc;
    ctx->triAccelValid = true;
    ctx->triAccelLineWidth = ctx->opt.lineWidth;
    return LV_OK;
}
