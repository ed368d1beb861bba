// lv_lines.hip -- data set -> device geometry without the host in between (gfx950 only).
//
// The reference prepares both geometries of a line set on ONE host thread and uploads them: getLinePassTubeAabbRenderData
// (src/LineData/LineDataFlow.cpp:2112-2277: the 48-byte line points + index pairs the ray tracer and the PPLL gather use) and
// createCappedTriangleTubesRenderDataCPU (src/Renderers/Tubes/CappedTriangleTubesCPU.cpp:214-383 via LineDataFlow.cpp:1912-2110: the
// capped N-gon tubes the RTAO pass traces) -- "seconds at 1 M segments", and again for every line-width change, because the tube
// radius is baked into the vertices.  Here the trajectories themselves live in HBM (lv_set_trajectories: 16 bytes per point) and both
// products are written by kernels, byte for byte what linevis_amd/host/LineData.cpp and host/Tubes.cpp produce:
//
//   a2   k_lp_tangents   one lane per point: central-difference tangent, its length, the "keep" flag (|t| >= 1e-4, :2160)
//        rocPRIM scan    rank of every kept point; per line: kept points (a line that keeps fewer than two keeps none, :2209-2221)
//        k_lp_normals    the only recurrence -- the line normal is carried from kept point to kept point (Gram-Schmidt against the
//                        tangent with the fallback axes of :2171-2177): one wave per line stages the tangents in LDS, lane 0 walks them
//        k_lp_records    one lane per point: 48-byte record, index pair
//   a14  k_tess_counts   vertices / indices per line incl. the reference's quirks for lines with < 2 valid points, two scans
//        k_tess_body     one lane per (line point, ring vertex): Tubes.cpp:53-85 (circle table from the host, incremental rotation as
//                        Tubes.cpp:34-51) + the two triangles of the side towards the next point
//        k_tess_caps     one lane per cap vertex / cap triangle: CappedTriangleTubesCPU.cpp:33-211
//        k_tess_points   the mesh's line-point table (LineDataFlow.cpp:1996-2020)
// Every formula has the host layer's evaluation order (-ffp-contract=off on both sides; + - * / sqrt are IEEE-exact on gfx950); the
// trigonometric tables (circle, cap rings) come from the host's libm once per tessellation, as the reference computes them on the CPU.
// Plain flow lines only: band data (elliptic tubes) and the rotating helicity bands (whose rotation is a running float sum across ALL
// lines, LineDataFlow.cpp:1994,2014-2028) stay on the host path (lv_set_lines + lv_set_tube_triangle_mesh).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include <rocprim/device/device_scan.hpp>

#include "lv_internal.h"

namespace {

const float kTwoPi = 6.28318530717958647692f;
const float kHalfPi = 1.57079632679489661923f;

__host__ __device__ inline uint32_t nblk(uint64_t n) { return uint32_t((n + LV_BLOCK - 1) / LV_BLOCK); }

__device__ __forceinline__ f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }

// line of point i: the last line whose offset is <= i and that is not empty at i (offsets are non-decreasing; empty lines repeat one)
__device__ __forceinline__ uint32_t lv_line_of(const uint32_t* __restrict__ off, uint32_t numLines, uint32_t i) {
    uint32_t lo = 0, hi = numLines; // first index in [0, numLines] with off[idx] > i
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off[mid] > i) hi = mid; else lo = mid + 1;
    }
    return lo - 1u;
}

// ---- a2, pass 1: tangents + keep flags.  tang[i] = {unit tangent, keep}; flags[i] = keep (scanned into ranks afterwards)
__global__ __launch_bounds__(LV_BLOCK) void k_lp_tangents(const float* __restrict__ pos, const uint32_t* __restrict__ off,
                                                          uint32_t numLines, uint32_t numPoints, float4* __restrict__ tang,
                                                          uint32_t* __restrict__ flags, uint32_t* __restrict__ lineOf) {
    const uint32_t i = blockIdx.x * LV_BLOCK + threadIdx.x;
    if (i > numPoints) return;
    if (i == numPoints) { flags[i] = 0u; return; } // the scan's last element: rank[numPoints] = number of kept points
    const uint32_t li = lv_line_of(off, numLines, i);
    const uint32_t b = off[li], n = off[li + 1] - b, j = i - b;
    lineOf[i] = li;
    uint32_t keep = 0u;
    f3 t = mk3(0.0f, 0.0f, 0.0f);
    if (n >= 2u) {
        const f3 ahead = ld3(pos + 3 * size_t(b + (j + 1u < n ? j + 1u : j)));
        const f3 behind = ld3(pos + 3 * size_t(b + (j > 0u ? j - 1u : j)));
        const f3 d = ahead - behind;          // one-sided at the two ends, central in between
        const float len = len3(d);
        if (!(len < 0.0001f)) { keep = 1u; t = mk3(d.x / len, d.y / len, d.z / len); }
    }
    tang[i] = make_float4(t.x, t.y, t.z, keep ? 1.0f : 0.0f);
    flags[i] = keep;
}

// per line: valid points m (the tessellation's count), kept records (m if >= 2) and segments, packed as (segments << 32) | records
__global__ __launch_bounds__(LV_BLOCK) void k_lp_line_counts(const uint32_t* __restrict__ off, const uint32_t* __restrict__ rank,
                                                             uint32_t numLines, uint32_t* __restrict__ lineValid,
                                                             unsigned long long* __restrict__ packed) {
    const uint32_t li = blockIdx.x * LV_BLOCK + threadIdx.x;
    if (li > numLines) return;
    if (li == numLines) { packed[li] = 0ull; return; }
    const uint32_t m = rank[off[li + 1]] - rank[off[li]];
    lineValid[li] = m;
    const uint32_t rec = m >= 2u ? m : 0u;
    packed[li] = (uint64_t(rec ? rec - 1u : 0u) << 32) | rec;
}

// ---- a2, pass 2: the carried line normal.  One wave per line; the tangents of LV_LP_CHUNK points at a time go through LDS, lane 0 walks
// them (the recurrence is ~90 dependent float operations per kept point) and the normals leave coalesced.
#define LV_LP_CHUNK 2048u
__global__ __launch_bounds__(LV_WAVE) void k_lp_normals(const float4* __restrict__ tang, const uint32_t* __restrict__ off,
                                                        const uint32_t* __restrict__ lineValid, float4* __restrict__ normalOut) {
    __shared__ float4 s[LV_LP_CHUNK];
    const uint32_t li = blockIdx.x;
    if (lineValid[li] < 2u) return;           // a tube of one point is dropped: no records, no normals
    const uint32_t b = off[li], n = off[li + 1] - b, lane = threadIdx.x;
    f3 carried = mk3(1.0f, 0.0f, 0.0f);       // lastLineNormal
    for (uint32_t base = 0; base < n; base += LV_LP_CHUNK) {
        const uint32_t cnt = min(LV_LP_CHUNK, n - base);
        for (uint32_t j = lane; j < cnt; j += LV_WAVE) s[j] = tang[size_t(b) + base + j];
        __syncthreads();
        if (lane == 0u) {
            for (uint32_t j = 0; j < cnt; j++) {
                const float4 v = s[j];
                if (v.w == 0.0f) continue;
                const f3 t = mk3(v.x, v.y, v.z);
                f3 axis = carried;
                if (len3(cross3(axis, t)) < 0.01f) {              // tangent (anti)parallel to the previous normal
                    axis = mk3(0.0f, 1.0f, 0.0f);
                    if (len3(cross3(axis, t)) < 0.01f) axis = mk3(0.0f, 0.0f, 1.0f);
                }
                carried = norm3(axis - dot3(axis, t) * t);        // Gram-Schmidt
                s[j] = make_float4(carried.x, carried.y, carried.z, 1.0f);
            }
        }
        __syncthreads();
        for (uint32_t j = lane; j < cnt; j += LV_WAVE) normalOut[size_t(b) + base + j] = s[j];
        __syncthreads();
    }
}

struct LvLineRef {       // per line, kept between tessellations
    uint32_t recOff;     // first record (= first entry of the mesh's line-point table)
    uint32_t segOff;     // first segment
    uint32_t firstIdx;   // min(first valid point, n - 2): the start cap's centre (CappedTriangleTubesCPU.cpp:262-263)
    uint32_t lastIdx;    // max(last valid point, 1)
};

// ---- a2, pass 3: records + index pairs
__global__ __launch_bounds__(LV_BLOCK) void k_lp_records(const float* __restrict__ pos, const float* __restrict__ attr,
                                                         const uint32_t* __restrict__ off, const uint32_t* __restrict__ lineOf,
                                                         const uint32_t* __restrict__ rank, const uint32_t* __restrict__ lineValid,
                                                         const unsigned long long* __restrict__ packedOff, const float4* __restrict__ tang,
                                                         const float4* __restrict__ normal, uint32_t numPoints,
                                                         float4* __restrict__ recOut, uint32_t* __restrict__ segOut,
                                                         uint32_t* __restrict__ recLine, LvLineRef* __restrict__ lineRef) {
    const uint32_t i = blockIdx.x * LV_BLOCK + threadIdx.x;
    if (i >= numPoints) return;
    const float4 t = tang[i];
    if (t.w == 0.0f) return;
    const uint32_t li = lineOf[i], m = lineValid[li];
    const uint32_t b = off[li], n = off[li + 1] - b, j = i - b;
    const uint32_t o = rank[i] - rank[b];
    // the tessellation's first / last valid point (also of lines that keep a single point: their caps are never built)
    if (o == 0u) lineRef[li].firstIdx = min(j, n - 2u);
    if (o == m - 1u) lineRef[li].lastIdx = max(j, 1u);
    if (m < 2u) return;
    const unsigned long long po = packedOff[li];
    const uint32_t recOff = uint32_t(po), segOff = uint32_t(po >> 32), r = recOff + o;
    if (o == 0u) { lineRef[li].recOff = recOff; lineRef[li].segOff = segOff; }
    const float4 nn = normal[i];
    const float a = attr ? attr[i] : 0.0f;
    recOut[3 * size_t(r)] = make_float4(pos[3 * size_t(i)], pos[3 * size_t(i) + 1], pos[3 * size_t(i) + 2], a);
    recOut[3 * size_t(r) + 1] = make_float4(t.x, t.y, t.z, 0.0f);                       // lineRotation = 0
    recOut[3 * size_t(r) + 2] = make_float4(nn.x, nn.y, nn.z, __uint_as_float(0u));     // lineStartIndex = 0 (LineDataFlow.cpp:2199-2207)
    recLine[r] = li;
    if (o > 0u) {
        const size_t sgm = size_t(segOff) + o - 1u;
        segOut[2 * sgm] = r - 1u;
        segOut[2 * sgm + 1] = r;
    }
}

// ---------------------------------------------------------------- a14: tessellation
struct LvTessParams {
    float circle[LV_PRISM_MAX_SUBDIV][3];   // initGlobalCircleVertexPositions (Tubes.cpp:34-51), radius included
    float radius;
    uint32_t n;                             // N = max(tube_num_subdivisions, 4)
    uint32_t nLat;                          // N / 2
    uint32_t capVerts, capIdx;              // per cap: nLon (nLat - 1) + 1 vertices, nLon (nLat - 1) 6 + nLon 3 indices
};

// vertices / indices per line, the reference's quirks kept (host/Tubes.cpp "placement"): a line with one valid point keeps the start
// cap's (zero) index range but no vertices; one with no valid point keeps the start cap's zero vertices AND zero indices
__global__ __launch_bounds__(LV_BLOCK) void k_tess_counts(const uint32_t* __restrict__ off, const uint32_t* __restrict__ lineValid,
                                                          uint32_t numLines, LvTessParams P, unsigned long long* __restrict__ nv,
                                                          unsigned long long* __restrict__ ni) {
    const uint32_t li = blockIdx.x * LV_BLOCK + threadIdx.x;
    if (li > numLines) return;
    unsigned long long v = 0ull, x = 0ull;
    if (li < numLines && off[li + 1] - off[li] >= 2u) {
        const unsigned long long m = lineValid[li];
        if (m >= 2ull) { v = 2ull * P.capVerts + m * P.n; x = 2ull * P.capIdx + (m - 1ull) * P.n * 6ull; }
        else if (m == 1ull) { x = P.capIdx; }
        else { v = P.capVerts; x = P.capIdx; }
    }
    nv[li] = v;
    ni[li] = x;
}

__device__ __forceinline__ f3 lv_combine(f3 pt, f3 a, f3 b, f3 c) {
    return mk3((pt.x * a.x + pt.y * b.x) + pt.z * c.x, (pt.x * a.y + pt.y * b.y) + pt.z * c.y, (pt.x * a.z + pt.y * b.z) + pt.z * c.z);
}
__device__ __forceinline__ void lv_store_vertex(lv_tube_vertex* V, size_t at, f3 p, uint32_t linePoint, f3 n, float phi) {
    float4* w = (float4*)(V + at);
    w[0] = make_float4(p.x, p.y, p.z, __uint_as_float(linePoint));
    w[1] = make_float4(n.x, n.y, n.z, phi);
}

// body: ring vertex j of line point r (insertOrientedCirclePoints, Tubes.cpp:53-85) and the two triangles of side j towards the next point
__global__ __launch_bounds__(LV_BLOCK) void k_tess_body(const float4* __restrict__ rec, const uint32_t* __restrict__ recLine,
                                                        const LvLineRef* __restrict__ lineRef, const uint32_t* __restrict__ lineValid,
                                                        const unsigned long long* __restrict__ vOff,
                                                        const unsigned long long* __restrict__ iOff, uint32_t numRecords,
                                                        LvTessParams P, lv_tube_vertex* __restrict__ V, uint32_t* __restrict__ I) {
    const uint64_t g = uint64_t(blockIdx.x) * LV_BLOCK + threadIdx.x;
    if (g >= uint64_t(numRecords) * P.n) return;
    const uint32_t r = uint32_t(g / P.n), j = uint32_t(g % P.n);
    const uint32_t li = recLine[r], k = r - lineRef[li].recOff, m = lineValid[li];
    const float4 c4 = rec[3 * size_t(r)], t4 = rec[3 * size_t(r) + 1], n4 = rec[3 * size_t(r) + 2];
    const f3 center = mk3(c4.x, c4.y, c4.z), tangent = mk3(t4.x, t4.y, t4.z), normal = mk3(n4.x, n4.y, n4.z);
    const f3 binormal = cross3(tangent, normal);
    const f3 off = lv_combine(mk3(P.circle[j][0], P.circle[j][1], P.circle[j][2]), normal, binormal, tangent);
    const f3 pos = mk3(off.x + center.x, off.y + center.y, off.z + center.z);
    const uint32_t bodyV = uint32_t(vOff[li]) + P.capVerts;
    lv_store_vertex(V, size_t(bodyV) + size_t(k) * P.n + j, pos, r, norm3(pos - center), float(j) / float(P.n) * kTwoPi);
    if (k + 1u < m) {
        const uint32_t j1 = (j + 1u) % P.n;
        const uint32_t a = bodyV + k * P.n + j, b = bodyV + k * P.n + j1, c = bodyV + (k + 1u) * P.n + j1, d = bodyV + (k + 1u) * P.n + j;
        uint32_t* w = I + (size_t(iOff[li]) + P.capIdx + (size_t(k) * P.n + j) * 6u);
        w[0] = a; w[1] = b; w[2] = c;
        w[3] = a; w[4] = c; w[5] = d;
    }
}

// caps: slot < 2 capVerts = a cap vertex (start cap first), the rest = a cap triangle; quirk lines write their zero ranges here
// capTable: per (lat - 1) * nLon + lon: {start point xyz, start phi}{stop point xyz, stop phi} (addHemisphereToMeshStart / Stop,
// CappedTriangleTubesCPU.cpp:33-211)
__global__ __launch_bounds__(LV_BLOCK) void k_tess_caps(const float* __restrict__ pos, const uint32_t* __restrict__ off,
                                                        const float4* __restrict__ rec, const LvLineRef* __restrict__ lineRef,
                                                        const uint32_t* __restrict__ lineValid, const unsigned long long* __restrict__ vOff,
                                                        const unsigned long long* __restrict__ iOff, uint32_t numLines, LvTessParams P,
                                                        const float4* __restrict__ capTable, lv_tube_vertex* __restrict__ V,
                                                        uint32_t* __restrict__ I) {
    const uint32_t capTris = P.capIdx / 3u, slots = 2u * P.capVerts + 2u * capTris;
    const uint64_t g = uint64_t(blockIdx.x) * LV_BLOCK + threadIdx.x;
    if (g >= uint64_t(numLines) * slots) return;
    const uint32_t li = uint32_t(g / slots), slot = uint32_t(g % slots);
    const uint32_t b = off[li], n = off[li + 1] - b;
    if (n < 2u) return;
    const uint32_t m = lineValid[li], nLon = P.n, nLat = P.nLat;
    const uint32_t v0 = uint32_t(vOff[li]);
    const size_t i0 = size_t(iOff[li]);
    if (m < 2u) {   // zero ranges of the degenerate lines
        if (slot < P.capVerts) {
            if (m == 0u) { float4* w = (float4*)(V + size_t(v0) + slot); w[0] = make_float4(0.f, 0.f, 0.f, 0.f); w[1] = w[0]; }
        } else if (slot >= 2u * P.capVerts && slot < 2u * P.capVerts + capTris) {
            uint32_t* w = I + i0 + 3u * size_t(slot - 2u * P.capVerts);
            w[0] = w[1] = w[2] = 0u;
        }
        return;
    }
    const LvLineRef L = lineRef[li];
    const uint32_t capStartV = v0, bodyV = v0 + P.capVerts, capEndV = bodyV + m * P.n;
    if (slot < 2u * P.capVerts) {
        const bool start = slot < P.capVerts;
        const uint32_t v = start ? slot : slot - P.capVerts;
        uint32_t lat, lon;
        if (start) {   // pole first, then the rings from the pole towards the tube
            if (v == 0u) { lat = nLat; lon = 0u; } else { lat = nLat - 1u - (v - 1u) / nLon; lon = (v - 1u) % nLon; }
        } else {       // rings from the tube towards the pole, pole last
            if (v == P.capVerts - 1u) { lat = nLat; lon = 0u; } else { lat = 1u + v / nLon; lon = v % nLon; }
        }
        const uint32_t r = start ? L.recOff : L.recOff + m - 1u;
        const float4 n4 = rec[3 * size_t(r) + 2];
        const f3 normal = mk3(n4.x, n4.y, n4.z);
        const float* c0 = pos + 3 * size_t(b + (start ? L.firstIdx : L.lastIdx));
        const float* c1 = pos + 3 * size_t(b + (start ? L.firstIdx + 1u : L.lastIdx - 1u));
        const f3 center = ld3(c0);
        const f3 tangent = norm3(center - ld3(c1));
        const f3 binormal = cross3(normal, tangent);
        const f3 sT = P.radius * tangent, sN = P.radius * normal, sB = P.radius * binormal;
        const float4 e = capTable[2u * ((lat - 1u) * nLon + lon) + (start ? 0u : 1u)];
        const f3 o = lv_combine(mk3(e.x, e.y, e.z), sN, sB, sT);
        lv_store_vertex(V, size_t(start ? capStartV : capEndV) + v, mk3(o.x + center.x, o.y + center.y, o.z + center.z),
                        r | 0x80000000u, norm3(o), e.w);
        return;
    }
    uint32_t q = slot - 2u * P.capVerts;
    uint32_t x, y, z;
    uint32_t* w;
    if (q < capTris) {   // start cap: fan at the pole, then quads ring by ring down to the tube's first circle
        w = I + i0 + 3u * size_t(q);
        const uint32_t ring0 = capStartV + 1u;
        if (q < nLon) {
            const uint32_t l0 = q, l1 = (q + 1u) % nLon;
            x = capStartV; y = ring0 + l1; z = ring0 + l0;
        } else {
            const uint32_t qq = q - nLon, lat = 1u + qq / (2u * nLon), rem = qq % (2u * nLon), l0 = rem >> 1, l1 = (l0 + 1u) % nLon;
            const uint32_t a = ring0 + (lat - 1u) * nLon, bb = ring0 + lat * nLon;
            if ((rem & 1u) == 0u) { x = a + l0; y = a + l1; z = bb + l0; } else { x = a + l1; y = bb + l1; z = bb + l0; }
        }
    } else {             // end cap: from the tube's last circle through the cap rings, fan at the pole
        q -= capTris;
        w = I + i0 + P.capIdx + size_t(m - 1u) * P.n * 6u + 3u * size_t(q);
        const uint32_t lastCircle = capEndV - P.n, quads = 2u * nLon * (nLat - 1u);
        if (q < quads) {
            const uint32_t lat = q / (2u * nLon), rem = q % (2u * nLon), l0 = rem >> 1, l1 = (l0 + 1u) % nLon;
            const uint32_t a = lastCircle + lat * nLon, bb = lastCircle + (lat + 1u) * nLon;
            if ((rem & 1u) == 0u) { x = a + l0; y = a + l1; z = bb + l0; } else { x = a + l1; y = bb + l1; z = bb + l0; }
        } else {
            const uint32_t l0 = q - quads, l1 = (l0 + 1u) % nLon;
            const uint32_t a = lastCircle + (nLat - 1u) * nLon, bb = lastCircle + nLat * nLon;
            x = a + l0; y = a + l1; z = bb;
        }
    }
    w[0] = x; w[1] = y; w[2] = z;
}

// the mesh's line-point table: the records with lineStartIndex = the line's first entry (LineDataFlow.cpp:1996-2020: it advances
// when the trajectory index changes, i.e. it is the first record of the line)
__global__ __launch_bounds__(LV_BLOCK) void k_tess_points(const float4* __restrict__ rec, const uint32_t* __restrict__ recLine,
                                                          const LvLineRef* __restrict__ lineRef, uint32_t numRecords,
                                                          float4* __restrict__ out) {
    const uint32_t r = blockIdx.x * LV_BLOCK + threadIdx.x;
    if (r >= numRecords) return;
    out[3 * size_t(r)] = rec[3 * size_t(r)];
    out[3 * size_t(r) + 1] = rec[3 * size_t(r) + 1];
    float4 n = rec[3 * size_t(r) + 2];
    n.w = __uint_as_float(lineRef[recLine[r]].recOff);
    out[3 * size_t(r) + 2] = n;
}

} // namespace

// LineRenderer::setLineData(LineDataPtr&, bool) (LineRenderer.hpp:98) for plain flow lines, with LineDataFlow::setTrajectoryData's
// arrays (LineDataFlow.cpp:468-578) instead of the host-built render data: see include/linevis_hip.h.
int lv_set_trajectories(lv_ctx* ctx, const float* positions, const float* attribute, const uint32_t* line_offsets, uint32_t num_lines) {
    if (!ctx) return LV_E_INVALID;
    if (!line_offsets) return lv_fail(ctx, LV_E_INVALID, "null line_offsets (num_lines + 1 entries, even for zero lines)");
    if (line_offsets[0] != 0u) return lv_fail(ctx, LV_E_INVALID, "line_offsets[0] must be 0");
    for (uint32_t l = 0; l < num_lines; l++)
        if (line_offsets[l + 1] < line_offsets[l])
            return lv_fail(ctx, LV_E_INVALID, "line_offsets must not decrease (line %u: %u -> %u)", l, line_offsets[l], line_offsets[l + 1]);
    const uint32_t numPoints = line_offsets[num_lines];
    if (numPoints && !positions) return lv_fail(ctx, LV_E_INVALID, "null positions");
    if (numPoints > 0x03FFFFFFu) return lv_fail(ctx, LV_E_CAPACITY, "at most 2^26-1 points (leaf index field of the AO work queue)");
    (void)hipSetDevice(ctx->device);
    lv_invalidate_bake(ctx);   // before any buffer is touched: a bake in flight on the second stream still reads the old ones
    hipStream_t st = ctx->stream;
    int rc;
    if ((rc = lv_buf_reserve(ctx, ctx->trajPos, size_t(numPoints) * 12))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->trajAttr, size_t(numPoints) * 4))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->trajOff, size_t(num_lines + 1u) * 4))) return rc;
    if (numPoints) LV_HIP(ctx, hipMemcpyAsync(ctx->trajPos.ptr, positions, size_t(numPoints) * 12, hipMemcpyHostToDevice, st));
    if (numPoints && attribute) LV_HIP(ctx, hipMemcpyAsync(ctx->trajAttr.ptr, attribute, size_t(numPoints) * 4, hipMemcpyHostToDevice, st));
    LV_HIP(ctx, hipMemcpyAsync(ctx->trajOff.ptr, line_offsets, size_t(num_lines + 1u) * 4, hipMemcpyHostToDevice, st));
    ctx->trajNumLines = num_lines;
    ctx->trajNumPoints = numPoints;
    ctx->trajHasAttr = attribute != nullptr;
    ctx->trajSet = false;

    // ---- a2 on the device
    const uint32_t* off = (const uint32_t*)ctx->trajOff.ptr;
    size_t scanBytes = 0, scanBytes64 = 0;
    LV_HIP(ctx, rocprim::exclusive_scan(nullptr, scanBytes, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, size_t(numPoints) + 1,
                                        rocprim::plus<uint32_t>(), st));
    LV_HIP(ctx, rocprim::exclusive_scan(nullptr, scanBytes64, (unsigned long long*)nullptr, (unsigned long long*)nullptr, 0ull,
                                        size_t(num_lines) + 1, rocprim::plus<unsigned long long>(), st));
    // scratch out of the build arena (dead once the records are written; the LBVH builds reuse it)
    struct Req { void** p; size_t bytes; };
    void *tang = nullptr, *normal = nullptr, *flags = nullptr, *rank = nullptr, *lineOf = nullptr, *packed = nullptr, *packedOff = nullptr,
         *scanTmp = nullptr;
    const Req reqs[] = {{&tang, size_t(numPoints) * 16}, {&normal, size_t(numPoints) * 16}, {&flags, (size_t(numPoints) + 1) * 4},
                        {&rank, (size_t(numPoints) + 1) * 4}, {&lineOf, size_t(numPoints) * 4}, {&packed, (size_t(num_lines) + 1) * 8},
                        {&packedOff, (size_t(num_lines) + 1) * 8}, {&scanTmp, std::max<size_t>(std::max(scanBytes, scanBytes64), 16)}};
    size_t total = 0;
    for (const Req& r : reqs) total += (r.bytes + 255) & ~size_t(255);
    if ((rc = lv_buf_reserve(ctx, ctx->buildArena, total))) return rc;
    {
        size_t o = 0;
        for (const Req& r : reqs) { *r.p = (char*)ctx->buildArena.ptr + o; o += (r.bytes + 255) & ~size_t(255); }
    }
    if ((rc = lv_buf_reserve(ctx, ctx->trajLineValid, size_t(num_lines ? num_lines : 1u) * 4))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->trajLineRef, size_t(num_lines ? num_lines : 1u) * sizeof(LvLineRef)))) return rc;
    if (!ctx->pinned) LV_HIP(ctx, hipHostMalloc((void**)&ctx->pinned, 64, hipHostMallocDefault));
    LV_HIP(ctx, hipEventRecord(ctx->ev[4], st));
    LV_HIP(ctx, hipMemsetAsync(ctx->trajLineRef.ptr, 0, size_t(num_lines ? num_lines : 1u) * sizeof(LvLineRef), st));
    k_lp_tangents<<<nblk(uint64_t(numPoints) + 1), LV_BLOCK, 0, st>>>((const float*)ctx->trajPos.ptr, off, num_lines, numPoints, (float4*)tang,
                                                                      (uint32_t*)flags, (uint32_t*)lineOf);
    {
        size_t tb = scanBytes;
        LV_HIP(ctx, rocprim::exclusive_scan(scanTmp, tb, (uint32_t*)flags, (uint32_t*)rank, 0u, size_t(numPoints) + 1,
                                            rocprim::plus<uint32_t>(), st));
    }
    k_lp_line_counts<<<nblk(uint64_t(num_lines) + 1), LV_BLOCK, 0, st>>>(off, (const uint32_t*)rank, num_lines, (uint32_t*)ctx->trajLineValid.ptr,
                                                                         (unsigned long long*)packed);
    {
        size_t tb = scanBytes64;
        LV_HIP(ctx, rocprim::exclusive_scan(scanTmp, tb, (unsigned long long*)packed, (unsigned long long*)packedOff, 0ull,
                                            size_t(num_lines) + 1, rocprim::plus<unsigned long long>(), st));
    }
    LV_HIP(ctx, hipMemcpyAsync((void*)ctx->pinned, (const unsigned long long*)packedOff + num_lines, 8, hipMemcpyDeviceToHost, st));
    if (num_lines)
        k_lp_normals<<<num_lines, LV_WAVE, 0, st>>>((const float4*)tang, off, (const uint32_t*)ctx->trajLineValid.ptr, (float4*)normal);
    LV_HIP(ctx, hipStreamSynchronize(st));   // the totals size the outputs (host arrays were borrowed for the call only anyway)
    const uint64_t totals = *(volatile unsigned long long*)ctx->pinned;
    const uint32_t numRecords = uint32_t(totals), numSegs = uint32_t(totals >> 32);
    if ((rc = lv_buf_reserve(ctx, ctx->points, size_t(numRecords) * sizeof(lv_line_point)))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->segIdx, size_t(numSegs) * 8))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->trajRecLine, size_t(numRecords ? numRecords : 1u) * 4))) return rc;
    if (numPoints)
        k_lp_records<<<nblk(numPoints), LV_BLOCK, 0, st>>>(
                (const float*)ctx->trajPos.ptr, attribute ? (const float*)ctx->trajAttr.ptr : nullptr, off, (const uint32_t*)lineOf,
                (const uint32_t*)rank, (const uint32_t*)ctx->trajLineValid.ptr, (const unsigned long long*)packedOff, (const float4*)tang,
                (const float4*)normal, numPoints, (float4*)ctx->points.ptr, (uint32_t*)ctx->segIdx.ptr, (uint32_t*)ctx->trajRecLine.ptr,
                (LvLineRef*)ctx->trajLineRef.ptr);
    LV_HIP(ctx, hipEventRecord(ctx->ev[6], st));
    LV_HIP(ctx, hipGetLastError());
    LV_HIP(ctx, hipStreamSynchronize(st));
    ctx->evLinePointsValid = true;
    ctx->numPoints = numRecords;
    ctx->numSegs = numSegs;
    ctx->accelValid = false;
    ctx->trajSet = true;
    ctx->triMeshSet = false;     // tessellated on demand (lv_ensure_tube_mesh)
    ctx->triMeshFromTraj = false;
    ctx->triAccelValid = false;
    ctx->aoGlobalFrameNumber = 0;   // VulkanRayTracedAmbientOcclusionPass::setLineData (.cpp:437-460), as lv_set_lines
    ctx->lastFrameViewProjValid = false;
    return lv_forward_to_ranks(ctx, [&](lv_ctx* p) { return lv_set_trajectories(p, positions, attribute, line_offsets, num_lines); });
}

// the triangle tubes of lv_set_trajectories' lines at the current line width / tube_num_subdivisions
static int lv_tessellate_tubes(lv_ctx* ctx) {
    lv_invalidate_bake(ctx);   // waits for a running asynchronous bake BEFORE the mesh it reads is overwritten
    hipStream_t st = ctx->stream;
    const uint32_t numLines = ctx->trajNumLines, numRecords = ctx->numPoints;
    LvTessParams P;
    memset(&P, 0, sizeof(P));
    const int N = std::max(int(ctx->opt.tubeNumSubdivisions), 4);
    if (N > LV_PRISM_MAX_SUBDIV) return lv_fail(ctx, LV_E_INVALID, "tube_num_subdivisions = %d: at most %d on the device tessellator", N, LV_PRISM_MAX_SUBDIV);
    const float tubeRadius = ctx->opt.lineWidth * 0.5f;
    P.radius = tubeRadius;
    P.n = uint32_t(N);
    P.nLat = uint32_t(N / 2);
    P.capVerts = P.n * (P.nLat - 1u) + 1u;
    P.capIdx = P.n * (P.nLat - 1u) * 6u + P.n * 3u;
    {   // initGlobalCircleVertexPositions: incremental rotation by tan / cos of the step angle (Tubes.cpp:34-51)
        const float theta = kTwoPi / float(N);
        const float tangentialFactor = std::tan(theta), radialFactor = std::cos(theta);
        float px = tubeRadius, py = 0.0f;
        for (int i = 0; i < N; i++) {
            P.circle[i][0] = px; P.circle[i][1] = py; P.circle[i][2] = 0.0f;
            const float tx = -py, ty = px;
            px = px + tangentialFactor * tx; py = py + tangentialFactor * ty;
            px = px * radialFactor; py = py * radialFactor;
        }
    }
    // unit-sphere points of the cap rings, shared by all caps (CappedTriangleTubesCPU.cpp:57-75,146-164)
    std::vector<float> table(size_t(P.nLat) * P.n * 8);
    for (uint32_t lat = 1; lat <= P.nLat; lat++) {
        const float phi = kHalfPi * (1.0f - float(lat) / float(P.nLat));
        for (uint32_t lon = 0; lon < P.n; lon++) {
            const float thetaA = kTwoPi * float(lon) / float(P.n);
            const float thetaB = -kTwoPi * float(lon) / float(P.n);
            float* e = table.data() + 8 * (size_t(lat - 1u) * P.n + lon);
            e[0] = std::cos(thetaA) * std::sin(phi); e[1] = std::sin(thetaA) * std::sin(phi); e[2] = std::cos(phi); e[3] = thetaA;
            e[4] = std::cos(thetaB) * std::sin(phi); e[5] = std::sin(thetaB) * std::sin(phi); e[6] = std::cos(phi); e[7] = -thetaB;
        }
    }
    int rc;
    size_t scanBytes64 = 0;
    LV_HIP(ctx, rocprim::exclusive_scan(nullptr, scanBytes64, (unsigned long long*)nullptr, (unsigned long long*)nullptr, 0ull,
                                        size_t(numLines) + 1, rocprim::plus<unsigned long long>(), st));
    const size_t lineBytes = ((size_t(numLines) + 1) * 8 + 255) & ~size_t(255), tableBytes = (table.size() * 4 + 255) & ~size_t(255);
    if ((rc = lv_buf_reserve(ctx, ctx->trajTess, 4 * lineBytes + tableBytes + std::max<size_t>(scanBytes64, 16)))) return rc;
    char* base = (char*)ctx->trajTess.ptr;
    unsigned long long *nv = (unsigned long long*)base, *ni = (unsigned long long*)(base + lineBytes),
                       *vOff = (unsigned long long*)(base + 2 * lineBytes), *iOff = (unsigned long long*)(base + 3 * lineBytes);
    float4* capTable = (float4*)(base + 4 * lineBytes);
    void* scanTmp = base + 4 * lineBytes + tableBytes;
    if (!ctx->pinned) LV_HIP(ctx, hipHostMalloc((void**)&ctx->pinned, 64, hipHostMallocDefault));
    const uint32_t* off = (const uint32_t*)ctx->trajOff.ptr;
    const uint32_t* lineValid = (const uint32_t*)ctx->trajLineValid.ptr;
    LV_HIP(ctx, hipEventRecord(ctx->ev[8], st));
    LV_HIP(ctx, hipMemcpyAsync(capTable, table.data(), table.size() * 4, hipMemcpyHostToDevice, st));
    k_tess_counts<<<nblk(uint64_t(numLines) + 1), LV_BLOCK, 0, st>>>(off, lineValid, numLines, P, nv, ni);
    {
        size_t tb = scanBytes64;
        LV_HIP(ctx, rocprim::exclusive_scan(scanTmp, tb, nv, vOff, 0ull, size_t(numLines) + 1, rocprim::plus<unsigned long long>(), st));
        tb = scanBytes64;
        LV_HIP(ctx, rocprim::exclusive_scan(scanTmp, tb, ni, iOff, 0ull, size_t(numLines) + 1, rocprim::plus<unsigned long long>(), st));
    }
    LV_HIP(ctx, hipMemcpyAsync((void*)ctx->pinned, vOff + numLines, 8, hipMemcpyDeviceToHost, st));
    LV_HIP(ctx, hipMemcpyAsync((void*)(ctx->pinned + 2), iOff + numLines, 8, hipMemcpyDeviceToHost, st));
    LV_HIP(ctx, hipStreamSynchronize(st));   // (also: the pageable table has been consumed)
    const uint64_t numVerts = *(volatile unsigned long long*)ctx->pinned, numIdx = *(volatile unsigned long long*)(ctx->pinned + 2);
    if (numIdx / 3u > 0x03FFFFFFull || numVerts > 0xFFFFFFFFull)
        return lv_fail(ctx, LV_E_CAPACITY, "the tube mesh would have %llu triangles / %llu vertices: at most 2^26-1 triangles (leaf index field "
                                           "of the AO work queue)", (unsigned long long)(numIdx / 3u), (unsigned long long)numVerts);
    if ((rc = lv_buf_reserve(ctx, ctx->triIdx, size_t(numIdx) * 4))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->triVerts, size_t(numVerts) * sizeof(lv_tube_vertex)))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->triPoints, size_t(numRecords) * sizeof(lv_line_point)))) return rc;
    const float4* rec = (const float4*)ctx->points.ptr;
    const uint32_t* recLine = (const uint32_t*)ctx->trajRecLine.ptr;
    const LvLineRef* lineRef = (const LvLineRef*)ctx->trajLineRef.ptr;
    if (numRecords) {
        k_tess_body<<<nblk(uint64_t(numRecords) * P.n), LV_BLOCK, 0, st>>>(rec, recLine, lineRef, lineValid, vOff, iOff, numRecords, P,
                                                                          (lv_tube_vertex*)ctx->triVerts.ptr, (uint32_t*)ctx->triIdx.ptr);
        k_tess_points<<<nblk(numRecords), LV_BLOCK, 0, st>>>(rec, recLine, lineRef, numRecords, (float4*)ctx->triPoints.ptr);
    }
    if (numLines) {
        const uint64_t slots = 2ull * P.capVerts + 2ull * (P.capIdx / 3u);
        k_tess_caps<<<nblk(uint64_t(numLines) * slots), LV_BLOCK, 0, st>>>((const float*)ctx->trajPos.ptr, off, rec, lineRef, lineValid, vOff, iOff,
                                                                           numLines, P, capTable, (lv_tube_vertex*)ctx->triVerts.ptr,
                                                                           (uint32_t*)ctx->triIdx.ptr);
    }
    LV_HIP(ctx, hipEventRecord(ctx->ev[9], st));
    LV_HIP(ctx, hipGetLastError());
    ctx->evTessValid = true;
    ctx->numTris = uint32_t(numIdx / 3u);
    ctx->numTriVerts = uint32_t(numVerts);
    ctx->numTriPoints = numRecords;
    ctx->triMeshSet = true;
    ctx->triMeshFromTraj = true;
    ctx->triMeshLineWidth = ctx->opt.lineWidth;
    ctx->triMeshSubdivisions = ctx->opt.tubeNumSubdivisions;
    ctx->triAccelValid = false;
    return LV_OK;
}

int lv_ensure_tube_mesh(lv_ctx* ctx) {
    if (!ctx->trajSet) return LV_OK;                      // meshes of lv_set_tube_triangle_mesh are the caller's
    if (ctx->triMeshSet && !ctx->triMeshFromTraj) return LV_OK;   // the caller replaced the mesh after lv_set_trajectories
    if (ctx->triMeshSet && ctx->triMeshLineWidth == ctx->opt.lineWidth && ctx->triMeshSubdivisions == ctx->opt.tubeNumSubdivisions)
        return LV_OK;
    if (ctx->opt.useRibbons || ctx->opt.helicityBands)
        return lv_fail(ctx, LV_E_STATE, "lv_set_trajectories tessellates plain flow lines; band data / rotating helicity bands need "
                                        "lv_set_lines + lv_set_tube_triangle_mesh (host/LineData.cpp)");
    return lv_tessellate_tubes(ctx);
}

int lv_get_lines(lv_ctx* ctx, lv_line_point* out_points, uint32_t max_points, uint32_t* out_segment_point_indices, uint32_t max_segments,
                 uint32_t* out_num_points, uint32_t* out_num_segments) {
    if (!ctx) return LV_E_INVALID;
    (void)hipSetDevice(ctx->device);
    if (out_num_points) *out_num_points = ctx->numPoints;
    if (out_num_segments) *out_num_segments = ctx->numSegs;
    if (out_points) {
        if (max_points < ctx->numPoints) return lv_fail(ctx, LV_E_CAPACITY, "%u points, room for %u", ctx->numPoints, max_points);
        if (ctx->numPoints)
            LV_HIP(ctx, hipMemcpyAsync(out_points, ctx->points.ptr, size_t(ctx->numPoints) * sizeof(lv_line_point), hipMemcpyDeviceToHost, ctx->stream));
    }
    if (out_segment_point_indices) {
        if (max_segments < ctx->numSegs) return lv_fail(ctx, LV_E_CAPACITY, "%u segments, room for %u", ctx->numSegs, max_segments);
        if (ctx->numSegs)
            LV_HIP(ctx, hipMemcpyAsync(out_segment_point_indices, ctx->segIdx.ptr, size_t(ctx->numSegs) * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    LV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LV_OK;
}

int lv_get_tube_triangle_mesh(lv_ctx* ctx, uint32_t* out_triangle_indices, uint32_t max_triangles, lv_tube_vertex* out_vertices,
                              uint32_t max_vertices, lv_line_point* out_line_points, uint32_t max_line_points, uint32_t* out_num_triangles,
                              uint32_t* out_num_vertices, uint32_t* out_num_line_points) {
    if (!ctx) return LV_E_INVALID;
    (void)hipSetDevice(ctx->device);
    int rc;
    if ((rc = lv_ensure_tube_mesh(ctx))) return rc;
    if (!ctx->triMeshSet) return lv_fail(ctx, LV_E_STATE, "no tube mesh: neither lv_set_trajectories nor lv_set_tube_triangle_mesh has been called");
    if (out_num_triangles) *out_num_triangles = ctx->numTris;
    if (out_num_vertices) *out_num_vertices = ctx->numTriVerts;
    if (out_num_line_points) *out_num_line_points = ctx->numTriPoints;
    if (out_triangle_indices) {
        if (max_triangles < ctx->numTris) return lv_fail(ctx, LV_E_CAPACITY, "%u triangles, room for %u", ctx->numTris, max_triangles);
        if (ctx->numTris)
            LV_HIP(ctx, hipMemcpyAsync(out_triangle_indices, ctx->triIdx.ptr, size_t(ctx->numTris) * 12, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (out_vertices) {
        if (max_vertices < ctx->numTriVerts) return lv_fail(ctx, LV_E_CAPACITY, "%u vertices, room for %u", ctx->numTriVerts, max_vertices);
        if (ctx->numTriVerts)
            LV_HIP(ctx, hipMemcpyAsync(out_vertices, ctx->triVerts.ptr, size_t(ctx->numTriVerts) * sizeof(lv_tube_vertex), hipMemcpyDeviceToHost, ctx->stream));
    }
    if (out_line_points) {
        if (max_line_points < ctx->numTriPoints) return lv_fail(ctx, LV_E_CAPACITY, "%u line points, room for %u", ctx->numTriPoints, max_line_points);
        if (ctx->numTriPoints)
            LV_HIP(ctx, hipMemcpyAsync(out_line_points, ctx->triPoints.ptr, size_t(ctx->numTriPoints) * sizeof(lv_line_point), hipMemcpyDeviceToHost, ctx->stream));
    }
    LV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LV_OK;
}
