/*
 * lv_oracle_tri.cpp -- CPU ORACLE, triangle-tube part: the tessellation the reference feeds to its RTAO pass (a14)
 * and RTAO traced against that mesh exactly as VulkanRayTracedAmbientOcclusion.glsl does (a13, reference geometry).
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT (see lv_oracle.h).  PARITY UNPINNED (no reference-held vectors for this path).
 * All paths are relative to /root/reference.
 *
 * Owned by the build because the Vulkan driver defines it: the ray-triangle test.  Definition used on both sides
 * (oracle and HIP), float32, fixed evaluation order:
 *   Moeller-Trumbore without culling (both faces, gl_RayFlagsOpaqueEXT): e1 = v1-v0, e2 = v2-v0, p = d x e2,
 *   det = e1.p (det == 0 -> miss), u = ((o-v0).p)/det in [0,1], q = (o-v0) x e1, v = (d.q)/det >= 0, u+v <= 1,
 *   t = (e2.q)/det; barycentrics (1-u-v, u, v) as rayQueryGetIntersectionBarycentricsEXT;
 *   AND t must lie inside the ray interval of the triangle's own padded AABB (pad = r*1e-3 + 1e-6, the same box the
 *   BVH is built from).  The last rule is what an acceleration structure does implicitly (a primitive is only ever
 *   tested when its box is hit); stating it as part of the test makes ANY conservative BVH agree with brute force
 *   bit for bit even when t of a grazing hit carries float32 noise larger than the pad.
 *   Closest hit: tMin <= t <= tMax, ties -> lowest triangle index.
 */
#include "lv_oracle_tri.h"

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

const float kPi = 3.14159265358979323846f;
const float kTwoPi = 6.28318530717958647692f;
const float kHalfPi = 1.57079632679489661923f;

// ---------------------------------------------------------------- a14: tessellation
// Tubes.cpp:34-51 initGlobalCircleVertexPositions: incremental rotation by tan/cos of the step angle
void circleOffsets(int n, float tubeRadius, std::vector<V3>& out) {
    out.clear();
    const float theta = kTwoPi / float(n);
    const float tangentialFactor = tanf(theta);
    const float radialFactor = cosf(theta);
    V3 position = v3(tubeRadius, 0.0f, 0.0f);
    for (int i = 0; i < n; i++) {
        out.push_back(position);
        V3 tangent = v3(-position.y, position.x, 0.0f);
        position = position + tangentialFactor * tangent;
        position = position * radialFactor;
    }
}

inline V3 frameCombine(V3 pt, V3 a, V3 b, V3 c) { // pt.x*a + pt.y*b + pt.z*c, summed left to right
    return v3((pt.x * a.x + pt.y * b.x) + pt.z * c.x, (pt.x * a.y + pt.y * b.y) + pt.z * c.y,
              (pt.x * a.z + pt.y * b.z) + pt.z * c.z);
}

struct Mesh {
    std::vector<uint32_t> idx;
    std::vector<lvo_tube_vertex> verts;
    std::vector<lvo_line_point> pts;
};

inline lvo_tube_vertex mkVertex(V3 p, uint32_t linePointIndex, V3 n, float phi) {
    lvo_tube_vertex v;
    v.vertexPosition[0] = p.x; v.vertexPosition[1] = p.y; v.vertexPosition[2] = p.z;
    v.vertexLinePointIndex = linePointIndex;
    v.vertexNormal[0] = n.x; v.vertexNormal[1] = n.y; v.vertexNormal[2] = n.z;
    v.phi = phi;
    return v;
}

// CappedTriangleTubesCPU.cpp:33-121 (start) and :123-211 (stop): hemisphere rings between pole and the tube's end circle
void hemisphere(bool start, V3 center, V3 tangent, V3 normal, uint32_t indexOffset, uint32_t vertexOffsetCap,
                uint32_t triOffsetCap, uint32_t linePointIndex, float tubeRadius, int nLon, int nLat, Mesh& m) {
    V3 binormal = cross(normal, tangent);
    V3 sT = tubeRadius * tangent, sN = tubeRadius * normal, sB = tubeRadius * binormal;
    uint32_t vo = vertexOffsetCap;
    auto ringVertex = [&](int lat, int lon) {
        float phi = kHalfPi * (1.0f - float(lat) / float(nLat));
        float theta = (start ? kTwoPi : -kTwoPi) * float(lon) / float(nLon);
        V3 pt = v3(cosf(theta) * sinf(phi), sinf(theta) * sinf(phi), cosf(phi));
        V3 off = frameCombine(pt, sN, sB, sT);
        V3 pos = v3(off.x + center.x, off.y + center.y, off.z + center.z);
        m.verts[vo++] = mkVertex(pos, linePointIndex | 0x80000000u, normalize(off), start ? theta : -theta);
    };
    if (start) {
        // pole first (lat = nLat), then rings towards the equator; the equator ring is the tube's first circle
        for (int lat = nLat; lat >= 1; lat--)
            for (int lon = 0; lon < nLon; lon++) { ringVertex(lat, lon); if (lat == nLat) break; }
        uint32_t ti = triOffsetCap, base = vertexOffsetCap + 1;
        for (int lat = 0; lat < nLat; lat++)
            for (int lon = 0; lon < nLon; lon++) {
                uint32_t l0 = uint32_t(lon % nLon), l1 = uint32_t((lon + 1) % nLon);
                if (lat > 0) {
                    uint32_t r0 = uint32_t(lat - 1) * nLon, r1 = uint32_t(lat) * nLon;
                    m.idx[ti++] = base + l0 + r0; m.idx[ti++] = base + l1 + r0; m.idx[ti++] = base + l0 + r1;
                    m.idx[ti++] = base + l1 + r0; m.idx[ti++] = base + l1 + r1; m.idx[ti++] = base + l0 + r1;
                } else {
                    m.idx[ti++] = vertexOffsetCap; m.idx[ti++] = base + l1; m.idx[ti++] = base + l0;
                }
            }
    } else {
        // rings from the tube's last circle towards the pole, pole last
        uint32_t ringBase = indexOffset + (vertexOffsetCap - indexOffset - uint32_t(nLon));
        for (int lat = 1; lat <= nLat; lat++)
            for (int lon = 0; lon < nLon; lon++) { ringVertex(lat, lon); if (lat == nLat) break; }
        uint32_t ti = triOffsetCap;
        for (int lat = 0; lat < nLat; lat++)
            for (int lon = 0; lon < nLon; lon++) {
                uint32_t l0 = uint32_t(lon % nLon), l1 = uint32_t((lon + 1) % nLon);
                uint32_t r0 = uint32_t(lat) * nLon, r1 = uint32_t(lat + 1) * nLon;
                if (lat < nLat - 1) {
                    m.idx[ti++] = ringBase + l0 + r0; m.idx[ti++] = ringBase + l1 + r0; m.idx[ti++] = ringBase + l0 + r1;
                    m.idx[ti++] = ringBase + l1 + r0; m.idx[ti++] = ringBase + l1 + r1; m.idx[ti++] = ringBase + l0 + r1;
                } else {
                    m.idx[ti++] = ringBase + l0 + r0; m.idx[ti++] = ringBase + l1 + r0; m.idx[ti++] = ringBase + 0 + r1;
                }
            }
    }
}

// createCappedTriangleTubesRenderDataCPU (open tubes), CappedTriangleTubesCPU.cpp:214-383, followed by the line-point
// table of LineDataFlow::getLinePassTubeTriangleMeshRenderDataPayload, LineDataFlow.cpp:1996-2020
void tessellate(const float* positions, const float* attributes, const uint32_t* lineOffsets, uint32_t nLines,
                float lineWidth, int numCircleSubdivisions, Mesh& m) {
    const float tubeRadius = lineWidth * 0.5f;
    numCircleSubdivisions = std::max(numCircleSubdivisions, 4);
    const int N = numCircleSubdivisions;
    std::vector<V3> circle;
    circleOffsets(N, tubeRadius, circle);
    const int nLon = N;
    const int nLat = N / 2; // int(std::ceil(numCircleSubdivisions / 2)): the division is integral (:230)
    const uint32_t numCapVertices = uint32_t(nLon * (nLat - 1) + 1);
    const uint32_t numCapIndices = uint32_t(nLon * (nLat - 1) * 6 + nLon * 3);

    std::vector<V3> lineTangents, lineNormals;
    std::vector<uint32_t> refLine, refPoint; // LinePointReference {trajectoryIndex, linePointIndex}
    for (uint32_t lineId = 0; lineId < nLines; lineId++) {
        const float* C = positions + 3 * size_t(lineOffsets[lineId]);
        const size_t n = lineOffsets[lineId + 1] - lineOffsets[lineId];
        const uint32_t lineIndexOffset = uint32_t(lineTangents.size());
        if (n < 2) continue;
        const uint32_t indexOffsetCapStart = uint32_t(m.verts.size());
        const uint32_t triOffsetCapStart = uint32_t(m.idx.size());
        m.verts.resize(m.verts.size() + numCapVertices, lvo_tube_vertex{});
        m.idx.resize(m.idx.size() + numCapIndices, 0u);
        const uint32_t indexOffset = uint32_t(m.verts.size());

        V3 lastLineNormal = v3(1.0f, 0.0f, 0.0f);
        int firstIdx = int(n) - 2, lastIdx = 1, numValid = 0;
        for (size_t i = 0; i < n; i++) {
            V3 tangent;
            if (i == 0) tangent = ld3(C + 3 * (i + 1)) - ld3(C + 3 * i);
            else if (i == n - 1) tangent = ld3(C + 3 * i) - ld3(C + 3 * (i - 1));
            else tangent = ld3(C + 3 * ((i + 1) % n)) - ld3(C + 3 * ((i + n - 1) % n));
            if (length(tangent) < 0.0001f) continue;
            firstIdx = std::min(int(i), firstIdx);
            lastIdx = std::max(int(i), lastIdx);
            tangent = normalize(tangent);
            // insertOrientedCirclePoints, Tubes.cpp:53-85
            V3 center = ld3(C + 3 * i);
            V3 helperAxis = lastLineNormal;
            if (length(cross(helperAxis, tangent)) < 0.01f) {
                helperAxis = v3(0.0f, 1.0f, 0.0f);
                if (length(cross(helperAxis, tangent)) < 0.01f) helperAxis = v3(0.0f, 0.0f, 1.0f);
            }
            V3 normal = normalize(helperAxis - dot(helperAxis, tangent) * tangent);
            lastLineNormal = normal;
            V3 binormal = cross(tangent, normal);
            const uint32_t linePointIndex = uint32_t(refLine.size());
            for (int k = 0; k < N; k++) {
                V3 off = frameCombine(circle[k], normal, binormal, tangent);
                V3 pos = v3(off.x + center.x, off.y + center.y, off.z + center.z);
                m.verts.push_back(mkVertex(pos, linePointIndex, normalize(pos - center), float(k) / float(N) * kTwoPi));
            }
            lineTangents.push_back(tangent);
            lineNormals.push_back(lastLineNormal);
            refLine.push_back(lineId);
            refPoint.push_back(uint32_t(i));
            numValid++;
        }
        if (numValid == 1) {
            // one point left: its circle and the reserved cap vertices are dropped, but the reference keeps the
            // reserved (zero) cap indices (:307-316) -> degenerate triangles, restated literally
            m.verts.resize(indexOffsetCapStart);
            lineTangents.pop_back(); lineNormals.pop_back(); refLine.pop_back(); refPoint.pop_back();
        }
        if (numValid <= 1) continue;

        for (int i = 0; i < numValid - 1; i++)
            for (int j = 0; j < N; j++) {
                uint32_t a = indexOffset + uint32_t(i * N + j), b = indexOffset + uint32_t(i * N + (j + 1) % N);
                uint32_t c = indexOffset + uint32_t(((i + 1) % numValid) * N + (j + 1) % N);
                uint32_t d = indexOffset + uint32_t(((i + 1) % numValid) * N + j);
                m.idx.push_back(a); m.idx.push_back(b); m.idx.push_back(c);
                m.idx.push_back(a); m.idx.push_back(c); m.idx.push_back(d);
            }
        const uint32_t indexOffsetCapEnd = uint32_t(m.verts.size());
        const uint32_t triOffsetCapEnd = uint32_t(m.idx.size());
        m.verts.resize(m.verts.size() + numCapVertices, lvo_tube_vertex{});
        m.idx.resize(m.idx.size() + numCapIndices, 0u);

        V3 center0 = ld3(C + 3 * firstIdx);
        V3 tangent0 = normalize(ld3(C + 3 * firstIdx) - ld3(C + 3 * (firstIdx + 1)));
        V3 normal0 = lineNormals[lineIndexOffset];
        V3 center1 = ld3(C + 3 * lastIdx);
        V3 tangent1 = normalize(ld3(C + 3 * lastIdx) - ld3(C + 3 * (lastIdx - 1)));
        V3 normal1 = lineNormals[lineIndexOffset + uint32_t(numValid) - 1];
        hemisphere(true, center0, tangent0, normal0, indexOffset, indexOffsetCapStart, triOffsetCapStart,
                   lineIndexOffset, tubeRadius, nLon, nLat, m);
        hemisphere(false, center1, tangent1, normal1, indexOffset, indexOffsetCapEnd, triOffsetCapEnd,
                   uint32_t(lineTangents.size() - 1), tubeRadius, nLon, nLat, m);
    }

    // LineDataFlow.cpp:1996-2020
    m.pts.resize(refLine.size());
    uint32_t lineStartIndex = 0, lastTrajectoryIndex = 0;
    float rotation = 0.0f; // useRotatingHelicityBands: NOT reset between trajectories here (LineDataFlow.cpp:1994)
    float maxHelicity = 1.0f;
    const float* helicities = lvo_get_helicity_source(&maxHelicity);
    for (size_t i = 0; i < refLine.size(); i++) {
        lvo_line_point lp;
        memset(&lp, 0, sizeof(lp));
        size_t src = size_t(lineOffsets[refLine[i]]) + refPoint[i];
        if (helicities) { // :2014-2028
            lp.lineRotation = rotation;
            const float helicity = helicities[src];
            float lineSegmentLength = 0.0f;
            if (i + 1 < refLine.size() && refLine[i] == refLine[i + 1]) {
                const size_t nxt = size_t(lineOffsets[refLine[i + 1]]) + refPoint[i + 1];
                lineSegmentLength = length(ld3(positions + 3 * nxt) - ld3(positions + 3 * src));
            }
            rotation += helicity / maxHelicity * 3.1415926535897932f * lineSegmentLength / 0.005f;
        }
        for (int k = 0; k < 3; k++) lp.linePosition[k] = positions[3 * src + k];
        lp.lineAttribute = attributes[src];
        lp.lineTangent[0] = lineTangents[i].x; lp.lineTangent[1] = lineTangents[i].y; lp.lineTangent[2] = lineTangents[i].z;
        lp.lineNormal[0] = lineNormals[i].x; lp.lineNormal[1] = lineNormals[i].y; lp.lineNormal[2] = lineNormals[i].z;
        if (lastTrajectoryIndex != refLine[i]) { lastTrajectoryIndex = refLine[i]; lineStartIndex = uint32_t(i); }
        lp.lineStartIndex = lineStartIndex;
        m.pts[i] = lp;
    }
}

} // namespace

namespace {

int32_t buildTriRange(lvo_tri_scene& sc, const std::vector<uint64_t>& keys, const std::vector<uint32_t>& order,
                      uint32_t lo, uint32_t hi, uint32_t depth, uint32_t& maxDepth) {
    if (depth > maxDepth) maxDepth = depth;
    if (hi - lo == 1) return ~int32_t(order[lo]);
    uint64_t first = keys[lo], last = keys[hi - 1];
    uint32_t split;
    if (first == last) {
        split = (lo + hi) / 2;
    } else {
        int common = __builtin_clzll(first ^ last);
        uint32_t a = lo, b = hi - 1;
        while (b - a > 1) {
            uint32_t mid = (a + b) / 2;
            uint64_t x = first ^ keys[mid];
            int pre = x == 0 ? 64 : __builtin_clzll(x);
            if (pre > common) a = mid; else b = mid;
        }
        split = a + 1;
    }
    int32_t idx = int32_t(sc.nodes.size());
    sc.nodes.push_back(TNode{});
    int32_t l = buildTriRange(sc, keys, order, lo, split, depth + 1, maxDepth);
    int32_t r = buildTriRange(sc, keys, order, split, hi, depth + 1, maxDepth);
    TNode nd;
    nd.left = l; nd.right = r;
    for (int k = 0; k < 3; k++) { nd.bmin[k] = 3.0e38f; nd.bmax[k] = -3.0e38f; }
    for (int32_t c : {l, r}) {
        const float* mn = c < 0 ? &sc.leafBoxes[6 * size_t(~c)] : sc.nodes[c].bmin;
        const float* mx = c < 0 ? &sc.leafBoxes[6 * size_t(~c) + 3] : sc.nodes[c].bmax;
        for (int k = 0; k < 3; k++) { nd.bmin[k] = fminf(nd.bmin[k], mn[k]); nd.bmax[k] = fmaxf(nd.bmax[k], mx[k]); }
    }
    sc.nodes[idx] = nd;
    return idx;
}

// ---------------------------------------------------------------- elliptic tubes of band data
// glm::inverse(mat3) (func_matrix.inl, cofactor form) followed by transpose: the normal matrix of the cap frame
struct Mat3 { float m[3][3]; }; // m[column][row]
inline Mat3 inverseTranspose(const Mat3& a) {
    const float (*m)[3] = a.m;
    const float oneOverDeterminant = 1.0f / (+ m[0][0] * (m[1][1] * m[2][2] - m[2][1] * m[1][2])
                                              - m[1][0] * (m[0][1] * m[2][2] - m[2][1] * m[0][2])
                                              + m[2][0] * (m[0][1] * m[1][2] - m[1][1] * m[0][2]));
    Mat3 inv;
    inv.m[0][0] = +(m[1][1] * m[2][2] - m[2][1] * m[1][2]) * oneOverDeterminant;
    inv.m[1][0] = -(m[1][0] * m[2][2] - m[2][0] * m[1][2]) * oneOverDeterminant;
    inv.m[2][0] = +(m[1][0] * m[2][1] - m[2][0] * m[1][1]) * oneOverDeterminant;
    inv.m[0][1] = -(m[0][1] * m[2][2] - m[2][1] * m[0][2]) * oneOverDeterminant;
    inv.m[1][1] = +(m[0][0] * m[2][2] - m[2][0] * m[0][2]) * oneOverDeterminant;
    inv.m[2][1] = -(m[0][0] * m[2][1] - m[2][0] * m[0][1]) * oneOverDeterminant;
    inv.m[0][2] = +(m[0][1] * m[1][2] - m[1][1] * m[0][2]) * oneOverDeterminant;
    inv.m[1][2] = -(m[0][0] * m[1][2] - m[1][0] * m[0][2]) * oneOverDeterminant;
    inv.m[2][2] = +(m[0][0] * m[1][1] - m[1][0] * m[0][1]) * oneOverDeterminant;
    Mat3 t;
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) t.m[c][r] = inv.m[r][c];
    return t;
}
inline V3 mulMat3(const Mat3& a, V3 v) { // column 0 * v.x + column 1 * v.y + column 2 * v.z
    return v3((a.m[0][0] * v.x + a.m[1][0] * v.y) + a.m[2][0] * v.z, (a.m[0][1] * v.x + a.m[1][1] * v.y) + a.m[2][1] * v.z,
              (a.m[0][2] * v.x + a.m[1][2] * v.y) + a.m[2][2] * v.z);
}

// addEllipticHemisphereToMeshStart / ...Stop, CappedTriangleTubesCPU.cpp:387-568: as the circular caps with the frame scaled by
// (normal radius, binormal radius, min of both), normals through the inverse-transposed frame, the ZENITH angle stored in phi
void ellipticHemisphere(bool start, V3 center, V3 tangent, V3 normal, uint32_t indexOffset, uint32_t vertexOffsetCap,
                        uint32_t triOffsetCap, uint32_t linePointIndex, float normalRadius, float binormalRadius, int nLon,
                        int nLat, Mesh& m) {
    const V3 binormal = cross(normal, tangent);
    const V3 sT = std::min(normalRadius, binormalRadius) * tangent, sN = normalRadius * normal, sB = binormalRadius * binormal;
    Mat3 frame;
    frame.m[0][0] = sN.x; frame.m[0][1] = sN.y; frame.m[0][2] = sN.z;
    frame.m[1][0] = sB.x; frame.m[1][1] = sB.y; frame.m[1][2] = sB.z;
    frame.m[2][0] = sT.x; frame.m[2][1] = sT.y; frame.m[2][2] = sT.z;
    const Mat3 normalFrame = inverseTranspose(frame);
    uint32_t vo = vertexOffsetCap;
    auto ringVertex = [&](int lat, int lon) {
        const float phi = kHalfPi * (1.0f - float(lat) / float(nLat));
        const float theta = (start ? kTwoPi : -kTwoPi) * float(lon) / float(nLon);
        const V3 pt = v3(cosf(theta) * sinf(phi), sinf(theta) * sinf(phi), cosf(phi));
        const V3 off = frameCombine(pt, sN, sB, sT);
        const V3 pos = v3(off.x + center.x, off.y + center.y, off.z + center.z);
        m.verts[vo++] = mkVertex(pos, linePointIndex | 0x80000000u, normalize(mulMat3(normalFrame, pt)), phi);
    };
    if (start) {
        for (int lat = nLat; lat >= 1; lat--)
            for (int lon = 0; lon < nLon; lon++) { ringVertex(lat, lon); if (lat == nLat) break; }
        uint32_t ti = triOffsetCap, base = vertexOffsetCap + 1;
        for (int lat = 0; lat < nLat; lat++)
            for (int lon = 0; lon < nLon; lon++) {
                uint32_t l0 = uint32_t(lon % nLon), l1 = uint32_t((lon + 1) % nLon);
                if (lat > 0) {
                    uint32_t r0 = uint32_t(lat - 1) * nLon, r1 = uint32_t(lat) * nLon;
                    m.idx[ti++] = base + l0 + r0; m.idx[ti++] = base + l1 + r0; m.idx[ti++] = base + l0 + r1;
                    m.idx[ti++] = base + l1 + r0; m.idx[ti++] = base + l1 + r1; m.idx[ti++] = base + l0 + r1;
                } else {
                    m.idx[ti++] = vertexOffsetCap; m.idx[ti++] = base + l1; m.idx[ti++] = base + l0;
                }
            }
    } else {
        uint32_t ringBase = indexOffset + (vertexOffsetCap - indexOffset - uint32_t(nLon));
        for (int lat = 1; lat <= nLat; lat++)
            for (int lon = 0; lon < nLon; lon++) { ringVertex(lat, lon); if (lat == nLat) break; }
        uint32_t ti = triOffsetCap;
        for (int lat = 0; lat < nLat; lat++)
            for (int lon = 0; lon < nLon; lon++) {
                uint32_t l0 = uint32_t(lon % nLon), l1 = uint32_t((lon + 1) % nLon);
                uint32_t r0 = uint32_t(lat) * nLon, r1 = uint32_t(lat + 1) * nLon;
                if (lat < nLat - 1) {
                    m.idx[ti++] = ringBase + l0 + r0; m.idx[ti++] = ringBase + l1 + r0; m.idx[ti++] = ringBase + l0 + r1;
                    m.idx[ti++] = ringBase + l1 + r0; m.idx[ti++] = ringBase + l1 + r1; m.idx[ti++] = ringBase + l0 + r1;
                } else {
                    m.idx[ti++] = ringBase + l0 + r0; m.idx[ti++] = ringBase + l1 + r0; m.idx[ti++] = ringBase + 0 + r1;
                }
            }
    }
}

// createCappedTriangleEllipticTubesRenderDataCPU (open tubes), CappedTriangleTubesCPU.cpp:570-745, with
// initGlobalEllipseVertexPositions / insertOrientedEllipsePoints (Tubes.cpp:121-170) and the line-point table of
// getLinePassTubeTriangleMeshRenderDataPayload (LineDataFlow.cpp:1949-2020): what the reference's triangle-mesh consumers
// (RTAO, "Triangle Mesh" geometry mode) get for a band data set.
void tessellateElliptic(const float* positions, const float* attributes, const uint32_t* lineOffsets, uint32_t nLines,
                        const float* ribbonDirections, float normalRadius, float binormalRadius, int numEllipseSubdivisions,
                        Mesh& m) {
    const int N = std::max(numEllipseSubdivisions, 4);
    std::vector<V3> ellipsePos, ellipseNrm;
    for (int i = 0; i < N; i++) {
        const float t = float(i) / float(N) * kTwoPi;
        const float cosAngle = cosf(t), sinAngle = sinf(t);
        ellipsePos.push_back(v3(normalRadius * cosAngle, binormalRadius * sinAngle, 0.0f));
        ellipseNrm.push_back(normalize(v3(binormalRadius * cosAngle, normalRadius * sinAngle, 0.0f)));
    }
    const int nLon = N, nLat = N / 2;
    const uint32_t numCapVertices = uint32_t(nLon * (nLat - 1) + 1);
    const uint32_t numCapIndices = uint32_t(nLon * (nLat - 1) * 6 + nLon * 3);
    std::vector<V3> lineTangents, lineNormals;
    std::vector<uint32_t> refLine, refPoint;
    for (uint32_t lineId = 0; lineId < nLines; lineId++) {
        const float* C = positions + 3 * size_t(lineOffsets[lineId]);
        const float* R = ribbonDirections + 3 * size_t(lineOffsets[lineId]);
        const size_t n = lineOffsets[lineId + 1] - lineOffsets[lineId];
        const uint32_t lineIndexOffset = uint32_t(lineTangents.size());
        if (n < 2) continue;
        const uint32_t indexOffsetCapStart = uint32_t(m.verts.size());
        const uint32_t triOffsetCapStart = uint32_t(m.idx.size());
        m.verts.resize(m.verts.size() + numCapVertices, lvo_tube_vertex{});
        m.idx.resize(m.idx.size() + numCapIndices, 0u);
        const uint32_t indexOffset = uint32_t(m.verts.size());
        int firstIdx = int(n) - 2, lastIdx = 1, numValid = 0;
        for (size_t i = 0; i < n; i++) {
            V3 tangent;
            if (i == 0) tangent = ld3(C + 3 * (i + 1)) - ld3(C + 3 * i);
            else if (i == n - 1) tangent = ld3(C + 3 * i) - ld3(C + 3 * (i - 1));
            else tangent = ld3(C + 3 * ((i + 1) % n)) - ld3(C + 3 * ((i + n - 1) % n));
            if (length(tangent) < 0.0001f) continue;
            firstIdx = std::min(int(i), firstIdx);
            lastIdx = std::max(int(i), lastIdx);
            tangent = normalize(tangent);
            const V3 normal = cross(ld3(R + 3 * i), tangent);
            // insertOrientedEllipsePoints
            const V3 center = ld3(C + 3 * i);
            const V3 binormal = cross(tangent, normal);
            const uint32_t linePointIndex = uint32_t(refLine.size());
            for (int k = 0; k < N; k++) {
                const V3 off = frameCombine(ellipsePos[size_t(k)], normal, binormal, tangent);
                const V3 pos = v3(off.x + center.x, off.y + center.y, off.z + center.z);
                m.verts.push_back(mkVertex(pos, linePointIndex, frameCombine(ellipseNrm[size_t(k)], normal, binormal, tangent),
                                           float(k) / float(N) * kTwoPi));
            }
            lineTangents.push_back(tangent);
            lineNormals.push_back(normal);
            refLine.push_back(lineId);
            refPoint.push_back(uint32_t(i));
            numValid++;
        }
        if (numValid == 1) {
            m.verts.resize(indexOffsetCapStart);
            lineTangents.pop_back(); lineNormals.pop_back(); refLine.pop_back(); refPoint.pop_back();
        }
        if (numValid <= 1) continue;
        for (int i = 0; i < numValid - 1; i++)
            for (int j = 0; j < N; j++) {
                uint32_t a = indexOffset + uint32_t(i * N + j), b = indexOffset + uint32_t(i * N + (j + 1) % N);
                uint32_t c = indexOffset + uint32_t(((i + 1) % numValid) * N + (j + 1) % N);
                uint32_t d = indexOffset + uint32_t(((i + 1) % numValid) * N + j);
                m.idx.push_back(a); m.idx.push_back(b); m.idx.push_back(c);
                m.idx.push_back(a); m.idx.push_back(c); m.idx.push_back(d);
            }
        const uint32_t indexOffsetCapEnd = uint32_t(m.verts.size());
        const uint32_t triOffsetCapEnd = uint32_t(m.idx.size());
        m.verts.resize(m.verts.size() + numCapVertices, lvo_tube_vertex{});
        m.idx.resize(m.idx.size() + numCapIndices, 0u);
        const V3 center0 = ld3(C + 3 * firstIdx);
        const V3 tangent0 = normalize(ld3(C + 3 * firstIdx) - ld3(C + 3 * (firstIdx + 1)));
        const V3 normal0 = lineNormals[lineIndexOffset];
        const V3 center1 = ld3(C + 3 * lastIdx);
        const V3 tangent1 = normalize(ld3(C + 3 * lastIdx) - ld3(C + 3 * (lastIdx - 1)));
        const V3 normal1 = lineNormals[lineIndexOffset + uint32_t(numValid) - 1];
        ellipticHemisphere(true, center0, tangent0, normal0, indexOffset, indexOffsetCapStart, triOffsetCapStart, lineIndexOffset,
                           normalRadius, binormalRadius, nLon, nLat, m);
        ellipticHemisphere(false, center1, tangent1, normal1, indexOffset, indexOffsetCapEnd, triOffsetCapEnd,
                           uint32_t(lineTangents.size() - 1), normalRadius, binormalRadius, nLon, nLat, m);
    }
    m.pts.resize(refLine.size());
    uint32_t lineStartIndex = 0, lastTrajectoryIndex = 0;
    for (size_t i = 0; i < refLine.size(); i++) {
        lvo_line_point lp;
        memset(&lp, 0, sizeof(lp));
        size_t src = size_t(lineOffsets[refLine[i]]) + refPoint[i];
        for (int k = 0; k < 3; k++) lp.linePosition[k] = positions[3 * src + k];
        lp.lineAttribute = attributes[src];
        lp.lineTangent[0] = lineTangents[i].x; lp.lineTangent[1] = lineTangents[i].y; lp.lineTangent[2] = lineTangents[i].z;
        lp.lineNormal[0] = lineNormals[i].x; lp.lineNormal[1] = lineNormals[i].y; lp.lineNormal[2] = lineNormals[i].z;
        if (lastTrajectoryIndex != refLine[i]) { lastTrajectoryIndex = refLine[i]; lineStartIndex = uint32_t(i); }
        lp.lineStartIndex = lineStartIndex;
        m.pts[i] = lp;
    }
}

} // namespace

extern "C" {

// band data: binormalRadius = bandWidth / 2, normalRadius = binormalRadius * minBandThickness (LineDataFlow.cpp:1959-1960)
void lvo_build_tube_triangle_render_data_ribbons(
        const float* positions, const float* attributes, const uint32_t* lineOffsets, uint32_t nLines,
        const float* ribbonDirections, float bandWidth, float minBandThickness, uint32_t tubeNumSubdivisions,
        uint32_t* outIndices, uint64_t* outNumIndices, lvo_tube_vertex* outVerts, uint64_t* outNumVerts,
        lvo_line_point* outPoints, uint64_t* outNumPoints) {
    Mesh m;
    const float binormalRadius = bandWidth * 0.5f;
    const float normalRadius = binormalRadius * minBandThickness;
    tessellateElliptic(positions, attributes, lineOffsets, nLines, ribbonDirections, normalRadius, binormalRadius,
                       int(tubeNumSubdivisions), m);
    if (outIndices) memcpy(outIndices, m.idx.data(), m.idx.size() * sizeof(uint32_t));
    if (outVerts) memcpy(outVerts, m.verts.data(), m.verts.size() * sizeof(lvo_tube_vertex));
    if (outPoints) memcpy(outPoints, m.pts.data(), m.pts.size() * sizeof(lvo_line_point));
    *outNumIndices = m.idx.size();
    *outNumVerts = m.verts.size();
    *outNumPoints = m.pts.size();
}

void lvo_build_tube_triangle_render_data(
        const float* positions, const float* attributes, const uint32_t* lineOffsets, uint32_t nLines, float lineWidth,
        uint32_t tubeNumSubdivisions, uint32_t* outIndices, uint64_t* outNumIndices, lvo_tube_vertex* outVerts,
        uint64_t* outNumVerts, lvo_line_point* outPoints, uint64_t* outNumPoints) {
    Mesh m;
    tessellate(positions, attributes, lineOffsets, nLines, lineWidth, int(tubeNumSubdivisions), m);
    if (outIndices) memcpy(outIndices, m.idx.data(), m.idx.size() * sizeof(uint32_t));
    if (outVerts) memcpy(outVerts, m.verts.data(), m.verts.size() * sizeof(lvo_tube_vertex));
    if (outPoints) memcpy(outPoints, m.pts.data(), m.pts.size() * sizeof(lvo_line_point));
    *outNumIndices = m.idx.size();
    *outNumVerts = m.verts.size();
    *outNumPoints = m.pts.size();
}

lvo_tri_scene* lvo_tri_scene_create(const uint32_t* indices, uint32_t nTri, const lvo_tube_vertex* verts, uint32_t nVerts,
                                    const lvo_line_point* pts, uint32_t nPts, float lineWidth) {
    lvo_tri_scene* sc = new lvo_tri_scene();
    sc->idx.assign(indices, indices + 3 * size_t(nTri));
    sc->verts.assign(verts, verts + nVerts);
    sc->pts.assign(pts, pts + nPts);
    sc->nTri = nTri;
    float r = lineWidth * 0.5f;
    sc->pad = r * 1e-3f + 1e-6f;
    return sc;
}
void lvo_tri_scene_destroy(lvo_tri_scene* sc) { delete sc; }

void lvo_tri_scene_build_bvh(lvo_tri_scene* sc) {
    sc->nodes.clear();
    sc->root = -1;
    sc->hasBvh = false;
    uint32_t n = sc->nTri;
    if (n == 0) return;
    sc->leafBoxes.resize(6 * size_t(n));
    float smn[3] = {3e38f, 3e38f, 3e38f}, smx[3] = {-3e38f, -3e38f, -3e38f};
    for (uint32_t t = 0; t < n; t++) {
        V3 a, b, c; triVerts(*sc, t, a, b, c);
        float* bx = &sc->leafBoxes[6 * size_t(t)];
        triBox(a, b, c, sc->pad, bx, bx + 3);
        for (int k = 0; k < 3; k++) { smn[k] = fminf(smn[k], bx[k]); smx[k] = fmaxf(smx[k], bx[3 + k]); }
    }
    std::vector<uint64_t> keys(n);
    std::vector<uint32_t> order(n);
    for (uint32_t t = 0; t < n; t++) {
        const float* bx = &sc->leafBoxes[6 * size_t(t)];
        uint64_t q[3];
        for (int k = 0; k < 3; k++) {
            double c = 0.5 * (double(bx[k]) + double(bx[3 + k]));
            double u = (c - smn[k]) / std::max(1e-30, double(smx[k]) - double(smn[k]));
            u = std::min(std::max(u, 0.0), 1.0);
            q[k] = uint64_t(std::min(2097151.0, u * 2097152.0));
        }
        keys[t] = (expandBits21(q[0]) << 2) | (expandBits21(q[1]) << 1) | expandBits21(q[2]);
        order[t] = t;
    }
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b] || (keys[a] == keys[b] && a < b); });
    std::vector<uint64_t> sk(n);
    for (uint32_t i = 0; i < n; i++) sk[i] = keys[order[i]];
    sc->nodes.reserve(n);
    uint32_t maxDepth = 0;
    sc->root = buildTriRange(*sc, sk, order, 0, n, 0, maxDepth);
    sc->bvhDepth = maxDepth;
    sc->hasBvh = true;
}

int lvo_intersect_triangle(const float o[3], const float d[3], const float v0[3], const float v1[3], const float v2[3],
                           float pad, float* outT, float* outU, float* outV) {
    V3 dd = ld3(d);
    V3 inv = v3(1.0f / dd.x, 1.0f / dd.y, 1.0f / dd.z);
    float t = 0.0f, u = 0.0f, v = 0.0f;
    bool h = rayTriangle(ld3(o), dd, inv, ld3(v0), ld3(v1), ld3(v2), pad, t, u, v);
    *outT = t; *outU = u; *outV = v;
    return h ? 1 : 0;
}

void lvo_trace_rays_tri(const lvo_tri_scene* sc, int useBvh, const float* origins, const float* dirs, float tMin,
                        float tMax, uint32_t n, float* outT, uint32_t* outTri, float* outUV) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < int64_t(n); i++) {
        Counters c;
        TriHit h;
        bool f = closestTri(*sc, useBvh != 0, ld3(origins + 3 * i), ld3(dirs + 3 * i), tMin, tMax, h, c);
        outT[i] = f ? h.t : tMax;
        outTri[i] = f ? h.tri : 0xFFFFFFFFu;
        if (outUV) { outUV[2 * i] = f ? h.u : 0.0f; outUV[2 * i + 1] = f ? h.v : 0.0f; }
    }
}

// VulkanRayTracedAmbientOcclusion.glsl:178-319 against the triangle tubes (the reference's own geometry)
void lvo_render_ao_tri(const lvo_tri_scene* sc, const lvo_params* Pp, int useBvh, uint32_t x0, uint32_t y0, uint32_t w,
                       uint32_t h, float* aoOut, lvo_stats* stats) {
    const lvo_params& P = *Pp;
    Frame F = makeFrame(P);
    uint64_t rays = 0, nodes = 0, prims = 0;
    for (uint32_t iter = 0; iter < P.aoIterations; iter++) {
        // SVGF: DISABLE_ACCUMULATION (no running means) + useGlobalFrameNumber (seeds from a counter that onHasMoved does not
        // reset), VulkanRayTracedAmbientOcclusion.cpp:415-421,576-581
        const uint32_t frameNumber = g_lvoAoFeatures.svgf ? 0u : iter;
        const uint32_t globalFrameNumber = g_lvoAoFeatures.svgf ? g_lvoAoFeatures.globalFrameNumber + iter : frameNumber;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : rays, nodes, prims)
        for (int64_t tileIdx = 0; tileIdx < lvoTileCount(w, h); tileIdx++) { // 16x16-pixel tiles
            Counters cnt;
            for (uint32_t tilePix = 0; tilePix < 256u; tilePix++) {
                uint32_t xx, yy;
                if (!lvoTilePixel(w, h, tileIdx, tilePix, xx, yy)) continue;
                uint32_t x = x0 + xx, y = y0 + uint32_t(yy);
                uint32_t pix = x + y * P.width;
                uint32_t seed = tea(pix, globalFrameNumber);
                float xix = 0.5f, xiy = 0.5f;
                if (P.aoJitterPrimary) { xix = rnd(seed); xiy = rnd(seed); }
                V3 o, d;
                primaryRay(P, F, x, y, xix, xiy, o, d);
                TriHit hit;
                float aoFactor = 1.0f;
                bool hasHitSurface = false;
                V3 featNormal = v3(0, 0, 0), featPosition = v3(0, 0, 0); // surfaceNormal / vertexPositionWorld of a miss, glsl:211-212
                if (closestTri(*sc, useBvh != 0, o, d, 0.0001f, 1000.0f, hit, cnt)) {
                    // glsl:219-263
                    const uint32_t* ti = &sc->idx[3 * size_t(hit.tri)];
                    const lvo_tube_vertex& vd0 = sc->verts[ti[0]];
                    const lvo_tube_vertex& vd1 = sc->verts[ti[1]];
                    const lvo_tube_vertex& vd2 = sc->verts[ti[2]];
                    V3 bc = v3((1.0f - hit.u) - hit.v, hit.u, hit.v);
                    const lvo_line_point& lp0 = sc->pts[vd0.vertexLinePointIndex & 0x7FFFFFFFu];
                    const lvo_line_point& lp1 = sc->pts[vd1.vertexLinePointIndex & 0x7FFFFFFFu];
                    const lvo_line_point& lp2 = sc->pts[vd2.vertexLinePointIndex & 0x7FFFFFFFu];
                    V3 vertexPositionWorld = interpolateVec3(ld3(vd0.vertexPosition), ld3(vd1.vertexPosition), ld3(vd2.vertexPosition), bc);
                    V3 surfaceNormal = normalize(interpolateVec3(ld3(vd0.vertexNormal), ld3(vd1.vertexNormal), ld3(vd2.vertexNormal), bc));
                    V3 linePosition = interpolateVec3(ld3(lp0.linePosition), ld3(lp1.linePosition), ld3(lp2.linePosition), bc);
                    V3 surfaceTangent = normalize(interpolateVec3(ld3(lp0.lineTangent), ld3(lp1.lineTangent), ld3(lp2.lineTangent), bc));
                    V3 surfaceBitangent = cross(surfaceNormal, surfaceTangent);
                    float offsetFactor = length(linePosition - vertexPositionWorld) / F.subdivisionCorrectionFactor; // glsl:280
                    hasHitSurface = true; featNormal = surfaceNormal; featPosition = vertexPositionWorld;
                    aoFactor = 0.0f;
                    for (uint32_t s = 0; s < P.aoSamplesPerFrame; s++) {
                        uint32_t sseed = tea(pix, globalFrameNumber * P.aoSamplesPerFrame + s);
                        float xi0 = rnd(sseed), xi1 = rnd(sseed);
                        float sn, cs;
                        sincos2pi(xi1, sn, cs);
                        float r = sqrtf(1.0f - xi0 * xi0);
                        V3 smp = v3(cs * r, sn * r, xi0);
                        V3 dirU = v3((surfaceTangent.x * smp.x + surfaceBitangent.x * smp.y) + surfaceNormal.x * smp.z,
                                     (surfaceTangent.y * smp.x + surfaceBitangent.y * smp.y) + surfaceNormal.y * smp.z,
                                     (surfaceTangent.z * smp.x + surfaceBitangent.z * smp.y) + surfaceNormal.z * smp.z);
                        V3 rd = normalize(dirU);
                        V3 ro = vertexPositionWorld + rd * offsetFactor;
                        TriHit ah;
                        float occ = 1.0f;
                        if (closestTri(*sc, useBvh != 0, ro, rd, 0.0f, P.aoRadius, ah, cnt))
                            occ = P.aoUseDistance ? ah.t / P.aoRadius : 0.0f;
                        aoFactor += occ;
                    }
                    aoFactor /= float(P.aoSamplesPerFrame);
                }
                size_t idx = size_t(y) * P.width + x;
                if (frameNumber != 0) aoFactor = mixf(aoOut[idx], aoFactor, 1.0f / float(frameNumber + 1));
                aoOut[idx] = aoFactor;
                writeAoFeatures(P, F, x, y, idx, frameNumber, hasHitSurface, featNormal, featPosition);
            }
            rays += cnt.rays; nodes += cnt.nodes; prims += cnt.prims;
        }
    }
    if (stats) { stats->raysTraced += rays; stats->nodesVisited += nodes; stats->primsTested += prims; stats->bvhDepth = sc->bvhDepth; }
}

} // extern "C"
