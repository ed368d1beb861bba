// Tubes.cpp -- triangle tessellation of line sets into capped N-gon tubes (host side, OpenMP over lines).
//
// Output is byte-identical to what the reference feeds to its triangle-mesh consumers (RTAO, triangle ray tracing):
//   createCappedTriangleTubesRenderDataCPU      src/Renderers/Tubes/CappedTriangleTubesCPU.cpp:214-383
//   addHemisphereToMeshStart / ...Stop          src/Renderers/Tubes/CappedTriangleTubesCPU.cpp:33-211
//   initGlobalCircleVertexPositions             src/Renderers/Tubes/Tubes.cpp:34-51
//   insertOrientedCirclePoints                  src/Renderers/Tubes/Tubes.cpp:53-85
//   createCappedTriangleEllipticTubesRenderDataCPU, addEllipticHemisphereToMeshStart / ...Stop (band data)
//                                               src/Renderers/Tubes/CappedTriangleTubesCPU.cpp:387-745
//   initGlobalEllipseVertexPositions, insertOrientedEllipsePoints     src/Renderers/Tubes/Tubes.cpp:121-170
// The reference appends line after line to growing vectors on one thread ("seconds at 1 M segments").  Here the work is
// split so that lines are independent: pass 1 finds every line's valid points and their frames (the normal is carried
// from point to point, so a line is the unit of parallelism), a serial prefix sum places each line's vertex / index /
// line-point ranges, pass 2 writes all ranges in parallel straight into the final arrays.
#include "Tubes.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace lv {

namespace {

const float kTwoPi = 6.28318530717958647692f;
const float kHalfPi = 1.57079632679489661923f;

struct LineFrames {
    std::vector<uint32_t> pointIndex;   // index of every valid point inside its trajectory
    std::vector<vec3> tangent, normal;
    int firstIdx = 0, lastIdx = 0;
    uint32_t pointsIn = 0;              // points of the trajectory (0: skipped entirely)
};

inline vec3 combine(vec3 pt, vec3 a, vec3 b, vec3 c) {
    return vec3((pt.x * a.x + pt.y * b.x) + pt.z * c.x, (pt.x * a.y + pt.y * b.y) + pt.z * c.y,
                (pt.x * a.z + pt.y * b.z) + pt.z * c.z);
}

inline TubeTriangleVertexData vertex(vec3 p, uint32_t linePoint, vec3 n, float phi) {
    TubeTriangleVertexData v;
    v.vertexPosition[0] = p.x; v.vertexPosition[1] = p.y; v.vertexPosition[2] = p.z;
    v.vertexLinePointIndex = linePoint;
    v.vertexNormal[0] = n.x; v.vertexNormal[1] = n.y; v.vertexNormal[2] = n.z;
    v.phi = phi;
    return v;
}

// unit-sphere points of the cap rings, shared by all caps: ring[lat-1][lon] for lat = 1..nLat (lat = nLat is the pole)
struct CapTable {
    std::vector<vec3> startPt, stopPt;
    std::vector<float> startPhi, stopPhi; // the angle stored in the vertex
    int nLon, nLat;
    CapTable(int nLon_, int nLat_) : nLon(nLon_), nLat(nLat_) {
        startPt.resize(size_t(nLat) * nLon); stopPt = startPt;
        startPhi.resize(size_t(nLat) * nLon); stopPhi = startPhi;
        for (int lat = 1; lat <= nLat; lat++) {
            float phi = kHalfPi * (1.0f - float(lat) / float(nLat));
            for (int lon = 0; lon < nLon; lon++) {
                float thetaA = kTwoPi * float(lon) / float(nLon);
                float thetaB = -kTwoPi * float(lon) / float(nLon);
                size_t k = size_t(lat - 1) * nLon + lon;
                startPt[k] = vec3(std::cos(thetaA) * std::sin(phi), std::sin(thetaA) * std::sin(phi), std::cos(phi));
                stopPt[k] = vec3(std::cos(thetaB) * std::sin(phi), std::sin(thetaB) * std::sin(phi), std::cos(phi));
                startPhi[k] = thetaA;
                stopPhi[k] = -thetaB;
            }
        }
    }
};

} // namespace

namespace {
// glm::inverse(mat3) in cofactor form, then transposed: the normal matrix of an elliptic cap's frame (columns a, b, c)
struct NormalFrame {
    float m[3][3]; // [column][row]
    NormalFrame(vec3 a, vec3 b, vec3 c) {
        const float f[3][3] = {{a.x, a.y, a.z}, {b.x, b.y, b.z}, {c.x, c.y, c.z}};
        const float oneOverDeterminant = 1.0f / (+ f[0][0] * (f[1][1] * f[2][2] - f[2][1] * f[1][2])
                                                  - f[1][0] * (f[0][1] * f[2][2] - f[2][1] * f[0][2])
                                                  + f[2][0] * (f[0][1] * f[1][2] - f[1][1] * f[0][2]));
        float inv[3][3];
        inv[0][0] = +(f[1][1] * f[2][2] - f[2][1] * f[1][2]) * oneOverDeterminant;
        inv[1][0] = -(f[1][0] * f[2][2] - f[2][0] * f[1][2]) * oneOverDeterminant;
        inv[2][0] = +(f[1][0] * f[2][1] - f[2][0] * f[1][1]) * oneOverDeterminant;
        inv[0][1] = -(f[0][1] * f[2][2] - f[2][1] * f[0][2]) * oneOverDeterminant;
        inv[1][1] = +(f[0][0] * f[2][2] - f[2][0] * f[0][2]) * oneOverDeterminant;
        inv[2][1] = -(f[0][0] * f[2][1] - f[2][0] * f[0][1]) * oneOverDeterminant;
        inv[0][2] = +(f[0][1] * f[1][2] - f[1][1] * f[0][2]) * oneOverDeterminant;
        inv[1][2] = -(f[0][0] * f[1][2] - f[1][0] * f[0][2]) * oneOverDeterminant;
        inv[2][2] = +(f[0][0] * f[1][1] - f[1][0] * f[0][1]) * oneOverDeterminant;
        for (int col = 0; col < 3; col++)
            for (int row = 0; row < 3; row++) m[col][row] = inv[row][col];
    }
    vec3 mul(vec3 v) const {
        return vec3((m[0][0] * v.x + m[1][0] * v.y) + m[2][0] * v.z, (m[0][1] * v.x + m[1][1] * v.y) + m[2][1] * v.z,
                    (m[0][2] * v.x + m[1][2] * v.y) + m[2][2] * v.z);
    }
};

// rightVectors == nullptr: circular tubes of radius `normalRadius`; otherwise the elliptic tubes of a band data set (semi-axes
// normalRadius along cross(rightVector, tangent), binormalRadius across)
void createTubes(
        const std::vector<std::vector<vec3>>& lineCentersList, const std::vector<std::vector<vec3>>* rightVectors,
        float normalRadius, float binormalRadius, int numCircleSubdivisions,
        std::vector<uint32_t>& triangleIndices, std::vector<TubeTriangleVertexData>& vertexDataList,
        std::vector<LinePointReference>& linePointReferenceList, std::vector<vec3>& lineTangents,
        std::vector<vec3>& lineNormals) {
    const bool elliptic = rightVectors != nullptr;
    const float tubeRadius = normalRadius;
    const int N = std::max(numCircleSubdivisions, 4);
    const int nLon = N, nLat = N / 2;
    const uint32_t capVerts = uint32_t(nLon * (nLat - 1) + 1);
    const uint32_t capIdx = uint32_t(nLon * (nLat - 1) * 6 + nLon * 3);
    const size_t numLines = lineCentersList.size();

    // circle offsets by incremental rotation (tan / cos of the step angle)
    std::vector<vec3> circle;
    {
        const float theta = kTwoPi / float(N);
        const float tangentialFactor = std::tan(theta), radialFactor = std::cos(theta);
        vec3 position(tubeRadius, 0.0f, 0.0f);
        for (int i = 0; i < N; i++) {
            circle.push_back(position);
            vec3 tangent(-position.y, position.x, 0.0f);
            position = position + tangentialFactor * tangent;
            position = position * radialFactor;
        }
    }
    std::vector<vec3> ellipseNormals;
    if (elliptic) { // initGlobalEllipseVertexPositions
        circle.clear();
        for (int i = 0; i < N; i++) {
            const float t = float(i) / float(N) * kTwoPi;
            const float cosAngle = std::cos(t), sinAngle = std::sin(t);
            circle.push_back(vec3(normalRadius * cosAngle, binormalRadius * sinAngle, 0.0f));
            ellipseNormals.push_back(normalize(vec3(binormalRadius * cosAngle, normalRadius * sinAngle, 0.0f)));
        }
    }
    const CapTable caps(nLon, nLat);

    // ---- pass 1: valid points + frames per line
    std::vector<LineFrames> frames(numLines);
#pragma omp parallel for schedule(dynamic, 16)
    for (long li = 0; li < long(numLines); li++) {
        const std::vector<vec3>& C = lineCentersList[size_t(li)];
        LineFrames& f = frames[size_t(li)];
        const size_t n = C.size();
        if (n < 2) continue;
        f.pointsIn = uint32_t(n);
        f.firstIdx = int(n) - 2;
        f.lastIdx = 1;
        vec3 lastLineNormal(1.0f, 0.0f, 0.0f);
        for (size_t i = 0; i < n; i++) {
            vec3 tangent;
            if (i == 0) tangent = C[i + 1] - C[i];
            else if (i == n - 1) tangent = C[i] - C[i - 1];
            else tangent = C[(i + 1) % n] - C[(i + n - 1) % n];
            if (length(tangent) < 0.0001f) continue;
            f.firstIdx = std::min(int(i), f.firstIdx);
            f.lastIdx = std::max(int(i), f.lastIdx);
            tangent = normalize(tangent);
            vec3 helperAxis = lastLineNormal;
            if (length(cross(helperAxis, tangent)) < 0.01f) {
                helperAxis = vec3(0.0f, 1.0f, 0.0f);
                if (length(cross(helperAxis, tangent)) < 0.01f) helperAxis = vec3(0.0f, 0.0f, 1.0f);
            }
            vec3 normal = normalize(helperAxis - dot(helperAxis, tangent) * tangent);
            if (elliptic) normal = cross((*rightVectors)[size_t(li)][i], tangent); // :640 (not normalised)
            lastLineNormal = normal;
            f.pointIndex.push_back(uint32_t(i));
            f.tangent.push_back(tangent);
            f.normal.push_back(normal);
        }
    }

    // ---- placement.  Quirks of the reference kept for byte identity: a line that ends up with one valid point keeps
    // the start cap's (zero) index range but no vertices; one with no valid point keeps both start-cap ranges.
    std::vector<size_t> vOff(numLines + 1, 0), iOff(numLines + 1, 0), pOff(numLines + 1, 0);
    for (size_t li = 0; li < numLines; li++) {
        const LineFrames& f = frames[li];
        size_t nv = 0, ni = 0, np = 0;
        if (f.pointsIn >= 2) {
            const size_t m = f.pointIndex.size();
            if (m >= 2) {
                nv = 2 * size_t(capVerts) + m * N;
                ni = 2 * size_t(capIdx) + (m - 1) * size_t(N) * 6;
                np = m;
            } else if (m == 1) {
                ni = capIdx;
            } else {
                nv = capVerts;
                ni = capIdx;
            }
        }
        vOff[li + 1] = vOff[li] + nv; iOff[li + 1] = iOff[li] + ni; pOff[li + 1] = pOff[li] + np;
    }
    TubeTriangleVertexData zeroVertex;
    memset(&zeroVertex, 0, sizeof(zeroVertex));
    vertexDataList.assign(vOff[numLines], zeroVertex);
    triangleIndices.assign(iOff[numLines], 0u);
    linePointReferenceList.resize(pOff[numLines]);
    lineTangents.resize(pOff[numLines]);
    lineNormals.resize(pOff[numLines]);

    // ---- pass 2: vertices and indices of every line
#pragma omp parallel for schedule(dynamic, 16)
    for (long lli = 0; lli < long(numLines); lli++) {
        const size_t li = size_t(lli);
        const LineFrames& f = frames[li];
        const size_t m = f.pointIndex.size();
        if (f.pointsIn < 2 || m < 2) continue;
        const std::vector<vec3>& C = lineCentersList[li];
        const uint32_t capStartV = uint32_t(vOff[li]), bodyV = capStartV + capVerts, capEndV = bodyV + uint32_t(m) * N;
        const uint32_t p0 = uint32_t(pOff[li]);
        uint32_t* idx = triangleIndices.data() + iOff[li];
        TubeTriangleVertexData* V = vertexDataList.data();

        for (size_t k = 0; k < m; k++) {
            const vec3 center = C[f.pointIndex[k]], tangent = f.tangent[k], normal = f.normal[k];
            const vec3 binormal = cross(tangent, normal);
            for (int j = 0; j < N; j++) {
                vec3 off = combine(circle[size_t(j)], normal, binormal, tangent);
                vec3 pos(off.x + center.x, off.y + center.y, off.z + center.z);
                const vec3 nrm = elliptic ? combine(ellipseNormals[size_t(j)], normal, binormal, tangent) : normalize(pos - center);
                V[bodyV + k * N + j] = vertex(pos, p0 + uint32_t(k), nrm, float(j) / float(N) * kTwoPi);
            }
            linePointReferenceList[p0 + k] = LinePointReference(uint32_t(li), f.pointIndex[k]);
            lineTangents[p0 + k] = tangent;
            lineNormals[p0 + k] = normal;
        }

        // caps: rings between the pole and the tube's first / last circle
        auto capVertices = [&](bool start, uint32_t base, vec3 center, vec3 tangent, vec3 normal, uint32_t linePoint) {
            const vec3 binormal = cross(normal, tangent);
            const vec3 sT = (elliptic ? std::min(normalRadius, binormalRadius) : tubeRadius) * tangent, sN = normalRadius * normal,
                       sB = (elliptic ? binormalRadius : tubeRadius) * binormal;
            const NormalFrame normalFrame(sN, sB, sT);
            uint32_t w = base;
            auto put = [&](int lat, int lon) {
                const size_t k = size_t(lat - 1) * nLon + lon;
                const vec3 pt = start ? caps.startPt[k] : caps.stopPt[k];
                vec3 off = combine(pt, sN, sB, sT);
                vec3 pos(off.x + center.x, off.y + center.y, off.z + center.z);
                if (elliptic) // normal through the inverse-transposed frame, the ZENITH angle in phi (:425-432,:513-520)
                    V[w++] = vertex(pos, linePoint | 0x80000000u, normalize(normalFrame.mul(pt)),
                                    kHalfPi * (1.0f - float(lat) / float(nLat)));
                else
                    V[w++] = vertex(pos, linePoint | 0x80000000u, normalize(off), start ? caps.startPhi[k] : caps.stopPhi[k]);
            };
            if (start) {
                put(nLat, 0); // pole first
                for (int lat = nLat - 1; lat >= 1; lat--)
                    for (int lon = 0; lon < nLon; lon++) put(lat, lon);
            } else {
                for (int lat = 1; lat < nLat; lat++)
                    for (int lon = 0; lon < nLon; lon++) put(lat, lon);
                put(nLat, 0); // pole last
            }
        };
        const vec3 center0 = C[size_t(f.firstIdx)];
        const vec3 tangent0 = normalize(C[size_t(f.firstIdx)] - C[size_t(f.firstIdx) + 1]);
        const vec3 center1 = C[size_t(f.lastIdx)];
        const vec3 tangent1 = normalize(C[size_t(f.lastIdx)] - C[size_t(f.lastIdx) - 1]);
        capVertices(true, capStartV, center0, tangent0, f.normal[0], p0);
        capVertices(false, capEndV, center1, tangent1, f.normal[m - 1], p0 + uint32_t(m) - 1);

        uint32_t* w = idx;
        // start cap: fan at the pole, then quads ring by ring; ring r of the cap starts at capStartV + 1 + r * nLon and
        // ring nLat - 1 is the tube's first circle (it follows the cap vertices directly)
        for (int lat = 0; lat < nLat; lat++)
            for (int lon = 0; lon < nLon; lon++) {
                const uint32_t l0 = uint32_t(lon), l1 = uint32_t((lon + 1) % nLon), ring0 = capStartV + 1;
                if (lat == 0) { *w++ = capStartV; *w++ = ring0 + l1; *w++ = ring0 + l0; continue; }
                const uint32_t a = ring0 + uint32_t(lat - 1) * nLon, b = ring0 + uint32_t(lat) * nLon;
                *w++ = a + l0; *w++ = a + l1; *w++ = b + l0;
                *w++ = a + l1; *w++ = b + l1; *w++ = b + l0;
            }
        // body: two triangles per side and segment
        for (uint32_t i = 0; i + 1 < uint32_t(m); i++)
            for (int j = 0; j < N; j++) {
                const uint32_t j1 = uint32_t((j + 1) % N);
                const uint32_t a = bodyV + i * N + uint32_t(j), b = bodyV + i * N + j1;
                const uint32_t c = bodyV + ((i + 1) % uint32_t(m)) * N + j1, d = bodyV + ((i + 1) % uint32_t(m)) * N + uint32_t(j);
                *w++ = a; *w++ = b; *w++ = c;
                *w++ = a; *w++ = c; *w++ = d;
            }
        // end cap: ring 0 is the tube's last circle, then the cap rings, fan at the pole
        const uint32_t lastCircle = capEndV - uint32_t(N);
        for (int lat = 0; lat < nLat; lat++)
            for (int lon = 0; lon < nLon; lon++) {
                const uint32_t l0 = uint32_t(lon), l1 = uint32_t((lon + 1) % nLon);
                const uint32_t a = lastCircle + uint32_t(lat) * nLon, b = lastCircle + uint32_t(lat + 1) * nLon;
                if (lat < nLat - 1) {
                    *w++ = a + l0; *w++ = a + l1; *w++ = b + l0;
                    *w++ = a + l1; *w++ = b + l1; *w++ = b + l0;
                } else {
                    *w++ = a + l0; *w++ = a + l1; *w++ = b;
                }
            }
    }
}
} // namespace

void createCappedTriangleTubesRenderData(
        const std::vector<std::vector<vec3>>& lineCentersList, float tubeRadius, int numCircleSubdivisions,
        std::vector<uint32_t>& triangleIndices, std::vector<TubeTriangleVertexData>& vertexDataList,
        std::vector<LinePointReference>& linePointReferenceList, std::vector<vec3>& lineTangents,
        std::vector<vec3>& lineNormals) {
    createTubes(lineCentersList, nullptr, tubeRadius, tubeRadius, numCircleSubdivisions, triangleIndices, vertexDataList,
                linePointReferenceList, lineTangents, lineNormals);
}

void createCappedTriangleEllipticTubesRenderData(
        const std::vector<std::vector<vec3>>& lineCentersList, const std::vector<std::vector<vec3>>& lineRightVectorsList,
        float tubeNormalRadius, float tubeBinormalRadius, int numEllipseSubdivisions,
        std::vector<uint32_t>& triangleIndices, std::vector<TubeTriangleVertexData>& vertexDataList,
        std::vector<LinePointReference>& linePointReferenceList, std::vector<vec3>& lineTangents,
        std::vector<vec3>& lineNormals) {
    createTubes(lineCentersList, &lineRightVectorsList, tubeNormalRadius, tubeBinormalRadius, numEllipseSubdivisions,
                triangleIndices, vertexDataList, linePointReferenceList, lineTangents, lineNormals);
}

} // namespace lv
