/*
 * lv_oracle_tri.h -- internal: triangle-tube scene, the build's ray-triangle test and closest-hit traversal, shared by
 * the CPU ORACLE translation units.  TEST INFRASTRUCTURE, NOT PRODUCT (see lv_oracle.h).
 */
#pragma once
#include "lv_oracle_common.h"

struct lvo_tri_node { float bmin[3], bmax[3]; int32_t left, right; };

namespace {

typedef lvo_tri_node TNode;

// ---------------------------------------------------------------- triangle scene + BVH
struct TriHit { float t, u, v; uint32_t tri; };

inline void triBox(V3 a, V3 b, V3 c, float pad, float mn[3], float mx[3]) {
    mn[0] = fminf(fminf(a.x, b.x), c.x) - pad; mx[0] = fmaxf(fmaxf(a.x, b.x), c.x) + pad;
    mn[1] = fminf(fminf(a.y, b.y), c.y) - pad; mx[1] = fmaxf(fmaxf(a.y, b.y), c.y) + pad;
    mn[2] = fminf(fminf(a.z, b.z), c.z) - pad; mx[2] = fmaxf(fmaxf(a.z, b.z), c.z) + pad;
}

// the build's ray-triangle test (header comment)
inline bool rayTriangle(V3 o, V3 d, V3 inv, V3 v0, V3 v1, V3 v2, float pad, float& tOut, float& uOut, float& vOut) {
    V3 e1 = v1 - v0, e2 = v2 - v0;
    V3 p = cross(d, e2);
    float det = dot(e1, p);
    if (det == 0.0f) return false;
    float r = 1.0f / det;
    V3 tv = o - v0;
    float u = dot(tv, p) * r;
    if (!(u >= 0.0f && u <= 1.0f)) return false;
    V3 q = cross(tv, e1);
    float v = dot(d, q) * r;
    if (!(v >= 0.0f && u + v <= 1.0f)) return false;
    float t = dot(e2, q) * r;
    float mn[3], mx[3];
    triBox(v0, v1, v2, pad, mn, mx);
    float tx0 = (mn[0] - o.x) * inv.x, tx1 = (mx[0] - o.x) * inv.x;
    float ty0 = (mn[1] - o.y) * inv.y, ty1 = (mx[1] - o.y) * inv.y;
    float tz0 = (mn[2] - o.z) * inv.z, tz1 = (mx[2] - o.z) * inv.z;
    float tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fminf(tz0, tz1));
    float tf = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1));
    if (!(t >= tn && t <= tf)) return false;
    tOut = t; uOut = u; vOut = v;
    return true;
}


} // namespace

struct lvo_tri_scene {
    std::vector<uint32_t> idx;
    std::vector<lvo_tube_vertex> verts;
    std::vector<lvo_line_point> pts;
    uint32_t nTri = 0;
    float pad = 0.0f;
    std::vector<lvo_tri_node> nodes;
    std::vector<float> leafBoxes;
    int32_t root = -1;
    bool hasBvh = false;
    uint32_t bvhDepth = 0;
};

namespace {

inline void triVerts(const lvo_tri_scene& sc, uint32_t tri, V3& a, V3& b, V3& c) {
    a = ld3(sc.verts[sc.idx[3 * size_t(tri)]].vertexPosition);
    b = ld3(sc.verts[sc.idx[3 * size_t(tri) + 1]].vertexPosition);
    c = ld3(sc.verts[sc.idx[3 * size_t(tri) + 2]].vertexPosition);
}

inline bool triChildBox(const lvo_tri_scene& sc, int32_t c, V3 o, V3 inv, float tMin, float tMax, float& tNear) {
    if (c < 0) { const float* b = &sc.leafBoxes[6 * size_t(~c)]; return rayBox(b, b + 3, o, inv, tMin, tMax, tNear); }
    return rayBox(sc.nodes[c].bmin, sc.nodes[c].bmax, o, inv, tMin, tMax, tNear);
}

inline bool closestTri(const lvo_tri_scene& sc, bool useBvh, V3 o, V3 d, float tMin, float tMax, TriHit& out,
                       Counters& cnt) {
    cnt.rays++;
    bool found = false;
    TriHit best{tMax, 0.0f, 0.0f, 0xFFFFFFFFu};
    const V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    auto test = [&](uint32_t tri) {
        cnt.prims++;
        V3 a, b, c; triVerts(sc, tri, a, b, c);
        float t, u, v;
        if (rayTriangle(o, d, inv, a, b, c, sc.pad, t, u, v) && t >= tMin && t <= tMax &&
            (!found || t < best.t || (t == best.t && tri < best.tri))) {
            found = true; best = TriHit{t, u, v, tri};
        }
    };
    if (!useBvh || !sc.hasBvh) {
        for (uint32_t tri = 0; tri < sc.nTri; tri++) test(tri);
    } else if (sc.root < 0) {
        test(uint32_t(~sc.root));
    } else {
        int32_t stack[192];
        int sp = 0;
        stack[sp++] = sc.root;
        while (sp > 0) {
            int32_t n = stack[--sp];
            if (n < 0) { test(uint32_t(~n)); continue; }
            const TNode& nd = sc.nodes[n];
            cnt.nodes++;
            float tl, tr;
            const float limit = found ? best.t : tMax;
            bool hl = triChildBox(sc, nd.left, o, inv, tMin, limit, tl);
            bool hr = triChildBox(sc, nd.right, o, inv, tMin, limit, tr);
            if (hl && hr) {
                if (tr < tl) { stack[sp++] = nd.left; stack[sp++] = nd.right; }
                else { stack[sp++] = nd.right; stack[sp++] = nd.left; }
            } else if (hl) stack[sp++] = nd.left;
            else if (hr) stack[sp++] = nd.right;
        }
    }
    out = best;
    return found;
}

// BarycentricInterpolation.glsl:38-40
inline V3 interpolateVec3(V3 a, V3 b, V3 c, V3 bc) { return (a * bc.x + b * bc.y) + c * bc.z; }

} // namespace
