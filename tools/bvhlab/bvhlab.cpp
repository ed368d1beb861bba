// bvhlab -- offline BVH-quality lab for the AO kernel (dev tool; not part of the product, not part of the oracle).
//
// Question it answers: how many node steps per AO ray would a full-sweep SAH tree need compared with
// the Morton-order LBVH the library builds on the GPU (k_ao_rays: 25.6 steps per capsule AO ray on config 3)?  It rebuilds
// the library's tree on the CPU (63-bit Morton keys of box centroids, highest-differing-bit splits, greedy area-guided
// collapse into 4-wide nodes), builds alternatives, and walks the same sample of AO rays through each with the kernel's
// visiting rule (nearest hit child first, the others pushed as stored, closest hit shrinks the interval), counting node
// steps and leaf tests.  Rays: primary rays of the default camera on a pixel grid, AO rays from their hits exactly like
// the RTAO pass (uniform hemisphere about the capsule normal, origin offset, length aoRadius).
//
//   bvhlab scene.bin [pixelStride=8] [spp=16] [builders=lbvh,sah,...]
// scene.bin: uint32 n, float32 radius, then n x {p0.xyz, p1.xyz} float32 (tools/bvhlab/run.py writes it).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <numeric>
#include <string>
#include <vector>
#include <omp.h>

struct V3 { float x, y, z; };
static inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static inline V3 norm(V3 a) { float l = std::sqrt(dot(a, a)); return {a.x / l, a.y / l, a.z / l}; }

struct Box {
    float mn[3], mx[3];
    void reset() { for (int k = 0; k < 3; k++) { mn[k] = 3e38f; mx[k] = -3e38f; } }
    void grow(const Box& b) { for (int k = 0; k < 3; k++) { mn[k] = std::min(mn[k], b.mn[k]); mx[k] = std::max(mx[k], b.mx[k]); } }
    float halfArea() const { float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2]; return dx * dy + dy * dz + dz * dx; }
    // measure of the rays of length L (random origin and direction) that meet the box: V + A L / 4 (LAB_RAYLEN; 0 = surface area)
    float hitMeasure(float L) const { float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2]; const float h = dx * dy + dy * dz + dz * dx; return L > 0.0f ? dx * dy * dz + h * 0.5f * L : h; }
};
static float g_rayLen = 0.0f;

struct Seg { V3 p0, p1; };
static std::vector<Seg> g_segs;       // capsule mode: g_segs.size() primitives
struct Tri { V3 a, b, c; };
static std::vector<Tri> g_tris;       // triangle mode (the reference's RTAO geometry): g_tris.size() primitives
static uint32_t g_leaf = 1;            // consecutive primitives per leaf (LAB_LEAF)
static uint32_t numRaw() { return uint32_t(g_tris.empty() ? g_segs.size() : g_tris.size()); }
static uint32_t numPrims() { return (numRaw() + g_leaf - 1) / g_leaf; }
static std::vector<Box> g_boxes;
static float g_radius;

// ---------------------------------------------------------------- binary trees: child >= 0 internal, < 0 leaf ~prim
struct BNode { Box box; int32_t l, r; };
struct BTree { std::vector<BNode> nodes; int32_t root; };

static uint64_t expand21(uint64_t v) {
    v &= 0x1fffffull;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

static int32_t lbvhRange(BTree& T, const std::vector<uint64_t>& keys, const std::vector<uint32_t>& order, uint32_t lo, uint32_t hi) {
    if (hi - lo == 1) return ~int32_t(order[lo]);
    const uint64_t first = keys[lo], last = keys[hi - 1];
    uint32_t split;
    if (first == last) split = (lo + hi) / 2;
    else {
        const int cp = __builtin_clzll(first ^ last);
        uint32_t a = lo, b = hi - 1;
        while (b - a > 1) {
            const uint32_t mid = (a + b) / 2;
            const uint64_t x = first ^ keys[mid];
            const int pre = x == 0 ? 64 : __builtin_clzll(x);
            if (pre > cp) a = mid; else b = mid;
        }
        split = a + 1;
    }
    const int32_t idx = int32_t(T.nodes.size());
    T.nodes.push_back(BNode{});
    const int32_t l = lbvhRange(T, keys, order, lo, split), r = lbvhRange(T, keys, order, split, hi);
    BNode nd; nd.l = l; nd.r = r; nd.box.reset();
    nd.box.grow(l < 0 ? g_boxes[~l] : T.nodes[l].box);
    nd.box.grow(r < 0 ? g_boxes[~r] : T.nodes[r].box);
    T.nodes[idx] = nd;
    return idx;
}

static BTree buildLbvh() {
    const uint32_t n = numPrims();
    Box sb; sb.reset();
    for (auto& b : g_boxes) sb.grow(b);
    std::vector<uint64_t> keys(n);
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; i++) {
        uint64_t q[3];
        for (int k = 0; k < 3; k++) {
            float c = 0.5f * (g_boxes[i].mn[k] + g_boxes[i].mx[k]);
            float u = (c - sb.mn[k]) / std::max(sb.mx[k] - sb.mn[k], 1e-30f);
            u = std::min(std::max(u, 0.0f), 1.0f);
            q[k] = uint64_t(std::min(2097151.0f, u * 2097152.0f));
        }
        keys[i] = (expand21(q[0]) << 2) | (expand21(q[1]) << 1) | expand21(q[2]);
        order[i] = i;
    }
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });
    std::vector<uint64_t> sk(n);
    for (uint32_t i = 0; i < n; i++) sk[i] = keys[order[i]];
    BTree T;
    T.nodes.reserve(n);
    T.root = lbvhRange(T, sk, order, 0, n);
    return T;
}

static const V3* centroids() {
    static std::vector<V3> cen;
    if (cen.empty()) {
        cen.resize(numPrims());
        for (uint32_t i = 0; i < numPrims(); i++)
            cen[i] = {0.5f * (g_boxes[i].mn[0] + g_boxes[i].mx[0]), 0.5f * (g_boxes[i].mn[1] + g_boxes[i].mx[1]), 0.5f * (g_boxes[i].mn[2] + g_boxes[i].mx[2])};
    }
    return cen.data();
}

static int g_bins = 64;               // bins of the binned SAH split (LAB_BINS)
static int g_hybSweepBelow = -1;      // hybrid builder: full sweep below this many leaves, binned above (LAB_HYB_SWEEP; -1 = always sweep)
// full-sweep SAH below `sweepBelow` primitives, binned SAH above; leaves hold one primitive (like the library's)
struct SahBuilder {
    BTree T;
    std::vector<uint32_t> idx;
    const V3* cen = nullptr;            // box centroids of all primitives (shared)
    uint32_t sweepBelow;
    int32_t build(uint32_t lo, uint32_t hi) {
        if (hi - lo == 1) return ~int32_t(idx[lo]);
        const uint32_t n = hi - lo;
        Box cb; cb.reset();
        for (uint32_t i = lo; i < hi; i++) { const V3 c = cen[idx[i]]; const float cc[3] = {c.x, c.y, c.z};
            for (int k = 0; k < 3; k++) { cb.mn[k] = std::min(cb.mn[k], cc[k]); cb.mx[k] = std::max(cb.mx[k], cc[k]); } }
        float bestCost = 3e38f; int bestAxis = -1, bestBin = 0; uint32_t bestSplit = 0;
        if (n <= sweepBelow) {
            std::vector<float> rightArea(n);
            for (int a = 0; a < 3; a++) {
                std::sort(idx.begin() + lo, idx.begin() + hi, [&](uint32_t x, uint32_t y) { return (&cen[x].x)[a] < (&cen[y].x)[a]; });
                Box b; b.reset();
                for (uint32_t i = n - 1; i > 0; i--) { b.grow(g_boxes[idx[lo + i]]); rightArea[i] = b.hitMeasure(g_rayLen); }
                b.reset();
                for (uint32_t i = 1; i < n; i++) {
                    b.grow(g_boxes[idx[lo + i - 1]]);
                    const float c = b.hitMeasure(g_rayLen) * float(i) + rightArea[i] * float(n - i);
                    if (c < bestCost) { bestCost = c; bestAxis = a; bestSplit = i; }
                }
            }
            if (bestAxis != 2) std::sort(idx.begin() + lo, idx.begin() + hi, [&](uint32_t x, uint32_t y) { return (&cen[x].x)[bestAxis] < (&cen[y].x)[bestAxis]; });
        } else {
            const int NB = g_bins;
            for (int a = 0; a < 3; a++) {
                const float ext = cb.mx[a] - cb.mn[a];
                if (!(ext > 0.0f)) continue;
                Box bb[64]; uint32_t cnt[64];
                for (int k = 0; k < NB; k++) { bb[k].reset(); cnt[k] = 0; }
                for (uint32_t i = lo; i < hi; i++) {
                    int k = int(((&cen[idx[i]].x)[a] - cb.mn[a]) / ext * NB); k = std::min(std::max(k, 0), NB - 1);
                    bb[k].grow(g_boxes[idx[i]]); cnt[k]++;
                }
                float ra[64]; uint32_t rc[64]; Box b; b.reset(); uint32_t c = 0;
                for (int k = NB - 1; k > 0; k--) { b.grow(bb[k]); c += cnt[k]; ra[k] = b.hitMeasure(g_rayLen); rc[k] = c; }
                b.reset(); c = 0;
                for (int k = 1; k < NB; k++) {
                    b.grow(bb[k - 1]); c += cnt[k - 1];
                    if (c == 0 || rc[k] == 0) continue;
                    const float cost = b.hitMeasure(g_rayLen) * float(c) + ra[k] * float(rc[k]);
                    if (cost < bestCost) { bestCost = cost; bestAxis = a; bestBin = k; bestSplit = c; }
                }
            }
            if (bestAxis >= 0) {
                const int a = bestAxis;
                const float ext = cb.mx[a] - cb.mn[a];
                auto mid = std::partition(idx.begin() + lo, idx.begin() + hi, [&](uint32_t x) {
                    int k = int(((&cen[x].x)[a] - cb.mn[a]) / ext * NB); k = std::min(std::max(k, 0), NB - 1);
                    return k < bestBin;
                });
                bestSplit = uint32_t(mid - (idx.begin() + lo));
            }
        }
        if (bestAxis < 0 || bestSplit == 0 || bestSplit >= n) bestSplit = n / 2; // all centroids equal / degenerate
        const int32_t me = int32_t(T.nodes.size());
        T.nodes.push_back(BNode{});
        const int32_t l = build(lo, lo + bestSplit), r = build(lo + bestSplit, hi);
        BNode nd; nd.l = l; nd.r = r; nd.box.reset();
        nd.box.grow(l < 0 ? g_boxes[~l] : T.nodes[l].box);
        nd.box.grow(r < 0 ? g_boxes[~r] : T.nodes[r].box);
        T.nodes[me] = nd;
        return me;
    }
};

static BTree buildSah(uint32_t sweepBelow) {
    SahBuilder B;
    const uint32_t n = numPrims();
    B.idx.resize(n); std::iota(B.idx.begin(), B.idx.end(), 0u);
    B.cen = centroids();
    B.sweepBelow = sweepBelow;
    B.T.nodes.reserve(n);
    B.T.root = B.build(0, n);
    return B.T;
}

// LBVH whose subtrees of at most K leaves are rebuilt by the full-sweep SAH builder (what a wave-per-treelet GPU kernel could do)
static void collectLeaves(const BTree& T, int32_t c, std::vector<uint32_t>& out) {
    if (c < 0) { out.push_back(uint32_t(~c)); return; }
    collectLeaves(T, T.nodes[c].l, out); collectLeaves(T, T.nodes[c].r, out);
}
static uint32_t countLeaves(const BTree& T, int32_t c, std::vector<uint32_t>& cnt) {
    if (c < 0) return 1;
    return cnt[c] = countLeaves(T, T.nodes[c].l, cnt) + countLeaves(T, T.nodes[c].r, cnt);
}
static BTree buildHybrid(uint32_t K) {
    BTree L = buildLbvh();
    std::vector<uint32_t> cnt(L.nodes.size(), 0);
    countLeaves(L, L.root, cnt);
    BTree T; T.nodes.reserve(L.nodes.size());
    std::function<int32_t(int32_t)> rec = [&](int32_t c) -> int32_t {
        if (c < 0) return c;
        if (cnt[c] <= K) {
            SahBuilder B; B.sweepBelow = g_hybSweepBelow >= 0 ? uint32_t(g_hybSweepBelow) : K + 1;
            collectLeaves(L, c, B.idx);
            B.cen = centroids();
            const int32_t r = B.build(0, uint32_t(B.idx.size()));
            const int32_t base = int32_t(T.nodes.size());
            for (auto nd : B.T.nodes) { if (nd.l >= 0) nd.l += base; if (nd.r >= 0) nd.r += base; T.nodes.push_back(nd); }
            return r >= 0 ? r + base : r;
        }
        const int32_t me = int32_t(T.nodes.size());
        T.nodes.push_back(BNode{});
        const int32_t l = rec(L.nodes[c].l), r = rec(L.nodes[c].r);
        BNode nd; nd.l = l; nd.r = r; nd.box = L.nodes[c].box;
        T.nodes[me] = nd;
        return me;
    };
    T.root = rec(L.root);
    return T;
}

static double sahCost(const BTree& T) {
    const double ra = T.nodes[T.root].box.halfArea();
    double c = 0.0;
    for (auto& n : T.nodes) c += n.box.halfArea() / ra;
    return c;
}

// ---------------------------------------------------------------- 4-wide collapse (k_collapse_select's greedy rule)
struct WNode { Box cb[8]; int32_t c[8]; int n; int level; float umn[3], ustep[3]; uint8_t qmn[8][3], qmx[8][3]; }; // child >= 0: wide node, < 0: ~prim, INT32_MIN: empty (width 4 or 8)
struct WTree { std::vector<WNode> nodes; int levels; };

static WTree collapse(const BTree& T, int width = 4) {
    WTree W; W.levels = 0;
    struct Item { int32_t bin; int32_t wide; };
    std::vector<Item> frontier{{T.root, 0}}, next;
    W.nodes.push_back(WNode{});
    while (!frontier.empty()) {
        next.clear();
        for (auto it : frontier) {
            int32_t s[8]; int ns = 2;
            s[0] = T.nodes[it.bin].l; s[1] = T.nodes[it.bin].r;
            while (ns < width) {
                float best = -1.0f; int bk = -1;
                for (int k = 0; k < ns; k++) if (s[k] >= 0) { const float a = T.nodes[s[k]].box.halfArea(); if (a > best) { best = a; bk = k; } }
                if (bk < 0) break;
                const int32_t c = s[bk];
                s[bk] = T.nodes[c].l; s[ns++] = T.nodes[c].r;
            }
            WNode wn; wn.n = ns; wn.level = W.levels;
            for (int k = 0; k < 8; k++) wn.c[k] = INT32_MIN;
            for (int k = 0; k < ns; k++) {
                if (s[k] < 0) { wn.c[k] = s[k]; wn.cb[k] = g_boxes[~s[k]]; }
                else { wn.cb[k] = T.nodes[s[k]].box; wn.c[k] = int32_t(W.nodes.size()); W.nodes.push_back(WNode{}); next.push_back({s[k], wn.c[k]}); }
            }
            if (getenv("LAB_QUANT")) { // the library's node format: child planes on the 8-bit grid of the node's own box, rounded outwards
                Box U; U.reset();
                for (int k = 0; k < ns; k++) U.grow(wn.cb[k]);
                for (int a = 0; a < 3; a++) {
                    const float step = (U.mx[a] - U.mn[a]) / 255.0f;
                    if (!(step > 0.0f)) { wn.umn[a] = U.mn[a]; wn.ustep[a] = 0.0f; for (int k = 0; k < ns; k++) { wn.qmn[k][a] = 0; wn.qmx[k][a] = 0; } continue; }
                    wn.umn[a] = U.mn[a]; wn.ustep[a] = step;
                    for (int k = 0; k < ns; k++) {
                        const float qa = std::floor((wn.cb[k].mn[a] - U.mn[a]) / step), qb = std::min(255.0f, std::ceil((wn.cb[k].mx[a] - U.mn[a]) / step));
                        wn.qmn[k][a] = uint8_t(std::max(0.0f, qa)); wn.qmx[k][a] = uint8_t(qb);
                        wn.cb[k].mn[a] = U.mn[a] + qa * step;
                        wn.cb[k].mx[a] = U.mn[a] + qb * step;
                    }
                }
            }
            W.nodes[it.wide] = wn;
        }
        frontier.swap(next);
        W.levels++;
    }
    return W;
}

// ---------------------------------------------------------------- capsule test (closest-approach form; plain float)
static bool raySphere(V3 o, V3 d, V3 c, float r, float& t) {
    V3 f = o - c; float A = dot(d, d), tc = -dot(f, d) / A; V3 l = f + d * tc; float disc = r * r - dot(l, l);
    if (disc < 0) return false; float h = std::sqrt(disc / A), t0 = tc - h, t1 = tc + h;
    if (t0 >= 0) { t = t0; return true; } if (t1 >= 0) { t = t1; return true; } return false;
}
static bool rayTube(V3 o, V3 d, V3 a, V3 b, float r, float& t) {
    V3 td = norm(b - a), dp = o - a, av = d - td * dot(d, td), cv = dp - td * dot(dp, td);
    float A = dot(av, av), tc = -dot(av, cv) / A; V3 l = cv + av * tc; float disc = r * r - dot(l, l);
    if (disc < 0) return false; float h = std::sqrt(disc / A);
    for (float tt : {tc - h, tc + h}) if (tt >= 0) { V3 ip = o + d * tt; if (dot(td, ip - a) > 0 && dot(td, ip - b) < 0) { t = tt; return true; } }
    return false;
}
static bool capsule(V3 o, V3 d, const Seg& s, float r, float& t, int& kind) {
    bool has = false; float best = 1e7f, x; kind = 0;
    if (rayTube(o, d, s.p0, s.p1, r, x)) { best = x; has = true; }
    if (raySphere(o, d, s.p0, r, x) && x < best) { best = x; has = true; kind = 1; }
    if (raySphere(o, d, s.p1, r, x) && x < best) { best = x; has = true; kind = 2; }
    t = best; return has;
}

static bool rayTri(V3 o, V3 d, const Tri& T, float& t) {
    const V3 e1 = T.b - T.a, e2 = T.c - T.a, p = cross(d, e2);
    const float det = dot(e1, p);
    if (det == 0.0f) return false;
    const float r = 1.0f / det; const V3 tv = o - T.a;
    const float u = dot(tv, p) * r; if (!(u >= 0.0f && u <= 1.0f)) return false;
    const V3 q = cross(tv, e1); const float v = dot(d, q) * r; if (!(v >= 0.0f && u + v <= 1.0f)) return false;
    t = dot(e2, q) * r; return true;
}
struct Counters { uint64_t rays = 0, nodes = 0, prims = 0, hits = 0, boxes = 0, barren = 0, stalePops = 0; uint64_t perLevel[48] = {0}; };

// closest hit with the kernel's visiting rule; immediate leaf tests (the kernel batches them: best shrinks a little later there)
// g_order: 0 = the kernel's rule (nearest hit child first, the rest as stored), 1 = all hit children sorted front to back,
// 2 = as stored, 3 = sort-free sign order: children are stored sorted along the axis on which their centroids spread most
// (2 bits per node) and visited in that order, reversed when the ray runs against the axis
static int g_order = 0;
static bool trace(const WTree& W, V3 o, V3 d, float tMin, float tMax, float& tHit, uint32_t& prim, int& kindOut, Counters& C, int32_t start = 0, bool foundIn = false, bool countRay = true) {
    const float inv[3] = {1.0f / d.x, 1.0f / d.y, 1.0f / d.z}, oo[3] = {o.x, o.y, o.z};
    int32_t stack[256]; float stackT[256]; int sp = 0;
    int32_t cur = start; float best = tMax; bool found = foundIn;
    if (countRay) C.rays++;
    while (true) {
        if (cur >= 0) {
            const WNode& n = W.nodes[cur];
            C.nodes++; C.perLevel[std::min(n.level, 47)]++;
            float key[8]; int32_t ch[8]; int nh = 0;
            C.boxes += uint64_t(n.n);
            static const int f16mode = getenv("LAB_F16") ? atoi(getenv("LAB_F16")) : 0;
            if (f16mode) {
                // packed-f16 slab test: node-local time t' = t - c (c = the node's own entry time), planes q in
                // f16 exactly, A = step / d and B' = (umn - o) / d - c rounded to f16, every fma result rounded to f16; boxes widened
                // by LAB_F16 grid units on each side to stay conservative
                auto h = [](float x) -> float {   // round to the nearest binary16 value (ties to even), overflow to infinity
                    if (!(std::fabs(x) < 65520.0f)) return x > 0 ? INFINITY : (x < 0 ? -INFINITY : x);
                    if (x == 0.0f) return x;
                    int e; std::frexp(x, &e); e = std::max(e - 1, -14);
                    const float quantum = std::ldexp(1.0f, e - 10);
                    return std::nearbyint(x / quantum) * quantum;
                };
                float A[3], B[3], c = tMin;
                for (int a = 0; a < 3; a++) {
                    A[a] = n.ustep[a] * inv[a]; B[a] = (n.umn[a] - oo[a]) * inv[a];
                    const float e0 = B[a], e1 = 255.0f * A[a] + B[a];
                    c = std::max(c, std::min(e0, e1));
                }
                const float tMinL = h(tMin - c) , tMaxL = h(best - c);
                for (int k = 0; k < n.n; k++) {
                    float tn = tMinL - 0.0f, tf = tMaxL;
                    for (int a = 0; a < 3; a++) {
                        const float Ah = h(A[a]), Bh = h(B[a] - c);
                        float qn = float(n.qmn[k][a]) - float(f16mode), qx = float(n.qmx[k][a]) + float(f16mode);
                        float t0 = h(qn * Ah + Bh), t1 = h(qx * Ah + Bh);
                        if (inv[a] < 0.0f) { t0 = h((float(n.qmx[k][a]) + float(f16mode)) * Ah + Bh); t1 = h((float(n.qmn[k][a]) - float(f16mode)) * Ah + Bh); }
                        tn = std::max(tn, t0); tf = std::min(tf, t1);
                    }
                    // exact f32 test for the statistics of false negatives
                    float en = tMin, ef = best;
                    for (int a = 0; a < 3; a++) { float t0 = (n.cb[k].mn[a] - oo[a]) * inv[a], t1 = (n.cb[k].mx[a] - oo[a]) * inv[a]; if (t0 > t1) std::swap(t0, t1); en = std::max(en, t0); ef = std::min(ef, t1); }
                    if (en <= ef && !(tn <= tf)) C.barren += 1000000ull;   // a box the exact test hits was culled: NOT conservative
                    if (tn <= tf) { key[nh] = tn; ch[nh] = n.c[k]; nh++; }
                }
            } else
            for (int k = 0; k < n.n; k++) {
                float tn = tMin, tf = best;
                for (int a = 0; a < 3; a++) {
                    float t0 = (n.cb[k].mn[a] - oo[a]) * inv[a], t1 = (n.cb[k].mx[a] - oo[a]) * inv[a];
                    if (t0 > t1) std::swap(t0, t1);
                    tn = std::max(tn, t0); tf = std::min(tf, t1);
                }
                if (tn <= tf) { key[nh] = tn; ch[nh] = n.c[k]; nh++; }
            }
            if (nh == 0) { C.barren++; if (sp == 0) break; cur = stack[--sp]; if (g_order == 0 && stackT[sp] > best) C.stalePops++; continue; }
            if (g_order == 1) {
                for (int a = 1; a < nh; a++) for (int b = a; b > 0 && key[b] < key[b - 1]; b--) { std::swap(key[b], key[b - 1]); std::swap(ch[b], ch[b - 1]); }
                for (int k = nh - 1; k >= 1; k--) stack[sp++] = ch[k];
                cur = ch[0];
            } else if (g_order == 2) {
                for (int k = nh - 1; k >= 1; k--) stack[sp++] = ch[k];
                cur = ch[0];
            } else if (g_order == 3) {
                // axis of the largest centroid spread over the node's children; order by centroid along it
                float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
                for (int k = 0; k < n.n; k++) for (int a = 0; a < 3; a++) { const float c = n.cb[k].mn[a] + n.cb[k].mx[a]; lo[a] = std::min(lo[a], c); hi[a] = std::max(hi[a], c); }
                int ax = 0; for (int a = 1; a < 3; a++) if (hi[a] - lo[a] > hi[ax] - lo[ax]) ax = a;
                const float sgn = inv[ax] < 0.0f ? -1.0f : 1.0f;
                int j = 0;
                for (int k = 0; k < n.n; k++) { // rebuild the hit list with the sign-order key
                    float tn = tMin, tf = best;
                    for (int a = 0; a < 3; a++) { float t0 = (n.cb[k].mn[a] - oo[a]) * inv[a], t1 = (n.cb[k].mx[a] - oo[a]) * inv[a]; if (t0 > t1) std::swap(t0, t1); tn = std::max(tn, t0); tf = std::min(tf, t1); }
                    if (tn <= tf) { key[j] = sgn * (n.cb[k].mn[ax] + n.cb[k].mx[ax]); ch[j] = n.c[k]; j++; }
                }
                for (int a = 1; a < nh; a++) for (int b = a; b > 0 && key[b] < key[b - 1]; b--) { std::swap(key[b], key[b - 1]); std::swap(ch[b], ch[b - 1]); }
                for (int k = nh - 1; k >= 1; k--) stack[sp++] = ch[k];
                cur = ch[0];
            } else if (g_order == 4) { // sort-free octant order (optimistic stand-in for the slot ^ octant order of compressed wide BVHs):
                // hit children by the projection of their centroid on the ray's octant diagonal
                int j = 0;
                for (int k = 0; k < n.n; k++) {
                    float tn = tMin, tf = best;
                    for (int a = 0; a < 3; a++) { float t0 = (n.cb[k].mn[a] - oo[a]) * inv[a], t1 = (n.cb[k].mx[a] - oo[a]) * inv[a]; if (t0 > t1) std::swap(t0, t1); tn = std::max(tn, t0); tf = std::min(tf, t1); }
                    if (tn <= tf) { float pr = 0; for (int a = 0; a < 3; a++) pr += (inv[a] < 0.0f ? -1.0f : 1.0f) * (n.cb[k].mn[a] + n.cb[k].mx[a]); key[j] = pr; ch[j] = n.c[k]; j++; }
                }
                for (int a = 1; a < nh; a++) for (int b = a; b > 0 && key[b] < key[b - 1]; b--) { std::swap(key[b], key[b - 1]); std::swap(ch[b], ch[b - 1]); }
                for (int k = nh - 1; k >= 1; k--) stack[sp++] = ch[k];
                cur = ch[0];
            } else {
                int m = 0; for (int k = 1; k < nh; k++) if (key[k] < key[m]) m = k;
                for (int k = nh - 1; k >= 0; k--) if (k != m) { stackT[sp] = key[k]; stack[sp++] = ch[k]; }
                cur = ch[m];
            }
        } else {
            for (uint32_t p = uint32_t(~cur) * g_leaf; p < std::min(numRaw(), (uint32_t(~cur) + 1u) * g_leaf); p++) {
                C.prims++;
                float t; int kind;
                const bool hitp = g_tris.empty() ? capsule(o, d, g_segs[p], g_radius, t, kind) : (kind = 0, rayTri(o, d, g_tris[p], t));
                if (hitp && t >= tMin && t <= best) {
                    if (t < best || !found || p < prim) { best = t; prim = p; kindOut = kind; found = true; }
                }
            }
            if (sp == 0) break;
            cur = stack[--sp];
            if (g_order == 0 && stackT[sp] > best) C.stalePops++;
        }
    }
    tHit = best;
    if (found && countRay) C.hits++;
    return found;
}

static uint32_t tea(uint32_t v0, uint32_t v1) {
    uint32_t s0 = 0;
    for (int n = 0; n < 16; n++) { s0 += 0x9e3779b9u; v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu); }
    return v0;
}
static float rnd(uint32_t& s) { s = 1664525u * s + 1013904223u; return float(s & 0x00FFFFFFu) / float(0x01000000); }

struct AoRay { V3 o, d; };
struct AoGroup { V3 hit, N; float off; };   // one per hit pixel: its spp AO rays share the hit point (LAB_CUT)
struct CutEntry { int32_t c; Box b; };
// LAB_CUT = L: per-pixel entry cut -- the nodes of wide level L (and the leaves above it) whose boxes meet the sphere (hit, aoRadius + off)
// and are not entirely behind the tangent plane of the hemisphere; the rays of the pixel start from synthetic 4-wide nodes over the cut
static void entryCut(const WTree& W, const AoGroup& g, float R, int L, std::vector<CutEntry>& cut, uint64_t& steps) {
    cut.clear();
    int32_t st[512]; int sp = 0; st[sp++] = 0;
    const float c[3] = {g.hit.x, g.hit.y, g.hit.z}, nn[3] = {g.N.x, g.N.y, g.N.z};
    while (sp) {
        const WNode& n = W.nodes[st[--sp]]; steps++;
        for (int k = 0; k < n.n; k++) {
            float d2 = 0, far = 0;
            for (int a = 0; a < 3; a++) {
                const float lo = n.cb[k].mn[a] - c[a], hi = n.cb[k].mx[a] - c[a];
                const float d = lo > 0 ? lo : (hi < 0 ? -hi : 0.0f); d2 += d * d;
                far += nn[a] > 0 ? nn[a] * hi : nn[a] * lo;
            }
            if (d2 > R * R || far < 0.0f) continue;
            if (n.c[k] >= 0 && W.nodes[n.c[k]].level < L) st[sp++] = n.c[k];
            else cut.push_back({n.c[k], n.cb[k]});
        }
    }
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: bvhlab scene.bin [pixelStride] [spp] [builders]\n"); return 1; }
    const int stride = argc > 2 ? atoi(argv[2]) : 8, spp = argc > 3 ? atoi(argv[3]) : 16;
    const std::string builders = argc > 4 ? argv[4] : "lbvh,sah";
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("scene"); return 1; }
    uint32_t n; float radius;
    if (fread(&n, 4, 1, f) != 1 || fread(&radius, 4, 1, f) != 1) return 1;
    const bool triMode = (n & 0x80000000u) != 0;   // top bit of the count: n x {a, b, c} triangles instead of segments
    n &= 0x7FFFFFFFu;
    g_radius = radius;
    g_boxes.resize(n);
    const float pad = radius * 1e-3f + 1e-6f;
    if (triMode) {
        g_tris.resize(n);
        if (fread(g_tris.data(), sizeof(Tri), n, f) != n) return 1;
        for (uint32_t i = 0; i < n; i++) {
            const V3* v = &g_tris[i].a;
            g_boxes[i].reset();
            for (int j = 0; j < 3; j++) { const float c[3] = {v[j].x, v[j].y, v[j].z};
                for (int k = 0; k < 3; k++) { g_boxes[i].mn[k] = std::min(g_boxes[i].mn[k], c[k] - pad); g_boxes[i].mx[k] = std::max(g_boxes[i].mx[k], c[k] + pad); } }
        }
    } else {
        g_segs.resize(n);
        if (fread(g_segs.data(), sizeof(Seg), n, f) != n) return 1;
    }
    fclose(f);
    for (uint32_t i = 0; i < n && !triMode; i++) {
        const float a[3] = {g_segs[i].p0.x, g_segs[i].p0.y, g_segs[i].p0.z}, b[3] = {g_segs[i].p1.x, g_segs[i].p1.y, g_segs[i].p1.z};
        for (int k = 0; k < 3; k++) { g_boxes[i].mn[k] = std::min(a[k], b[k]) - radius - pad; g_boxes[i].mx[k] = std::max(a[k], b[k]) + radius + pad; }
    }
    if (getenv("LAB_RAYLEN")) g_rayLen = float(atof(getenv("LAB_RAYLEN")));
    if (getenv("LAB_BINS")) g_bins = std::min(64, std::max(2, atoi(getenv("LAB_BINS"))));
    if (getenv("LAB_HYB_SWEEP")) g_hybSweepBelow = atoi(getenv("LAB_HYB_SWEEP"));
    if (getenv("LAB_LEAF") && atoi(getenv("LAB_LEAF")) > 1) {
        g_leaf = uint32_t(atoi(getenv("LAB_LEAF")));
        std::vector<Box> merged(numPrims());
        for (uint32_t g = 0; g < numPrims(); g++) {
            merged[g].reset();
            for (uint32_t p = g * g_leaf; p < std::min(n, (g + 1) * g_leaf); p++)
                for (int k = 0; k < 3; k++) { merged[g].mn[k] = std::min(merged[g].mn[k], g_boxes[p].mn[k]); merged[g].mx[k] = std::max(merged[g].mx[k], g_boxes[p].mx[k]); }
        }
        g_boxes.swap(merged);
    }
    printf("%u primitives, %u per leaf, radius %g, pixel stride %d, %d AO samples per hit pixel\n", n, g_leaf, radius, stride, spp);
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };

    std::vector<AoRay> rays;   // generated once, with the first tree
    std::vector<AoGroup> groups;
    const int Wd = 1920, Ht = 1080;
    const float aoRadius = 0.1f, corr = std::cos(3.14159265358979f / 6.0f);
    size_t pos = 0;
    while (pos <= builders.size()) {
        size_t e = builders.find(',', pos); if (e == std::string::npos) e = builders.size();
        const std::string name = builders.substr(pos, e - pos); pos = e + 1;
        if (name.empty()) continue;
        double t0 = now();
        BTree T;
        if (name == "lbvh") T = buildLbvh();
        else if (name == "sah") T = buildSah(1u << 14);
        else if (name == "sahbin") T = buildSah(0);
        else if (name.rfind("hyb", 0) == 0) T = buildHybrid(uint32_t(atoi(name.c_str() + 3)));
        else { fprintf(stderr, "unknown builder %s\n", name.c_str()); continue; }
        const double tb = now() - t0;
        const int width = getenv("LAB_WIDTH") ? atoi(getenv("LAB_WIDTH")) : 4;
        WTree W = collapse(T, width);
        double fill = 0; for (auto& w : W.nodes) fill += w.n;
        printf("[%s] build %.1f s, binary SAH node cost %.2f, wide nodes %zu (%.2f children/node), %d wide levels\n", name.c_str(), tb,
               sahCost(T), W.nodes.size(), fill / W.nodes.size(), W.levels);
        if (rays.empty()) {
            Counters C;
            for (int y = stride / 2; y < Ht; y += stride)
                for (int x = stride / 2; x < Wd; x += stride) {
                    const uint32_t pix = uint32_t(x + y * Wd);
                    uint32_t seed = tea(pix, 0);
                    const float xi = rnd(seed), yi = rnd(seed);
                    const float ndcx = 2.0f * ((x + xi) / Wd) - 1.0f, ndcy = 2.0f * ((y + yi) / Ht) - 1.0f;
                    const V3 o = {0, 0, 0.8f}, d = norm({ndcx * 0.5f * float(Wd) / float(Ht), -ndcy * 0.5f, -1.0f});
                    float t; uint32_t p = 0; int kind = 0;
                    if (!trace(W, o, d, 1e-4f, 1000.0f, t, p, kind, C)) continue;
                    const V3 hit = o + d * t;
                    V3 N, Tn; float off;
                    if (!g_tris.empty()) {   // geometric normal towards the camera, an edge as the tangent, mean offset of a 6-gon face
                        const Tri& T = g_tris[p];
                        N = norm(cross(T.b - T.a, T.c - T.a)); if (dot(N, d) > 0.0f) N = N * -1.0f;
                        Tn = norm(T.b - T.a); off = g_radius * 0.93f / corr;
                    } else {
                        const Seg& s = g_segs[p];
                        const V3 v = s.p1 - s.p0;
                        const float ts = kind == 0 ? dot(v, hit - s.p0) / dot(v, v) : (kind == 1 ? 0.0f : 1.0f);
                        const V3 lp = s.p0 + v * ts;
                        N = norm(hit - lp); Tn = norm(v); off = std::sqrt(dot(lp - hit, lp - hit)) / corr;
                    }
                    const V3 B = cross(N, Tn);
                    groups.push_back({hit, N, off});
                    for (int sidx = 0; sidx < spp; sidx++) {
                        uint32_t sd = tea(pix, uint32_t(sidx));
                        const float x0 = rnd(sd), x1 = rnd(sd), rs = std::sqrt(1.0f - x0 * x0), ph = 6.2831853f * x1;
                        const V3 smp = {std::cos(ph) * rs, std::sin(ph) * rs, x0};
                        const V3 dd = norm(Tn * smp.x + B * smp.y + N * smp.z);
                        rays.push_back({hit + dd * off, dd});
                    }
                }
            printf("  primary: %llu rays, %.1f node steps and %.1f leaf tests per ray, %llu hit -> %zu AO rays\n", (unsigned long long)C.rays,
                   double(C.nodes) / C.rays, double(C.prims) / C.rays, (unsigned long long)C.hits, rays.size());
        }
        if (getenv("LAB_CUT")) {
            const int L = atoi(getenv("LAB_CUT"));
            Counters C; uint64_t cutSteps = 0, cutSize = 0, synth = 0, cutMax = 0; std::vector<uint64_t> hist(65, 0);
            t0 = now();
#pragma omp parallel
            {
                Counters Lc; uint64_t cs = 0, cz = 0, sy = 0, cm = 0; std::vector<CutEntry> cut; std::vector<uint64_t> h(65, 0);
#pragma omp for schedule(dynamic, 64)
                for (size_t gi = 0; gi < groups.size(); gi++) {
                    entryCut(W, groups[gi], aoRadius + groups[gi].off, L, cut, cs);
                    cz += cut.size(); cm = std::max<uint64_t>(cm, cut.size()); h[std::min<size_t>(64, cut.size())]++;
                    for (int sidx = 0; sidx < spp; sidx++) {
                        const AoRay& r = rays[gi * size_t(spp) + sidx];
                        const float inv[3] = {1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z}, oo[3] = {r.o.x, r.o.y, r.o.z};
                        float best = aoRadius; bool found = false; uint32_t prim = 0; int kind = 0;
                        Lc.rays++;
                        for (size_t b = 0; b < cut.size(); b += 4) {       // one synthetic node step per four cut entries
                            sy++;
                            float key[4]; int idx[4]; int nh = 0;
                            for (size_t k = b; k < std::min(cut.size(), b + 4); k++) {
                                float tn = 0.0f, tf = best;
                                for (int a = 0; a < 3; a++) { float t0 = (cut[k].b.mn[a] - oo[a]) * inv[a], t1 = (cut[k].b.mx[a] - oo[a]) * inv[a]; if (t0 > t1) std::swap(t0, t1); tn = std::max(tn, t0); tf = std::min(tf, t1); }
                                if (tn <= tf) { key[nh] = tn; idx[nh] = int(k); nh++; }
                            }
                            for (int a = 1; a < nh; a++) for (int q = a; q > 0 && key[q] < key[q - 1]; q--) { std::swap(key[q], key[q - 1]); std::swap(idx[q], idx[q - 1]); }
                            for (int q = 0; q < nh; q++) {
                                if (key[q] > best) continue;
                                float t; found = trace(W, r.o, r.d, 0.0f, best, t, prim, kind, Lc, cut[idx[q]].c, found, false); best = t;
                            }
                        }
                        if (found) Lc.hits++;
                    }
                }
#pragma omp critical
                { C.rays += Lc.rays; C.nodes += Lc.nodes; C.prims += Lc.prims; C.hits += Lc.hits; cutSteps += cs; cutSize += cz; synth += sy; cutMax = std::max(cutMax, cm); for (int k = 0; k < 65; k++) hist[k] += h[k]; }
            }
            printf("  AO with a per-pixel entry cut at wide level %d: cut %.1f entries per pixel (max %llu), %.1f node steps per pixel to find it; per ray %.2f synthetic + %.2f tree node steps = %.2f (+ %.2f amortised cut steps), %.2f leaf tests, %.1f %% hit (%.1f s)\n",
                   L, double(cutSize) / groups.size(), (unsigned long long)cutMax, double(cutSteps) / groups.size(), double(synth) / C.rays, double(C.nodes) / C.rays,
                   double(synth + C.nodes) / C.rays, double(cutSteps) / C.rays, double(C.prims) / C.rays, 100.0 * C.hits / C.rays, now() - t0);
            printf("    cut size histogram (pixels):"); for (int k = 0; k < 65; k++) if (hist[k]) printf(" %d:%llu", k, (unsigned long long)hist[k]); printf("\n");
        }
        for (g_order = 0; g_order < (getenv("LAB_ALL_ORDERS") ? 5 : 1); g_order++) {
        Counters C;
        t0 = now();
#pragma omp parallel
        {
            Counters L;
#pragma omp for schedule(dynamic, 4096)
            for (size_t i = 0; i < rays.size(); i++) { float t; uint32_t p = 0; int kind; trace(W, rays[i].o, rays[i].d, 0.0f, aoRadius, t, p, kind, L); }
#pragma omp critical
            { C.rays += L.rays; C.nodes += L.nodes; C.prims += L.prims; C.hits += L.hits; C.boxes += L.boxes; C.barren += L.barren; C.stalePops += L.stalePops; for (int k = 0; k < 48; k++) C.perLevel[k] += L.perLevel[k]; }
        }
        static const char* ORD[5] = {"nearest first (kernel)", "fully sorted", "as stored", "sign order along the node's widest axis", "octant-diagonal order"};
        printf("  AO, %s: %.2f node steps (%.2f of them with no child hit; %.2f pops whose entry distance lies beyond the hit found meanwhile), %.2f child boxes, %.2f leaf tests per ray, %.1f %% of the rays hit (%.1f s)\n", ORD[g_order], double(C.nodes) / C.rays,
               double(C.barren) / C.rays, double(C.stalePops) / C.rays, double(C.boxes) / C.rays, double(C.prims) / C.rays, 100.0 * C.hits / C.rays, now() - t0);
        if (g_order == 0) {
            printf("    node steps per ray by wide level:");
            for (int k = 0; k < W.levels && k < 48; k++) printf(" %.2f", double(C.perLevel[k]) / C.rays);
            printf("\n");
        }
        fflush(stdout);
        }
        g_order = 0;
    }
    return 0;
}
