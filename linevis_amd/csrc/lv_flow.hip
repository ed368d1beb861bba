// lv_flow.hip -- streamline tracing through a regular grid on the GPU (the producer of the line sets the hot path renders).
//
// Replaces the OpenMP/TBB loop over seeds of the reference's flow tracer, src/LineData/Flow/StreamlineTracingGrid.cpp:
//   setGridExtent / addVectorField (max |v|)          :81-218      -> lv_set_flow_grid (k_max_magnitude)
//   _getVectorAtPosition, _getScalarFieldAtPosition   :857-913     -> lv_vector_at / lv_scalar_at (trilinear, 8 gathers)
//   _rayBoxIntersection                               :949-1010    -> lv_ray_box
//   _trace + _integrationStep{ExplicitEuler,ImplicitEuler,Heun,Midpoint,RK4,RKF45} :1193-1396
//                                                                  -> k_trace_streamlines (one thread per seed and
//                                                                  direction; the steps of a line are inherently serial)
//   traceStreamlines: backward reversal, forward/backward merge, minimum-length filter :344-426,1118-1166 -> host part of
//                                                                  lv_trace_streamlines
// float32, fixed evaluation order (library-wide -ffp-contract=off): positions and attributes are bit-identical to a
// host evaluation of the same formulas.  Output of the kernel is [step][thread] so that the lanes of a wave, which
// advance in lock step, write neighbouring addresses.
#include <algorithm>
#include <cstring>

#include "lv_internal.h"

namespace {

struct LvFlowGrid {
    const float* V;          // xs*ys*zs*3
    const float* scalars;    // numScalars * xs*ys*zs
    int xs, ys, zs;
    float dx, dy, dz;
    float bx, by, bz;        // box maximum (minimum is the origin), setGridExtent :113-115
    uint32_t numScalars;
};

struct LvCell { int x, y, z; float fx, fy, fz, ix, iy, iz; };

__device__ __forceinline__ LvCell lv_locate(const LvFlowGrid& g, f3 p) {
    // gridPositionFloat = (p - box.min) * (1/dx, 1/dy, 1/dz); ivec3() truncates towards zero, fract() = x - floor(x)
    const float qx = (p.x - 0.0f) * (1.0f / g.dx), qy = (p.y - 0.0f) * (1.0f / g.dy), qz = (p.z - 0.0f) * (1.0f / g.dz);
    LvCell c;
    c.x = int(qx); c.y = int(qy); c.z = int(qz);
    c.fx = qx - floorf(qx); c.fy = qy - floorf(qy); c.fz = qz - floorf(qz);
    c.ix = 1.0f - c.fx; c.iy = 1.0f - c.fy; c.iz = 1.0f - c.fz;
    return c;
}

__device__ __forceinline__ f3 lv_vector_at_idx(const LvFlowGrid& g, int x, int y, int z, bool fw) {
    if (x < 0 || y < 0 || z < 0 || x >= g.xs || y >= g.ys || z >= g.zs) return mk3(0.0f, 0.0f, 0.0f);
    const float* p = g.V + 3 * (size_t(x) + size_t(y) * g.xs + size_t(z) * g.xs * g.ys);
    return fw ? mk3(p[0], p[1], p[2]) : mk3(-p[0], -p[1], -p[2]);
}

__device__ __forceinline__ f3 lv_vector_at(const LvFlowGrid& g, f3 p, bool fw) {
    const LvCell c = lv_locate(g, p);
    f3 r = (c.ix * c.iy * c.iz) * lv_vector_at_idx(g, c.x, c.y, c.z, fw);
    r = r + (c.fx * c.iy * c.iz) * lv_vector_at_idx(g, c.x + 1, c.y, c.z, fw);
    r = r + (c.ix * c.fy * c.iz) * lv_vector_at_idx(g, c.x, c.y + 1, c.z, fw);
    r = r + (c.fx * c.fy * c.iz) * lv_vector_at_idx(g, c.x + 1, c.y + 1, c.z, fw);
    r = r + (c.ix * c.iy * c.fz) * lv_vector_at_idx(g, c.x, c.y, c.z + 1, fw);
    r = r + (c.fx * c.iy * c.fz) * lv_vector_at_idx(g, c.x + 1, c.y, c.z + 1, fw);
    r = r + (c.ix * c.fy * c.fz) * lv_vector_at_idx(g, c.x, c.y + 1, c.z + 1, fw);
    r = r + (c.fx * c.fy * c.fz) * lv_vector_at_idx(g, c.x + 1, c.y + 1, c.z + 1, fw);
    return r;
}

__device__ __forceinline__ float lv_scalar_at_idx(const LvFlowGrid& g, const float* f, int x, int y, int z) {
    if (x < 0 || y < 0 || z < 0 || x >= g.xs || y >= g.ys || z >= g.zs) return 0.0f;
    return f[size_t(x) + size_t(y) * g.xs + size_t(z) * g.xs * g.ys];
}

__device__ __forceinline__ float lv_scalar_at(const LvFlowGrid& g, const float* f, f3 p) {
    const LvCell c = lv_locate(g, p);
    float r = (c.ix * c.iy * c.iz) * lv_scalar_at_idx(g, f, c.x, c.y, c.z);
    r = r + (c.fx * c.iy * c.iz) * lv_scalar_at_idx(g, f, c.x + 1, c.y, c.z);
    r = r + (c.ix * c.fy * c.iz) * lv_scalar_at_idx(g, f, c.x, c.y + 1, c.z);
    r = r + (c.fx * c.fy * c.iz) * lv_scalar_at_idx(g, f, c.x + 1, c.y + 1, c.z);
    r = r + (c.ix * c.iy * c.fz) * lv_scalar_at_idx(g, f, c.x, c.y, c.z + 1);
    r = r + (c.fx * c.iy * c.fz) * lv_scalar_at_idx(g, f, c.x + 1, c.y, c.z + 1);
    r = r + (c.ix * c.fy * c.fz) * lv_scalar_at_idx(g, f, c.x, c.y + 1, c.z + 1);
    r = r + (c.fx * c.fy * c.fz) * lv_scalar_at_idx(g, f, c.x + 1, c.y + 1, c.z + 1);
    return r;
}

__device__ __forceinline__ bool lv_box_contains(const LvFlowGrid& g, f3 p) {
    return p.x >= 0.0f && p.y >= 0.0f && p.z >= 0.0f && p.x <= g.bx && p.y <= g.by && p.z <= g.bz;
}

__device__ __forceinline__ bool lv_ray_box_plane(float o, float d, float lower, float upper, float& tNear, float& tFar) {
    if (fabsf(d) < 0.00001f) {
        if (o < lower || o > upper) return false;
    } else {
        float t0 = (lower - o) / d, t1 = (upper - o) / d;
        if (t0 > t1) { float tmp = t0; t0 = t1; t1 = tmp; }
        if (t0 > tNear) tNear = t0;
        if (t1 < tFar) tFar = t1;
        if (tNear > tFar) return false;
        if (tFar < 0) return false;
    }
    return true;
}
__device__ __forceinline__ bool lv_ray_box(f3 o, f3 d, f3 upper, float& tNear, float& tFar) {
    tNear = -3.402823466e+38f;
    tFar = 3.402823466e+38f;
    if (!lv_ray_box_plane(o.x, d.x, 0.0f, upper.x, tNear, tFar)) return false;
    if (!lv_ray_box_plane(o.y, d.y, 0.0f, upper.y, tNear, tFar)) return false;
    if (!lv_ray_box_plane(o.z, d.z, 0.0f, upper.z, tNear, tFar)) return false;
    return true;
}

// _getVectorAtIdxDouble / _getVectorAtPositionDouble, :915-944
struct d3t { double x, y, z; };
__device__ __forceinline__ d3t mkd3(double x, double y, double z) { d3t r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ d3t operator+(d3t a, d3t b) { return mkd3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ d3t operator-(d3t a, d3t b) { return mkd3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ d3t operator*(d3t a, double s) { return mkd3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ d3t operator*(double s, d3t a) { return mkd3(s * a.x, s * a.y, s * a.z); }

__device__ __forceinline__ d3t lv_vector_at_idx_d(const LvFlowGrid& g, int x, int y, int z, bool fw) {
    const f3 v = lv_vector_at_idx(g, x, y, z, fw);
    return mkd3(double(v.x), double(v.y), double(v.z));
}

__device__ __forceinline__ d3t lv_vector_at_d(const LvFlowGrid& g, d3t p, bool fw) {
    const double qx = (p.x - 0.0) * (1.0 / double(g.dx)), qy = (p.y - 0.0) * (1.0 / double(g.dy)),
                 qz = (p.z - 0.0) * (1.0 / double(g.dz));
    const int x = int(qx), y = int(qy), z = int(qz);
    const double fx = qx - floor(qx), fy = qy - floor(qy), fz = qz - floor(qz);
    const double ix = 1.0 - fx, iy = 1.0 - fy, iz = 1.0 - fz;
    d3t r = (ix * iy * iz) * lv_vector_at_idx_d(g, x, y, z, fw);
    r = r + (fx * iy * iz) * lv_vector_at_idx_d(g, x + 1, y, z, fw);
    r = r + (ix * fy * iz) * lv_vector_at_idx_d(g, x, y + 1, z, fw);
    r = r + (fx * fy * iz) * lv_vector_at_idx_d(g, x + 1, y + 1, z, fw);
    r = r + (ix * iy * fz) * lv_vector_at_idx_d(g, x, y, z + 1, fw);
    r = r + (fx * iy * fz) * lv_vector_at_idx_d(g, x + 1, y, z + 1, fw);
    r = r + (ix * fy * fz) * lv_vector_at_idx_d(g, x, y + 1, z + 1, fw);
    r = r + (fx * fy * fz) * lv_vector_at_idx_d(g, x + 1, y + 1, z + 1, fw);
    return r;
}

// _integrationStepRKF45, :1341-1396: double precision; the step only ever shrinks and stays shrunk (dt by reference)
__device__ __forceinline__ void lv_integration_step_rkf45(const LvFlowGrid& g, float timeStepScale, f3& fP0, float& fDt,
                                                          bool fw) {
    const double EPSILON = double(2.0 * 1e-5) * double(fminf(g.dx, fminf(g.dy, g.dz))) * double(timeStepScale);
    const int MAX_NUM_ITERATIONS = 100;
    double dt = fDt;
    int iteration = 0;
    const d3t p0 = mkd3(fP0.x, fP0.y, fP0.z);
    d3t rk5;
    bool adapt;
    do {
        const d3t k1 = dt * lv_vector_at_d(g, p0, fw);
        const d3t k2 = dt * lv_vector_at_d(g, p0 + k1 * double(1.0 / 4.0), fw);
        const d3t k3 = dt * lv_vector_at_d(g, p0 + k1 * double(3.0 / 32.0) + k2 * double(9.0 / 32.0), fw);
        const d3t k4 = dt * lv_vector_at_d(
                g, p0 + k1 * double(1932.0 / 2197.0) - k2 * double(7200.0 / 2197.0) + k3 * double(7296.0 / 2197.0), fw);
        const d3t k5 = dt * lv_vector_at_d(
                g, p0 + k1 * double(439.0 / 216.0) - k2 * double(8.0) + k3 * double(3680.0 / 513.0) - k4 * double(845.0 / 4104.0), fw);
        const d3t k6 = dt * lv_vector_at_d(
                g, p0 - k1 * double(8.0 / 27.0) + k2 * double(2.0) - k3 * double(3544.0 / 2565.0) + k4 * double(1859.0 / 4104.0)
                           - k5 * double(11.0 / 40.0), fw);
        rk5 = p0 + k1 * double(16.0 / 135.0) + k3 * double(6656.0 / 12825.0) + k4 * double(28561.0 / 56430.0)
              - k5 * double(9.0 / 50.0) + k6 * double(2.0 / 55.0);
        const d3t e = k1 * double(1.0 / 360.0) + k3 * double(-128.0 / 4275.0) + k4 * double(-2197.0 / 75240.0)
                      + k5 * (1.0 / 50.0) + k6 * double(2.0 / 55.0);
        const double TE = sqrt(e.x * e.x + e.y * e.y + e.z * e.z);
        adapt = TE > EPSILON;
        if (adapt) dt = 0.9 * dt * pow(EPSILON / TE, double(1.0 / 5.0));
        iteration++;
    } while (adapt && iteration < MAX_NUM_ITERATIONS);
    fP0 = mk3(float(rk5.x), float(rk5.y), float(rk5.z));
    fDt = float(dt);
}

__device__ __forceinline__ void lv_integration_step(const LvFlowGrid& g, uint32_t method, f3& p0, float& dt, bool fw,
                                                    float timeStepScale) {
    if (method == 0u) {
        p0 = p0 + dt * lv_vector_at(g, p0, fw);
    } else if (method == 1u) { // implicit Euler by fixed-point iteration, :1285-1307
        const float EPSILON = 1e-6f;
        const int MAX_NUM_ITERATIONS = 100;
        int iteration = 0;
        f3 pLast = p0;
        float diff;
        do {
            const f3 pNext = p0 + dt * lv_vector_at(g, pLast, fw);
            diff = len3(pLast - pNext);
            pLast = pNext;
            iteration++;
        } while (diff > EPSILON && iteration < MAX_NUM_ITERATIONS);
        p0 = pLast;
    } else if (method == 5u) {
        lv_integration_step_rkf45(g, timeStepScale, p0, dt, fw);
    } else if (method == 2u) {
        const f3 v0 = lv_vector_at(g, p0, fw);
        const f3 p1 = p0 + dt * v0;
        const f3 v1 = lv_vector_at(g, p1, fw);
        p0 = p0 + (dt * 0.5f) * (v0 + v1);
    } else if (method == 3u) {
        const f3 pp = p0 + (dt * 0.5f) * lv_vector_at(g, p0, fw);
        p0 = p0 + dt * lv_vector_at(g, pp, fw);
    } else {
        const f3 k1 = dt * lv_vector_at(g, p0, fw);
        const f3 k2 = dt * lv_vector_at(g, p0 + k1 * 0.5f, fw);
        const f3 k3 = dt * lv_vector_at(g, p0 + k2 * 0.5f, fw);
        const f3 k4 = dt * lv_vector_at(g, p0 + k3, fw);
        const float s6 = 6.0f, s3 = 3.0f;
        const f3 a = mk3(k1.x / s6, k1.y / s6, k1.z / s6), b = mk3(k2.x / s3, k2.y / s3, k2.z / s3);
        const f3 c = mk3(k3.x / s3, k3.y / s3, k3.z / s3), d = mk3(k4.x / s6, k4.y / s6, k4.z / s6);
        p0 = p0 + (((a + b) + c) + d);
    }
}

// thread t: seed t % numSeeds, forward for t < numForward else backward.  Point i of thread t is stored at
// positions[(i * numThreads + t) * 3], attribute a at attributes[(a * capacity + i) * numThreads + t].
__global__ __launch_bounds__(LV_WAVE) void k_trace_streamlines(const LvFlowGrid g, const float* __restrict__ seeds,
                                                               uint32_t numSeeds, uint32_t numThreads,
                                                               uint32_t firstBackward, uint32_t method, float dt,
                                                               float timeStepScale, float terminationDistance,
                                                               int maxIterations,
                                                               float maxLineLength, uint32_t capacity,
                                                               float* __restrict__ positions,
                                                               float* __restrict__ attributes,
                                                               uint32_t* __restrict__ counts) {
    const uint32_t t = blockIdx.x * LV_WAVE + threadIdx.x;
    if (t >= numThreads) return;
    const bool fw = t < firstBackward;
    const uint32_t s = t % numSeeds;
    f3 p = mk3(seeds[3 * s], seeds[3 * s + 1], seeds[3 * s + 2]), old, last = p;
    uint32_t n = 0;
    auto push = [&](f3 q) {
        if (n < capacity) {
            float* o = positions + (size_t(n) * numThreads + t) * 3;
            o[0] = q.x; o[1] = q.y; o[2] = q.z;
            for (uint32_t a = 0; a < g.numScalars; a++)
                attributes[(size_t(a) * capacity + n) * numThreads + t] =
                        lv_scalar_at(g, g.scalars + size_t(a) * g.xs * g.ys * g.zs, q);
        }
        last = q;
        n++;
    };
    int iterationCounter = 0;
    float lineLength = 0.0f;
    while (iterationCounter <= maxIterations && lineLength <= maxLineLength) {
        old = p;
        if (!lv_box_contains(g, p)) {
            if (n != 0) { // clamp the position to the boundary, :1218-1233
                const f3 ro = last, rd = norm3(p - ro);
                float tNear, tFar;
                lv_ray_box(ro, rd, mk3(g.bx, g.by, g.bz), tNear, tFar);
                push(tNear > 0.0f ? ro + tNear * rd : ro + tFar * rd);
            }
            break;
        }
        push(p);
        lv_integration_step(g, method, p, dt, fw, timeStepScale); // (RKF45 shrinks this thread's dt)
        const float segmentLength = len3(p - old);
        lineLength += segmentLength;
        if (segmentLength < terminationDistance) break;
        iterationCounter++;
    }
    counts[t] = n;
}

// Cell of the seeder's occupancy grid (StreamlineMaxHelicityFirstSeeder::isPointTerminated, StreamlineSeeder.cpp:514-523): (xs - 1) x
// (ys - 1) x (zs - 1) cells, position / cell size truncated towards zero, clamped
__host__ __device__ __forceinline__ uint32_t lv_occupancy_cell(int xs, int ys, int zs, float dx, float dy, float dz, float px, float py,
                                                               float pz) {
    int x = int(px * (1.0f / dx)), y = int(py * (1.0f / dy)), z = int(pz * (1.0f / dz));
    x = x < 0 ? 0 : (x > xs - 2 ? xs - 2 : x);
    y = y < 0 ? 0 : (y > ys - 2 ? ys - 2 : y);
    z = z < 0 ? 0 : (z > zs - 2 ? zs - 2 : z);
    return uint32_t(x) + uint32_t(y) * uint32_t(xs - 1) + uint32_t(z) * uint32_t(xs - 1) * uint32_t(ys - 1);
}

// Point-based termination checks (TerminationCheckType NAIVE / KD_TREE_BASED / HASHED_GRID_BASED, StreamlineTracingGrid.cpp:676-689,
// StreamlineSeeder.cpp:514-529): "some point of a finished trajectory lies closer than minimumSeparationDistance".  The reference answers it
// by a loop over every finished point, a k-d tree or a hashed grid (sgl's, not vendored) -- three searches for ONE predicate, here
// distance(p, q) = sqrt((dx dx + dy dy) + dz dz) < r in float32.  MI355X form: the finished points live in HBM in per-cell linked lists of a
// uniform grid with cells >= r (a point within r of p lies in the 3 x 3 x 3 cells around p's), appended by k_flow_points_insert after
// every committed batch; a traced line walks the 27 lists of its new point.
struct LvFlowPoints {
    const float* pts;        // 3 floats per finished point
    const int32_t* next;     // next point of the same cell, -1 = end
    const int32_t* head;     // first point of a cell, -1 = empty
    int nx, ny, nz;
    float invCell, r;
};
__host__ __device__ __forceinline__ void lv_flow_points_cell(int nx, int ny, int nz, float invCell, float px, float py, float pz, int& x,
                                                             int& y, int& z) {
    x = int(px * invCell); y = int(py * invCell); z = int(pz * invCell);
    x = x < 0 ? 0 : (x > nx - 1 ? nx - 1 : x);
    y = y < 0 ? 0 : (y > ny - 1 ? ny - 1 : y);
    z = z < 0 ? 0 : (z > nz - 1 ? nz - 1 : z);
}
__host__ __device__ __forceinline__ bool lv_flow_points_closer(const float* pts, const int32_t* next, const int32_t* head, int nx, int ny,
                                                               int nz, float invCell, float r, float px, float py, float pz) {
    int cx, cy, cz;
    lv_flow_points_cell(nx, ny, nz, invCell, px, py, pz, cx, cy, cz);
    for (int z = (cz > 0 ? cz - 1 : 0); z <= (cz < nz - 1 ? cz + 1 : nz - 1); z++)
        for (int y = (cy > 0 ? cy - 1 : 0); y <= (cy < ny - 1 ? cy + 1 : ny - 1); y++)
            for (int x = (cx > 0 ? cx - 1 : 0); x <= (cx < nx - 1 ? cx + 1 : nx - 1); x++)
                for (int32_t i = head[(size_t(z) * ny + y) * nx + x]; i >= 0; i = next[i]) {
                    const float ddx = px - pts[3 * size_t(i)], ddy = py - pts[3 * size_t(i) + 1], ddz = pz - pts[3 * size_t(i) + 2];
                    if (sqrtf((ddx * ddx + ddy * ddy) + ddz * ddz) < r) return true;
                }
    return false;
}
__global__ __launch_bounds__(LV_BLOCK) void k_flow_points_insert(const float* __restrict__ pts, int32_t* __restrict__ next,
                                                                 int32_t* __restrict__ head, int nx, int ny, int nz, float invCell,
                                                                 uint32_t first, uint32_t count) {
    const uint32_t i = first + blockIdx.x * LV_BLOCK + threadIdx.x;
    if (i >= first + count) return;
    int x, y, z;
    lv_flow_points_cell(nx, ny, nz, invCell, pts[3 * size_t(i)], pts[3 * size_t(i) + 1], pts[3 * size_t(i) + 2], x, y, z);
    next[i] = atomicExch(&head[(size_t(z) * ny + y) * nx + x], int32_t(i));
}

// _traceStreamlineDecreasingHelicity + _isTerminated (StreamlineTracingGrid.cpp:546-738) for a batch of seeds, traced SPECULATIVELY
// against the occupancy grid as it stood when the batch was formed (lv_flow_trace_max_helicity_first commits the lines in seeding
// order and cuts each one where a line committed before it has claimed the cell: a point is pushed after its own termination
// test, so the part of a line in front of the cut is what the sequential algorithm would have traced).  Order of the tests as in
// the reference: iteration limit, cumulated length below the termination distance, leaving the box (boundary point appended, flagged
// in bit 31 of the count), loop check "Start Point", occupied cell.
__global__ __launch_bounds__(LV_WAVE) void k_trace_streamlines_seeded(const LvFlowGrid g, const float* __restrict__ seeds,
                                                                      uint32_t numSeeds, uint32_t numThreads, uint32_t firstBackward,
                                                                      uint32_t method, float dt, float timeStepScale,
                                                                      float terminationDistance, int maxIterations,
                                                                      uint32_t loopCheckMode, float terminationDistanceStart,
                                                                      const uint8_t* __restrict__ occupancy, uint32_t capacity,
                                                                      float* __restrict__ positions, float* __restrict__ attributes,
                                                                      uint32_t* __restrict__ counts, uint32_t* __restrict__ selfGrid,
                                                                      uint32_t selfGridWords, const LvFlowPoints fp) {
    const uint32_t t = blockIdx.x * LV_WAVE + threadIdx.x;
    if (t >= numThreads) return;
    // state of the other loop checks (members of the reference's tracer, reset after every line, :727-737): "Grid" = the cells this line
    // has visited (one bit per cell, zeroed by the host before the batch) + the last 32 cells it entered (CircularQueue<size_t>(32):
    // sgl's class, not vendored -- here a FIFO of 32, `contains` = membership); "Curvature" = a double sum and a segment count
    uint32_t* myGrid = selfGrid ? selfGrid + size_t(t) * selfGridWords : nullptr;
    uint32_t cellQueue[32];
    uint32_t queueHead = 0u, queueSize = 0u, oldCell = 0xFFFFFFFFu, segmentSum = 0u;
    double curvatureSum = 0.0;
    const bool fw = t < firstBackward;
    const uint32_t s = t % numSeeds;
    f3 cur = mk3(seeds[3 * s], seeds[3 * s + 1], seeds[3 * s + 2]), lastPoint = cur, back = cur, pt0 = cur, pt1 = cur, prev = cur;
    uint32_t n = 0;
    bool boundary = false;
    auto push = [&](f3 q) {
        if (n < capacity) {
            float* o = positions + (size_t(n) * numThreads + t) * 3;
            o[0] = q.x; o[1] = q.y; o[2] = q.z;
            for (uint32_t a = 0; a < g.numScalars; a++)
                attributes[(size_t(a) * capacity + n) * numThreads + t] =
                        lv_scalar_at(g, g.scalars + size_t(a) * g.xs * g.ys * g.zs, q);
        }
        if (n == 0) pt0 = q;
        if (n == 1) pt1 = q;
        prev = back;
        back = q;
        n++;
    };
    float segmentLength = 0.0f;   // (the reference's name: the length of the whole line so far)
    int iterationCounter = 0;
    for (;;) {
        if (iterationCounter > maxIterations) break;
        if (n != 0 && segmentLength < terminationDistance) break;
        if (!lv_box_contains(g, cur)) {
            if (n != 0) {
                const f3 ro = back, rd = norm3(cur - ro);
                float tNear, tFar;
                lv_ray_box(ro, rd, mk3(g.bx, g.by, g.bz), tNear, tFar);
                push(tNear > 0.0f ? ro + tNear * rd : ro + tFar * rd);
                boundary = true;
            }
            break;
        }
        if (n > 1 && loopCheckMode == 1u) {   // LoopCheckMode::START_POINT, :590-607
            f3 dir0 = pt1 - pt0;
            const float dist0 = len3(dir0);
            dir0 = mk3(dir0.x / dist0, dir0.y / dist0, dir0.z / dist0);
            f3 dirNow = cur - back;
            const float distNow = len3(dirNow);
            dirNow = mk3(dirNow.x / distNow, dirNow.y / distNow, dirNow.z / distNow);
            const float distToStart = len3(cur - pt0);
            const float planeDistance = dot3(dir0, cur) + (-dot3(dir0, pt0));   // sgl::Plane(normal, point).getDistance
            if (planeDistance < 0.0f && distToStart < terminationDistanceStart && dot3(dir0, dirNow) > 0.0f) break;
        }
        if (n > 1 && loopCheckMode == 2u) {   // LoopCheckMode::ALL_POINTS, :609-626: the points this line has pushed after its first one,
            // each with the direction it was reached in (hashedGridLoop->add, :689-692); sgl::HashedGrid is not vendored: the sphere
            // query is defined here as distance <= radius, evaluated over the line's own points
            f3 dirNow = cur - back;
            const float distNow = len3(dirNow);
            dirNow = mk3(dirNow.x / distNow, dirNow.y / distNow, dirNow.z / distNow);
            bool loop = false;
            const uint32_t stored = n < capacity ? n : capacity;
            // (the scan is O(points) per step -- the reference asks a hashed grid; here the common case of a point is a 12-B load and a
            // coordinate-wise reject: |p - cur| <= r needs every |component| <= r.  The direction the point was reached in -- three IEEE
            // divisions and a square root -- is evaluated only for the points inside the sphere, from the point stored before it:
            // the same values)
            for (uint32_t i = 1; i < stored && !loop; i++) {
                const float* o = positions + (size_t(i) * numThreads + t) * 3;
                const f3 p = mk3(o[0], o[1], o[2]);
                const f3 dp = p - cur;
                if (!(fabsf(dp.x) <= terminationDistanceStart && fabsf(dp.y) <= terminationDistanceStart &&
                      fabsf(dp.z) <= terminationDistanceStart)) continue;
                if (!(len3(dp) <= terminationDistanceStart)) continue;
                const float* ob = positions + (size_t(i - 1u) * numThreads + t) * 3;
                const f3 dir0 = norm3(p - mk3(ob[0], ob[1], ob[2]));
                const float planeDistance = dot3(dir0, cur) + (-dot3(dir0, p));
                loop = planeDistance < 0.0f && distNow < terminationDistanceStart && dot3(dir0, dirNow) > 0.0f;
            }
            if (loop) break;
        } else if (n > 1 && loopCheckMode == 3u) {   // LoopCheckMode::GRID, :627-649
            const uint32_t cell = lv_occupancy_cell(g.xs, g.ys, g.zs, g.dx, g.dy, g.dz, cur.x, cur.y, cur.z);
            const uint32_t word = myGrid[cell >> 5], bit = 1u << (cell & 31u);
            myGrid[cell >> 5] = word | bit;
            bool inQueue = false;
            for (uint32_t i = 0; i < queueSize; i++) inQueue = inQueue || cellQueue[(queueHead + i) & 31u] == cell;
            if ((word & bit) && !inQueue) break;
            if (cell != oldCell) {
                if (queueSize == 32u) { queueHead = (queueHead + 1u) & 31u; queueSize--; }
                cellQueue[(queueHead + queueSize) & 31u] = cell;
                queueSize++;
            }
            oldCell = cell;
        } else if (n > 1 && loopCheckMode == 4u) {   // LoopCheckMode::CURVATURE, :650-671 (acos: the build's fixed formula, NaN outside [-1, 1] like std::acos)
            f3 dir0 = back - prev, dir1 = cur - back;
            const float length0 = len3(dir0), length1 = len3(dir1);
            if (length0 > 1e-8f) dir0 = mk3(dir0.x / length0, dir0.y / length0, dir0.z / length0);
            if (length1 > 1e-8f) dir1 = mk3(dir1.x / length1, dir1.y / length1, dir1.z / length1);
            const float c = dot3(dir0, dir1);
            const float ang = fabsf(c) <= 1.0f ? lv_atan2_det(sqrtf((1.0f - c) * (1.0f + c)), c) : __uint_as_float(0x7FC00000u);
            curvatureSum += double(ang) * double(length0 + length1);
            segmentSum++;
            if (segmentSum > 100u && curvatureSum > double(2.5f)) break;
        }
        // the termination check against the lines finished before this batch: occupied cell (grid-based), or a finished point closer than r
        if (occupancy ? occupancy[lv_occupancy_cell(g.xs, g.ys, g.zs, g.dx, g.dy, g.dz, cur.x, cur.y, cur.z)] != 0
                      : lv_flow_points_closer(fp.pts, fp.next, fp.head, fp.nx, fp.ny, fp.nz, fp.invCell, fp.r, cur.x, cur.y, cur.z)) break;
        push(cur);
        lv_integration_step(g, method, cur, dt, fw, timeStepScale);
        iterationCounter++;
        segmentLength += len3(cur - lastPoint);
        lastPoint = cur;
    }
    counts[t] = n | (boundary ? 0x80000000u : 0u);
}

// max |v| over the grid (addVectorField, :189-216): max of non-negative floats = max of their bit patterns
__global__ __launch_bounds__(LV_BLOCK) void k_max_magnitude(const float* __restrict__ v, uint64_t numCells, uint32_t* out) {
    float m = 0.0f;
    for (uint64_t i = uint64_t(blockIdx.x) * LV_BLOCK + threadIdx.x; i < numCells; i += uint64_t(gridDim.x) * LV_BLOCK) {
        const float vx = v[3 * i], vy = v[3 * i + 1], vz = v[3 * i + 2];
        m = fmaxf(m, sqrtf((vx * vx + vy * vy) + vz * vz));
    }
    m = lv_wave_max(m);
    if (lv_lane() == 0) atomicMax(out, __float_as_uint(m));
}

} // namespace

int lv_flow_set_grid(lv_ctx* ctx, const float* vectorField, uint32_t xs, uint32_t ys, uint32_t zs, float dx, float dy,
                     float dz, const float* const* scalarFields, uint32_t numScalarFields) {
    const uint64_t cells = uint64_t(xs) * ys * zs;
    hipStream_t st = ctx->stream;
    int rc;
    if ((rc = lv_buf_reserve(ctx, ctx->flowVectors, size_t(cells) * 12))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->flowScalars, size_t(cells) * 4 * (numScalarFields ? numScalarFields : 1)))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->flowMisc, 16))) return rc;
    LV_HIP(ctx, hipMemcpyAsync(ctx->flowVectors.ptr, vectorField, size_t(cells) * 12, hipMemcpyHostToDevice, st));
    for (uint32_t a = 0; a < numScalarFields; a++)
        LV_HIP(ctx, hipMemcpyAsync((float*)ctx->flowScalars.ptr + size_t(a) * cells, scalarFields[a], size_t(cells) * 4,
                                   hipMemcpyHostToDevice, st));
    LV_HIP(ctx, hipMemsetAsync(ctx->flowMisc.ptr, 0, 16, st));
    const uint32_t grid = uint32_t(std::min<uint64_t>((cells + LV_BLOCK - 1) / LV_BLOCK, 4096));
    k_max_magnitude<<<grid, LV_BLOCK, 0, st>>>((const float*)ctx->flowVectors.ptr, cells, (uint32_t*)ctx->flowMisc.ptr);
    LV_HIP(ctx, hipGetLastError());
    uint32_t bitsMax = 0;
    LV_HIP(ctx, hipMemcpyAsync(&bitsMax, ctx->flowMisc.ptr, 4, hipMemcpyDeviceToHost, st));
    LV_HIP(ctx, hipStreamSynchronize(st)); // host arrays are borrowed for the call only
    memcpy(&ctx->flowMaxMagnitude, &bitsMax, 4);
    ctx->flowXs = xs; ctx->flowYs = ys; ctx->flowZs = zs;
    ctx->flowDx = dx; ctx->flowDy = dy; ctx->flowDz = dz;
    ctx->flowNumScalars = numScalarFields;
    ctx->flowGridSet = true;
    return LV_OK;
}

int lv_flow_trace(lv_ctx* ctx, const float* seeds, uint32_t numSeeds, const lv_streamline_settings* S) {
    ctx->flowPositions.clear();
    ctx->flowAttributes.clear();
    ctx->flowOffsets.assign(1, 0u);
    ctx->flowSeedIndex.clear();
    if (numSeeds == 0) return LV_OK;
    hipStream_t st = ctx->stream;
    LvFlowGrid g;
    g.V = (const float*)ctx->flowVectors.ptr;
    g.scalars = (const float*)ctx->flowScalars.ptr;
    g.xs = int(ctx->flowXs); g.ys = int(ctx->flowYs); g.zs = int(ctx->flowZs);
    g.dx = ctx->flowDx; g.dy = ctx->flowDy; g.dz = ctx->flowDz;
    g.bx = float(g.xs - 1) * g.dx; g.by = float(g.ys - 1) * g.dy; g.bz = float(g.zs - 1) * g.dz;
    g.numScalars = ctx->flowNumScalars;
    // _trace, :1198-1213
    const float dt = 1.0f / ctx->flowMaxMagnitude * std::min(g.dx, std::min(g.dy, g.dz)) * S->time_step_scale;
    const float terminationDistance = 1e-6f * S->termination_distance;
    const int maxIterations = std::min(int(roundf(float(S->max_num_iterations) / S->time_step_scale)),
                                       S->max_num_iterations * 10);
    const float diag = sqrtf((g.bx * g.bx + g.by * g.by) + g.bz * g.bz);
    const float maxLineLength = diag * (float(S->max_num_iterations) / float(2000));
    const uint32_t dirs = S->integration_direction == 2u ? 2u : 1u;
    const uint32_t numThreads = numSeeds * dirs;
    const uint32_t firstBackward = S->integration_direction == 0u ? numThreads : (S->integration_direction == 1u ? 0u : numSeeds);
    const uint64_t capacity64 = uint64_t(maxIterations) + 2;
    const uint32_t k = g.numScalars;
    const uint64_t bytes = capacity64 * numThreads * (12 + 4 * uint64_t(k));
    if (capacity64 > 0x7FFFFFFFull || bytes > (64ull << 30))
        return lv_fail(ctx, LV_E_CAPACITY, "streamline buffers would need %llu bytes", (unsigned long long)bytes);
    const uint32_t capacity = uint32_t(capacity64);
    int rc;
    if ((rc = lv_buf_reserve(ctx, ctx->flowSeeds, size_t(numSeeds) * 12))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->flowOutPos, size_t(capacity) * numThreads * 12))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->flowOutAtt, size_t(capacity) * numThreads * 4 * (k ? k : 1)))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->flowCounts, size_t(numThreads) * 4))) return rc;
    LV_HIP(ctx, hipMemcpyAsync(ctx->flowSeeds.ptr, seeds, size_t(numSeeds) * 12, hipMemcpyHostToDevice, st));
    k_trace_streamlines<<<(numThreads + LV_WAVE - 1) / LV_WAVE, LV_WAVE, 0, st>>>(
            g, (const float*)ctx->flowSeeds.ptr, numSeeds, numThreads, firstBackward, S->integration_method, dt,
            S->time_step_scale, terminationDistance, maxIterations, maxLineLength, capacity, (float*)ctx->flowOutPos.ptr,
            (float*)ctx->flowOutAtt.ptr, (uint32_t*)ctx->flowCounts.ptr);
    LV_HIP(ctx, hipGetLastError());
    std::vector<uint32_t> counts(numThreads);
    LV_HIP(ctx, hipMemcpyAsync(counts.data(), ctx->flowCounts.ptr, size_t(numThreads) * 4, hipMemcpyDeviceToHost, st));
    LV_HIP(ctx, hipStreamSynchronize(st));
    uint32_t maxCount = 0;
    for (uint32_t c : counts) maxCount = std::max(maxCount, c);
    if (maxCount > capacity) return lv_fail(ctx, LV_E_CAPACITY, "streamline longer than its buffer (%u > %u)", maxCount, capacity);
    // only the rows that hold points travel back: [step][thread] -> the first maxCount steps
    std::vector<float> hp(size_t(maxCount) * numThreads * 3), ha(size_t(k) * maxCount * numThreads);
    if (maxCount) {
        LV_HIP(ctx, hipMemcpyAsync(hp.data(), ctx->flowOutPos.ptr, hp.size() * 4, hipMemcpyDeviceToHost, st));
        for (uint32_t a = 0; a < k; a++)
            LV_HIP(ctx, hipMemcpyAsync(ha.data() + size_t(a) * maxCount * numThreads,
                                       (const float*)ctx->flowOutAtt.ptr + size_t(a) * capacity * numThreads,
                                       size_t(maxCount) * numThreads * 4, hipMemcpyDeviceToHost, st));
        LV_HIP(ctx, hipStreamSynchronize(st));
    }
    // traceStreamlines, :367-426: order forward / reversed backward, merge, drop lines that are too short
    ctx->flowAttributes.assign(k, std::vector<float>());
    std::vector<uint32_t> order; // (thread, step) pairs of one merged line, as step indices per part
    for (uint32_t s = 0; s < numSeeds; s++) {
        struct Part { uint32_t thread, begin, end; bool reversed; };
        Part parts[2];
        int np = 0;
        if (S->integration_direction == 0u) {
            parts[np++] = {s, 0u, counts[s], false};
        } else if (S->integration_direction == 1u) {
            parts[np++] = {s, 0u, counts[s], counts[s] > 1};
        } else {
            const uint32_t tb = numSeeds + s, nb = counts[tb];
            // the reversed backward line without its last point (= the seed), then the forward line
            if (nb > 1) parts[np++] = {tb, 1u, nb, true};
            parts[np++] = {s, 0u, counts[s], false};
        }
        size_t total = 0;
        for (int i = 0; i < np; i++) total += parts[i].end - parts[i].begin;
        if (total == 0) continue;
        auto at = [&](const Part& pt, uint32_t j) { // j-th point of the part in output order
            const uint32_t step = pt.reversed ? (pt.end - 1 - j) : (pt.begin + j);
            return size_t(step) * numThreads + pt.thread;
        };
        float len = 0.0f;
        bool have = false;
        float px = 0, py = 0, pz = 0;
        for (int i = 0; i < np; i++)
            for (uint32_t j = 0; j < parts[i].end - parts[i].begin; j++) {
                const float* q = &hp[at(parts[i], j) * 3];
                if (have) {
                    const float ddx = q[0] - px, ddy = q[1] - py, ddz = q[2] - pz;
                    len += sqrtf((ddx * ddx + ddy * ddy) + ddz * ddz);
                }
                px = q[0]; py = q[1]; pz = q[2];
                have = true;
            }
        if (!(len > S->minimum_length)) continue;
        for (int i = 0; i < np; i++)
            for (uint32_t j = 0; j < parts[i].end - parts[i].begin; j++) {
                const size_t src = at(parts[i], j);
                ctx->flowPositions.insert(ctx->flowPositions.end(), &hp[src * 3], &hp[src * 3] + 3);
                for (uint32_t a = 0; a < k; a++)
                    ctx->flowAttributes[a].push_back(ha[size_t(a) * maxCount * numThreads + src]);
            }
        ctx->flowOffsets.push_back(uint32_t(ctx->flowPositions.size() / 3));
        // where the seed sits in the merged line: behind the reversed backward part (streamribbons carry their ribbon
        // direction outwards from the seed in both parts, StreamlineTracingGrid.cpp:479-486)
        uint32_t seedIndex = 0u;
        if (S->integration_direction == 1u) seedIndex = counts[s] ? counts[s] - 1u : 0u;
        else if (S->integration_direction == 2u) seedIndex = counts[numSeeds + s] > 1u ? counts[numSeeds + s] - 1u : 0u;
        ctx->flowSeedIndex.push_back(seedIndex);
    }
    return LV_OK;
}

// StreamlineTracingGrid::_traceStreamribbonsDecreasingHelicity with StreamlineMaxHelicityFirstSeeder (see include/linevis_hip.h):
// sample queue on the host, batches of LV_HELICITY_BATCH seeds traced speculatively by k_trace_streamlines_seeded, sequential commit.
#define LV_HELICITY_BATCH 256u
int lv_flow_trace_max_helicity_first(lv_ctx* ctx, const float* helicityField, const lv_streamline_settings* S,
                                     const lv_helicity_seeding_settings* H) {
    ctx->flowPositions.clear();
    ctx->flowAttributes.clear();
    ctx->flowOffsets.assign(1, 0u);
    ctx->flowSeedIndex.clear();
    hipStream_t st = ctx->stream;
    LvFlowGrid g;
    g.V = (const float*)ctx->flowVectors.ptr;
    g.scalars = (const float*)ctx->flowScalars.ptr;
    g.xs = int(ctx->flowXs); g.ys = int(ctx->flowYs); g.zs = int(ctx->flowZs);
    g.dx = ctx->flowDx; g.dy = ctx->flowDy; g.dz = ctx->flowDz;
    g.bx = float(g.xs - 1) * g.dx; g.by = float(g.ys - 1) * g.dy; g.bz = float(g.zs - 1) * g.dz;
    g.numScalars = ctx->flowNumScalars;
    const int xs = g.xs, ys = g.ys, zs = g.zs;
    const uint32_t k = g.numScalars;
    ctx->flowAttributes.assign(k, std::vector<float>());
    // ---- StreamlineMaxHelicityFirstSeeder::reset, StreamlineSeeder.cpp:364-426 (box minimum = origin, dimensions = (bx, by, bz))
    struct Sample { float value; float px, py, pz; uint32_t index; };
    std::vector<Sample> queue;
    const int f = H->seeding_subsampling_factor;
    if (f == 1) {
        for (int z = 1; z < zs - 1; z++)
            for (int y = 1; y < ys - 1; y++)
                for (int x = 1; x < xs - 1; x++)
                    queue.push_back({helicityField[size_t(x) + size_t(y) * xs + size_t(z) * xs * ys], g.bx * float(x) / float(xs),
                                     g.by * float(y) / float(ys), g.bz * float(z) / float(zs), uint32_t(queue.size())});
    } else {
        const int ncx = (xs - 1) / f, ncy = (ys - 1) / f, ncz = (zs - 1) / f;
        for (int z = 0; z < ncz; z++)
            for (int y = 0; y < ncy; y++)
                for (int x = 0; x < ncx; x++) {
                    const int xg = std::min(x * f, xs - 1), yg = std::min(y * f, ys - 1), zg = std::min(z * f, zs - 1);
                    queue.push_back({fabsf(helicityField[size_t(xg) + size_t(yg) * xs + size_t(zg) * xs * ys]),
                                     g.bx * (float(x) + 0.5f) / float(ncx), g.by * (float(y) + 0.5f) / float(ncy),
                                     g.bz * (float(z) + 0.5f) / float(ncz), uint32_t(queue.size())});
                }
    }
    // ascending, taken from the back (std::sort in the reference: the order of equal values is unspecified there; here: creation order)
    std::stable_sort(queue.begin(), queue.end(), [](const Sample& a, const Sample& b) { return a.value < b.value; });
    const size_t numCells = size_t(xs - 1) * (ys - 1) * (zs - 1);
    // termination_check_type (TerminationCheckType): 1 = the occupancy grid; 0 / 2 / 3 = naive / k-d tree / hashed grid = the point
    // predicate above (0 does not filter the seeds: StreamlineMaxHelicityFirstSeeder::hasNextPoint's last branch, StreamlineSeeder.cpp:452-454)
    const bool gridCheck = H->termination_check_type == 1u;
    const bool filterSeeds = H->termination_check_type != 0u;
    std::vector<uint8_t> occupancy(gridCheck ? numCells : 0, 0);
    int rc;
    if (gridCheck) {
        if ((rc = lv_buf_reserve(ctx, ctx->flowOccupancy, numCells))) return rc;
        LV_HIP(ctx, hipMemsetAsync(ctx->flowOccupancy.ptr, 0, numCells, st));
    }
    // the finished points' grid: cells of max(r, longest box edge / 128), on the host (exact cut of the speculative lines) and in HBM
    const float rSep = H->minimum_separation_distance;
    LvFlowPoints fp{};
    std::vector<int32_t> fpHead, fpNext;
    std::vector<float> fpPts;
    size_t fpOnDevice = 0;
    if (!gridCheck) {
        const float cell = std::max(rSep, std::max(g.bx, std::max(g.by, g.bz)) / 128.0f);
        if (!(cell > 0.0f)) return lv_fail(ctx, LV_E_INVALID, "termination_check_type %u needs a grid with an extent", H->termination_check_type);
        fp.invCell = 1.0f / cell;
        fp.r = rSep;
        fp.nx = std::max(1, int(g.bx * fp.invCell) + 1); fp.ny = std::max(1, int(g.by * fp.invCell) + 1); fp.nz = std::max(1, int(g.bz * fp.invCell) + 1);
        fpHead.assign(size_t(fp.nx) * fp.ny * fp.nz, -1);
        if ((rc = lv_buf_reserve(ctx, ctx->flowOccupancy, fpHead.size() * 4))) return rc;   // (the occupancy buffer holds the list heads)
        LV_HIP(ctx, hipMemsetAsync(ctx->flowOccupancy.ptr, 0xFF, fpHead.size() * 4, st));
        fp.head = (const int32_t*)ctx->flowOccupancy.ptr;   // (all lists empty: pts / next are not read before the first insert)
    }
    auto pointTerminated = [&](const float* q) {
        return lv_flow_points_closer(fpPts.data(), fpNext.data(), fpHead.data(), fp.nx, fp.ny, fp.nz, fp.invCell, fp.r, q[0], q[1], q[2]);
    };
    // ---- constants of _traceStreamribbonsDecreasingHelicity / _isTerminated, StreamlineTracingGrid.cpp:553-557,767-773
    const float dt = 1.0f / ctx->flowMaxMagnitude * std::min(g.dx, std::min(g.dy, g.dz)) * S->time_step_scale;
    const float terminationDistance = 1e-6f * S->termination_distance;
    const int maxIterations = std::min(int(roundf(float(S->max_num_iterations) / S->time_step_scale)), S->max_num_iterations * 10) * 10;
    const float diag = sqrtf((g.bx * g.bx + g.by * g.by) + g.bz * g.bz);
    const float terminationDistanceStart = diag / 100.0f * H->termination_distance_self;
    const uint32_t dirs = S->integration_direction == 2u ? 2u : 1u;
    const uint64_t capacity64 = uint64_t(maxIterations) + 3;
    const uint64_t bytes = capacity64 * LV_HELICITY_BATCH * dirs * (12 + 4 * uint64_t(k));
    if (capacity64 > 0x7FFFFFFFull || bytes > (64ull << 30))
        return lv_fail(ctx, LV_E_CAPACITY, "streamline buffers would need %llu bytes", (unsigned long long)bytes);
    const uint32_t capacity = uint32_t(capacity64);
    const uint32_t maxThreads = LV_HELICITY_BATCH * dirs;
    if ((rc = lv_buf_reserve(ctx, ctx->flowSeeds, size_t(LV_HELICITY_BATCH) * 12))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->flowOutPos, size_t(capacity) * maxThreads * 12))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->flowOutAtt, size_t(capacity) * maxThreads * 4 * (k ? k : 1)))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->flowCounts, size_t(maxThreads) * 4))) return rc;
    // loop check "Grid": one bit per cell and traced line of a batch (the reference keeps ONE such grid, its tracer is sequential)
    const uint32_t selfGridWords = H->loop_check_mode == 3u ? uint32_t((numCells + 31) / 32) : 0u;
    if (size_t(selfGridWords) * maxThreads * 4 > (16ull << 30))
        return lv_fail(ctx, LV_E_CAPACITY, "loop_check_mode 3 would need %llu bytes of per-line cell masks",
                       (unsigned long long)(size_t(selfGridWords) * maxThreads * 4));
    if (selfGridWords && (rc = lv_buf_reserve(ctx, ctx->flowSelfGrid, size_t(selfGridWords) * maxThreads * 4))) return rc;
    auto cellOf = [&](const float* q) { return lv_occupancy_cell(xs, ys, zs, g.dx, g.dy, g.dz, q[0], q[1], q[2]); };
    std::vector<float> seeds, hp, ha;
    std::vector<uint32_t> counts;
    const float r = H->minimum_separation_distance;
    while (!queue.empty()) {
        // the next seeds whose cell is free now (hasNextPoint, :428-458); taken cells can only become more
        seeds.clear();
        while (!queue.empty() && seeds.size() < size_t(LV_HELICITY_BATCH) * 3) {
            const Sample sm = queue.back();
            queue.pop_back();
            const float q[3] = {sm.px, sm.py, sm.pz};
            if (gridCheck ? !occupancy[cellOf(q)] : !(filterSeeds && pointTerminated(q))) seeds.insert(seeds.end(), q, q + 3);
        }
        const uint32_t numSeeds = uint32_t(seeds.size() / 3);
        if (numSeeds == 0) break;
        const uint32_t numThreads = numSeeds * dirs;
        const uint32_t firstBackward = S->integration_direction == 0u ? numThreads : (S->integration_direction == 1u ? 0u : numSeeds);
        LV_HIP(ctx, hipMemcpyAsync(ctx->flowSeeds.ptr, seeds.data(), seeds.size() * 4, hipMemcpyHostToDevice, st));
        if (selfGridWords) LV_HIP(ctx, hipMemsetAsync(ctx->flowSelfGrid.ptr, 0, size_t(selfGridWords) * numThreads * 4, st));
        k_trace_streamlines_seeded<<<(numThreads + LV_WAVE - 1) / LV_WAVE, LV_WAVE, 0, st>>>(
                g, (const float*)ctx->flowSeeds.ptr, numSeeds, numThreads, firstBackward, S->integration_method, dt, S->time_step_scale,
                terminationDistance, maxIterations, H->loop_check_mode, terminationDistanceStart,
                gridCheck ? (const uint8_t*)ctx->flowOccupancy.ptr : nullptr,
                capacity, (float*)ctx->flowOutPos.ptr, (float*)ctx->flowOutAtt.ptr, (uint32_t*)ctx->flowCounts.ptr,
                selfGridWords ? (uint32_t*)ctx->flowSelfGrid.ptr : nullptr, selfGridWords, fp);
        LV_HIP(ctx, hipGetLastError());
        counts.resize(numThreads);
        LV_HIP(ctx, hipMemcpyAsync(counts.data(), ctx->flowCounts.ptr, size_t(numThreads) * 4, hipMemcpyDeviceToHost, st));
        LV_HIP(ctx, hipStreamSynchronize(st));
        uint32_t maxCount = 0;
        for (uint32_t c : counts) maxCount = std::max(maxCount, c & 0x7FFFFFFFu);
        if (maxCount > capacity) return lv_fail(ctx, LV_E_CAPACITY, "streamline longer than its buffer (%u > %u)", maxCount, capacity);
        hp.resize(size_t(maxCount) * numThreads * 3);
        ha.resize(size_t(k) * maxCount * numThreads);
        if (maxCount) {
            LV_HIP(ctx, hipMemcpyAsync(hp.data(), ctx->flowOutPos.ptr, hp.size() * 4, hipMemcpyDeviceToHost, st));
            for (uint32_t a = 0; a < k; a++)
                LV_HIP(ctx, hipMemcpyAsync(ha.data() + size_t(a) * maxCount * numThreads,
                                           (const float*)ctx->flowOutAtt.ptr + size_t(a) * capacity * numThreads,
                                           size_t(maxCount) * numThreads * 4, hipMemcpyDeviceToHost, st));
            LV_HIP(ctx, hipStreamSynchronize(st));
        }
        // ---- commit in seeding order
        bool dirty = false;
        for (uint32_t s = 0; s < numSeeds; s++) {
            // an earlier line of this batch took the seed's cell / came within r of the seed: hasNextPoint skips it
            if (gridCheck ? occupancy[cellOf(&seeds[3 * size_t(s)])] != 0 : (filterSeeds && pointTerminated(&seeds[3 * size_t(s)]))) continue;
            // length of a thread's part as the sequential tracer would have produced it: up to the first point in a taken cell
            // (the boundary point is appended without that test)
            auto cut = [&](uint32_t thread) {
                const uint32_t n = counts[thread] & 0x7FFFFFFFu, tested = (counts[thread] >> 31) ? n - 1u : n;
                for (uint32_t i = 0; i < tested; i++) {
                    const float* q = &hp[(size_t(i) * numThreads + thread) * 3];
                    if (gridCheck ? occupancy[cellOf(q)] != 0 : pointTerminated(q)) return i;
                }
                return n;
            };
            struct Part { uint32_t thread, begin, end; bool reversed; };
            Part parts[2];
            int np = 0;
            uint32_t seedIndex = 0u;
            uint32_t nPart[2] = {0u, 0u};   // points of each direction's part after the cut
            for (uint32_t d = 0; d < dirs; d++) nPart[d] = cut(d * numSeeds + s);
            if (S->integration_direction == 0u) {
                parts[np++] = {s, 0u, nPart[0], false};
            } else if (S->integration_direction == 1u) {
                parts[np++] = {s, 0u, nPart[0], true};            // _reverseTrajectory
                seedIndex = nPart[0] ? nPart[0] - 1u : 0u;
            } else {
                // _reverseTrajectory(backward) + _insertBackwardTrajectory: the reversed backward line without its last point (= the
                // seed), then the forward line (:1118-1160, as lv_flow_trace)
                if (nPart[1] > 1) parts[np++] = {numSeeds + s, 1u, nPart[1], true};
                parts[np++] = {s, 0u, nPart[0], false};
                seedIndex = nPart[1] > 1u ? nPart[1] - 1u : 0u;
            }
            auto at = [&](const Part& pt, uint32_t j) {
                const uint32_t step = pt.reversed ? (pt.end - 1 - j) : (pt.begin + j);
                return size_t(step) * numThreads + pt.thread;
            };
            // isValid: _computeTrajectoryLength(forward) [+ _computeTrajectoryLength(backward)] >= minimumLength, :783-823
            float total = 0.0f;
            for (uint32_t d = 0; d < dirs; d++) {
                const uint32_t thread = d * numSeeds + s;
                float len = 0.0f;
                for (uint32_t i = 0; i + 1 < nPart[d]; i++) {
                    const float* pa = &hp[(size_t(i) * numThreads + thread) * 3];
                    const float* pb = &hp[(size_t(i + 1) * numThreads + thread) * 3];
                    const float ddx = pa[0] - pb[0], ddy = pa[1] - pb[1], ddz = pa[2] - pb[2];
                    len += sqrtf((ddx * ddx + ddy * ddy) + ddz * ddz);
                }
                total += len;
            }
            if (!(total >= S->minimum_length)) continue;
            size_t points = 0;
            for (int i = 0; i < np; i++) points += parts[i].end - parts[i].begin;
            if (points == 0) continue;
            const size_t firstPoint = ctx->flowPositions.size() / 3;
            for (int i = 0; i < np; i++)
                for (uint32_t j = 0; j < parts[i].end - parts[i].begin; j++) {
                    const size_t src = at(parts[i], j);
                    ctx->flowPositions.insert(ctx->flowPositions.end(), &hp[src * 3], &hp[src * 3] + 3);
                    for (uint32_t a = 0; a < k; a++) ctx->flowAttributes[a].push_back(ha[size_t(a) * maxCount * numThreads + src]);
                }
            ctx->flowOffsets.push_back(uint32_t(ctx->flowPositions.size() / 3));
            ctx->flowSeedIndex.push_back(seedIndex);
            // addFinishedTrajectory, StreamlineSeeder.cpp:464-502: every cell the sphere (point, minimumSeparationDistance) touches
            // (sgl::Sphere::intersects(AABB) is un-vendored; build-owned: squared distance from the centre to the box <= r^2)
            if (!gridCheck) {   // kdTree.build / hashedGrid.add (:503-511), filteredTrajectories.push_back (:819): the line's points join the lists
                for (size_t pnt = firstPoint; pnt < ctx->flowPositions.size() / 3; pnt++) {
                    const float* q = &ctx->flowPositions[pnt * 3];
                    int x, y, z;
                    lv_flow_points_cell(fp.nx, fp.ny, fp.nz, fp.invCell, q[0], q[1], q[2], x, y, z);
                    int32_t& hd = fpHead[(size_t(z) * fp.ny + y) * fp.nx + x];
                    fpNext.push_back(hd);
                    hd = int32_t(fpPts.size() / 3);
                    fpPts.insert(fpPts.end(), q, q + 3);
                }
            }
            for (size_t pnt = firstPoint; gridCheck && pnt < ctx->flowPositions.size() / 3; pnt++) {
                const float* q = &ctx->flowPositions[pnt * 3];
                auto cellCoord = [](float v, float cell, int hi) { int c = int(v * (1.0f / cell)); return c < 0 ? 0 : (c > hi ? hi : c); };
                const int x0 = cellCoord(q[0] - r, g.dx, xs - 2), x1 = cellCoord(q[0] + r, g.dx, xs - 2);
                const int y0 = cellCoord(q[1] - r, g.dy, ys - 2), y1 = cellCoord(q[1] + r, g.dy, ys - 2);
                const int z0 = cellCoord(q[2] - r, g.dz, zs - 2), z1 = cellCoord(q[2] + r, g.dz, zs - 2);
                for (int z = z0; z <= z1; z++)
                    for (int y = y0; y <= y1; y++)
                        for (int x = x0; x <= x1; x++) {
                            const float lo[3] = {float(x) * g.dx, float(y) * g.dy, float(z) * g.dz};
                            const float hi[3] = {float(x + 1) * g.dx, float(y + 1) * g.dy, float(z + 1) * g.dz};
                            float d2 = 0.0f;
                            for (int c = 0; c < 3; c++) {
                                const float dd = q[c] < lo[c] ? lo[c] - q[c] : (q[c] > hi[c] ? q[c] - hi[c] : 0.0f);
                                d2 += dd * dd;
                            }
                            if (d2 <= r * r) occupancy[size_t(x) + size_t(y) * (xs - 1) + size_t(z) * (xs - 1) * (ys - 1)] = 1;
                        }
            }
            dirty = true;
        }
        if (dirty && gridCheck) {
            LV_HIP(ctx, hipMemcpyAsync(ctx->flowOccupancy.ptr, occupancy.data(), numCells, hipMemcpyHostToDevice, st));
            LV_HIP(ctx, hipStreamSynchronize(st));
        }
        if (dirty && !gridCheck) {
            // the batch's finished points: appended to the device arrays (grown geometrically; the lists are rebuilt after a move) and
            // linked into their cells' lists
            const size_t have = fpPts.size() / 3;
            if (have > 0x7FFFFFF0ull) return lv_fail(ctx, LV_E_CAPACITY, "more than 2^31 finished points");
            if (have > ctx->flowPointsCapacity) {
                size_t cap = std::max<size_t>(have * 2, 1u << 16);
                if ((rc = lv_buf_reserve(ctx, ctx->flowPoints, cap * 12))) return rc;
                if ((rc = lv_buf_reserve(ctx, ctx->flowPointsNext, cap * 4))) return rc;
                ctx->flowPointsCapacity = cap;
                fpOnDevice = 0;
                LV_HIP(ctx, hipMemsetAsync(ctx->flowOccupancy.ptr, 0xFF, fpHead.size() * 4, st));
            }
            LV_HIP(ctx, hipMemcpyAsync((float*)ctx->flowPoints.ptr + 3 * fpOnDevice, fpPts.data() + 3 * fpOnDevice, (have - fpOnDevice) * 12,
                                       hipMemcpyHostToDevice, st));
            k_flow_points_insert<<<uint32_t((have - fpOnDevice + LV_BLOCK - 1) / LV_BLOCK), LV_BLOCK, 0, st>>>(
                    (const float*)ctx->flowPoints.ptr, (int32_t*)ctx->flowPointsNext.ptr, (int32_t*)ctx->flowOccupancy.ptr, fp.nx, fp.ny, fp.nz,
                    fp.invCell, uint32_t(fpOnDevice), uint32_t(have - fpOnDevice));
            LV_HIP(ctx, hipGetLastError());
            LV_HIP(ctx, hipStreamSynchronize(st));
            fpOnDevice = have;
            fp.pts = (const float*)ctx->flowPoints.ptr;
            fp.next = (const int32_t*)ctx->flowPointsNext.ptr;
            fp.head = (const int32_t*)ctx->flowOccupancy.ptr;
        }
    }
    return LV_OK;
}

