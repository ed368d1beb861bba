// lv_api.hip -- extern "C" entry points of include/linevis_hip.h (context, settings, orchestration glue).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "lv_internal.h"

int lv_fail(lv_ctx* ctx, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->lastError = buf;
    return code;
}

int lv_buf_reserve(lv_ctx* ctx, LvDeviceBuffer& b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b.ptr && b.bytes >= bytes) return LV_OK;
    if (b.ptr) {
        // the buffer may still be in use by work queued on the stream
        LV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(b.ptr);
        b.ptr = nullptr;
        b.bytes = 0;
    }
    LV_HIP(ctx, hipMalloc(&b.ptr, bytes));
    b.bytes = bytes;
    return LV_OK;
}

void lv_buf_free(LvDeviceBuffer& b) {
    if (b.ptr) (void)hipFree(b.ptr);
    b.ptr = nullptr;
    b.bytes = 0;
}

// 4x4 inverse by cofactor expansion (2x2 sub-determinants -> adjugate -> 1/det), the scheme of glm::inverse that
// LineData::updateVulkanUniformBuffers applies (src/LineData/LineData.cpp:1290-1291).  Column-major.
// glm operator*(mat4, mat4), column major: column j of the product = sum over k of A's column k times B[j][k], left to right
void lv_mat4_mul(const float* A, const float* B, float* out) {
    float r[16];
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++)
            r[4 * j + i] = ((A[i] * B[4 * j] + A[4 + i] * B[4 * j + 1]) + A[8 + i] * B[4 * j + 2]) + A[12 + i] * B[4 * j + 3];
    memcpy(out, r, sizeof r);
}

void lv_mat4_inverse(const float* m, float* inv) {
    float c00 = m[10] * m[15] - m[14] * m[11];
    float c02 = m[6] * m[15] - m[14] * m[7];
    float c03 = m[6] * m[11] - m[10] * m[7];
    float c04 = m[9] * m[15] - m[13] * m[11];
    float c06 = m[5] * m[15] - m[13] * m[7];
    float c07 = m[5] * m[11] - m[9] * m[7];
    float c08 = m[9] * m[14] - m[13] * m[10];
    float c10 = m[5] * m[14] - m[13] * m[6];
    float c11 = m[5] * m[10] - m[9] * m[6];
    float c12 = m[8] * m[15] - m[12] * m[11];
    float c14 = m[4] * m[15] - m[12] * m[7];
    float c15 = m[4] * m[11] - m[8] * m[7];
    float c16 = m[8] * m[14] - m[12] * m[10];
    float c18 = m[4] * m[14] - m[12] * m[6];
    float c19 = m[4] * m[10] - m[8] * m[6];
    float c20 = m[8] * m[13] - m[12] * m[9];
    float c22 = m[4] * m[13] - m[12] * m[5];
    float c23 = m[4] * m[9] - m[8] * m[5];

    float i00 = +((m[5] * c00 - m[6] * c04) + m[7] * c08);
    float i01 = -((m[1] * c00 - m[2] * c04) + m[3] * c08);
    float i02 = +((m[1] * c02 - m[2] * c06) + m[3] * c10);
    float i03 = -((m[1] * c03 - m[2] * c07) + m[3] * c11);
    float i10 = -((m[4] * c00 - m[6] * c12) + m[7] * c16);
    float i11 = +((m[0] * c00 - m[2] * c12) + m[3] * c16);
    float i12 = -((m[0] * c02 - m[2] * c14) + m[3] * c18);
    float i13 = +((m[0] * c03 - m[2] * c15) + m[3] * c19);
    float i20 = +((m[4] * c04 - m[5] * c12) + m[7] * c20);
    float i21 = -((m[0] * c04 - m[1] * c12) + m[3] * c20);
    float i22 = +((m[0] * c06 - m[1] * c14) + m[3] * c22);
    float i23 = -((m[0] * c07 - m[1] * c15) + m[3] * c23);
    float i30 = -((m[4] * c08 - m[5] * c16) + m[6] * c20);
    float i31 = +((m[0] * c08 - m[1] * c16) + m[2] * c20);
    float i32 = -((m[0] * c10 - m[1] * c18) + m[2] * c22);
    float i33 = +((m[0] * c11 - m[1] * c19) + m[2] * c23);

    float det = ((m[0] * i00 + m[1] * i10) + m[2] * i20) + m[3] * i30;
    float r = 1.0f / det;
    inv[0] = i00 * r;  inv[1] = i01 * r;  inv[2] = i02 * r;  inv[3] = i03 * r;
    inv[4] = i10 * r;  inv[5] = i11 * r;  inv[6] = i12 * r;  inv[7] = i13 * r;
    inv[8] = i20 * r;  inv[9] = i21 * r;  inv[10] = i22 * r; inv[11] = i23 * r;
    inv[12] = i30 * r; inv[13] = i31 * r; inv[14] = i32 * r; inv[15] = i33 * r;
}

// every device buffer a context owns: freed by lv_destroy, summed by lv_get_stats (device_bytes)
static std::vector<LvDeviceBuffer*> lv_all_buffers(lv_ctx* ctx) {
    return {&ctx->points, &ctx->segIdx, &ctx->nodes, &ctx->segs, &ctx->segAxis, &ctx->prismFrames, &ctx->leafSeg, &ctx->segToLeaf, &ctx->tf,
            &ctx->depthMinMax, &ctx->ao, &ctx->aoAlt, &ctx->featNormal, &ctx->featNormalAlt, &ctx->featPosition, &ctx->featPositionAlt,
            &ctx->eawPing, &ctx->eawPong, &ctx->tilesHaloDev, &ctx->fullFrameTile, &ctx->svgf.normalDepth, &ctx->svgf.normalDepthHistory,
            &ctx->svgf.flowFwidth, &ctx->svgf.moments, &ctx->svgf.momentsHistory, &ctx->svgf.colorHistory, &ctx->svgf.tempAccum,
            &ctx->svgf.tempAccumFiltered, &ctx->svgf.ping, &ctx->svgf.pong, &ctx->svgf.result, &ctx->aoGbuf, &ctx->aoList, &ctx->aoSamples,
            &ctx->counters, &ctx->ppllNodes, &ctx->ppllStart, &ctx->ppllCount, &ctx->ppllScratch, &ctx->prismRecords, &ctx->scanTemp, &ctx->ppllOverflow, &ctx->ppllCoarse, &ctx->prismLeafList, &ctx->flowOccupancy, &ctx->flowSelfGrid, &ctx->twistTex, &ctx->tilesDev, &ctx->outDev,
            &ctx->scratchRays, &ctx->stackOverflow, &ctx->trajPos, &ctx->trajAttr, &ctx->trajOff, &ctx->trajLineValid, &ctx->trajLineRef, &ctx->trajRecLine, &ctx->trajTess, &ctx->triIdx, &ctx->triVerts, &ctx->triPoints, &ctx->triNodes, &ctx->tris, &ctx->triPairFlag,
            &ctx->flowVectors, &ctx->flowScalars, &ctx->flowMisc, &ctx->flowSeeds, &ctx->flowOutPos, &ctx->flowOutAtt, &ctx->flowCounts,
            &ctx->bakeBlendingWeights, &ctx->bakeSamplingLocations, &ctx->bakedAo, &ctx->bakeLcgSkip, &ctx->bakedAoPending, &ctx->bakeCounters,
            &ctx->bakeGbuf, &ctx->bakeSamples, &ctx->bakeOverflow, &ctx->mlatTrace, &ctx->buildArena, &ctx->firstHit,
            &ctx->accum, &ctx->groupOrder[0].cost, &ctx->groupOrder[0].order, &ctx->groupOrder[1].cost, &ctx->groupOrder[1].order};
}

namespace {

typedef LvDevCounters LvDevCountersHost;

bool parseBool(const char* v) { return strcmp(v, "true") == 0 || strcmp(v, "1") == 0; } // InternalState.hpp:64-71

bool parseFloat(const char* v, float& out) {
    char* end = nullptr;
    float f = strtof(v, &end);
    if (end == v) return false;
    out = f;
    return true;
}
bool parseUint(const char* v, uint32_t& out) {
    char* end = nullptr;
    long long x = strtoll(v, &end, 10);
    if (end == v || x < 0 || x > 0xFFFFFFFFll) return false;
    out = uint32_t(x);
    return true;
}

void updateAoMode(lv_ctx* ctx) {
    // LineRenderer::setNewSettings, LineRenderer.cpp:462-488: AO is on iff a baker is set and strength > 0
    ctx->opt.useAmbientOcclusion = ctx->opt.aoBakerIsRtao && ctx->opt.aoStrength > 0.0f;
}

} // namespace

extern "C" {

const char* lv_version(void) { return "linevis_hip 0.1 (gfx950)"; }

lv_ctx* lv_create(int device_ordinal, int* err) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0 || device_ordinal < 0 || device_ordinal >= n) {
        if (err) *err = LV_E_HIP;
        return nullptr;
    }
    if (hipSetDevice(device_ordinal) != hipSuccess) {
        if (err) *err = LV_E_HIP;
        return nullptr;
    }
    lv_ctx* ctx = new lv_ctx();
    ctx->device = device_ordinal;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess && prop.multiProcessorCount > 0)
            ctx->numCUs = prop.multiProcessorCount;
    }
    memset(&ctx->stats, 0, sizeof(ctx->stats));
    if (hipStreamCreateWithFlags(&ctx->ownStream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        if (err) *err = LV_E_HIP;
        return nullptr;
    }
    ctx->stream = ctx->ownStream;
    for (int i = 0; i < 16; i++) {
        if (hipEventCreate(&ctx->ev[i]) != hipSuccess) {
            if (err) *err = LV_E_HIP;
            (void)hipStreamDestroy(ctx->ownStream);
            delete ctx;
            return nullptr;
        }
    }
    for (int k = 0; k < lv_ctx::kNumKernels; k++)
        for (int i = 0; i < 2 * lv_ctx::kRing; i++)
            if (hipEventCreate(&ctx->evKernel[k][i]) != hipSuccess) {
                if (err) *err = LV_E_HIP;
                return nullptr; // context leaked deliberately: the device is unusable
            }
    ctx->evCreated = true;
    if (err) *err = LV_OK;
    return ctx;
}

void lv_destroy(lv_ctx* ctx) {
    if (!ctx) return;
    if (ctx->multi) lv_multi_destroy(ctx); // the other ranks, the communicators and the gather buffers
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->bakeStream) {
        (void)hipStreamSynchronize(ctx->bakeStream);
        (void)hipStreamDestroy(ctx->bakeStream);
        (void)hipEventDestroy(ctx->evBakePrereq);
        (void)hipEventDestroy(ctx->evBakeDone);
    }
    for (LvDeviceBuffer* b : lv_all_buffers(ctx))
        lv_buf_free(*b);
    if (ctx->evCreated) {
        for (int i = 0; i < 16; i++) (void)hipEventDestroy(ctx->ev[i]);
        for (int k = 0; k < lv_ctx::kNumKernels; k++)
            for (int i = 0; i < 2 * lv_ctx::kRing; i++) (void)hipEventDestroy(ctx->evKernel[k][i]);
    }
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->ownStream) (void)hipStreamDestroy(ctx->ownStream);
    delete ctx;
}

const char* lv_last_error(const lv_ctx* ctx) { return ctx ? ctx->lastError.c_str() : "null context"; }

int lv_set_stream(lv_ctx* ctx, void* hip_stream) {
    if (!ctx) return LV_E_INVALID;
    (void)hipSetDevice(ctx->device);
    LV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->ownStream;
    ctx->evBuildValid = false;
    ctx->evTriBuildValid = false;
    ctx->evLinePointsValid = ctx->evTessValid = false;
    ctx->evFrameValid = false;
    for (int k = 0; k < lv_ctx::kNumKernels; k++) ctx->kernelLaunches[k] = 0;
    return LV_OK;
}

int lv_set_lines(lv_ctx* ctx, const lv_line_point* points, uint32_t num_points, const uint32_t* seg, uint32_t num_segments) {
    if (!ctx) return LV_E_INVALID;
    if ((num_points && !points) || (num_segments && !seg)) return lv_fail(ctx, LV_E_INVALID, "null input array");
    if (num_segments > 0x03FFFFFFu) return lv_fail(ctx, LV_E_CAPACITY, "at most 2^26-1 segments (leaf index field of the AO work queue)");
    for (uint64_t i = 0; i < 2ull * num_segments; i++)
        if (seg[i] >= num_points)
            return lv_fail(ctx, LV_E_INVALID, "segment %llu references point %u >= num_points %u",
                           (unsigned long long)(i / 2), seg[i], num_points);
    (void)hipSetDevice(ctx->device);
    lv_invalidate_bake(ctx);   // before any buffer is touched: a bake in flight on the second stream still reads the old ones
    int rc;
    if ((rc = lv_buf_reserve(ctx, ctx->points, size_t(num_points) * sizeof(lv_line_point)))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->segIdx, size_t(num_segments) * 8))) return rc;
    if (num_points)
        LV_HIP(ctx, hipMemcpyAsync(ctx->points.ptr, points, size_t(num_points) * sizeof(lv_line_point),
                                   hipMemcpyHostToDevice, ctx->stream));
    if (num_segments)
        LV_HIP(ctx, hipMemcpyAsync(ctx->segIdx.ptr, seg, size_t(num_segments) * 8, hipMemcpyHostToDevice, ctx->stream));
    LV_HIP(ctx, hipStreamSynchronize(ctx->stream)); // host arrays are borrowed for the call only
    ctx->numPoints = num_points;
    ctx->numSegs = num_segments;
    ctx->accelValid = false;
    // the triangle tubes belong to the lines they were tessellated from (the reference's passes fetch both in setLineData): new
    // lines need lv_set_tube_triangle_mesh again before rtao_geometry = auto / triangle_tubes, Triangle Mesh or the prebaker use it
    ctx->triMeshSet = false;
    ctx->triAccelValid = false;
    ctx->trajSet = false;        // these lines are the caller's: nothing to tessellate from
    ctx->triMeshFromTraj = false;
    // VulkanRayTracedAmbientOcclusionPass::setLineData (.cpp:437-460): denoiser->resetFrameNumber(), globalFrameNumber = 0,
    // lastFrameViewProjectionMatrix = the current camera's
    ctx->aoGlobalFrameNumber = 0;
    ctx->lastFrameViewProjValid = false;
    return lv_forward_to_ranks(ctx, [&](lv_ctx* p) { return lv_set_lines(p, points, num_points, seg, num_segments); });
}

int lv_set_tube_triangle_mesh(lv_ctx* ctx, const uint32_t* triangle_indices, uint32_t num_triangles,
                              const lv_tube_vertex* vertices, uint32_t num_vertices, const lv_line_point* line_points,
                              uint32_t num_line_points) {
    if (!ctx) return LV_E_INVALID;
    if ((num_triangles && !triangle_indices) || (num_vertices && !vertices) || (num_line_points && !line_points))
        return lv_fail(ctx, LV_E_INVALID, "null input array");
    if (num_triangles > 0x03FFFFFFu)
        return lv_fail(ctx, LV_E_CAPACITY, "at most 2^26-1 triangles (leaf index field of the AO work queue)");
    for (uint64_t i = 0; i < 3ull * num_triangles; i++)
        if (triangle_indices[i] >= num_vertices)
            return lv_fail(ctx, LV_E_INVALID, "triangle %llu references vertex %u >= num_vertices %u",
                           (unsigned long long)(i / 3), triangle_indices[i], num_vertices);
    for (uint32_t i = 0; i < num_vertices; i++)
        if ((vertices[i].vertexLinePointIndex & 0x7FFFFFFFu) >= num_line_points && num_triangles)
            return lv_fail(ctx, LV_E_INVALID, "vertex %u references line point %u >= num_line_points %u", i,
                           vertices[i].vertexLinePointIndex & 0x7FFFFFFFu, num_line_points);
    (void)hipSetDevice(ctx->device);
    lv_invalidate_bake(ctx);   // waits for a running asynchronous bake BEFORE its inputs are overwritten
    int rc;
    if ((rc = lv_buf_reserve(ctx, ctx->triIdx, size_t(num_triangles) * 12))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->triVerts, size_t(num_vertices) * sizeof(lv_tube_vertex)))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->triPoints, size_t(num_line_points) * sizeof(lv_line_point)))) return rc;
    if (num_triangles)
        LV_HIP(ctx, hipMemcpyAsync(ctx->triIdx.ptr, triangle_indices, size_t(num_triangles) * 12, hipMemcpyHostToDevice,
                                   ctx->stream));
    if (num_vertices)
        LV_HIP(ctx, hipMemcpyAsync(ctx->triVerts.ptr, vertices, size_t(num_vertices) * sizeof(lv_tube_vertex),
                                   hipMemcpyHostToDevice, ctx->stream));
    if (num_line_points)
        LV_HIP(ctx, hipMemcpyAsync(ctx->triPoints.ptr, line_points, size_t(num_line_points) * sizeof(lv_line_point),
                                   hipMemcpyHostToDevice, ctx->stream));
    LV_HIP(ctx, hipStreamSynchronize(ctx->stream)); // host arrays are borrowed for the call only
    ctx->numTris = num_triangles;
    ctx->numTriVerts = num_vertices;
    ctx->numTriPoints = num_line_points;
    ctx->triMeshSet = true;
    ctx->triMeshFromTraj = false;
    ctx->triAccelValid = false;
    return lv_forward_to_ranks(ctx, [&](lv_ctx* p) { return lv_set_tube_triangle_mesh(p, triangle_indices, num_triangles, vertices, num_vertices, line_points, num_line_points); });
}

int lv_set_ao_parametrization(lv_ctx* ctx, const float* blending_weights, uint32_t num_line_vertices,
                              const float* sampling_locations, uint32_t num_parametrization_vertices) {
    if (!ctx) return LV_E_INVALID;
    if ((num_line_vertices && !blending_weights) || (num_parametrization_vertices && !sampling_locations))
        return lv_fail(ctx, LV_E_INVALID, "null input array");
    for (uint32_t i = 0; i < num_line_vertices; i++)
        if (!(blending_weights[i] >= 0.0f) || !(blending_weights[i] < float(num_parametrization_vertices)))
            return lv_fail(ctx, LV_E_INVALID, "blending weight %u = %g is outside [0, %u)", i, double(blending_weights[i]),
                           num_parametrization_vertices);
    for (uint32_t i = 0; i < num_parametrization_vertices; i++)
        if (!(sampling_locations[i] >= 0.0f) || !(sampling_locations[i] < float(num_line_vertices)))
            return lv_fail(ctx, LV_E_INVALID, "sampling location %u = %g is outside [0, %u)", i,
                           double(sampling_locations[i]), num_line_vertices);
    (void)hipSetDevice(ctx->device);
    lv_invalidate_bake(ctx);   // waits for a running asynchronous bake BEFORE its inputs are overwritten
    int rc;
    if ((rc = lv_buf_reserve(ctx, ctx->bakeBlendingWeights, size_t(num_line_vertices ? num_line_vertices : 1) * 4))) return rc;
    if ((rc = lv_buf_reserve(ctx, ctx->bakeSamplingLocations, size_t(num_parametrization_vertices ? num_parametrization_vertices : 1) * 4))) return rc;
    if (num_line_vertices)
        LV_HIP(ctx, hipMemcpyAsync(ctx->bakeBlendingWeights.ptr, blending_weights, size_t(num_line_vertices) * 4,
                                   hipMemcpyHostToDevice, ctx->stream));
    if (num_parametrization_vertices)
        LV_HIP(ctx, hipMemcpyAsync(ctx->bakeSamplingLocations.ptr, sampling_locations, size_t(num_parametrization_vertices) * 4,
                                   hipMemcpyHostToDevice, ctx->stream));
    LV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->bakeNumLineVertices = num_line_vertices;
    ctx->bakeNumParametrizationVertices = num_parametrization_vertices;
    ctx->bakeParamSet = true;
    return lv_forward_to_ranks(ctx, [&](lv_ctx* p) { return lv_set_ao_parametrization(p, blending_weights, num_line_vertices, sampling_locations, num_parametrization_vertices); });
}

int lv_get_baked_ao(lv_ctx* ctx, float* out, uint64_t max_values) {
    if (!ctx || !out) return LV_E_INVALID;
    (void)hipSetDevice(ctx->device);
    int rc;
    if (!ctx->bakeValid && ctx->bakeAsyncPending && (rc = lv_bake_poll(ctx, true))) return rc;   // a started bake: wait for it
    if (!ctx->bakeValid && (rc = lv_bake_ambient_occlusion(ctx))) return rc;
    const uint64_t n = uint64_t(ctx->bakeNumParametrizationVertices) * ctx->opt.bakeNumTubeSubdivisions;
    if (max_values < n) return lv_fail(ctx, LV_E_CAPACITY, "baked AO table holds %llu values", (unsigned long long)n);
    if (n) LV_HIP(ctx, hipMemcpyAsync(out, ctx->bakedAo.ptr, size_t(n) * 4, hipMemcpyDeviceToHost, ctx->stream));
    LV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LV_OK;
}

int lv_bake_ao_start(lv_ctx* ctx) {
    if (!ctx) return LV_E_INVALID;
    (void)hipSetDevice(ctx->device);
    if (ctx->bakeAsyncPending && ctx->bakePendingGeneration == ctx->bakeGeneration) return LV_OK; // already on its way
    int rc;
    if (ctx->bakeAsyncPending && (rc = lv_bake_poll(ctx, true))) return rc;   // a stale one: let it finish, its table is dropped
    if ((rc = lv_bake_ambient_occlusion(ctx, true))) return rc;
    return lv_forward_to_ranks(ctx, [&](lv_ctx* p) { return lv_bake_ao_start(p); });
}

int lv_bake_ao_poll(lv_ctx* ctx, int* out_running, int* out_ready) {
    if (!ctx) return LV_E_INVALID;
    (void)hipSetDevice(ctx->device);
    int running = 0, ready = 1;
    const int n = lv_multi_num_ranks(ctx);
    for (int r = 0; r < n; r++) {
        lv_ctx* c = lv_multi_rank(ctx, r);
        (void)hipSetDevice(c->device);
        const int rc = lv_bake_poll(c, false);
        if (rc) return c == ctx ? rc : lv_fail(ctx, rc, "rank %d: %s", r, c->lastError.c_str());
        running |= c->bakeAsyncPending ? 1 : 0;
        ready &= c->bakeValid ? 1 : 0;
    }
    (void)hipSetDevice(ctx->device);
    if (out_running) *out_running = running;
    if (out_ready) *out_ready = ready;
    return LV_OK;
}

int lv_set_transfer_function(lv_ctx* ctx, const float* rgba, uint32_t n, float attr_min, float attr_max) {
    if (!ctx) return LV_E_INVALID;
    if (!rgba || n == 0) return lv_fail(ctx, LV_E_INVALID, "empty transfer function");
    (void)hipSetDevice(ctx->device);
    int rc;
    if ((rc = lv_buf_reserve(ctx, ctx->tf, size_t(n) * 16))) return rc;
    LV_HIP(ctx, hipMemcpyAsync(ctx->tf.ptr, rgba, size_t(n) * 16, hipMemcpyHostToDevice, ctx->stream));
    LV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->tfN = n;
    ctx->attrMin = attr_min;
    ctx->attrMax = attr_max;
    return lv_forward_to_ranks(ctx, [&](lv_ctx* p) { return lv_set_transfer_function(p, rgba, n, attr_min, attr_max); });
}

// LineDataFlow::loadTwistLineTexture (LineDataFlow.cpp:93-171) without the PNG decoder: level 0 = the RGBA8 pixels as UNORM floats, then
// intlog2(max(w, h)) levels in all (:161-163) of 2 x 2 box averages -- in float, (a + b) + (c + d) times 0.25, odd extents clamp the second
// texel (sgl generates the chain with linear blits, whose rounding Vulkan leaves open).  rgba8 == NULL unloads the texture.
int lv_set_twist_line_texture(lv_ctx* ctx, const uint8_t* rgba8, uint32_t width, uint32_t height) {
    if (!ctx) return LV_E_INVALID;
    (void)hipSetDevice(ctx->device);
    if (!rgba8) {
        ctx->twistW = ctx->twistH = ctx->twistLevels = 0;
        return lv_forward_to_ranks(ctx, [&](lv_ctx* p) { return lv_set_twist_line_texture(p, nullptr, 0, 0); });
    }
    if (width == 0 || height == 0 || width > 16384 || height > 16384) return lv_fail(ctx, LV_E_INVALID, "twist-line texture extent");
    uint32_t levels = 0;
    for (uint32_t m = std::max(width, height); m > 1; m >>= 1) levels++;   // intlog2
    if (levels == 0) levels = 1;
    std::vector<float> tex;
    tex.reserve(size_t(width) * height * 4 * 2);
    for (size_t i = 0; i < size_t(width) * height * 4; i++) tex.push_back(float(rgba8[i]) / 255.0f);
    size_t prev = 0;
    uint32_t w = width, h = height;
    for (uint32_t l = 1; l < levels; l++) {
        const uint32_t nw = std::max(w >> 1, 1u), nh = std::max(h >> 1, 1u);
        const size_t cur = tex.size();
        tex.resize(cur + size_t(nw) * nh * 4);
        for (uint32_t j = 0; j < nh; j++)
            for (uint32_t i = 0; i < nw; i++)
                for (uint32_t c = 0; c < 4; c++) {
                    const uint32_t i0 = std::min(2 * i, w - 1), i1 = std::min(2 * i + 1, w - 1), j0 = std::min(2 * j, h - 1), j1 = std::min(2 * j + 1, h - 1);
                    const float a = tex[prev + (size_t(j0) * w + i0) * 4 + c], b = tex[prev + (size_t(j0) * w + i1) * 4 + c];
                    const float cc = tex[prev + (size_t(j1) * w + i0) * 4 + c], d = tex[prev + (size_t(j1) * w + i1) * 4 + c];
                    tex[cur + (size_t(j) * nw + i) * 4 + c] = ((a + b) + (cc + d)) * 0.25f;
                }
        prev = cur; w = nw; h = nh;
    }
    int rc;
    if ((rc = lv_buf_reserve(ctx, ctx->twistTex, tex.size() * 4))) return rc;
    LV_HIP(ctx, hipMemcpyAsync(ctx->twistTex.ptr, tex.data(), tex.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    LV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->twistW = width; ctx->twistH = height; ctx->twistLevels = levels;
    return lv_forward_to_ranks(ctx, [&](lv_ctx* p) { return lv_set_twist_line_texture(p, rgba8, width, height); });
}

int lv_set_camera(lv_ctx* ctx, const float view[16], const float proj[16], float fov_y, float near_dist, float far_dist,
                  uint32_t w, uint32_t h) {
    if (!ctx) return LV_E_INVALID;
    if (!view || !proj || w == 0 || h == 0) return lv_fail(ctx, LV_E_INVALID, "invalid camera / viewport");
    // (pixel coordinates travel as 16-bit pairs through the wave-local queues and the PPLL fragment records)
    if (w > 0xFFFFu || h > 0xFFFFu) return lv_fail(ctx, LV_E_CAPACITY, "viewport %u x %u: at most 65535 pixels per side", w, h);
    memcpy(ctx->view, view, 64);
    memcpy(ctx->proj, proj, 64);
    lv_mat4_inverse(ctx->view, ctx->invView);
    lv_mat4_inverse(ctx->proj, ctx->invProj);
    ctx->fovY = fov_y;
    ctx->nearDist = near_dist;
    ctx->farDist = far_dist;
    ctx->width = w;
    ctx->height = h;
    ctx->cameraSet = true;
    return lv_forward_to_ranks(ctx, [&](lv_ctx* p) { return lv_set_camera(p, view, proj, fov_y, near_dist, far_dist, w, h); });
}

int lv_set_background(lv_ctx* ctx, const float rgba[4]) {
    if (!ctx || !rgba) return LV_E_INVALID;
    memcpy(ctx->background, rgba, 16);
    return lv_forward_to_ranks(ctx, [&](lv_ctx* p) { return lv_set_background(p, rgba); });
}

int lv_set_option(lv_ctx* ctx, const char* key, const char* value) {
    if (!ctx) return LV_E_INVALID;
    if (!key || !value) return lv_fail(ctx, LV_E_INVALID, "null key/value");
    LvOptions& o = ctx->opt;
    auto bad = [&]() { return lv_fail(ctx, LV_E_INVALID, "invalid value '%s' for option '%s'", value, key); };
    std::string k(key);
    float f;
    uint32_t u;
    if (k == "line_width") {
        if (!parseFloat(value, f) || !(f > 0.0f)) return bad();
        if (f != o.lineWidth) lv_invalidate_bake(ctx);
        o.lineWidth = f; // accel rebuilt lazily (setTriangleRepresentationDirty, LineRenderer.cpp:436-441)
    } else if (k == "depth_cue_strength") {
        if (!parseFloat(value, f)) return bad();
        o.depthCueStrength = f;
    } else if (k == "ambient_occlusion_mode") {
        // AMBIENT_OCCLUSION_BAKER_TYPE_NAMES, AmbientOcclusionBaker.hpp:78-95
        if (strcmp(value, "RTAO (Screen Space)") == 0) { o.aoBakerIsRtao = true; o.aoPrebaked = false; }
        else if (strcmp(value, "RTAO (Prebaker)") == 0) { o.aoBakerIsRtao = true; o.aoPrebaked = true; }
        else if (strcmp(value, "None") == 0) { o.aoBakerIsRtao = false; o.aoPrebaked = false; }
        else return lv_fail(ctx, LV_E_INVALID, "ambient_occlusion_mode '%s' is not provided (None | RTAO (Screen Space) | "
                                               "RTAO (Prebaker))", value);
        updateAoMode(ctx);
    } else if (k == "ambient_occlusion_strength") {
        if (!parseFloat(value, f)) return bad();
        o.aoStrength = f;
        updateAoMode(ctx);
    } else if (k == "ambient_occlusion_gamma") {
        if (!parseFloat(value, f)) return bad();
        o.aoGamma = f;
    } else if (k == "ambient_occlusion_iterations") {
        if (!parseUint(value, u) || u == 0) return bad();
        o.aoIterations = u;
    } else if (k == "ambient_occlusion_samples_per_frame") {
        if (!parseUint(value, u) || u == 0) return bad();
        o.aoSamplesPerFrame = u;
    } else if (k == "ambient_occlusion_radius") {
        if (!parseFloat(value, f) || !(f > 0.0f)) return bad();
        if (f != o.aoRadius) lv_invalidate_bake(ctx);
        o.aoRadius = f;
    } else if (k == "ambient_occlusion_distance_based") {
        if (parseBool(value) != o.aoUseDistance) lv_invalidate_bake(ctx);
        o.aoUseDistance = parseBool(value);
    } else if (k == "rtao_prebaker_iterations") {            // VulkanAmbientOcclusionBaker.hpp:108 (GUI-only there)
        if (!parseUint(value, u) || u == 0) return bad();
        if (u != o.bakeIterations) lv_invalidate_bake(ctx);
        o.bakeIterations = u;
    } else if (k == "rtao_prebaker_samples_per_frame") {     // :166
        if (!parseUint(value, u) || u == 0 || u > 4096) return bad();
        if (u != o.bakeSamplesPerFrame) lv_invalidate_bake(ctx);
        o.bakeSamplesPerFrame = u;
    } else if (k == "rtao_prebaker_num_tube_subdivisions") { // :165
        if (!parseUint(value, u) || u < 3 || u > 64) return bad();
        if (u != o.bakeNumTubeSubdivisions) lv_invalidate_bake(ctx);
        o.bakeNumTubeSubdivisions = u;
    } else if (k == "use_jittered_primary_rays") {
        o.aoJitterPrimary = parseBool(value);
    } else if (k == "ambient_occlusion_denoiser") {
        // DENOISER_NAMES, Denoiser.hpp:61-65,96-100 (VulkanRayTracedAmbientOcclusion.cpp:683-696)
        const bool wasSvgf = o.svgfEnabled, wasEaw = o.eawEnabled;
        if (strcmp(value, "None") == 0) { o.eawEnabled = false; o.svgfEnabled = false; }
        else if (strcmp(value, "Edge-Avoiding \xC3\x80-Trous Wavelet Transform") == 0 || strcmp(value, "EAW") == 0) { o.eawEnabled = true; o.svgfEnabled = false; }
        else if (strcmp(value, "SVGF") == 0) { o.eawEnabled = false; o.svgfEnabled = true; }
        else return lv_fail(ctx, LV_E_INVALID, "ambient_occlusion_denoiser '%s' is not provided (None | Edge-Avoiding \xC3\x80-Trous "
                                               "Wavelet Transform | SVGF)", value);
        if (o.svgfEnabled != wasSvgf) ctx->svgf.historyValid = false; // createDenoiser(): fresh history
        if (o.svgfEnabled != wasSvgf || o.eawEnabled != wasEaw) {
            // any change of the denoiser: the image the colour pass samples (raw / EAW ping-pong / SVGF result) is stale, and a
            // running accumulation would mix feature maps that were never written -- setDenoiserType resets the frame number
            // (VulkanRayTracedAmbientOcclusion.cpp:683-696 -> onHasMoved): the accumulation restarts with the next frame
            ctx->aoResult = nullptr;
            ctx->aoRestart = true;
        }
    } else if (k == "use_ribbons") {                              // LineDataFlow.cpp:588 (here: USE_BANDS = ribbons on AND band data set)
        if (parseBool(value) != o.useRibbons) lv_invalidate_bake(ctx);   // the prebaker bakes the band cross-section (k_bake_setup)
        o.useRibbons = parseBool(value);
    } else if (k == "thick_bands") {                              // :592
        o.thickBands = parseBool(value);
    } else if (k == "min_band_thickness") {                       // :596
        if (!parseFloat(value, f) || !(f > 0.0f) || f > 1.0f) return bad();
        if (f != o.minBandThickness) lv_invalidate_bake(ctx);
        o.minBandThickness = f;
    } else if (k == "band_width") {                               // LineRenderer.cpp:443
        if (!parseFloat(value, f) || !(f > 0.0f)) return bad();
        if (f != o.bandWidth) lv_invalidate_bake(ctx);
        o.bandWidth = f;
    } else if (k == "use_analytic_elliptic_tubes") {              // "Elliptic Tubes" checkbox, VulkanRayTracer.cpp:198-201
        o.ellipticTubes = parseBool(value);
    } else if (k == "rotating_helicity_bands") {                  // LineDataFlow.cpp:601 (= USE_ROTATING_HELICITY_BANDS: the line
        o.helicityBands = parseBool(value);                       // points then carry lineRotation)
    } else if (k == "use_uniform_twist_line_width") {             // :618 (UNIFORM_HELICITY_BAND_WIDTH; "Triangle Mesh" geometry only)
        o.uniformTwistLineWidth = parseBool(value);
    } else if (k == "use_twist_line_texture") {                   // :626 (the texture itself: lv_set_twist_line_texture)
        o.useTwistLineTexture = parseBool(value);
    } else if (k == "twist_line_texture_filtering_mode" || k == "twist_line_texture_filtering_mode_index") {   // :640-669
        static const char* const names[] = {"Nearest", "Linear", "Nearest Mipmap Nearest", "Linear Mipmap Nearest",
                                            "Nearest Mipmap Linear", "Linear Mipmap Linear"};
        uint32_t mode = 6;
        if (k == "twist_line_texture_filtering_mode") { for (uint32_t i = 0; i < 6; i++) if (std::string(value) == names[i]) mode = i; }
        else if (!parseUint(value, mode)) return bad();
        if (mode >= 6) return bad();
        o.twistFilterMode = mode;
    } else if (k == "twist_line_texture_max_anisotropy") {        // :671 (anisotropic filtering is implementation-defined: 1 only)
        if (!parseUint(value, u) || u != 1) return bad();
    } else if (k == "separator_width") {                          // :609
        if (!parseFloat(value, f) || !(f >= 0.0f)) return bad();
        o.separatorWidth = f;
    } else if (k == "band_subdivisions") {                        // :613
        if (!parseUint(value, u) || u == 0) return bad();
        o.bandSubdivisions = u;
    } else if (k == "helicity_rotation_factor") {                 // :622
        if (!parseFloat(value, f)) return bad();
        o.helicityRotationFactor = f;
    } else if (k == "svgf_denoiser_iterations") {                // maxNumIterations, SVGF.cpp:427-436 (GUI only in the reference)
        if (!parseUint(value, u) || u > 5) return bad();
        o.svgfIterations = u;
    } else if (k == "svgf_denoiser_allowed_z_dist") {            // SVGF.hpp:70
        if (!parseFloat(value, f) || !(f >= 0.0f)) return bad();
        o.svgfAllowedZDist = f;
    } else if (k == "svgf_denoiser_allowed_normal_dist") {       // SVGF.hpp:71
        if (!parseFloat(value, f) || !(f >= 0.0f)) return bad();
        o.svgfAllowedNormalDist = f;
    } else if (k == "eaw_denoiser_iterations") {                 // EAWDenoiser.cpp:402-430
        if (!parseUint(value, u) || u > 5) return bad();
        o.eawIterations = u;
    } else if (k == "eaw_denoiser_color_weights") {
        o.eawColorWeights = parseBool(value);
    } else if (k == "eaw_denoiser_position_weights") {
        o.eawPositionWeights = parseBool(value);
    } else if (k == "eaw_denoiser_normal_weights") {
        o.eawNormalWeights = parseBool(value);
    } else if (k == "eaw_denoiser_phi_color") {
        if (!parseFloat(value, f) || !(f > 0.0f)) return bad();
        o.eawPhiColor = f;
    } else if (k == "eaw_denoiser_phi_position") {
        if (!parseFloat(value, f) || !(f > 0.0f)) return bad();
        o.eawPhiPosition = f;
    } else if (k == "eaw_denoiser_phi_normal") {
        if (!parseFloat(value, f) || !(f > 0.0f)) return bad();
        o.eawPhiNormal = f;
    } else if (k == "eaw_denoiser_use_shared_memory") {
        o.eawUseSharedMemory = parseBool(value);
    } else if (k == "num_samples_per_frame") {
        if (!parseUint(value, u) || u == 0) return bad();
        o.numSamplesPerFrame = u;
    } else if (k == "num_accumulated_frames") {
        if (!parseUint(value, u) || u == 0) return bad();
        o.numAccumulatedFrames = u; // VulkanRayTracer.cpp:257-260
        o.frameNumber = 0;
    } else if (k == "frame_number") {
        if (!parseUint(value, u)) return bad();
        o.frameNumber = u;
    } else if (k == "use_deterministic_sampling") {
        o.useDeterministicSampling = parseBool(value);
    } else if (k == "use_analytic_intersections") {
        o.rtTriangleMesh = !parseBool(value); // VulkanRayTracer.cpp:243-245
        o.rtLss = false;
    } else if (k == "geometry_mode") {
        // RAY_TRACING_GEOMETRY_MODE_NAMES, VulkanRayTracer.hpp:58-63 ("AABBs" / "Analytic" kept as short forms)
        if (strcmp(value, "Triangle Mesh") == 0) { o.rtTriangleMesh = true; o.rtLss = false; }
        else if (strcmp(value, "AABBs (analytic)") == 0 || strcmp(value, "AABBs") == 0 || strcmp(value, "Analytic") == 0) {
            o.rtTriangleMesh = false; o.rtLss = false;
        } else if (strcmp(value, "Linear Swept Spheres") == 0) {
            // the NVIDIA hardware primitive (chained end caps) = the exact union of the segments' capsules: the capsule path with
            // its caps always on and the exact roots; ClosestHitTubeLinearSweptSpheres (TubeRayTracing.glsl:621-737) is
            // ClosestHitTubeAnalytic with isCap from the hit's position along the segment
            o.rtTriangleMesh = false; o.rtLss = true;
        } else
            return lv_fail(ctx, LV_E_INVALID, "geometry_mode '%s' is not provided (Triangle Mesh | AABBs (analytic) | Linear Swept Spheres)", value);
    } else if (k == "use_mlat") {
        o.useMlat = parseBool(value); // VulkanRayTracer.cpp:266-270
    } else if (k == "mlat_num_nodes") {
        // addSliderIntPowerOfTwo("#MLAT Nodes", 1, 32), VulkanRayTracer.cpp:218
        if (!parseUint(value, u) || u == 0 || u > 32 || (u & (u - 1)) != 0) return bad();
        o.mlatNumNodes = u;
    } else if (k == "mlat_record_trace") {
        o.mlatRecordTrace = parseBool(value);
    } else if (k == "mlat_trace_capacity") {
        if (!parseUint(value, u) || u == 0 || u > (1u << 28)) return bad();
        o.mlatTraceCapacity = u;
    } else if (k == "max_depth_complexity") {
        if (!parseUint(value, u) || u == 0) return bad();
        o.maxDepthComplexity = u;
    } else if (k == "use_capped_tubes") {
        o.useCappedTubes = parseBool(value);
    } else if (k == "use_halos") {
        o.useHalos = parseBool(value);
    } else if (k == "shading_numerics") {
        if (strcmp(value, "fast") == 0) o.fastShading = true;
        else if (strcmp(value, "exact") == 0) o.fastShading = false;
        else return bad();
    } else if (k == "overlap_primary_passes") {
        o.overlapPrimaryPasses = strcmp(value, "auto") == 0 ? 2 : (parseBool(value) ? 1 : 0);
    } else if (k == "tube_num_subdivisions") {
        if (!parseUint(value, u) || u < 3) return bad();
        o.tubeNumSubdivisions = u;
    } else if (k == "sorting_mode") {
        // the "Sorting Mode" combo box of the PPLL renderer (PerPixelLinkedListLineRenderer.cpp:470-475): a name of
        // SORTING_MODE_NAMES (src/Renderers/PPLL.hpp:32-35) or its index
        static const char* const names[] = {"Priority Queue", "Bubble Sort", "Insertion Sort", "Shell Sort", "Max Heap",
                                            "Bitonic Sort", "Quicksort", "Quicksort Hybrid"};
        uint32_t m = 8;
        for (uint32_t i = 0; i < 8; i++) if (std::string(value) == names[i]) m = i;
        if (m == 8 && (!parseUint(value, m) || m > 7)) return bad();
        o.ppllSortingMode = m;
    } else if (k == "ppll_max_num_frags") {
        if (!parseUint(value, u)) return bad();
        o.ppllMaxNumFrags = u;
    } else if (k == "ppll_expected_avg_depth_complexity") {
        if (!parseUint(value, u)) return bad();
        o.ppllExpectedAvgDepthComplexity = u;
    } else if (k == "ppll_tile_width" || k == "ppll_tile_height") {
        if (!parseUint(value, u) || u == 0 || (u & (u - 1)) != 0) return bad(); // power of two (TiledAddress.glsl uses &)
        (k == "ppll_tile_width" ? o.ppllTileW : o.ppllTileH) = u;
    } else if (k == "collect_stats") {
        o.collectStats = parseBool(value);
    } else if (k == "intersection_form") {
        if (std::string(value) == "auto") o.intersectionForm = 0;
        else if (std::string(value) == "closest_approach") o.intersectionForm = 1;
        else if (std::string(value) == "literal") o.intersectionForm = 2;
        else return bad();
    } else if (k == "ppll_fragment_colour") {
        // which computeFragmentColor the PPLL gather runs: "raster" = the raster tube shader's (LinePassGeometryShaderTubes.glsl, what
        // the reference's gather pass executes; default), "ray_tracer" = RayHitCommon.glsl's (rounds 1-2 of this build; deviation probe)
        if (std::string(value) == "raster") o.ppllRayTracerColour = false;
        else if (std::string(value) == "ray_tracer") o.ppllRayTracerColour = true;
        else return bad();
    } else if (k == "ppll_fragment_source") {
        // where the PPLL gather's fragments come from: "raster_prism" = the rasterised N-gon prism of the reference's default "Tube
        // (Programmable Pull)" mode (lv_prism.h), "capsule_entry" = entry hits of the pixel-centre ray against the analytic capsules
        // (rounds 1-3; kept as the probe), "auto" (default) = raster_prism wherever it is built
        if (std::string(value) == "auto") o.ppllFragmentSource = 0;
        else if (std::string(value) == "capsule_entry") o.ppllFragmentSource = 1;
        else if (std::string(value) == "raster_prism") o.ppllFragmentSource = 2;
        else return bad();
    } else if (k == "ppll_prism_rasteriser") {
        // front end of ppll_fragment_source = raster_prism: which (pixel, segment) pairs the coverage test sees
        if (std::string(value) == "segments") o.ppllPrismLbvhWalk = false;
        else if (std::string(value) == "lbvh") o.ppllPrismLbvhWalk = true;
        else return bad();
    } else if (k == "triangle_leaf_size") {
        // triangles per leaf of the triangle LBVH (build-owned; the hits do not depend on it)
        uint32_t g;
        if (!parseUint(value, g) || g < 1 || g > 8) return bad();
        if (g != o.triLeafSize) { ctx->triAccelValid = false; lv_invalidate_bake(ctx); }
        o.triLeafSize = g;
    } else if (k == "triangle_leaf_records") {
        // pairs: 64-B records of four vertices for leaves of two triangles that share two (the build falls back to 48-B records
        // for a mesh with a pair that does not); triangles: always 48-B records.  Build-owned; the hits do not depend on it.
        bool p;
        if (std::string(value) == "pairs") p = true;
        else if (std::string(value) == "triangles") p = false;
        else return bad();
        if (p != o.triLeafPairs) { ctx->triAccelValid = false; lv_invalidate_bake(ctx); }
        o.triLeafPairs = p;
    } else if (k == "treelet_leaves") {
        uint32_t t;
        if (!parseUint(value, t) || t < 3 || t > 4096) return bad();
        if (t != o.treeletLeaves) { ctx->accelValid = false; ctx->triAccelValid = false; lv_invalidate_bake(ctx); }
        o.treeletLeaves = t;
    } else if (k == "treelet_lane_leaves") {
        // build-time tunable of the treelet pass (the tree does not depend on it): ranges of <= this many leaves are built by one lane
        // each instead of by the whole wave; 0 = the wave splits every range
        uint32_t t;
        if (!parseUint(value, t) || t == 1 || t > 64) return bad();
        if (t != o.treeletLaneLeaves) { ctx->accelValid = false; ctx->triAccelValid = false; lv_invalidate_bake(ctx); }
        o.treeletLaneLeaves = t;
    } else if (k == "treelet_group_leaves") {
        // build-time tunable of the treelet pass (the tree does not depend on it): ranges of <= 8 leaves are built by groups of 8 lanes
        // (8 ranges per pass of the wave); 16 (default): also those of 9 ... 16 leaves, by groups of 16 lanes that hand their small
        // children down to the groups of 8; 0 = see treelet_lane_leaves
        uint32_t t;
        if (!parseUint(value, t) || (t != 0 && t != 8 && t != 16)) return bad();
        if (t != o.treeletGroupLeaves) { ctx->accelValid = false; ctx->triAccelValid = false; lv_invalidate_bake(ctx); }
        o.treeletGroupLeaves = t;
    } else if (k == "kernel_timers") {
        // which launches of a frame are bracketed by HIP events (lv_get_kernel_times, the ms_* fields of lv_get_stats): "all" (default)
        // | "none" | a comma-separated list of LV_KERNEL_* numbers and / or "phases".  Every event record costs the stream 2 - 4 us.
        const std::string v(value);
        uint32_t mask = 0u;
        if (v == "all") mask = 0xFFFFFFFFu;
        else if (v != "none") {
            size_t pos = 0;
            while (pos <= v.size()) {
                size_t e = v.find(',', pos);
                if (e == std::string::npos) e = v.size();
                const std::string t = v.substr(pos, e - pos);
                pos = e + 1;
                uint32_t id;
                if (t == "phases") mask |= 1u << 31;
                else if (parseUint(t.c_str(), id) && id < 31u) mask |= 1u << id;
                else return bad();
            }
        }
        o.timerMask = mask;
    } else if (k == "accel_collapse_top") {
        // build-time tunable (same tree): the top levels of the 4-wide collapse in one launch ("true", default) or one pass per level
        const bool b = parseBool(value);
        if (b != o.collapseTop) { ctx->accelValid = false; ctx->triAccelValid = false; lv_invalidate_bake(ctx); }
        o.collapseTop = b;
    } else if (k == "treelet_plane_eval") {
        // how a wave evaluates the 45 split planes of a range (same planes, same costs, same tree): "scan" | "loop"
        bool sc;
        if (std::string(value) == "scan") sc = true;
        else if (std::string(value) == "loop") sc = false;
        else return bad();
        if (sc != o.treeletPlaneScan) { ctx->accelValid = false; ctx->triAccelValid = false; lv_invalidate_bake(ctx); }
        o.treeletPlaneScan = sc;
    } else if (k == "accel_build") {
        // the analogue of the reference's VK_BUILD_ACCELERATION_STRUCTURE_PREFER_FAST_TRACE_BIT_KHR (LineData.cpp:740-741): "fast_trace"
        // (default) rebuilds the LBVH's subtrees of <= treelet_leaves (512) leaves with a binned SAH, "fast_build" keeps the plain LBVH
        bool ft;
        if (std::string(value) == "fast_trace") ft = true;
        else if (std::string(value) == "fast_build") ft = false;
        else return bad();
        if (ft != o.accelFastTrace) { ctx->accelValid = false; ctx->triAccelValid = false; lv_invalidate_bake(ctx); }
        o.accelFastTrace = ft;
    } else if (k == "dispatch_order") {
        // tile kernels: "cost" = the 64x64-pixel groups start in the order of what they cost in the previous frame, heaviest
        // first (default); "as_numbered" = in tile-list order (measurement knob; the image is the same)
        if (std::string(value) == "cost") o.dispatchByCost = true;
        else if (std::string(value) == "as_numbered") o.dispatchByCost = false;
        else return bad();
    } else if (k == "rtao_geometry") {
        if (std::string(value) == "auto") o.rtaoGeometry = 0;
        else if (std::string(value) == "capsules") o.rtaoGeometry = 1;
        else if (std::string(value) == "triangle_tubes") o.rtaoGeometry = 2;
        else return bad();
    } else {
        return lv_fail(ctx, LV_E_INVALID, "unknown option '%s'", key);
    }
    return lv_forward_to_ranks(ctx, [&](lv_ctx* p) { return lv_set_option(p, key, value); });
}

int lv_build_accel(lv_ctx* ctx) {
    if (!ctx) return LV_E_INVALID;
    (void)hipSetDevice(ctx->device);
    int rc = lv_bvh_build(ctx);
    if (rc) return rc;
    // LineData::getRayTracingTubeTriangleTopLevelAS (LineData.cpp:986-1013): the triangle LBVH of the tube mesh, if there is one
    if ((rc = lv_ensure_tube_mesh(ctx))) return rc;
    if (ctx->triMeshSet && (rc = lv_bvh_build_triangles(ctx))) return rc;
    return lv_forward_to_ranks(ctx, [&](lv_ctx* p) { return lv_build_accel(p); });
}

lv_ctx* lv_create_multi(const int* device_ordinals, int num_devices, const char* transport, int* err) {
    if (!device_ordinals || num_devices <= 0) {
        if (err) *err = LV_E_INVALID;
        return nullptr;
    }
    lv_ctx* ctx = lv_create(device_ordinals[0], err);
    if (!ctx) return nullptr;
    const int rc = lv_multi_create(ctx, device_ordinals, num_devices, transport);
    if (rc) {
        fprintf(stderr, "lv_create_multi: %s\n", ctx->lastError.c_str());
        lv_destroy(ctx);
        if (err) *err = rc;
        return nullptr;
    }
    if (err) *err = LV_OK;
    return ctx;
}

int lv_multi_ranks(const lv_ctx* ctx) { return ctx ? lv_multi_num_ranks(ctx) : 0; }

int lv_multi_rebalance(lv_ctx* ctx, double base_cost_per_tile) {
    if (!ctx) return LV_E_INVALID;
    return lv_multi_rebalance_impl(ctx, base_cost_per_tile);
}

int lv_multi_deal(lv_ctx* ctx, uint32_t* out_owner, uint32_t capacity, uint32_t* out_count) {
    if (!ctx) return LV_E_INVALID;
    return lv_multi_get_deal(ctx, out_owner, capacity, out_count);
}

int lv_render_tiles_device(lv_ctx* ctx, int mode, const uint32_t* tiles_xy, uint32_t num_tiles, uint32_t tile_w,
                           uint32_t tile_h, void* out) {
    if (!ctx) return LV_E_INVALID;
    if (!tiles_xy || !out) return lv_fail(ctx, LV_E_INVALID, "null tile list / output");
    (void)hipSetDevice(ctx->device);
    if (ctx->multi) { // the caller's tiles dealt over the ranks, one gather, tile-major output on rank 0's device
        if (num_tiles == 0 || tile_w == 0 || tile_h == 0) return lv_fail(ctx, LV_E_INVALID, "empty tile list");
        return lv_multi_render(ctx, mode, tiles_xy, num_tiles, tile_w, tile_h, false, 0, 0, 0, 0, out);
    }
    return lv_frame_render(ctx, mode, tiles_xy, num_tiles, tile_w, tile_h, out);
}

int lv_render_device(lv_ctx* ctx, int mode, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, void* out) {
    if (ctx && ctx->multi) { // the rectangle in 64 x 64 tiles over the ranks
        if (!out || w == 0 || h == 0) return lv_fail(ctx, LV_E_INVALID, "null output / empty rectangle");
        (void)hipSetDevice(ctx->device);
        return lv_multi_render_rect(ctx, mode, x0, y0, w, h, out);
    }
    uint32_t xy[2] = {x0, y0};
    return lv_render_tiles_device(ctx, mode, xy, 1, w, h, out);
}

int lv_render(lv_ctx* ctx, int mode, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint8_t* out) {
    if (!ctx) return LV_E_INVALID;
    if (!out || w == 0 || h == 0) return lv_fail(ctx, LV_E_INVALID, "null output / empty rectangle");
    (void)hipSetDevice(ctx->device);
    int rc;
    if ((rc = lv_buf_reserve(ctx, ctx->outDev, size_t(w) * h * 4))) return rc;
    if ((rc = lv_render_device(ctx, mode, x0, y0, w, h, ctx->outDev.ptr))) return rc;
    LV_HIP(ctx, hipMemcpyAsync(out, ctx->outDev.ptr, size_t(w) * h * 4, hipMemcpyDeviceToHost, ctx->stream));
    LV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LV_OK;
}

static int lv_get_stats_impl(lv_ctx* ctx, lv_stats* out, bool aggregate) {
    if (!ctx || !out) return LV_E_INVALID;
    (void)hipSetDevice(ctx->device);
    LV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    lv_stats& s = ctx->stats;
    s.bvh_depth = ctx->bvhDepth;
    s.num_segments = ctx->numSegs;
    s.num_nodes = ctx->numNodes;
    s.num_tube_triangles = ctx->numTris;
    s.ppll_pool_nodes = ctx->ppllPoolNodes;
    auto ms = [&](int a, int b) {
        float t = 0.0f;
        if (hipEventElapsedTime(&t, ctx->ev[a], ctx->ev[b]) != hipSuccess) t = 0.0f;
        return t;
    };
    if (ctx->evBuildValid) s.ms_accel_build = ms(0, 1);
    if (ctx->evTriBuildValid) s.ms_tri_accel_build = ms(14, 15);
    if (ctx->evLinePointsValid) s.ms_line_points = ms(4, 6);
    if (ctx->evTessValid) s.ms_tessellate = ms(8, 9);
    s.num_tri_nodes = ctx->numTriNodes;
    s.tri_leaf_bytes = !ctx->triAccelValid ? 0u : ctx->triLeafPairs ? 64u : 48u * ctx->triLeafSize;
    if (ctx->evFrameValid) {
        // phase marks on the stream: 2 start, 5 depth range done, 7 RTAO done, 11 PPLL lists cleared, 13 gathered, 3 end
        s.ms_total = s.ms_depth_range = s.ms_ao = 0.0f;
        s.ms_color = s.ms_ppll_clear = s.ms_ppll_gather = s.ms_ppll_resolve = 0.0f;
        if (ctx->evPhaseRecorded) {   // (kernel_timers without the phase marks: the times stay 0, the counters below are still the frame's)
            s.ms_total = ms(2, 3);
            s.ms_depth_range = ms(2, 5);
            s.ms_ao = ms(5, 7);
            if (ctx->lastMode == LV_RENDERING_MODE_VULKAN_RAY_TRACER) s.ms_color = ms(7, 3);
            else { s.ms_ppll_clear = ms(7, 11); s.ms_ppll_gather = ms(11, 13); s.ms_ppll_resolve = ms(13, 3); }
        }
        LvDevCountersHost hc;
        memset(&hc, 0, sizeof(hc));
        if (ctx->counters.ptr) {
            LV_HIP(ctx, hipMemcpy(&hc, ctx->counters.ptr, sizeof(hc), hipMemcpyDeviceToHost));
        }
        s.rays_traced = hc.rays;
        s.nodes_visited = hc.nodes;
        s.prims_tested = hc.prims;
        s.hits_shaded = hc.hits;
        s.fragments = hc.fragCounter;
        s.ao_hit_pixels = hc.aoCount;
        s.max_depth_complexity = hc.maxDepthComplexity;
        s.ao_rays_traced = hc.aoRays;
        s.ao_nodes_visited = hc.aoNodes;
        s.ao_prims_tested = hc.aoPrims;
        s.max_nodes_per_pixel = hc.maxNodesPerPixel;
        s.ao_prim_hits = hc.aoPrimHits;
        s.ao_prim_may_axis = hc.aoPrimMayAxis;
        s.ao_prim_may_both = hc.aoPrimMayBoth;
        for (int k = 0; k < 3; k++) { s.ao_phase_iterations[k] = hc.aoPhaseIters[k]; s.ao_phase_lanes[k] = hc.aoPhaseLanes[k]; }
    }
    for (int k = 0; k < 8; k++) { s.ms_kernel_avg[k] = 0.0f; s.kernel_launches[k] = 0; }
    for (int k = 0; k < lv_ctx::kNumKernels; k++) {
        const uint64_t n = ctx->kernelLaunches[k];
        const uint64_t m = n < uint64_t(lv_ctx::kRing) ? n : uint64_t(lv_ctx::kRing);
        double sum = 0.0;
        for (uint64_t i = 0; i < m; i++) {
            const uint64_t slot = (n - 1 - i) % lv_ctx::kRing;
            float t = 0.0f;
            if (hipEventElapsedTime(&t, ctx->evKernel[k][2 * slot], ctx->evKernel[k][2 * slot + 1]) == hipSuccess) sum += t;
        }
        s.ms_kernel_avg[k] = m ? float(sum / double(m)) : 0.0f;
        s.kernel_launches[k] = uint32_t(n);
    }
    uint64_t bytes = 0;
    for (const LvDeviceBuffer* b : lv_all_buffers(ctx)) bytes += b->bytes; // the same list lv_destroy frees
    s.device_bytes = bytes;
    *out = s;
    // a multi-device handle reports the work of all its ranks (counters and memory summed; times are rank 0's)
    for (int r = 1; aggregate && r < lv_multi_num_ranks(ctx); r++) {
        lv_stats ps;
        const int rc = lv_get_stats_impl(lv_multi_rank(ctx, r), &ps, false);
        if (rc) return lv_fail(ctx, rc, "rank %d: %s", r, lv_multi_rank(ctx, r)->lastError.c_str());
        out->rays_traced += ps.rays_traced; out->nodes_visited += ps.nodes_visited; out->prims_tested += ps.prims_tested;
        out->hits_shaded += ps.hits_shaded; out->fragments += ps.fragments; out->ao_hit_pixels += ps.ao_hit_pixels;
        out->ao_rays_traced += ps.ao_rays_traced; out->ao_nodes_visited += ps.ao_nodes_visited; out->ao_prims_tested += ps.ao_prims_tested;
        out->ao_prim_hits += ps.ao_prim_hits; out->ao_prim_may_axis += ps.ao_prim_may_axis; out->ao_prim_may_both += ps.ao_prim_may_both;
        out->max_depth_complexity = ps.max_depth_complexity > out->max_depth_complexity ? ps.max_depth_complexity : out->max_depth_complexity;
        out->device_bytes += ps.device_bytes;
    }
    (void)hipSetDevice(ctx->device);
    return LV_OK;
}

int lv_get_stats(lv_ctx* ctx, lv_stats* out) { return lv_get_stats_impl(ctx, out, true); }

int lv_multi_rank_stats(lv_ctx* ctx, int rank, lv_stats* out) {
    if (!ctx || !out) return LV_E_INVALID;
    lv_ctx* c = lv_multi_rank(ctx, rank);
    if (!c) return lv_fail(ctx, LV_E_INVALID, "rank %d of %d", rank, lv_multi_num_ranks(ctx));
    const int rc = lv_get_stats_impl(c, out, false);
    if (rc && c != ctx) return lv_fail(ctx, rc, "rank %d: %s", rank, c->lastError.c_str());
    (void)hipSetDevice(ctx->device);
    return rc;
}

int lv_reset_timers(lv_ctx* ctx) {
    if (!ctx) return LV_E_INVALID;
    (void)hipSetDevice(ctx->device);
    LV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < lv_ctx::kNumKernels; k++) ctx->kernelLaunches[k] = 0;
    return LV_OK;
}

int lv_get_kernel_times(lv_ctx* ctx, int kernel_id, float* out_ms, uint32_t capacity, uint32_t* out_count) {
    if (!ctx) return LV_E_INVALID;
    if (kernel_id < 0 || kernel_id >= lv_ctx::kNumKernels || (!out_ms && capacity)) return lv_fail(ctx, LV_E_INVALID, "invalid kernel id / buffer");
    (void)hipSetDevice(ctx->device);
    LV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint64_t n = ctx->kernelLaunches[kernel_id];
    uint64_t m = n < uint64_t(lv_ctx::kRing) ? n : uint64_t(lv_ctx::kRing);
    if (m > capacity) m = capacity;
    for (uint64_t i = 0; i < m; i++) {
        const uint64_t slot = (n - m + i) % lv_ctx::kRing;
        float t = 0.0f;
        if (hipEventElapsedTime(&t, ctx->evKernel[kernel_id][2 * slot], ctx->evKernel[kernel_id][2 * slot + 1]) != hipSuccess) t = 0.0f;
        out_ms[i] = t;
    }
    if (out_count) *out_count = uint32_t(m);
    return LV_OK;
}

int lv_trace_rays(lv_ctx* ctx, const float* origins, const float* dirs, float t_min, float t_max, uint32_t n, float* out_t,
                  uint32_t* out_segment, uint32_t* out_kind) {
    if (!ctx) return LV_E_INVALID;
    if (n && (!origins || !dirs || !out_t || !out_segment || !out_kind)) return lv_fail(ctx, LV_E_INVALID, "null array");
    if (!ctx->points.ptr && ctx->numSegs == 0 && !ctx->accelValid && !ctx->segIdx.ptr)
        return lv_fail(ctx, LV_E_STATE, "lv_set_lines has not been called");
    (void)hipSetDevice(ctx->device);
    return lv_frame_trace_rays(ctx, origins, dirs, t_min, t_max, n, out_t, out_segment, out_kind);
}

int lv_trace_rays_triangles(lv_ctx* ctx, const float* origins, const float* dirs, float t_min, float t_max, uint32_t n,
                            float* out_t, uint32_t* out_triangle, float* out_uv) {
    if (!ctx) return LV_E_INVALID;
    if (n && (!origins || !dirs || !out_t || !out_triangle)) return lv_fail(ctx, LV_E_INVALID, "null array");
    (void)hipSetDevice(ctx->device);
    return lv_frame_trace_rays_triangles(ctx, origins, dirs, t_min, t_max, n, out_t, out_triangle, out_uv);
}

int lv_set_flow_grid(lv_ctx* ctx, const float* vector_field, uint32_t xs, uint32_t ys, uint32_t zs, float dx, float dy,
                     float dz, const float* const* scalar_fields, uint32_t num_scalar_fields) {
    if (!ctx) return LV_E_INVALID;
    if (!vector_field || xs < 2 || ys < 2 || zs < 2 || !(dx > 0.0f) || !(dy > 0.0f) || !(dz > 0.0f))
        return lv_fail(ctx, LV_E_INVALID, "flow grid needs a vector field, at least 2 cells per axis and positive spacing");
    if (uint64_t(xs) * ys * zs > 0x7FFFFFFFull) return lv_fail(ctx, LV_E_CAPACITY, "flow grid too large");
    if (num_scalar_fields && !scalar_fields) return lv_fail(ctx, LV_E_INVALID, "null scalar field table");
    for (uint32_t a = 0; a < num_scalar_fields; a++)
        if (!scalar_fields[a]) return lv_fail(ctx, LV_E_INVALID, "scalar field %u is null", a);
    (void)hipSetDevice(ctx->device);
    return lv_flow_set_grid(ctx, vector_field, xs, ys, zs, dx, dy, dz, scalar_fields, num_scalar_fields);
}

int lv_trace_streamlines(lv_ctx* ctx, const float* seed_points, uint32_t num_seeds, const lv_streamline_settings* settings,
                         uint64_t* out_num_lines, uint64_t* out_num_points) {
    if (!ctx) return LV_E_INVALID;
    if (!ctx->flowGridSet) return lv_fail(ctx, LV_E_STATE, "lv_set_flow_grid has not been called");
    if (!settings || (num_seeds && !seed_points)) return lv_fail(ctx, LV_E_INVALID, "null argument");
    const uint32_t m = settings->integration_method;
    if (m > 5u)
        return lv_fail(ctx, LV_E_INVALID, "integration method %u does not exist (0 explicit Euler, 1 implicit Euler, 2 Heun, "
                                          "3 midpoint, 4 RK4, 5 Runge-Kutta-Fehlberg)", m);
    if (settings->integration_direction > 2u) return lv_fail(ctx, LV_E_INVALID, "integration direction must be 0, 1 or 2");
    if (!(settings->time_step_scale > 0.0f) || settings->max_num_iterations <= 0 || settings->max_num_iterations > 10000000)
        return lv_fail(ctx, LV_E_INVALID, "time_step_scale must be > 0 and max_num_iterations in 1..1e7");
    if (!(ctx->flowMaxMagnitude > 0.0f)) return lv_fail(ctx, LV_E_STATE, "the vector field is zero everywhere");
    (void)hipSetDevice(ctx->device);
    int rc = lv_flow_trace(ctx, seed_points, num_seeds, settings);
    if (rc) return rc;
    if (out_num_lines) *out_num_lines = ctx->flowOffsets.size() - 1;
    if (out_num_points) *out_num_points = ctx->flowPositions.size() / 3;
    return LV_OK;
}

int lv_trace_streamlines_max_helicity_first(lv_ctx* ctx, const float* helicity_field, const lv_streamline_settings* settings,
                                            const lv_helicity_seeding_settings* seeding, uint64_t* out_num_lines,
                                            uint64_t* out_num_points) {
    if (!ctx) return LV_E_INVALID;
    if (!ctx->flowGridSet) return lv_fail(ctx, LV_E_STATE, "lv_set_flow_grid has not been called");
    if (!settings || !seeding || !helicity_field) return lv_fail(ctx, LV_E_INVALID, "null argument");
    if (settings->integration_method > 4u)
        return lv_fail(ctx, LV_E_INVALID, "max-helicity-first seeding: integration method %u is not built (0 ... 4; the adaptive step "
                                          "of Runge-Kutta-Fehlberg carries over from line to line)", settings->integration_method);
    if (settings->integration_direction > 2u) return lv_fail(ctx, LV_E_INVALID, "integration direction must be 0, 1 or 2");
    if (!(settings->time_step_scale > 0.0f) || settings->max_num_iterations <= 0 || settings->max_num_iterations > 1000000)
        return lv_fail(ctx, LV_E_INVALID, "time_step_scale must be > 0 and max_num_iterations in 1..1e6");
    if (seeding->termination_check_type > 3u)
        return lv_fail(ctx, LV_E_INVALID, "termination_check_type %u (0 naive, 1 grid-based, 2 k-d tree-based, 3 hashed grid-based)",
                       seeding->termination_check_type);
    if (seeding->loop_check_mode > 4u)
        return lv_fail(ctx, LV_E_INVALID, "loop_check_mode %u (0 none, 1 start point, 2 all points, 3 grid, 4 curvature)", seeding->loop_check_mode);
    if (!(seeding->minimum_separation_distance >= 0.0f) || seeding->seeding_subsampling_factor < 1)
        return lv_fail(ctx, LV_E_INVALID, "minimum_separation_distance must be >= 0 and seeding_subsampling_factor >= 1");
    if (!(ctx->flowMaxMagnitude > 0.0f)) return lv_fail(ctx, LV_E_STATE, "the vector field is zero everywhere");
    if (ctx->flowXs < 3 || ctx->flowYs < 3 || ctx->flowZs < 3) return lv_fail(ctx, LV_E_STATE, "the grid needs at least 3 points per axis");
    (void)hipSetDevice(ctx->device);
    int rc = lv_flow_trace_max_helicity_first(ctx, helicity_field, settings, seeding);
    if (rc) return rc;
    if (out_num_lines) *out_num_lines = ctx->flowOffsets.size() - 1;
    if (out_num_points) *out_num_points = ctx->flowPositions.size() / 3;
    return LV_OK;
}

int lv_get_streamlines(lv_ctx* ctx, float* positions, float* attributes, uint32_t* line_offsets) {
    if (!ctx) return LV_E_INVALID;
    if (ctx->flowOffsets.empty()) return lv_fail(ctx, LV_E_STATE, "lv_trace_streamlines has not been called");
    const size_t n = ctx->flowPositions.size() / 3;
    if (positions && n) memcpy(positions, ctx->flowPositions.data(), n * 12);
    if (attributes)
        for (size_t a = 0; a < ctx->flowAttributes.size(); a++)
            if (n) memcpy(attributes + a * n, ctx->flowAttributes[a].data(), n * 4);
    if (line_offsets) memcpy(line_offsets, ctx->flowOffsets.data(), ctx->flowOffsets.size() * 4);
    return LV_OK;
}

int lv_get_streamline_seed_indices(lv_ctx* ctx, uint32_t* out_seed_index) {
    if (!ctx || !out_seed_index) return LV_E_INVALID;
    if (ctx->flowOffsets.empty()) return lv_fail(ctx, LV_E_STATE, "lv_trace_streamlines has not been called");
    if (!ctx->flowSeedIndex.empty()) memcpy(out_seed_index, ctx->flowSeedIndex.data(), ctx->flowSeedIndex.size() * 4);
    return LV_OK;
}

int lv_compute_depth_range(lv_ctx* ctx, float out_min_max[2]) {
    if (!ctx || !out_min_max) return LV_E_INVALID;
    if (!ctx->cameraSet) return lv_fail(ctx, LV_E_STATE, "lv_set_camera has not been called");
    (void)hipSetDevice(ctx->device);
    int rc;
    if ((rc = lv_frame_depth_range(ctx))) return rc;
    LV_HIP(ctx, hipMemcpyAsync(out_min_max, ctx->depthMinMax.ptr, 8, hipMemcpyDeviceToHost, ctx->stream));
    LV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LV_OK;
}

int lv_get_ao(lv_ctx* ctx, float* out) {
    if (!ctx || !out) return LV_E_INVALID;
    if (!ctx->ao.ptr || !ctx->cameraSet) return lv_fail(ctx, LV_E_STATE, "no AO texture (render with RTAO first)");
    (void)hipSetDevice(ctx->device);
    if (ctx->aoW != ctx->width || ctx->aoH != ctx->height)
        return lv_fail(ctx, LV_E_STATE, "the AO texture was rendered at %ux%u, the viewport is now %ux%u (render again)",
                       ctx->aoW, ctx->aoH, ctx->width, ctx->height);
    // the image the colour pass samples: the accumulated AO factors, or their denoised version (ambient_occlusion_denoiser)
    const void* src = ctx->aoResult ? (const void*)ctx->aoResult : ctx->ao.ptr;
    LV_HIP(ctx, hipMemcpyAsync(out, src, size_t(ctx->aoW) * ctx->aoH * 4, hipMemcpyDeviceToHost, ctx->stream));
    LV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LV_OK;
}

int lv_get_ao_tile_costs(lv_ctx* ctx, uint32_t* out_counts, uint32_t capacity, uint32_t* out_count, uint32_t* out_groups_per_tile) {
    if (!ctx) return LV_E_INVALID;
    if (!ctx->aoNumGroups || !ctx->aoList.ptr) return lv_fail(ctx, LV_E_STATE, "no RTAO pass in the last render call");
    if (out_count) *out_count = ctx->aoNumGroups;
    if (out_groups_per_tile) *out_groups_per_tile = ctx->aoGroupsPerTile;
    if (out_counts) {
        if (capacity < ctx->aoNumGroups) return lv_fail(ctx, LV_E_CAPACITY, "out_counts holds %u entries, need %u", capacity, ctx->aoNumGroups);
        (void)hipSetDevice(ctx->device);
        LV_HIP(ctx, hipMemcpyAsync(out_counts, ctx->aoList.ptr, size_t(ctx->aoNumGroups) * 4, hipMemcpyDeviceToHost, ctx->stream));
        LV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return LV_OK;
}

int lv_get_dispatch_order(lv_ctx* ctx, uint32_t* out_order, uint32_t* out_cost, uint32_t capacity, uint32_t* out_count) {
    if (!ctx) return LV_E_INVALID;
    if (ctx->multi) ctx = lv_multi_rank(ctx, 0);
    const lv_ctx::GroupOrder& G = ctx->groupOrder[0];
    const uint32_t n = (G.active && ctx->evFrameValid) ? G.n : 0u;
    if (out_count) *out_count = n;
    if (n && (out_order || out_cost)) {
        if (capacity < n) return lv_fail(ctx, LV_E_CAPACITY, "the launch has %u groups", n);
        (void)hipSetDevice(ctx->device);
        if (out_order) LV_HIP(ctx, hipMemcpyAsync(out_order, G.order.ptr, size_t(n) * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (out_cost) LV_HIP(ctx, hipMemcpyAsync(out_cost, G.cost.ptr, size_t(n) * 4, hipMemcpyDeviceToHost, ctx->stream));
        LV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return LV_OK;
}

int lv_ppll_get_buffers(lv_ctx* ctx, uint32_t* out_nodes, uint64_t max_nodes, uint32_t* out_start, uint64_t max_pixels,
                        uint32_t* out_frag_counter) {
    if (!ctx) return LV_E_INVALID;
    if (!ctx->ppllNodes.ptr || !ctx->ppllStart.ptr || ctx->lastMode != LV_RENDERING_MODE_PER_PIXEL_LINKED_LIST)
        return lv_fail(ctx, LV_E_STATE, "no PPLL buffers (render mode 2 first)");
    (void)hipSetDevice(ctx->device);
    LV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    LvDevCountersHost hc;
    LV_HIP(ctx, hipMemcpy(&hc, ctx->counters.ptr, sizeof(hc), hipMemcpyDeviceToHost));
    if (out_frag_counter) *out_frag_counter = hc.fragCounter;
    if (ctx->ppllArrays) {
        // raster_prism frames keep every pixel's fragments as ONE contiguous run of 8-B {colour, depth} entries (offsets = exclusive
        // scan of the per-pixel counts) instead of a linked list, in no defined order; the caller gets the reference's buffers
        // (LinkedListHeader.glsl:36-85): node index = position in the fragment array, `next` chains a pixel's live entries in
        // ascending (depth, colour) key order -- the order the resolve pass gives them, so that a literal walk of these lists
        // (first ppllMaxNumFrags nodes) sees what the resolve pass kept.  Dead entries (discarded fragments) come out as unlinked nodes.
        const uint64_t np = uint64_t(ctx->ppllPaddedW) * ctx->ppllPaddedH;
        std::vector<uint32_t> cnt(np), off(np);
        if (np) {
            LV_HIP(ctx, hipMemcpy(cnt.data(), ctx->ppllCount.ptr, size_t(np) * 4, hipMemcpyDeviceToHost));
            LV_HIP(ctx, hipMemcpy(off.data(), ctx->ppllStart.ptr, size_t(np) * 4, hipMemcpyDeviceToHost));
            // run offsets are relative to the pixel's scan block (k_ppll_scan): add the blocks' bases
            std::vector<uint32_t> base(ctx->ppllScanBlocks);
            if (!base.empty())
                LV_HIP(ctx, hipMemcpy(base.data(), (const uint32_t*)ctx->scanTemp.ptr + ctx->ppllScanBlocks, base.size() * 4, hipMemcpyDeviceToHost));
            const uint64_t itemsPerBlock = LV_SCAN_ITEMS;
            for (uint64_t p = 0; p < np; p++)
                if (p / itemsPerBlock < base.size()) off[p] += base[p / itemsPerBlock];
        }
        const uint64_t total = np ? uint64_t(off[np - 1]) + (cnt[np - 1] & 0xFFFFu) : 0u;
        if (out_nodes && max_nodes < total) return lv_fail(ctx, LV_E_CAPACITY, "out_nodes holds %llu nodes, %llu stored",
                                                           (unsigned long long)max_nodes, (unsigned long long)total);
        if (out_start && max_pixels < np) return lv_fail(ctx, LV_E_CAPACITY, "out_start_offset holds %llu entries, need %llu",
                                                         (unsigned long long)max_pixels, (unsigned long long)np);
        std::vector<uint32_t> fr(size_t(total) * 2);
        if (total) LV_HIP(ctx, hipMemcpy(fr.data(), ctx->ppllNodes.ptr, size_t(total) * 8, hipMemcpyDeviceToHost));
        std::vector<uint32_t> live;
        for (uint64_t p = 0; p < np; p++) {
            const uint32_t n = cnt[p] & 0xFFFFu;
            live.clear();
            for (uint32_t j = 0; j < n; j++) {
                const uint64_t idx = uint64_t(off[p]) + j;
                const uint32_t c = fr[2 * idx], d = fr[2 * idx + 1];
                const bool dead = c == 0u && d == LV_PPLL_DEAD;
                if (out_nodes) { out_nodes[3 * idx] = c; out_nodes[3 * idx + 1] = d; out_nodes[3 * idx + 2] = 0xFFFFFFFFu; }
                if (!dead) live.push_back(uint32_t(idx));
            }
            std::sort(live.begin(), live.end(), [&](uint32_t a, uint32_t b) {
                float da, db;
                std::memcpy(&da, &fr[2 * size_t(a) + 1], 4);
                std::memcpy(&db, &fr[2 * size_t(b) + 1], 4);
                if (da != db) return da < db;
                if (fr[2 * size_t(a)] != fr[2 * size_t(b)]) return fr[2 * size_t(a)] < fr[2 * size_t(b)];
                return a < b;
            });
            if (out_nodes)
                for (size_t k = 0; k + 1 < live.size(); k++) out_nodes[3 * size_t(live[k]) + 2] = live[k + 1];
            if (out_start) out_start[p] = live.empty() ? 0xFFFFFFFFu : live[0];
        }
        return LV_OK;
    }
    // node slots are handed out to waves in chunks (k_ppll_gather): copy up to the allocator's high-water mark
    uint64_t stored = hc.fragAlloc < ctx->ppllPoolNodes ? hc.fragAlloc : ctx->ppllPoolNodes;
    if (out_nodes) {
        if (max_nodes < stored) return lv_fail(ctx, LV_E_CAPACITY, "out_nodes holds %llu nodes, %llu stored",
                                               (unsigned long long)max_nodes, (unsigned long long)stored);
        if (stored) LV_HIP(ctx, hipMemcpy(out_nodes, ctx->ppllNodes.ptr, size_t(stored) * 12, hipMemcpyDeviceToHost));
    }
    const uint64_t np = uint64_t(ctx->ppllPaddedW) * ctx->ppllPaddedH; // extents of the last gather, not the current camera's
    if (out_start) {
        if (max_pixels < np) return lv_fail(ctx, LV_E_CAPACITY, "out_start_offset holds %llu entries, need %llu",
                                            (unsigned long long)max_pixels, (unsigned long long)np);
        LV_HIP(ctx, hipMemcpy(out_start, ctx->ppllStart.ptr, size_t(np) * 4, hipMemcpyDeviceToHost));
    }
    return LV_OK;
}

namespace {
// every float argument once: grid-stride over the 2^32 bit patterns
__global__ __launch_bounds__(256) void k_selftest_rsqrt(unsigned long long* __restrict__ result) {
    unsigned long long bad = 0ull;
    uint32_t firstBad = 0u;
    for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < (1ull << 32); i += gridDim.x * 256ull) {
        const float x = __uint_as_float(uint32_t(i));
        // (the argument must not be a compile-time constant of either side; both are plain functions of a loaded bit pattern)
        const float a = lv_rsqrt_shade(x);
        float xs = x;
        asm volatile("" : "+v"(xs));   // keeps the two evaluations apart (no common subexpressions)
        const float b = lv_rsqrt_shade_reference(xs);
        const bool same = __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);
        if (!same) { if (!bad) firstBad = uint32_t(i); bad++; }
    }
    if (bad) { atomicAdd(&result[0], bad); result[1] = firstBad; }
}
}   // namespace

int lv_selftest_rsqrt(lv_ctx* ctx, uint64_t* out_mismatches, uint32_t* out_first_argument) {
    if (!ctx || !out_mismatches) return LV_E_INVALID;
    (void)hipSetDevice(ctx->device);
    int rc;
    if ((rc = lv_buf_reserve(ctx, ctx->counters, sizeof(LvDevCounters)))) return rc;   // scratch the context owns (nothing to leak on an error path)
    unsigned long long* dev = (unsigned long long*)ctx->counters.ptr;
    LV_HIP(ctx, hipMemsetAsync(dev, 0, 16, ctx->stream));
    k_selftest_rsqrt<<<uint32_t(ctx->numCUs) * 16u, 256, 0, ctx->stream>>>(dev);
    unsigned long long host[2] = {0ull, 0ull};
    LV_HIP(ctx, hipMemcpyAsync(host, dev, 16, hipMemcpyDeviceToHost, ctx->stream));
    LV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out_mismatches = host[0];
    if (out_first_argument) *out_first_argument = uint32_t(host[1]);
    return LV_OK;
}

int lv_get_mlat_trace(lv_ctx* ctx, uint32_t* out_records, uint64_t max_records, uint64_t* out_count) {
    if (!ctx) return LV_E_INVALID;
    if (!ctx->mlatTrace.ptr || !ctx->counters.ptr || ctx->lastMode != LV_RENDERING_MODE_VULKAN_RAY_TRACER ||
        !ctx->opt.useMlat || !ctx->opt.collectStats || !ctx->opt.mlatRecordTrace)
        return lv_fail(ctx, LV_E_STATE, "no MLAT trace (render mode 11 with use_mlat, collect_stats and mlat_record_trace first)");
    (void)hipSetDevice(ctx->device);
    LV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    LvDevCountersHost hc;
    LV_HIP(ctx, hipMemcpy(&hc, ctx->counters.ptr, sizeof(hc), hipMemcpyDeviceToHost));
    if (out_count) *out_count = hc.mlatTraceCount;
    if (hc.mlatTraceCount > ctx->opt.mlatTraceCapacity)
        return lv_fail(ctx, LV_E_CAPACITY, "the frame produced %u trace records, mlat_trace_capacity is %u", hc.mlatTraceCount,
                       ctx->opt.mlatTraceCapacity);
    if (out_records) {
        if (max_records < hc.mlatTraceCount)
            return lv_fail(ctx, LV_E_CAPACITY, "out_records holds %llu records, %u stored", (unsigned long long)max_records,
                           hc.mlatTraceCount);
        if (hc.mlatTraceCount)
            LV_HIP(ctx, hipMemcpy(out_records, ctx->mlatTrace.ptr, size_t(hc.mlatTraceCount) * 16, hipMemcpyDeviceToHost));
    }
    return LV_OK;
}

int lv_ppll_resolve_buffers(lv_ctx* ctx, const uint32_t* nodes, uint64_t num_nodes, const uint32_t* start_offset,
                            uint64_t num_pixels, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint8_t* out) {
    if (!ctx) return LV_E_INVALID;
    if ((num_nodes && !nodes) || !start_offset || !out || w == 0 || h == 0) return lv_fail(ctx, LV_E_INVALID, "null array");
    (void)hipSetDevice(ctx->device);
    return lv_frame_ppll_resolve_only(ctx, nodes, num_nodes, start_offset, num_pixels, x0, y0, w, h, out);
}

int lv_get_accel(lv_ctx* ctx, void* out_nodes, uint64_t max_nodes, uint32_t* out_leaf_segment, uint64_t max_leaves) {
    if (!ctx) return LV_E_INVALID;
    if (!ctx->accelValid) return lv_fail(ctx, LV_E_STATE, "no acceleration structure (lv_build_accel first)");
    (void)hipSetDevice(ctx->device);
    LV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (out_nodes) {
        if (max_nodes < ctx->numNodes) return lv_fail(ctx, LV_E_CAPACITY, "out_nodes too small");
        if (ctx->numNodes) LV_HIP(ctx, hipMemcpy(out_nodes, ctx->nodes.ptr, size_t(ctx->numNodes) * 64, hipMemcpyDeviceToHost));
    }
    if (out_leaf_segment) {
        if (max_leaves < ctx->numSegs) return lv_fail(ctx, LV_E_CAPACITY, "out_leaf_segment too small");
        if (ctx->numSegs) LV_HIP(ctx, hipMemcpy(out_leaf_segment, ctx->leafSeg.ptr, size_t(ctx->numSegs) * 4, hipMemcpyDeviceToHost));
    }
    return LV_OK;
}

} // extern "C"
