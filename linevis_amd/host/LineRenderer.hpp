// LineRenderer.hpp -- headless renderer plugins over the C-ABI (include/linevis_hip.h).
//
// Mirrors the reference's plugin surface without Vulkan / ImGui types:
//   class LineRenderer                         src/Renderers/LineRenderer.hpp:66-277
//   struct SceneData                           src/Renderers/SceneData.hpp:49-85
//   enum RenderingMode                         src/Renderers/RenderingModes.hpp:32-53
//   class AmbientOcclusionBaker (+ type enum)  src/Renderers/AmbientOcclusion/AmbientOcclusionBaker.hpp:78-141
//   class VulkanRayTracedAmbientOcclusion      src/Renderers/AmbientOcclusion/VulkanRayTracedAmbientOcclusion.hpp:61-110
//   class VulkanRayTracer                      src/Renderers/RayTracing/VulkanRayTracer.hpp:73-145
//   class PerPixelLinkedListLineRenderer       src/Renderers/OIT/PerPixelLinkedListLineRenderer.hpp
// A renderer owns one lv_ctx (one HIP device).  render() writes RGBA8 into SceneData::sceneTexture.
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "../../include/linevis_hip.h"
#include "InternalState.hpp"
#include "LineData.hpp"
#include "SettingsMap.hpp"

namespace lv {

enum RenderingMode : int32_t {
    RENDERING_MODE_NONE = -1,
    RENDERING_MODE_PER_PIXEL_LINKED_LIST = 2,
    RENDERING_MODE_VULKAN_RAY_TRACER = 11,
};

/// Stand-in for sgl::Camera (not in the reference tree); conventions: LvMath.hpp / linevis_amd/camera.py.
class Camera {
public:
    void setPosition(vec3 p) { position = p; }
    void setLookAtLocation(vec3 c) { lookAtLocation = c; }
    void setFOVy(float f) { fovy = f; }
    void setNearClipDistance(float n) { nearDist = n; }
    void setFarClipDistance(float f) { farDist = f; }
    void setAspectRatio(float a) { aspect = a; }
    vec3 getPosition() const { return position; }
    float getFOVy() const { return fovy; }
    float getNearClipDistance() const { return nearDist; }
    float getFarClipDistance() const { return farDist; }
    mat4 getViewMatrix() const { return overwritten ? viewOverride : lookAt(position, lookAtLocation, vec3(0.0f, 1.0f, 0.0f)); }
    mat4 getProjectionMatrix() const { return overwritten ? projOverride : perspectiveVulkan(fovy, aspect, nearDist, farDist); }
    /// LineRenderer::overwriteCameraMatrices (LineRenderer.cpp:814-819)
    void overwriteMatrices(const float view[16], const float proj[16]);

private:
    vec3 position = vec3(0.0f, 0.0f, 0.8f); // test/VolumetricPathTracingTestRenderer.cpp:34-41
    vec3 lookAtLocation = vec3(0.0f, 0.0f, 0.0f);
    float fovy = 0.9272952180016122f;       // 2 * atan(1/2)
    float nearDist = 0.01f, farDist = 100.0f, aspect = 1.0f;
    bool overwritten = false;
    mat4 viewOverride, projOverride;
};
typedef std::shared_ptr<Camera> CameraPtr;

/// Stand-in for sgl::TransferFunctionWindow: the RGBA float table the renderers sample.
class TransferFunctionWindow {
public:
    void setTable(const float* rgba, uint32_t n) { table.assign(rgba, rgba + 4 * size_t(n)); dirty = true; }
    const std::vector<float>& getTable() const { return table; }
    bool getIsDirty() const { return dirty; }
    void resetDirty() { dirty = false; }

private:
    std::vector<float> table;
    bool dirty = true;
};

struct Color { float r = 1.0f, g = 1.0f, b = 1.0f, a = 1.0f; };

struct SceneData {
    uint32_t* viewportWidth = nullptr;
    uint32_t* viewportHeight = nullptr;
    CameraPtr camera;
    Color* clearColor = nullptr;
    std::vector<uint8_t>* sceneTexture = nullptr; ///< RGBA8, viewportWidth * viewportHeight * 4, row 0 = top
    int deviceOrdinal = 0;
    /// Several GPUs behind one renderer (lv_create_multi): when not empty the plugin drives one context per listed device, the
    /// frame is sharded by screen tiles and gathered on the first device (transport "rccl" or "memcpy").  The reference is
    /// single-GPU (sgl::AppSettings::getPrimaryDevice()); this is north_star's multi-GPU requirement behind the plugin surface.
    std::vector<int> deviceOrdinals;
    std::string multiGpuTransport = "rccl";
};

// ---------------------------------------------------------------- ambient occlusion
enum class AmbientOcclusionBakerType { NONE = -1, RTAO_PREBAKER = 0, RTAO = 1, SSAO = 2, GTAO = 3 };
extern const char* const AMBIENT_OCCLUSION_BAKER_TYPE_NAMES[4];

/// AmbientOcclusionBaker.hpp:66-76
enum class BakingMode { IMMEDIATE, ITERATIVE_UPDATE, MULTI_THREADED };
const char* const BAKING_MODE_NAMES[] = {"Immediate", "Iterative", "Multi-Threaded"};

class AmbientOcclusionBaker {
public:
    virtual ~AmbientOcclusionBaker() = default;
    virtual AmbientOcclusionBakerType getType() = 0;
    virtual bool getIsStaticPrebaker() = 0;
    virtual void startAmbientOcclusionBaking(LineDataPtr& lineData, bool isNewData) = 0;
    virtual bool getIsDataReady() = 0;
    virtual bool getHasComputationFinished() = 0;
    virtual void onHasMoved() {}
    virtual void onResolutionChanged() {}
    virtual bool setNewSettings(const SettingsMap& settings) { (void)settings; return false; }
};
typedef std::shared_ptr<AmbientOcclusionBaker> AmbientOcclusionBakerPtr;

/// "RTAO (Screen Space)".  The iterations run inside lv_render (k_ao_primary / k_ao_rays); this class holds the
/// settings of VulkanRayTracedAmbientOcclusionPass (hpp:150-153,108) and forwards them to the context.
class HipRayTracedAmbientOcclusion : public AmbientOcclusionBaker {
public:
    explicit HipRayTracedAmbientOcclusion(lv_ctx* ctx) : ctx(ctx) {}
    AmbientOcclusionBakerType getType() override { return AmbientOcclusionBakerType::RTAO; }
    bool getIsStaticPrebaker() override { return false; }
    void startAmbientOcclusionBaking(LineDataPtr& lineData, bool isNewData) override;
    bool getIsDataReady() override { return isDataReady; }
    bool getHasComputationFinished() override { return hasComputationFinished; }
    void onHasMoved() override { isDataReady = false; hasComputationFinished = false; }
    void onResolutionChanged() override { onHasMoved(); }
    bool setNewSettings(const SettingsMap& settings) override;
    void pushSettings();
    void notifyRendered() { isDataReady = true; hasComputationFinished = true; }

    int maxNumAccumulatedFrames = 64;            // VulkanRayTracedAmbientOcclusion.hpp:108
    int numAmbientOcclusionSamplesPerFrame = 4;  // :150
    float ambientOcclusionRadius = 0.1f;         // :151
    bool useDistance = true;                     // :152
    bool useJitteredPrimaryRays = true;          // :153
    bool useTriangleTubes = true;                // build-owned key "rtao_geometry": triangle_tubes (the reference's RTAO geometry) | capsules

private:
    lv_ctx* ctx;
    bool isDataReady = false, hasComputationFinished = false;
};

/// "RTAO (Prebaker)": view-independent AO baked once per geometry (VulkanAmbientOcclusionBaker.{hpp,cpp}).  The host part
/// is the blending-weight parametrisation of the polylines; baking and the render-time lookup run in the library.
class HipAmbientOcclusionBaker : public AmbientOcclusionBaker {
public:
    explicit HipAmbientOcclusionBaker(lv_ctx* ctx) : ctx(ctx) {}
    AmbientOcclusionBakerType getType() override { return AmbientOcclusionBakerType::RTAO_PREBAKER; }
    bool getIsStaticPrebaker() override { return true; }
    void startAmbientOcclusionBaking(LineDataPtr& lineData, bool isNewData) override;
    /// BakingMode::IMMEDIATE (default here): the table is baked on the context's stream by the first frame that needs it.
    /// BakingMode::MULTI_THREADED (AmbientOcclusionBaker.hpp:66-73; the reference's worker thread, VulkanAmbientOcclusionBaker.cpp:
    /// 266-346): lv_bake_ao_start on a second HIP stream -- frames rendered meanwhile show no AO, getIsDataReady() turns true (and
    /// needsReRender() once) when the table is in place.  Settings key (build-owned): rtao_prebaker_baking_mode.
    BakingMode getBakingMode() const { return bakingMode; }
    bool getIsDataReady() override;
    bool getIsComputationRunning();
    bool getHasComputationFinished() override { return getIsDataReady(); }
    /// true once after an asynchronous bake has finished (the frame has to be rendered again with the AO)
    bool getHasThreadUpdate();
    bool setNewSettings(const SettingsMap& settings) override;
    void pushSettings();
    /// uploads the parametrisation when it is out of date (new data / new expectedParamSegmentLength); MULTI_THREADED: starts the bake
    bool uploadParametrization(LineDataPtr& lineData);
    bool startAsyncBakeIfNeeded();
    /// the mesh / line width changed: a MULTI_THREADED bake has to start again
    void notifyInputsChanged() { bakeSettingsDirty = true; if (bakingMode == BakingMode::MULTI_THREADED) isDataReady = false; }
    uint32_t getNumParametrizationVertices() const { return numParametrizationVertices; }

    int maxNumIterations = 128;                       // VulkanAmbientOcclusionBaker.hpp:108
    float expectedParamSegmentLength = 0.001f;        // :163
    uint32_t numTubeSubdivisions = 8;                 // :165
    uint32_t numAmbientOcclusionSamplesPerFrame = 4;  // :166
    float ambientOcclusionRadius = 0.1f;              // :167
    bool useDistance = true;                          // :168

private:
    lv_ctx* ctx;
    BakingMode bakingMode = BakingMode::IMMEDIATE;
    bool isDataReady = false, parametrizationDirty = true, bakeStarted = false, threadUpdatePending = false, bakeSettingsDirty = false;
    uint32_t numParametrizationVertices = 0;
};

/// AmbientOcclusionComputeRenderPass::generateBlendingWeightParametrization + recomputeStaticParametrization
/// (VulkanAmbientOcclusionBaker.cpp:513-653): per line vertex a fractional index into the parametrisation, per
/// parametrisation vertex a fractional line-vertex position.  Lines are independent: lengths and counts first, a prefix
/// sum over lines, then every line writes its own ranges (OpenMP).
void computeAmbientOcclusionParametrization(const std::vector<std::vector<vec3>>& lines, float expectedParamSegmentLength,
                                            std::vector<float>& blendingWeightParametrizationData,
                                            std::vector<float>& samplingLocations);

// ---------------------------------------------------------------- renderers
class LineRenderer {
public:
    LineRenderer(std::string windowName, SceneData* sceneData, TransferFunctionWindow& transferFunctionWindow);
    virtual void initialize();
    virtual ~LineRenderer();
    virtual RenderingMode getRenderingMode() const = 0;
    virtual bool getIsTransparencyUsed() { return true; }
    bool isDirty() const { return dirty; }
    virtual bool needsReRender();
    virtual bool getIsTriangleRepresentationUsed() const { return false; }
    /// the ray tracer's "Elliptic Tubes" switch for band data (VulkanRayTracer.hpp:127)
    virtual bool getUseAnalyticEllipticTubes() const { return false; }
    bool getIsRasterizer() const { return isRasterizer; }

    virtual void setLineData(LineDataPtr& lineData, bool isNewData) = 0;
    virtual void renderBase();
    virtual void render() = 0;
    virtual void onResolutionChanged();
    virtual void onClearColorChanged() {}
    virtual void onHasMoved();
    virtual void notifyReRenderTriggeredExternally() { internalReRender = false; }
    /// LineRenderer.hpp:162 -- called by setNewState of the application before setNewSettings(newState.rendererSettings)
    virtual void setNewState(const InternalState& newState) { (void)newState; }
    virtual bool setNewSettings(const SettingsMap& settings);
    /// LineRenderer::setNewTilingMode (LineRenderer.cpp:739-812): the pixel addressing of the per-pixel lists
    static void setNewTilingMode(int newTileWidth, int newTileHeight, bool useMortonCode = false);
    static int getTilingWidth() { return tileWidth; }
    static int getTilingHeight() { return tileHeight; }
    /// multi-GPU renderer: re-deal the screen tiles by the cost measured in the last frame (lv_multi_rebalance)
    bool rebalanceTiles(double baseCostPerTile = 4.0 * 64 * 64);
    int getNumDevices() const;
    virtual void onTransferFunctionMapRebuilt() { tfDirty = true; reRender = true; }

    SceneData* getSceneData() { return sceneData; }
    const std::string& getWindowName() const { return windowName; }
    lv_ctx* getContext() { return ctx; }
    /// last error reported by the C-ABI (empty when none)
    const std::string& getLastError() const { return lastError; }
    lv_stats getStatistics();

    static float getLineWidth() { return lineWidth; }
    static float getBandWidth() { return bandWidth; }
    static void setLineWidth(float w) { lineWidth = w; }
    static void setBandWidth(float w) { bandWidth = w; }
    AmbientOcclusionBaker* getAmbientOcclusionBaker() { return ambientOcclusionBaker.get(); }

protected:
    void updateNewLineData(LineDataPtr& lineData, bool isNewData);
    void setAmbientOcclusionBaker();
    bool check(int rc, const char* what);
    bool uploadFrameState();
    bool setOption(const char* key, const std::string& value);
    bool renderMode(int mode);

    std::string windowName;
    SceneData* sceneData;
    TransferFunctionWindow& transferFunctionWindow;
    LineDataPtr lineData;
    lv_ctx* ctx = nullptr;
    std::string lastError;
    bool isRasterizer = false;
    bool dirty = true, reRender = true, internalReRender = false;
    bool tfDirty = true, linesDirty = true, triangleMeshDirty = true;
    // device geometry (lv_set_trajectories): on by default for plain flow lines; use_device_geometry = false keeps the host-built
    // render data (build-owned key).  uploadedTrajectory*: which data set's arrays sit in HBM
    bool useDeviceGeometry = true, deviceGeometry = false;
    const LineData* uploadedTrajectoryData = nullptr;
    uint64_t uploadedTrajectoryGeneration = 0;
    // twist-line texture last handed to the library (dimensions + FNV-1a of the pixels): lv_set_twist_line_texture rebuilds a mip chain
    // and synchronises, so it is called only when the pixels changed, not with every dirty line setting
    uint32_t uploadedTwistW = 0xFFFFFFFFu, uploadedTwistH = 0xFFFFFFFFu;
    uint64_t uploadedTwistHash = 0;

    // LineRenderer.hpp:220-231 (depth cues default on with strength 0.8 in the GUI application; the headless
    // default is off until "depth_cue_strength" is set, like a fresh SettingsMap-driven benchmark state)
    bool useDepthCues = false;
    float depthCueStrength = 0.0f;
    bool useAmbientOcclusion = false;
    float ambientOcclusionStrength = 0.0f;
    float ambientOcclusionGamma = 1.0f;
    AmbientOcclusionBakerType ambientOcclusionBakerType = AmbientOcclusionBakerType::NONE;
    AmbientOcclusionBakerPtr ambientOcclusionBaker;

    static float lineWidth; // LineRenderer.hpp:266-269 (static: shared by all renderers, as in the reference)
    static float bandWidth;
    static int tileWidth, tileHeight; // LineRenderer.cpp:739-740 (2 x 8)
    static bool tilingModeDirty;
};

/// "Vulkan Ray Tracer" plugin re-hosted on HIP: the reference's three geometry modes (analytic AABBs -- the default here, the
/// reference starts in "Triangle Mesh" --, triangle mesh, linear swept spheres), elliptic tubes for band data, MLAT.
class HipRayTracer : public LineRenderer {
public:
    HipRayTracer(SceneData* sceneData, TransferFunctionWindow& transferFunctionWindow);
    RenderingMode getRenderingMode() const override { return RENDERING_MODE_VULKAN_RAY_TRACER; }
    bool getIsTransparencyUsed() override { return false; }
    bool getIsTriangleRepresentationUsed() const override { return useTriangleMesh; } // VulkanRayTracer.hpp: geometry mode
    bool getUseAnalyticEllipticTubes() const override { return useAnalyticEllipticTubes; }
    bool needsReRender() override;
    void setLineData(LineDataPtr& lineData, bool isNewData) override;
    void render() override;
    void onHasMoved() override;
    /// VulkanRayTracer::setNewState (VulkanRayTracer.cpp:280-330): camelCase keys of the canned benchmark states
    void setNewState(const InternalState& newState) override;
    bool setNewSettings(const SettingsMap& settings) override;

private:
    uint32_t numSamplesPerFrame = 1;       // VulkanRayTracer.hpp:137 (2 in the interactive application)
    uint32_t maxDepthComplexity = 1024;    // :139
    uint32_t maxNumAccumulatedFrames = 1;  // :142 (32 interactive; offline frames use spp instead)
    uint32_t accumulatedFramesCounter = 0; // :143
    bool useDeterministicSampling = false;
    bool useMlat = false;                  // multi-layer alpha tracing, VulkanRayTracer.hpp:133-134
    int mlatNumNodes = 8;
    bool useTriangleMesh = false;          // RayTracingGeometryMode::TRIANGLE_MESH (VulkanRayTracer.hpp:52-63)
    bool useAnalyticEllipticTubes = false; // VulkanRayTracer.hpp:127
};

/// "Per-Pixel Linked Lists" plugin re-hosted on HIP.
class HipPerPixelLinkedListLineRenderer : public LineRenderer {
public:
    HipPerPixelLinkedListLineRenderer(SceneData* sceneData, TransferFunctionWindow& transferFunctionWindow);
    RenderingMode getRenderingMode() const override { return RENDERING_MODE_PER_PIXEL_LINKED_LIST; }
    void setLineData(LineDataPtr& lineData, bool isNewData) override;
    void render() override;
    /// PerPixelLinkedListLineRenderer::setNewState (.cpp:98-107): a new state starts new per-phase timers
    void setNewState(const InternalState& newState) override;
    const std::string& getCurrentStateName() const { return currentStateName; }
    bool setNewSettings(const SettingsMap& settings) override;
    /// band data: the rasterisers always draw the elliptic tubes of the ribbon primitive mode (LineDataFlow.cpp:476-481)
    bool getUseAnalyticEllipticTubes() const override { return true; }
    /// computeStatistics (PerPixelLinkedListLineRenderer.cpp:578-663): fragments, max depth complexity
    void computeStatistics(uint64_t& totalNumFragments, uint32_t& maxComplexity);

private:
    std::string currentStateName;
};

} // namespace lv
