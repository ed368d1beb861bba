// LineRenderer.cpp -- see LineRenderer.hpp for the reference classes mirrored here.
#include "LineRenderer.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace lv {

const char* const AMBIENT_OCCLUSION_BAKER_TYPE_NAMES[4] = {"RTAO (Prebaker)", "RTAO (Screen Space)", "SSAO",
                                                           "GTAO (reference)"};

float LineRenderer::lineWidth = 0.002f; // src/Loaders/DataSetList.hpp:46
float LineRenderer::bandWidth = 0.005f;

void Camera::overwriteMatrices(const float view[16], const float proj[16]) {
    overwritten = true;
    memcpy(viewOverride.m, view, 64);
    memcpy(projOverride.m, proj, 64);
}

// ---------------------------------------------------------------- RTAO settings holder
void HipRayTracedAmbientOcclusion::startAmbientOcclusionBaking(LineDataPtr&, bool) {
    isDataReady = false;
    hasComputationFinished = false;
}

// VulkanRayTracedAmbientOcclusion::setNewSettings, VulkanRayTracedAmbientOcclusion.cpp:115-144
bool HipRayTracedAmbientOcclusion::setNewSettings(const SettingsMap& settings) {
    bool optionChanged = false;
    if (settings.getValueOpt("ambient_occlusion_iterations", maxNumAccumulatedFrames)) optionChanged = true;
    if (settings.getValueOpt("ambient_occlusion_samples_per_frame", numAmbientOcclusionSamplesPerFrame)) optionChanged = true;
    if (settings.getValueOpt("ambient_occlusion_radius", ambientOcclusionRadius)) optionChanged = true;
    if (settings.getValueOpt("ambient_occlusion_distance_based", useDistance)) optionChanged = true;
    if (settings.getValueOpt("use_jittered_primary_rays", useJitteredPrimaryRays)) optionChanged = true;
    std::string geometry;
    if (settings.getValueOpt("rtao_geometry", geometry)) {
        const bool tri = geometry == "triangle_tubes";
        useTriangleTubes = tri;
        optionChanged = true;
    }
    // the denoiser of the RTAO pass (VulkanRayTracedAmbientOcclusion::setNewSettings, .cpp:115-144,683-696) and its parameters
    // (EAWDenoiser.cpp:400-432; the svgf_* keys are this build's): forwarded as they are, a change restarts the accumulation
    for (const auto& kv : settings.getMap()) {
        const std::string& k = kv.first;
        if (k == "ambient_occlusion_denoiser" || k.rfind("eaw_denoiser_", 0) == 0 || k.rfind("svgf_denoiser_", 0) == 0) {
            lv_set_option(ctx, k.c_str(), kv.second.c_str());
            optionChanged = true;
        }
    }
    if (optionChanged) {
        onHasMoved();
        pushSettings();
    }
    return optionChanged;
}

void HipRayTracedAmbientOcclusion::pushSettings() {
    SettingsMap m;
    m.addKeyValue("ambient_occlusion_iterations", maxNumAccumulatedFrames);
    m.addKeyValue("ambient_occlusion_samples_per_frame", numAmbientOcclusionSamplesPerFrame);
    m.addKeyValue("ambient_occlusion_radius", ambientOcclusionRadius);
    m.addKeyValue("ambient_occlusion_distance_based", useDistance);
    m.addKeyValue("use_jittered_primary_rays", useJitteredPrimaryRays);
    m.addKeyValue("rtao_geometry", useTriangleTubes ? "triangle_tubes" : "capsules");
    for (const auto& kv : m.getMap()) lv_set_option(ctx, kv.first.c_str(), kv.second.c_str());
}

// ---------------------------------------------------------------- static RTAO prebaker
void computeAmbientOcclusionParametrization(const std::vector<std::vector<vec3>>& lines, float expectedParamSegmentLength,
                                            std::vector<float>& blendingWeights, std::vector<float>& samplingLocations) {
    const float EPSILON = 1e-5f;
    const size_t numLines = lines.size();
    std::vector<float> polylineLengths(numLines);
    std::vector<uint32_t> numSubdivs(numLines);
    std::vector<size_t> vertexOffset(numLines + 1, 0), paramOffset(numLines + 1, 0);
#pragma omp parallel for schedule(static)
    for (long li = 0; li < long(numLines); li++) {
        const std::vector<vec3>& line = lines[size_t(li)];
        float polylineLength = 0.0f;
        for (size_t i = 1; i < line.size(); i++) polylineLength += length(line[i] - line[i - 1]);
        polylineLengths[size_t(li)] = polylineLength;
        numSubdivs[size_t(li)] = std::max(1u, uint32_t(std::ceil(polylineLength / expectedParamSegmentLength)));
    }
    for (size_t li = 0; li < numLines; li++) {
        vertexOffset[li + 1] = vertexOffset[li] + lines[li].size();
        paramOffset[li + 1] = paramOffset[li] + numSubdivs[li] + 1;
    }
    blendingWeights.assign(vertexOffset[numLines], 0.0f);
    samplingLocations.assign(paramOffset[numLines], 0.0f);
    // Definition (VulkanAmbientOcclusionBaker.cpp:563-653).  A polyline of n vertices with float32 running arc lengths a_0 = 0,
    // a_k = a_(k-1) + |v_k - v_(k-1)| is cut into N = numSubdivs pieces of length h = L / N; `first` = index of its first
    // parametrisation vertex, `v0` = index of its first line vertex.
    //   blendingWeight(v_k)      = first + clamp(a_k / h, 0, N - eps)                         -- arc length in units of h
    //   samplingLocation(i), i>0 = v0 + min((j - 1) + (i h - a_(j-1)) / (a_j - a_(j-1)), (n - 1) - eps), where j is the first vertex
    //                              (1 <= j <= n - 1) whose cell index uint(a_j / h) reaches i, or n - 1 when none does
    // The cell indices are non-decreasing, so j is a lower bound found by bisection over a per-line prefix table; every quotient is
    // evaluated in float32 in the order written above, which is what makes the tables bit-identical to the reference's.
#pragma omp parallel
    {
        std::vector<float> arc;        // a_k
        std::vector<uint32_t> cell;    // uint(a_k / h)
#pragma omp for schedule(dynamic, 16)
        for (long lli = 0; lli < long(numLines); lli++) {
            const std::vector<vec3>& line = lines[size_t(lli)];
            const size_t n = line.size();
            if (n < 2) continue;
            const uint32_t N = numSubdivs[size_t(lli)];
            const float h = polylineLengths[size_t(lli)] / float(N);
            const float first = float(paramOffset[size_t(lli)]), v0 = float(uint32_t(vertexOffset[size_t(lli)]));
            arc.resize(n);
            cell.resize(n);
            arc[0] = 0.0f;
            cell[0] = 0u;
            for (size_t k = 1; k < n; k++) {
                arc[k] = arc[k - 1] + length(line[k] - line[k - 1]);
                cell[k] = uint32_t(arc[k] / h);
            }
            float* bw = blendingWeights.data() + vertexOffset[size_t(lli)];
            bw[0] = first;
            const float topWeight = float(N) - EPSILON;
            for (size_t k = 1; k < n; k++) bw[k] = first + std::fmin(std::fmax(arc[k] / h, 0.0f), topWeight);
            float* sl = samplingLocations.data() + paramOffset[size_t(lli)];
            sl[0] = v0;
            const float topLocation = float(uint32_t(n) - 1u) - EPSILON;
            for (uint32_t i = 1; i <= N; i++) {
                const size_t j = std::min(size_t(std::lower_bound(cell.begin() + 1, cell.end(), i) - cell.begin()), n - 1);
                const float along = (float(i) * h - arc[j - 1]) / (arc[j] - arc[j - 1]);
                sl[i] = v0 + std::min(float(j - 1) + along, topLocation);
            }
        }
    }
}

void HipAmbientOcclusionBaker::startAmbientOcclusionBaking(LineDataPtr&, bool) {
    isDataReady = false;
    parametrizationDirty = true;
    bakeSettingsDirty = true;
}

bool HipAmbientOcclusionBaker::uploadParametrization(LineDataPtr& lineData) {
    if (!parametrizationDirty) return true;
    std::vector<std::vector<vec3>> lines = lineData->getFilteredLines(nullptr);
    std::vector<float> blendingWeights, samplingLocations;
    computeAmbientOcclusionParametrization(lines, expectedParamSegmentLength, blendingWeights, samplingLocations);
    int rc = lv_set_ao_parametrization(ctx, blendingWeights.data(), uint32_t(blendingWeights.size()),
                                       samplingLocations.data(), uint32_t(samplingLocations.size()));
    if (rc != LV_OK) return false;
    numParametrizationVertices = uint32_t(samplingLocations.size());
    parametrizationDirty = false;
    if (bakingMode == BakingMode::MULTI_THREADED) {
        bakeSettingsDirty = true;   // started below, once mesh and parametrisation are both in place
    } else {
        isDataReady = true; // baking itself runs inside the next lv_render call
    }
    return true;
}

bool HipAmbientOcclusionBaker::getIsComputationRunning() {
    if (bakingMode != BakingMode::MULTI_THREADED || !bakeStarted) return false;
    int running = 0, ready = 0;
    if (lv_bake_ao_poll(ctx, &running, &ready) != LV_OK) return false;
    return running != 0;
}

bool HipAmbientOcclusionBaker::getIsDataReady() {
    if (bakingMode != BakingMode::MULTI_THREADED) return isDataReady;
    if (!bakeStarted) return false;
    int running = 0, ready = 0;
    if (lv_bake_ao_poll(ctx, &running, &ready) != LV_OK) return false;
    if (ready && !isDataReady) threadUpdatePending = true;
    isDataReady = ready != 0;
    return isDataReady;
}

bool HipAmbientOcclusionBaker::getHasThreadUpdate() {
    getIsDataReady();
    const bool u = threadUpdatePending;
    threadUpdatePending = false;
    return u;
}

// The reference exposes these only through its GUI (renderGuiPropertyEditorNodes, VulkanAmbientOcclusionBaker.cpp:412-470);
// here they are SettingsMap keys so that a headless run can set them.
bool HipAmbientOcclusionBaker::setNewSettings(const SettingsMap& settings) {
    bool changed = false;
    if (settings.getValueOpt("rtao_prebaker_iterations", maxNumIterations)) changed = true;
    if (settings.getValueOpt("rtao_prebaker_samples_per_frame", numAmbientOcclusionSamplesPerFrame)) changed = true;
    if (settings.getValueOpt("rtao_prebaker_num_tube_subdivisions", numTubeSubdivisions)) changed = true;
    if (settings.getValueOpt("ambient_occlusion_radius", ambientOcclusionRadius)) changed = true;
    if (settings.getValueOpt("ambient_occlusion_distance_based", useDistance)) changed = true;
    float len = expectedParamSegmentLength;
    if (settings.getValueOpt("rtao_prebaker_line_resolution", len) && len != expectedParamSegmentLength && len > 0.0f) {
        expectedParamSegmentLength = len;
        parametrizationDirty = true;
        changed = true;
    }
    std::string mode;
    if (settings.getValueOpt("rtao_prebaker_baking_mode", mode)) {
        const BakingMode m = mode == BAKING_MODE_NAMES[2] ? BakingMode::MULTI_THREADED : BakingMode::IMMEDIATE;
        if (m != bakingMode) { bakingMode = m; isDataReady = false; bakeStarted = false; bakeSettingsDirty = true; changed = true; }
    }
    if (changed) { pushSettings(); bakeSettingsDirty = true; isDataReady = bakingMode != BakingMode::MULTI_THREADED && !parametrizationDirty; }
    return changed;
}

void HipAmbientOcclusionBaker::pushSettings() {
    SettingsMap m;
    m.addKeyValue("rtao_prebaker_iterations", maxNumIterations);
    m.addKeyValue("rtao_prebaker_samples_per_frame", numAmbientOcclusionSamplesPerFrame);
    m.addKeyValue("rtao_prebaker_num_tube_subdivisions", numTubeSubdivisions);
    m.addKeyValue("ambient_occlusion_radius", ambientOcclusionRadius);
    m.addKeyValue("ambient_occlusion_distance_based", useDistance);
    for (const auto& kv : m.getMap()) lv_set_option(ctx, kv.first.c_str(), kv.second.c_str());
}

// ---------------------------------------------------------------- LineRenderer
LineRenderer::LineRenderer(std::string windowName, SceneData* sceneData, TransferFunctionWindow& tfw)
        : windowName(std::move(windowName)), sceneData(sceneData), transferFunctionWindow(tfw) {}

void LineRenderer::initialize() {
    int err = 0;
    if (sceneData && !sceneData->deviceOrdinals.empty()) {
        ctx = lv_create_multi(sceneData->deviceOrdinals.data(), int(sceneData->deviceOrdinals.size()),
                              sceneData->multiGpuTransport.c_str(), &err);
        if (!ctx) lastError = "lv_create_multi failed (a listed HIP device or the gather transport is not usable)";
        return;
    }
    ctx = lv_create(sceneData ? sceneData->deviceOrdinal : 0, &err);
    if (!ctx) lastError = "lv_create failed (no HIP device; there is no CPU fallback)";
}

int LineRenderer::tileWidth = 2;
int LineRenderer::tileHeight = 8;
bool LineRenderer::tilingModeDirty = false;

// LineRenderer::setNewTilingMode, LineRenderer.cpp:739-812 (the Morton variant of TiledAddress.glsl is not built: the flag is
// accepted and ignored, the row-major tile order is used)
void LineRenderer::setNewTilingMode(int newTileWidth, int newTileHeight, bool) {
    if (newTileWidth <= 0 || newTileHeight <= 0) return;
    if (newTileWidth != tileWidth || newTileHeight != tileHeight) tilingModeDirty = true;
    tileWidth = newTileWidth;
    tileHeight = newTileHeight;
}

bool LineRenderer::rebalanceTiles(double baseCostPerTile) {
    if (!ctx) return false;
    return check(lv_multi_rebalance(ctx, baseCostPerTile), "lv_multi_rebalance");
}

int LineRenderer::getNumDevices() const { return ctx ? lv_multi_ranks(ctx) : 0; }

LineRenderer::~LineRenderer() {
    if (ctx) lv_destroy(ctx);
}

bool LineRenderer::check(int rc, const char* what) {
    if (rc == LV_OK) return true;
    lastError = std::string(what) + ": " + (ctx ? lv_last_error(ctx) : "no context");
    return false;
}

bool LineRenderer::setOption(const char* key, const std::string& value) {
    if (!ctx) return false;
    return check(lv_set_option(ctx, key, value.c_str()), key);
}

bool LineRenderer::needsReRender() {
    bool tmp = reRender;
    reRender = false;
    // LineRenderer::renderBase, LineRenderer.cpp:265-269: a finished multi-threaded bake asks for one more frame (now with the AO)
    if (useAmbientOcclusion && ambientOcclusionBaker && ambientOcclusionBaker->getIsStaticPrebaker() &&
        static_cast<HipAmbientOcclusionBaker*>(ambientOcclusionBaker.get())->getHasThreadUpdate())
        tmp = true;
    return tmp;
}

void LineRenderer::onResolutionChanged() {
    if (ambientOcclusionBaker) ambientOcclusionBaker->onResolutionChanged();
    reRender = true;
}

void LineRenderer::onHasMoved() {
    if (ambientOcclusionBaker) ambientOcclusionBaker->onHasMoved();
    reRender = true;
}

// LineRenderer::setAmbientOcclusionBaker, LineRenderer.cpp:310-357 (only the screen-space RTAO baker exists here)
void LineRenderer::setAmbientOcclusionBaker() {
    ambientOcclusionBaker = {};
    if (ambientOcclusionBakerType == AmbientOcclusionBakerType::RTAO_PREBAKER && ctx) {
        auto baker = std::make_shared<HipAmbientOcclusionBaker>(ctx);
        baker->pushSettings();
        ambientOcclusionBaker = baker;
        setOption("ambient_occlusion_mode", AMBIENT_OCCLUSION_BAKER_TYPE_NAMES[0]);
    } else if (ambientOcclusionBakerType == AmbientOcclusionBakerType::RTAO && ctx) {
        auto baker = std::make_shared<HipRayTracedAmbientOcclusion>(ctx);
        baker->pushSettings();
        ambientOcclusionBaker = baker;
        setOption("ambient_occlusion_mode", AMBIENT_OCCLUSION_BAKER_TYPE_NAMES[1]);
    } else {
        if (ctx) setOption("ambient_occlusion_mode", "None");
    }
    reRender = true;
    if (useAmbientOcclusion && ambientOcclusionBaker && lineData) ambientOcclusionBaker->startAmbientOcclusionBaking(lineData, true);
}

// LineRenderer::setNewSettings, LineRenderer.cpp:433-498
bool LineRenderer::setNewSettings(const SettingsMap& settings) {
    bool shallReloadGatherShader = false;
    float newLineWidth = lineWidth;
    if (settings.getValueOpt("line_width", lineWidth)) {
        if (newLineWidth != lineWidth && lineData) {
            lineData->setTriangleRepresentationDirty();
            linesDirty = true;
        }
    }
    bool deviceGeometryWanted = useDeviceGeometry;   // build-owned key: false keeps the host-built render data (LineData.cpp / Tubes.cpp)
    if (settings.getValueOpt("use_device_geometry", deviceGeometryWanted) && deviceGeometryWanted != useDeviceGeometry) {
        useDeviceGeometry = deviceGeometryWanted;
        uploadedTrajectoryData = nullptr;
        linesDirty = true;
    }
    float newBandWidth = bandWidth;
    if (settings.getValueOpt("band_width", bandWidth)) { // LineRenderer.cpp:442-449
        if (newBandWidth != bandWidth && lineData) {
            lineData->setTriangleRepresentationDirty();
            linesDirty = true;
        }
    }

    if (settings.getValueOpt("depth_cue_strength", depthCueStrength)) {
        if (depthCueStrength <= 0.0f && useDepthCues) { useDepthCues = false; shallReloadGatherShader = true; }
        if (depthCueStrength > 0.0f && !useDepthCues) { useDepthCues = true; shallReloadGatherShader = true; }
        setOption("depth_cue_strength", settings.getValue("depth_cue_strength"));
    }

    std::string ambientOcclusionModeName;
    if (settings.getValueOpt("ambient_occlusion_mode", ambientOcclusionModeName)) {
        AmbientOcclusionBakerType newType = AmbientOcclusionBakerType::NONE;
        for (int i = 0; i < 4; i++)
            if (ambientOcclusionModeName == AMBIENT_OCCLUSION_BAKER_TYPE_NAMES[i]) newType = AmbientOcclusionBakerType(i);
        if (newType != AmbientOcclusionBakerType::NONE && newType != AmbientOcclusionBakerType::RTAO &&
            newType != AmbientOcclusionBakerType::RTAO_PREBAKER) {
            lastError = "ambient_occlusion_mode '" + ambientOcclusionModeName + "' is not provided by the HIP renderers";
        } else if (newType != ambientOcclusionBakerType || !ambientOcclusionBaker) {
            ambientOcclusionBakerType = newType;
            setAmbientOcclusionBaker();
        }
    }
    if (settings.getValueOpt("ambient_occlusion_strength", ambientOcclusionStrength)) {
        if (ambientOcclusionStrength <= 0.0f && useAmbientOcclusion) { useAmbientOcclusion = false; shallReloadGatherShader = true; }
        if (ambientOcclusionStrength > 0.0f && !useAmbientOcclusion) { useAmbientOcclusion = true; shallReloadGatherShader = true; }
        setOption("ambient_occlusion_strength", settings.getValue("ambient_occlusion_strength"));
    }
    if (settings.getValueOpt("ambient_occlusion_gamma", ambientOcclusionGamma))
        setOption("ambient_occlusion_gamma", settings.getValue("ambient_occlusion_gamma"));
    if (ambientOcclusionBaker) ambientOcclusionBaker->setNewSettings(settings);
    reRender = true;
    return shallReloadGatherShader;
}

// LineRenderer::updateNewLineData, LineRenderer.cpp:676-708
void LineRenderer::updateNewLineData(LineDataPtr& newLineData, bool isNewData) {
    lineData = newLineData;
    linesDirty = true;
    if (useAmbientOcclusion && ambientOcclusionBaker) ambientOcclusionBaker->startAmbientOcclusionBaking(lineData, isNewData);
    dirty = false;
    reRender = true;
}

// camera / clear colour / transfer function / geometry -> context (LineData::updateVulkanUniformBuffers,
// LineData.cpp:1275-1319, and the buffer fetches of RayTracingRenderPass::setLineData, VulkanRayTracer.cpp:370-411)
bool LineRenderer::uploadFrameState() {
    if (!ctx) { lastError = "renderer not initialised"; return false; }
    if (!lineData) { lastError = "no line data"; return false; }
    const uint32_t w = *sceneData->viewportWidth, h = *sceneData->viewportHeight;
    Camera& cam = *sceneData->camera;
    cam.setAspectRatio(float(w) / float(h));
    mat4 view = cam.getViewMatrix(), proj = cam.getProjectionMatrix();
    if (!check(lv_set_camera(ctx, view.m, proj.m, cam.getFOVy(), cam.getNearClipDistance(), cam.getFarClipDistance(), w, h),
               "lv_set_camera"))
        return false;
    const Color& c = *sceneData->clearColor;
    const float bg[4] = {c.r, c.g, c.b, c.a};
    if (!check(lv_set_background(ctx, bg), "lv_set_background")) return false;
    if (tfDirty || transferFunctionWindow.getIsDirty()) {
        const std::vector<float>& t = transferFunctionWindow.getTable();
        float mn = 0.0f, mx = 1.0f;
        lineData->getMinMaxAttributeValues(mn, mx);
        if (!check(lv_set_transfer_function(ctx, t.data(), uint32_t(t.size() / 4), mn, mx), "lv_set_transfer_function"))
            return false;
        transferFunctionWindow.resetDirty();
        tfDirty = false;
    }
    if (linesDirty || lineData->isDirty()) {
        char buf[64];
        snprintf(buf, sizeof(buf), "%.9g", double(lineWidth));
        if (!setOption("line_width", buf)) return false;
        // band data: USE_BANDS + the "Elliptic Tubes" geometry of the ray tracer (RayTracingRenderPass::setLineData,
        // VulkanRayTracer.cpp:370-381; LineDataFlow::getVulkanShaderPreprocessorDefines, LineDataFlow.cpp:2420-2431)
        // (for band data the reference switches to the ribbon primitive mode, LineDataFlow.cpp:476-481, and its rasterisers then draw
        // elliptic tubes with USE_BANDS; the ray-entry PPLL of this build gathers the entry hits of the analytic elliptic tubelets)
        const bool bands = lineData->getUseBands();
        const bool elliptic = bands && getUseAnalyticEllipticTubes();
        // plain flow lines: the trajectories themselves go to HBM (once per data set) and the device writes the line points, the index
        // pairs and -- whenever a frame needs them at another line width -- the triangle tubes; everything else: host-built render data
        deviceGeometry = false;
        if (useDeviceGeometry && !bands && !lineData->getUseRotatingHelicityBands()) {
            if (uploadedTrajectoryData == lineData.get() && uploadedTrajectoryGeneration == lineData->getDataGeneration()) {
                deviceGeometry = true;     // only the line width / the tessellation settings changed: nothing to upload
            } else {
                std::vector<float> positions, attribute;
                std::vector<uint32_t> lineOffsets;
                if (lineData->getTrajectoryArrays(positions, attribute, lineOffsets)) {
                    if (!check(lv_set_trajectories(ctx, positions.data(), attribute.data(), lineOffsets.data(), uint32_t(lineOffsets.size() - 1)),
                               "lv_set_trajectories"))
                        return false;
                    uploadedTrajectoryData = lineData.get();
                    uploadedTrajectoryGeneration = lineData->getDataGeneration();
                    deviceGeometry = true;
                }
            }
        }
        if (!deviceGeometry) {
            uploadedTrajectoryData = nullptr;
            TubeAabbRenderData d = lineData->getLinePassTubeAabbRenderData(false, elliptic);
            if (!check(lv_set_lines(ctx, d.linePointDataBuffer.data(), uint32_t(d.linePointDataBuffer.size()),
                                    d.indexBuffer.data(), uint32_t(d.indexBuffer.size() / 2)), "lv_set_lines"))
                return false;
        }
        setOption("use_ribbons", bands ? "true" : "false");
        setOption("use_analytic_elliptic_tubes", elliptic ? "true" : "false");
        setOption("thick_bands", LineData::getRenderThickBands() ? "true" : "false");
        snprintf(buf, sizeof(buf), "%.9g", double(LineData::getMinBandThickness()));
        setOption("min_band_thickness", buf);
        snprintf(buf, sizeof(buf), "%.9g", double(bandWidth));
        setOption("band_width", buf);
        // USE_ROTATING_HELICITY_BANDS + its LineUniformData members (LineDataFlow.cpp:979-984,2432-2440); the points carry lineRotation
        setOption("rotating_helicity_bands", lineData->getUseRotatingHelicityBands() ? "true" : "false");
        snprintf(buf, sizeof(buf), "%.9g", double(lineData->getSeparatorWidth()));
        setOption("separator_width", buf);
        setOption("band_subdivisions", std::to_string(lineData->getNumSubdivisionsBands()));
        setOption("use_uniform_twist_line_width", lineData->getUseUniformTwistLineWidth() ? "true" : "false");
        {   // USE_HELICITY_BANDS_TEXTURE (LineDataFlow.cpp:974-977,2437-2439)
            uint32_t tw = 0, th = 0;
            const std::vector<uint8_t>* tex = lineData->getTwistLineTexture(tw, th);
            uint64_t hash = 1469598103934665603ull;
            if (tex) for (uint8_t b : *tex) hash = (hash ^ b) * 1099511628211ull;
            if (tw != uploadedTwistW || th != uploadedTwistH || hash != uploadedTwistHash) {
                if (!check(lv_set_twist_line_texture(ctx, tex ? tex->data() : nullptr, tw, th), "lv_set_twist_line_texture")) return false;
                uploadedTwistW = tw; uploadedTwistH = th; uploadedTwistHash = hash;
            }
            setOption("use_twist_line_texture", lineData->getUseTwistLineTexture() ? "true" : "false");
            setOption("twist_line_texture_filtering_mode_index", std::to_string(lineData->getTwistLineTextureFilteringModeIndex()));
        }
        snprintf(buf, sizeof(buf), "%.9g", double(lineData->getHelicityRotationFactor()));
        setOption("helicity_rotation_factor", buf);
        // getVulkanShaderPreprocessorDefines, LineData.cpp:1209-1256
        setOption("use_capped_tubes", lineData->getUseCappedTubesDefine(isRasterizer) ? "true" : "false");
        setOption("use_halos", lineData->getUseHalos() ? "true" : "false");
        setOption("tube_num_subdivisions", std::to_string(lineData->getTubeNumSubdivisions()));
        lineData->setDirty(false);
        linesDirty = false;
        tfDirty = true; // attribute range may have changed with the data
        triangleMeshDirty = true;
        return uploadFrameState();
    }
    // Triangle tubes are fetched from the line data only by the consumers that trace them: the ray tracer in "Triangle
    // Mesh" geometry mode (RayTracingRenderPass::setLineData, VulkanRayTracer.cpp:370-411) and the RTAO pass when it
    // uses the reference's geometry (VulkanRayTracedAmbientOcclusionPass::setLineData, ...AmbientOcclusion.cpp:437-456).
    bool wantMesh = getIsTriangleRepresentationUsed();
    if (useAmbientOcclusion && ambientOcclusionBaker && ambientOcclusionBaker->getType() == AmbientOcclusionBakerType::RTAO)
        wantMesh = wantMesh || static_cast<HipRayTracedAmbientOcclusion*>(ambientOcclusionBaker.get())->useTriangleTubes;
    const bool prebaker = useAmbientOcclusion && ambientOcclusionBaker && ambientOcclusionBaker->getIsStaticPrebaker();
    wantMesh = wantMesh || prebaker; // the baker traces the triangle tubes (VulkanAmbientOcclusionBaker.cpp:480)
    if (wantMesh && triangleMeshDirty && deviceGeometry) {
        // lv_set_trajectories' lines: the library tessellates on the device when the frame asks for the mesh (lv_ensure_tube_mesh)
        triangleMeshDirty = false;
        if (prebaker) static_cast<HipAmbientOcclusionBaker*>(ambientOcclusionBaker.get())->notifyInputsChanged();
    }
    if (wantMesh && triangleMeshDirty) {
        TubeTriangleRenderData d = lineData->getLinePassTubeTriangleMeshRenderData(false, true);
        if (!check(lv_set_tube_triangle_mesh(ctx, d.indexBuffer.data(), uint32_t(d.indexBuffer.size() / 3),
                                             d.vertexBuffer.data(), uint32_t(d.vertexBuffer.size()),
                                             d.linePointDataBuffer.data(), uint32_t(d.linePointDataBuffer.size())),
                   "lv_set_tube_triangle_mesh"))
            return false;
        triangleMeshDirty = false;
        if (prebaker) static_cast<HipAmbientOcclusionBaker*>(ambientOcclusionBaker.get())->notifyInputsChanged();
    }
    if (prebaker) {
        HipAmbientOcclusionBaker* baker = static_cast<HipAmbientOcclusionBaker*>(ambientOcclusionBaker.get());
        if (!baker->uploadParametrization(lineData)) {
            check(LV_E_INVALID, "lv_set_ao_parametrization");
            return false;
        }
        // BakingMode::MULTI_THREADED: (re)start the bake on the second stream whenever its inputs changed; this frame and the
        // following ones come out without AO until the table is in place
        if (!baker->startAsyncBakeIfNeeded()) { check(LV_E_HIP, "lv_bake_ao_start"); return false; }
    }
    return true;
}

bool HipAmbientOcclusionBaker::startAsyncBakeIfNeeded() {
    if (bakingMode != BakingMode::MULTI_THREADED || !bakeSettingsDirty) return true;
    bakeSettingsDirty = false;
    isDataReady = false;
    if (lv_bake_ao_start(ctx) != LV_OK) return false;
    bakeStarted = true;
    return true;
}

// LineRenderer::renderBase (LineRenderer.cpp:248-277): depth range + AO iterations happen inside lv_render.
void LineRenderer::renderBase() {}

bool LineRenderer::renderMode(int mode) {
    if (!uploadFrameState()) return false;
    const uint32_t w = *sceneData->viewportWidth, h = *sceneData->viewportHeight;
    sceneData->sceneTexture->resize(size_t(w) * h * 4);
    if (!check(lv_render(ctx, mode, 0, 0, w, h, sceneData->sceneTexture->data()), "lv_render")) return false;
    if (useAmbientOcclusion && ambientOcclusionBaker && ambientOcclusionBaker->getType() == AmbientOcclusionBakerType::RTAO)
        static_cast<HipRayTracedAmbientOcclusion*>(ambientOcclusionBaker.get())->notifyRendered();
    return true;
}

lv_stats LineRenderer::getStatistics() {
    lv_stats s;
    memset(&s, 0, sizeof(s));
    if (ctx) lv_get_stats(ctx, &s);
    return s;
}

// ---------------------------------------------------------------- HipRayTracer
HipRayTracer::HipRayTracer(SceneData* sceneData, TransferFunctionWindow& tfw)
        : LineRenderer("Vulkan Ray Tracer", sceneData, tfw) {
    isRasterizer = false;
}

void HipRayTracer::setLineData(LineDataPtr& newLineData, bool isNewData) {
    updateNewLineData(newLineData, isNewData);
    accumulatedFramesCounter = 0;
}

bool HipRayTracer::needsReRender() {
    // VulkanRayTracer::needsReRender, VulkanRayTracer.cpp:330-336
    if (accumulatedFramesCounter < maxNumAccumulatedFrames) return true;
    return LineRenderer::needsReRender();
}

void HipRayTracer::onHasMoved() {
    LineRenderer::onHasMoved();
    accumulatedFramesCounter = 0; // VulkanRayTracer.cpp:343-346
}

void HipRayTracer::render() {
    LineRenderer::renderBase();
    // rayTracingRenderPass->setFrameNumber(accumulatedFramesCounter), VulkanRayTracer.cpp:141
    if (maxNumAccumulatedFrames > 1) setOption("frame_number", std::to_string(accumulatedFramesCounter));
    if (renderMode(LV_RENDERING_MODE_VULKAN_RAY_TRACER)) accumulatedFramesCounter++;
}

// VulkanRayTracer::setNewSettings, VulkanRayTracer.cpp:226-278
bool HipRayTracer::setNewSettings(const SettingsMap& settings) {
    bool shallReloadGatherShader = LineRenderer::setNewSettings(settings);
    // any change of the AO pipeline (baker type, its sampling, its denoiser) restarts the progressive accumulation, like
    // onHasMoved() does in the reference when the baker reports new settings
    for (const auto& kv : settings.getMap()) {
        const std::string& k = kv.first;
        if (k.rfind("ambient_occlusion_", 0) == 0 || k.rfind("eaw_denoiser_", 0) == 0 || k.rfind("svgf_denoiser_", 0) == 0 ||
            k == "rtao_geometry" || k == "use_jittered_primary_rays")
            accumulatedFramesCounter = 0;
    }
    std::string s;
    bool useAnalyticIntersections = true;
    if (settings.getValueOpt("geometry_mode", s)) {
        if (setOption("geometry_mode", s)) useTriangleMesh = s == "Triangle Mesh";
        accumulatedFramesCounter = 0;
    } else if (settings.getValueOpt("use_analytic_intersections", useAnalyticIntersections)) {
        if (setOption("use_analytic_intersections", useAnalyticIntersections ? "true" : "false"))
            useTriangleMesh = !useAnalyticIntersections;
        accumulatedFramesCounter = 0;
    }
    // the "Elliptic Tubes" checkbox of the AABB geometry mode has no settings key in the reference (VulkanRayTracer.cpp:198-201)
    bool elliptic = useAnalyticEllipticTubes;
    if (settings.getValueOpt("use_analytic_elliptic_tubes", elliptic) && elliptic != useAnalyticEllipticTubes) {
        useAnalyticEllipticTubes = elliptic;
        if (lineData) lineData->setTriangleRepresentationDirty(); // other point normals, other boxes
        linesDirty = true;
        accumulatedFramesCounter = 0;
    }
    if (settings.getValueOpt("num_samples_per_frame", numSamplesPerFrame)) {
        setOption("num_samples_per_frame", std::to_string(numSamplesPerFrame));
        accumulatedFramesCounter = 0;
    }
    if (settings.getValueOpt("num_accumulated_frames", maxNumAccumulatedFrames)) {
        setOption("num_accumulated_frames", std::to_string(maxNumAccumulatedFrames));
        accumulatedFramesCounter = 0;
    }
    if (settings.getValueOpt("use_deterministic_sampling", useDeterministicSampling)) {
        setOption("use_deterministic_sampling", useDeterministicSampling ? "true" : "false");
        accumulatedFramesCounter = 0;
    }
    if (settings.getValueOpt("use_mlat", useMlat)) { // VulkanRayTracer.cpp:266-275
        setOption("use_mlat", useMlat ? "true" : "false");
        accumulatedFramesCounter = 0;
    }
    if (settings.getValueOpt("mlat_num_nodes", mlatNumNodes)) {
        setOption("mlat_num_nodes", std::to_string(mlatNumNodes));
        accumulatedFramesCounter = 0;
    }
    if (settings.getValueOpt("max_depth_complexity", maxDepthComplexity))
        setOption("max_depth_complexity", std::to_string(maxDepthComplexity));
    return shallReloadGatherShader;
}

// VulkanRayTracer::setNewState, VulkanRayTracer.cpp:280-330: the camelCase keys the canned benchmark states carry
// (InternalState.cpp:276-297), translated to the settings keys of setNewSettings
void HipRayTracer::setNewState(const InternalState& newState) {
    const SettingsMap& rs = newState.rendererSettings;
    SettingsMap m;
    std::string geometryModeString;
    bool useAnalyticIntersections = true, b = false;
    uint32_t u = 0;
    if (rs.getValueOpt("geometryMode", geometryModeString)) m.addKeyValue("geometry_mode", geometryModeString);
    else if (rs.getValueOpt("useAnalyticIntersections", useAnalyticIntersections)) m.addKeyValue("use_analytic_intersections", useAnalyticIntersections);
    if (rs.getValueOpt("numSamplesPerFrame", u)) m.addKeyValue("num_samples_per_frame", u);
    if (rs.getValueOpt("maxNumAccumulatedFrames", u)) m.addKeyValue("num_accumulated_frames", u);
    if (rs.getValueOpt("useDeterministicSampling", b)) m.addKeyValue("use_deterministic_sampling", b);
    if (rs.getValueOpt("useMlat", b)) m.addKeyValue("use_mlat", b);
    if (rs.getValueOpt("mlatNumNodes", u)) m.addKeyValue("mlat_num_nodes", u);
    if (!m.isEmpty()) setNewSettings(m);
    accumulatedFramesCounter = 0;
}

// ---------------------------------------------------------------- PPLL
HipPerPixelLinkedListLineRenderer::HipPerPixelLinkedListLineRenderer(SceneData* sceneData, TransferFunctionWindow& tfw)
        : LineRenderer("Per-Pixel Linked List Renderer", sceneData, tfw) {
    isRasterizer = true;
}

void HipPerPixelLinkedListLineRenderer::setLineData(LineDataPtr& newLineData, bool isNewData) {
    updateNewLineData(newLineData, isNewData);
    // updateLargeMeshMode (PerPixelLinkedListLineRenderer.cpp:109-126) is applied by the context from the
    // segment count unless the ppll_* options override it.
}

// PerPixelLinkedListLineRenderer::setNewState, .cpp:98-107: remember the state's name, start fresh per-phase timers (the
// reference creates a new sgl::vk::Timer for "PPLLClear" / "FCGather" / "PPLLResolve"; here: the context's per-kernel event rings)
void HipPerPixelLinkedListLineRenderer::setNewState(const InternalState& newState) {
    currentStateName = newState.name;
    if (ctx) lv_reset_timers(ctx);
}

void HipPerPixelLinkedListLineRenderer::render() {
    // LineRenderer::setNewTilingMode -> the addressing of the per-pixel lists
    setOption("ppll_tile_width", std::to_string(tileWidth));
    setOption("ppll_tile_height", std::to_string(tileHeight));
    LineRenderer::renderBase();
    renderMode(LV_RENDERING_MODE_PER_PIXEL_LINKED_LIST); // clear -> gather -> resolve
}

bool HipPerPixelLinkedListLineRenderer::setNewSettings(const SettingsMap& settings) {
    bool r = LineRenderer::setNewSettings(settings);
    for (const char* key : {"ppll_max_num_frags", "ppll_expected_avg_depth_complexity", "ppll_tile_width", "ppll_tile_height",
                            "sorting_mode", // sortingAlgorithmMode, PerPixelLinkedListLineRenderer.hpp:113
                            "ppll_fragment_source", "ppll_fragment_colour"}) { // build-owned probes (include/linevis_hip.h)
        std::string s;
        if (settings.getValueOpt(key, s)) setOption(key, s);
    }
    return r;
}

void HipPerPixelLinkedListLineRenderer::computeStatistics(uint64_t& totalNumFragments, uint32_t& maxComplexity) {
    lv_stats s = getStatistics();
    totalNumFragments = s.fragments;
    maxComplexity = s.max_depth_complexity;
}

} // namespace lv
