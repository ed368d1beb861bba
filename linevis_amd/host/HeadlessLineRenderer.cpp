#include "HeadlessLineRenderer.hpp"

namespace lv {

HeadlessLineRenderer::HeadlessLineRenderer(RenderingMode mode, int deviceOrdinal) {
    sceneData.viewportWidth = &viewportWidth;
    sceneData.viewportHeight = &viewportHeight;
    sceneData.camera = std::make_shared<Camera>(); // (0,0,0.8), FOVy = 2 atan(1/2), near 0.01, far 100
    sceneData.clearColor = &clearColor;
    sceneData.sceneTexture = &sceneTexture;
    sceneData.deviceOrdinal = deviceOrdinal;
    createRenderer(mode);
}

HeadlessLineRenderer::HeadlessLineRenderer(RenderingMode mode, const std::vector<int>& deviceOrdinals, const std::string& transport) {
    sceneData.viewportWidth = &viewportWidth;
    sceneData.viewportHeight = &viewportHeight;
    sceneData.camera = std::make_shared<Camera>();
    sceneData.clearColor = &clearColor;
    sceneData.sceneTexture = &sceneTexture;
    sceneData.deviceOrdinal = deviceOrdinals.empty() ? 0 : deviceOrdinals[0];
    sceneData.deviceOrdinals = deviceOrdinals;
    sceneData.multiGpuTransport = transport;
    createRenderer(mode);
}

// renderer factory of MainApp::setRenderer (src/MainApp.cpp:732-862), reduced to the two hot-path plugins
void HeadlessLineRenderer::createRenderer(RenderingMode mode) {
    lineRenderer.reset(); // one context at a time: the old plugin releases its device memory first
    if (mode == RENDERING_MODE_PER_PIXEL_LINKED_LIST)
        lineRenderer.reset(new HipPerPixelLinkedListLineRenderer(&sceneData, transferFunctionWindow));
    else
        lineRenderer.reset(new HipRayTracer(&sceneData, transferFunctionWindow));
    lineRenderer->initialize();
    lineRenderer->onResolutionChanged();
}

void HeadlessLineRenderer::setNewState(const InternalState& newState) {
    // 1. resolution (MainApp.cpp:597-610), 1.1 tiling mode of the SSBO accesses (:612-615)
    if (newState.windowResolution[0] > 0 && newState.windowResolution[1] > 0 &&
        (uint32_t(newState.windowResolution[0]) != viewportWidth || uint32_t(newState.windowResolution[1]) != viewportHeight))
        setRenderingResolution(uint32_t(newState.windowResolution[0]), uint32_t(newState.windowResolution[1]));
    LineRenderer::setNewTilingMode(newState.tilingWidth, newState.tilingHeight, newState.useMortonCodeForTiling);
    // 2.1 a new renderer when the mode changes (:624-643; the reference also recreates it when only the renderer settings
    // change -- its renderers keep GUI state --, here the settings are applied to the living plugin)
    if (firstState || newState.renderingMode != lastState.renderingMode) {
        if (RenderingMode(newState.renderingMode) != lineRenderer->getRenderingMode()) {
            createRenderer(RenderingMode(newState.renderingMode));
            transferFunctionWindow.setTable(transferFunctionWindow.getTable().data(), uint32_t(transferFunctionWindow.getTable().size() / 4));
            if (lineData) lineRenderer->setLineData(lineData, false);
        }
    }
    // 2.2 renderer settings (:645-663)
    lineRenderer->setNewState(newState);
    lineRenderer->setNewSettings(newState.rendererSettings);
    // 5. data-set settings (:700-702)
    if (lineData && lineData->setNewSettings(newState.dataSetSettings)) lineRenderer->setLineData(lineData, false);
    lastState = newState;
    firstState = false;
}

HeadlessLineRenderer::~HeadlessLineRenderer() = default;

void HeadlessLineRenderer::setRenderingResolution(uint32_t width, uint32_t height) {
    viewportWidth = width;
    viewportHeight = height;
    lineRenderer->onResolutionChanged();
}

void HeadlessLineRenderer::setLineData(LineDataPtr& ld, bool isNewData) {
    lineData = ld;
    lineRenderer->setLineData(lineData, isNewData);
}

void HeadlessLineRenderer::setTransferFunction(const float* rgba, uint32_t n) {
    transferFunctionWindow.setTable(rgba, n);
    lineRenderer->onTransferFunctionMapRebuilt();
}

void HeadlessLineRenderer::setClearColor(float r, float g, float b, float a) {
    clearColor = Color{r, g, b, a};
    lineRenderer->onClearColorChanged();
}

void HeadlessLineRenderer::setCameraPosition(vec3 position, vec3 lookAt) {
    sceneData.camera->setPosition(position);
    sceneData.camera->setLookAtLocation(lookAt);
    lineRenderer->onHasMoved();
}

void HeadlessLineRenderer::setNewSettings(const SettingsMap& settings) {
    // MainApp::setNewState order: renderer settings, then data-set settings (src/MainApp.cpp:656-702)
    lineRenderer->setNewSettings(settings);
    if (lineData && lineData->setNewSettings(settings)) lineRenderer->setLineData(lineData, false);
}

const uint8_t* HeadlessLineRenderer::renderFrame() {
    lineRenderer->render();
    if (!lineRenderer->getLastError().empty()) return nullptr;
    return sceneTexture.data();
}

} // namespace lv
