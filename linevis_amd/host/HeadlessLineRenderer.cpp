#include "HeadlessLineRenderer.hpp"

namespace lv {

HeadlessLineRenderer::HeadlessLineRenderer(RenderingMode mode, int deviceOrdinal) {
    sceneData.viewportWidth = &viewportWidth;
    sceneData.viewportHeight = &viewportHeight;
    sceneData.camera = std::make_shared<Camera>(); // (0,0,0.8), FOVy = 2 atan(1/2), near 0.01, far 100
    sceneData.clearColor = &clearColor;
    sceneData.sceneTexture = &sceneTexture;
    sceneData.deviceOrdinal = deviceOrdinal;
    // renderer factory of MainApp::setRenderer (src/MainApp.cpp:732-862), reduced to the two hot-path plugins
    if (mode == RENDERING_MODE_PER_PIXEL_LINKED_LIST)
        lineRenderer.reset(new HipPerPixelLinkedListLineRenderer(&sceneData, transferFunctionWindow));
    else
        lineRenderer.reset(new HipRayTracer(&sceneData, transferFunctionWindow));
    lineRenderer->initialize();
    lineRenderer->onResolutionChanged();
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
