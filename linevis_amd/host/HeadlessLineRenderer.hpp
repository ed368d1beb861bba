// HeadlessLineRenderer.hpp -- test / benchmark harness shaped like the reference's
// test/VolumetricPathTracingTestRenderer.{hpp,cpp}:47-92,34-41,120-165: owns a fixed default camera, a viewport and
// one renderer plugin; setRenderingResolution(w, h) + renderFrame() -> RGBA8 image in host memory.
#pragma once

#include <memory>
#include <vector>

#include "LineRenderer.hpp"

namespace lv {

class HeadlessLineRenderer {
public:
    HeadlessLineRenderer(RenderingMode mode, int deviceOrdinal);
    /// several GPUs behind the one plugin (SceneData::deviceOrdinals): tiles dealt over the devices, one gather per frame
    HeadlessLineRenderer(RenderingMode mode, const std::vector<int>& deviceOrdinals, const std::string& transport);
    ~HeadlessLineRenderer();
    bool isValid() const { return lineRenderer && lineRenderer->getContext() != nullptr; }
    void setRenderingResolution(uint32_t width, uint32_t height);
    void setLineData(LineDataPtr& lineData, bool isNewData);
    void setTransferFunction(const float* rgba, uint32_t n);
    void setClearColor(float r, float g, float b, float a);
    void setCameraPosition(vec3 position, vec3 lookAt);
    void setNewSettings(const SettingsMap& settings);
    /// MainApp::setNewState (src/MainApp.cpp:584-726) for the headless harness: resolution, tiling mode, renderer (re)creation when
    /// the rendering mode changes, lineRenderer->setNewState + setNewSettings(rendererSettings), lineData->setNewSettings(dataSetSettings).
    /// The transfer function is the embedder's to load (transferFunctionName is reported back by getLastState()).
    void setNewState(const InternalState& newState);
    const InternalState& getLastState() const { return lastState; }
    RenderingMode getRenderingMode() const { return lineRenderer->getRenderingMode(); }
    /// Renders one complete frame; returns width*height*4 bytes (row 0 = top) or nullptr on error.
    const uint8_t* renderFrame();
    uint32_t getWidth() const { return viewportWidth; }
    uint32_t getHeight() const { return viewportHeight; }
    LineRenderer* getLineRenderer() { return lineRenderer.get(); }
    const std::string& getLastError() const { return lineRenderer->getLastError(); }

private:
    uint32_t viewportWidth = 128, viewportHeight = 128;
    Color clearColor;
    std::vector<uint8_t> sceneTexture;
    SceneData sceneData;
    TransferFunctionWindow transferFunctionWindow;
    std::unique_ptr<LineRenderer> lineRenderer;
    LineDataPtr lineData;
    InternalState lastState;
    bool firstState = true;
    void createRenderer(RenderingMode mode);
};

} // namespace lv
