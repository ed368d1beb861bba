// InternalState.hpp -- the benchmark / replay state of the reference's --perf harness, for the two hot-path plugins.
//
//   struct InternalState                 src/Utils/InternalState.hpp:176-199  (name, renderingMode, rendererSettings, dataSetSettings,
//                                        tilingWidth / tilingHeight / useMortonCodeForTiling, transferFunctionName, windowResolution)
//   getTestModesPerPixelLinkedLists      src/Utils/InternalState.cpp:46-51    ("PPLL", empty renderer settings)
//   getTestModesVulkanRayTracing         src/Utils/InternalState.cpp:276-297  ("VRT Triangle Mesh": useAnalyticIntersections=false,
//                                        numSamplesPerFrame=1; "VRT Analytic" is commented out there and listed here as well: it is
//                                        north_star's path)
//   every state twice, "(2)" appended    src/Utils/InternalState.cpp:187-197  (runStatesTwoTimesForErrorMeasure)
// AutomaticPerformanceMeasurer walks such a list, calling MainApp::setNewState (src/MainApp.cpp:584-726) per entry;
// lv::HeadlessLineRenderer::setNewState is that function for the headless harness, bench.py --states iterates the table.
// DataSetDescriptor / filterSettings belong to the data-set menu and the line filters (outside the hot path) and are omitted.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "SettingsMap.hpp"

namespace lv {

struct InternalState {
    std::string name, nameRaw;
    int32_t renderingMode = 11;            // RenderingMode (RenderingModes.hpp:32-53): 11 = Vulkan ray tracer, 2 = PPLL
    SettingsMap rendererSettings;          // LineRenderer::setNewState reads camelCase keys, setNewSettings snake_case ones
    SettingsMap dataSetSettings;           // LineData::setNewSettings
    int tilingWidth = 2;                   // InternalState.hpp:193-195
    int tilingHeight = 8;
    bool useMortonCodeForTiling = false;
    std::string transferFunctionName;      // e.g. "Standard.xml" / "Transparent_Aneurysm.xml"; the embedder resolves the name
    int windowResolution[2] = {0, 0};      // 0: keep the current one

    bool operator==(const InternalState& rhs) const {
        return name == rhs.name && renderingMode == rhs.renderingMode && rendererSettings == rhs.rendererSettings &&
               dataSetSettings == rhs.dataSetSettings && tilingWidth == rhs.tilingWidth && tilingHeight == rhs.tilingHeight &&
               useMortonCodeForTiling == rhs.useMortonCodeForTiling && transferFunctionName == rhs.transferFunctionName &&
               windowResolution[0] == rhs.windowResolution[0] && windowResolution[1] == rhs.windowResolution[1];
    }
    bool operator!=(const InternalState& rhs) const { return !(*this == rhs); }
};

inline void getTestModesPerPixelLinkedLists(std::vector<InternalState>& states, InternalState state) {
    state.renderingMode = 2;
    state.name = "PPLL";
    state.rendererSettings = SettingsMap();
    states.push_back(state);
}

inline void getTestModesVulkanRayTracing(std::vector<InternalState>& states, InternalState state) {
    state.renderingMode = 11;
    state.name = "VRT Analytic";
    state.rendererSettings = SettingsMap({{"useAnalyticIntersections", "true"}, {"numSamplesPerFrame", "1"}});
    states.push_back(state);
    state.name = "VRT Triangle Mesh";
    state.rendererSettings = SettingsMap({{"useAnalyticIntersections", "false"}, {"numSamplesPerFrame", "1"}});
    states.push_back(state);
}

/// The hot-path subset of getTestModes() (InternalState.cpp:641-644 -> getTestModesOIT, :149-214): 1920 x 1080, PPLL + the ray tracer
/// states, every state twice for the error measure.
inline std::vector<InternalState> getTestModes(bool runStatesTwoTimesForErrorMeasure = true) {
    std::vector<InternalState> states;
    InternalState state;
    state.windowResolution[0] = 1920;
    state.windowResolution[1] = 1080;
    getTestModesPerPixelLinkedLists(states, state);
    getTestModesVulkanRayTracing(states, state);
    if (runStatesTwoTimesForErrorMeasure) {
        std::vector<InternalState> oldStates = states;
        states.clear();
        for (InternalState s : oldStates) {
            states.push_back(s);
            s.name += "(2)";
            states.push_back(s);
        }
    }
    for (InternalState& s : states) s.nameRaw = s.name;
    return states;
}

} // namespace lv
