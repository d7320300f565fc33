// Render (01_Basic): same signature as Sources/01_Basic/Main.cpp:83-87; the pixel loop and the TraceTay recursion
// (:324-359 -> :372-496) run in k_whitted<false> on the GPU.
#pragma once
#include <memory>
#include "Plane.hpp"
#include "Sphere.hpp"
#include "TraceTay.hpp"

inline void Render(ImageBuffer<RGBQUAD>* imageBuffer, const float& fov, const std::vector<std::shared_ptr<SceneElement>>& sceneElements,
                   const std::vector<LightSource>& lightSources) {
    whitted_detail::render(imageBuffer, fov, sceneElements, lightSources, 1, 1u);
}
