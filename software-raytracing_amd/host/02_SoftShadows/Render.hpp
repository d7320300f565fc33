// Render (02_SoftShadows): signature of Sources/02_SoftShadows/Main.cpp:88-92; MAX_SHADOW_SAMPLES = 16 disk-light
// samples per shaded hit (:15, :448-476).  The reference seeds its engine from the wall clock (:191-192); here the
// seed is a process-wide setting so that images are reproducible.
#pragma once
#include <memory>
#include "Plane.hpp"
#include "Sphere.hpp"
#include "TraceTay.hpp"

inline uint32_t& RenderSeed() { static uint32_t seed = 1u; return seed; }

inline void Render(ImageBuffer<RGBQUAD>* imageBuffer, const float& fov, const std::vector<std::shared_ptr<SceneElement>>& sceneElements,
                   const std::vector<LightSource>& lightSources) {
    whitted_detail::render(imageBuffer, fov, sceneElements, lightSources, 2, RenderSeed());
}
