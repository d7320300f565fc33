// TraceTay (01_Basic / 02_SoftShadows) on the HOST: the reference's secondary seam - one ray against the scene, external linkage
// (Sources/01_Basic/Main.cpp:100-107 -> :372-496; 02_SoftShadows/Main.cpp:100-107 -> :389-531).  Render() never calls it (the
// pixel loop and the whole recursion run on the GPU); it is here for callers that trace single rays through the host classes,
// and it is the CPU statement of what the kernels compute: same operator order, so 01's colours match the reference's floats
// bit for bit (tests/test_host_mirror.py against tests/golden/frame_01.npz).
// The depth limit is the reference's MAX_RECURSION_DEPTH (4, a #define there), a process-wide setting here.
#pragma once
#include <algorithm>
#include <cmath>
#include <limits>
#include <memory>
#include <vector>
#include "Types.h"

inline uint32_t& TraceTayMaxDepth() { static uint32_t depth = 4u; return depth; }

namespace whitted_detail {
// visibility(point, light, toLight, depth) -> the factor that scales the light's diffuse and specular terms, or a negative value
// for "skip this light" (01: an occluded light contributes nothing, :437; 02: 1 - occluded samples / samples, :457-467)
template <class Visibility>
inline bool trace_tay(const math::Ray& ray, const std::vector<std::shared_ptr<SceneElement>>& elements, const std::vector<LightSource>& lights,
                      float minDistance, float maxDistance, math::Vec3<float>* outColor, uint32_t depth, const Visibility& visibility) {
    using V = math::Vec3<float>;
    bool hit = false;
    float nearest = std::numeric_limits<float>::max();
    V normal{0.0f, 0.0f, 0.0f};
    size_t index = 0;
    if (depth <= TraceTayMaxDepth())                               // beyond the limit nothing is tested: "no hit", i.e. the sky colour below
        for (size_t i = 0; i < elements.size(); ++i) {
            float t; V n;
            if (elements[i]->intersectsRay(ray, minDistance, maxDistance, &t, &n) && t < nearest) { hit = true; nearest = t; normal = n; index = i; }   // strict <: the first element wins ties
        }
    if (!outColor) return hit;                                     // a shadow ray
    if (!hit) { *outColor = {0.2f, 0.7f, 0.8f}; return false; }
    const Material& m = elements[index]->getMaterial();
    const V point = ray.getOrigin() + (ray.getDirection() * nearest);
    V primary{0.0f, 0.0f, 0.0f}, secondary{0.0f, 0.0f, 0.0f};
    if (m.primaryToSecondary > 0.0f) {                             // Phong with shadows
        V diffuse{0.0f, 0.0f, 0.0f}, specular{0.0f, 0.0f, 0.0f};
        for (const LightSource& light : lights) {
            const V toLight = light.position - point;
            const V toLightDir = math::Normalize(toLight);
            const float scale = visibility(point, light, toLight, depth);
            if (scale < 0.0f) continue;
            // (x * 1.0f is x to the last bit, so 01 - scale 1 - and 02 share one expression)
            diffuse = diffuse + (light.color * std::max(0.0f, math::Dot(toLightDir, normal)) * scale);
            specular = specular + (light.color * std::pow(std::max(0.0f, math::Dot(math::Reflect(-toLightDir, normal), -ray.getDirection())), m.shininess) * scale);
        }
        primary = (diffuse * m.albedo) + (specular * m.specularIntensity);
    }
    if (m.primaryToSecondary < 1.0f) {                             // reflection / refraction, mixed by reflectToRefract
        V reflection{0.0f, 0.0f, 0.0f}, refraction{0.0f, 0.0f, 0.0f};
        if (m.reflectToRefract > 0.0f)
            trace_tay(math::Ray(point, math::Reflect(ray.getDirection(), normal)), elements, lights, 0.001f, maxDistance, &reflection, depth + 1, visibility);
        if (m.reflectToRefract < 1.0f) {
            float eta = m.refractionEta;
            if (math::Dot(normal, -ray.getDirection()) < 0.0f) { normal = -normal; eta = 1.0f / eta; }      // leaving the medium (after the reflected ray was built)
            trace_tay(math::Ray(point, math::Refract(ray.getDirection(), normal, eta)), elements, lights, 0.001f, maxDistance, &refraction, depth + 1, visibility);
        }
        secondary = math::Mix(reflection, refraction, std::max(1.0f - m.reflectToRefract, 0.0f));
    }
    *outColor = math::Mix(primary, secondary, std::max(1.0f - m.primaryToSecondary, 0.0f));
    return true;
}
// a shadow ray: any element inside [minDistance, maxDistance], one level deeper, no lights, no colour
inline bool occluded(const math::Ray& ray, const std::vector<std::shared_ptr<SceneElement>>& elements, float minDistance, float maxDistance, uint32_t depth) {
    if (depth > TraceTayMaxDepth()) return false;
    for (const auto& e : elements) { float t; math::Vec3<float> n; if (e->intersectsRay(ray, minDistance, maxDistance, &t, &n)) return true; }
    return false;
}
} // namespace whitted_detail

#ifndef RT_HOST_SOFT_SHADOWS      // 02_SoftShadows/TraceTay.hpp defines this name with its own shadow term
inline bool TraceTay(const math::Ray& ray, const std::vector<std::shared_ptr<SceneElement>>& sceneElements, const std::vector<LightSource>& lightSources,
                     float minDistance, float maxDistance, math::Vec3<float>* outColor, uint32_t recursionDepth = 0) {
    return whitted_detail::trace_tay(ray, sceneElements, lightSources, minDistance, maxDistance, outColor, recursionDepth,
        [&](const math::Vec3<float>& point, const LightSource&, const math::Vec3<float>& toLight, uint32_t depth) {
            return whitted_detail::occluded(math::Ray(point, toLight), sceneElements, 0.01f, math::Length(toLight), depth + 1) ? -1.0f : 1.0f;      // :435-437
        });
}
#endif
