// TraceTay (04_PathTracingLights) on the HOST: the reference's secondary seam (Sources/04_PathTracingLights/Main.cpp:102-106 ->
// :431-502) - one ray against a scene::Hittable: scattered light (attenuation x child, averaged over SAMPLES_PER_RAY) plus what
// the material emits, black background, black beyond the depth limit.  Render() never calls it (the bounce loop runs in
// k_path<4>).  The materials draw from the clock-seeded engines of Utils.h, as the reference's do: deterministic chains (lights,
// mirror metal, an element without material, the depth limit) compare value for value, the rest in distribution - the test
// renders the committed scene through this function and holds it against the converged image of the reference's own code.
// MAX_RECURSION_DEPTH (6) and SAMPLES_PER_RAY (1) are #defines there, process-wide settings here.
#pragma once
#include "Utils.h"

inline unsigned& TraceTayMaxDepth() { static unsigned depth = 6u; return depth; }
inline unsigned& TraceTaySamplesPerRay() { static unsigned samples = 1u; return samples; }

inline bool TraceTay(const math::Ray& ray, const scene::Hittable& sceneElement, math::Vec3<float>* outColor, unsigned int recursionDepth = 0) {
    using V = math::Vec3<float>;
    if (recursionDepth > TraceTayMaxDepth()) { *outColor = {0.0f, 0.0f, 0.0f}; return false; }
    HitInfo hit{};
    if (!sceneElement.intersectsRay(ray, 0.01f, 1000.0f, &hit)) { *outColor = {0.0f, 0.0f, 0.0f}; return false; }      // the background is black
    if (hit.materialPtr == nullptr) { *outColor = {0.0f, 0.0f, 0.0f}; return true; }
    V result{0.0f, 0.0f, 0.0f};
    if (hit.materialPtr->isScatters(ray, hit)) {
        const unsigned n = TraceTaySamplesPerRay();
        for (unsigned s = 0; s < n; ++s) {
            V attenuation{0.0f, 0.0f, 0.0f};
            const math::Ray scattered = hit.materialPtr->scatteredRay(ray, hit, &attenuation);
            V child{0.0f, 0.0f, 0.0f};
            TraceTay(scattered, sceneElement, &child, recursionDepth + 1);
            result = result + (attenuation * child);
        }
        result = result / static_cast<float>(n);
    }
    if (hit.materialPtr->isEmits(ray, hit)) result = result + hit.materialPtr->emittedColor();
    *outColor = result;
    return true;
}
