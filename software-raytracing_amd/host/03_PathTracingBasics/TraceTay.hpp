// TraceTay (03_PathTracingBasics) on the HOST: the reference's secondary seam (Sources/03_PathTracingBasics/Main.cpp:93-97 ->
// :380-435) - one ray against a HittableElement, its colour by recursion through Material::scatter().  Render() never calls it
// (the bounce loop runs in k_path<3>); it serves callers that trace single rays through the host classes.  The materials draw
// from the clock-seeded engines of Utils.h, as the reference's do, so only deterministic chains (sky, mirror metal, an element
// without material, the depth limit) are comparable value for value; everything else in distribution.
// MAX_RECURSION_DEPTH (24) and SAMPLES_PER_RAY (1) are #defines there, process-wide settings here.
#pragma once
#include "Utils.h"

inline unsigned& TraceTayMaxDepth() { static unsigned depth = 24u; return depth; }
inline unsigned& TraceTaySamplesPerRay() { static unsigned samples = 1u; return samples; }

inline bool TraceTay(const math::Ray& ray, const HittableElement& hittableElement, math::Vec3<float>* outColor, unsigned recursionDepth = 0) {
    using V = math::Vec3<float>;
    if (recursionDepth > TraceTayMaxDepth()) { *outColor = {0.0f, 0.0f, 0.0f}; return false; }
    HitInfo hit{};
    if (hittableElement.intersectsRay(ray, 0.001f, 1000.0f, &hit)) {
        if (hit.materialPtr == nullptr) { *outColor = {0.0f, 0.0f, 0.0f}; return true; }      // no material: absorbs
        const unsigned n = TraceTaySamplesPerRay();
        V sum{0.0f, 0.0f, 0.0f};
        for (unsigned s = 0; s < n; ++s) {
            math::Ray scattered{};
            V attenuation{};
            if (!hit.materialPtr->scatter(ray, hit, &attenuation, &scattered)) continue;      // absorbed: contributes black
            V child{};
            TraceTay(scattered, hittableElement, &child, recursionDepth + 1);
            sum = sum + (attenuation * child);
        }
        *outColor = sum / static_cast<float>(n);
        return true;
    }
    // the sky: white at the horizon's underside to blue overhead; the ratio is a double there too (:430)
    const auto h = 0.5 * (ray.getDirection().y + 1.0);
    *outColor = math::Mix(V(1.0f, 1.0f, 1.0f), V(0.5f, 0.7f, 1.0f), h);
    return false;
}
