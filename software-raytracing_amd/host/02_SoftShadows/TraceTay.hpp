// TraceTay + RandomVectorToLightSphere (02_SoftShadows) on the HOST (Sources/02_SoftShadows/Main.cpp:100-117 -> :389-566): 01's
// tracer with MAX_SHADOW_SAMPLES random shadow rays per light toward the light's disk.  As in the reference the samples come
// from ONE process-wide std::default_random_engine (:40, :191-192: seeded from the clock there; SeedShadowEngine() here), so
// only statistics are comparable, never single values.  Render() does not use it (the kernels draw from a counter-based generator).
#pragma once
#define RT_HOST_SOFT_SHADOWS 1
#include <random>
#include "../01_Basic/TraceTay.hpp"

inline unsigned& TraceTayShadowSamples() { static unsigned samples = 16u; return samples; }      // MAX_SHADOW_SAMPLES (:15)
inline std::default_random_engine& ShadowEngine() { static std::default_random_engine engine(1u); return engine; }
inline void SeedShadowEngine(unsigned seed) { ShadowEngine().seed(seed); }

// a unit vector from the shaded point to a random point of the light's disk: radius linear in one uniform (centre-heavy), as written (:539-566)
inline math::Vec3<float> RandomVectorToLightSphere(const math::Vec3<float>& shadedPoint, const LightSource& lightSource) {
    const math::Vec3<float> toLight = math::Normalize(lightSource.position - shadedPoint);
    std::uniform_real_distribution<float> rBiasDist(0.0f, 1.0f), vBiasDist(-1.0f, 1.0f);
    auto& engine = ShadowEngine();
    const float bx = vBiasDist(engine), by = vBiasDist(engine), bz = vBiasDist(engine);      // drawn in this order
    const float rBias = rBiasDist(engine);
    const math::Vec3<float> inDisk = math::Normalize(math::Cross(toLight, toLight + math::Vec3<float>(bx, by, bz)));
    return math::Normalize((lightSource.position + inDisk * lightSource.radius * rBias) - shadedPoint);
}

inline bool TraceTay(const math::Ray& ray, const std::vector<std::shared_ptr<SceneElement>>& sceneElements, const std::vector<LightSource>& lightSources,
                     float minDistance, float maxDistance, math::Vec3<float>* outColor, uint32_t recursionDepth = 0) {
    return whitted_detail::trace_tay(ray, sceneElements, lightSources, minDistance, maxDistance, outColor, recursionDepth,
        [&](const math::Vec3<float>& point, const LightSource& light, const math::Vec3<float>& toLight, uint32_t depth) {
            const unsigned n = TraceTayShadowSamples();
            float intensity = 1.0f;
            const float per = 1.0f / static_cast<float>(n);
            for (unsigned i = 0; i < n; ++i)                                                               // :457-467
                if (whitted_detail::occluded(math::Ray(point, RandomVectorToLightSphere(point, light)), sceneElements, 0.01f, math::Length(toLight) - light.radius, depth + 1))
                    intensity -= per;
            return intensity;
        });
}
