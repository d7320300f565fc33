// materials::Refractive (04): Schlick-weighted choice between mirror reflection and refraction, no attenuation
// (Materials/Refractive.hpp:80-105).
#pragma once
#include "../Utils.h"
namespace materials {
class Refractive : public Material {
    float refractionIndex_ = 1.0f;
public:
    Refractive() = default;
    explicit Refractive(float refractionIndex) : refractionIndex_(refractionIndex) {}
    bool isScatters(const math::Ray&, const HitInfo&) const override { return true; }
    bool isEmits(const math::Ray&, const HitInfo&) const override { return false; }
    math::Vec3<float> emittedColor() const override { return {0.0f, 0.0f, 0.0f}; }
    static float reflectance(const math::Vec3<float>& v, const math::Vec3<float>& normal, const float& refractionRatio) {
        const float cosine = std::max(math::Dot(-v, normal), 0.0f);
        float r0 = (1.0f - refractionRatio) / (1.0f + refractionRatio);
        r0 = r0 * r0;
        return r0 + (1.0f - r0) * powf(1.0f - cosine, 5);
    }
    // reflect, or - with probability 1 - reflectance (Schlick) - refract; total internal reflection reflects; no attenuation
    math::Ray scatteredRay(const math::Ray& rayIn, const HitInfo& hitInfo, math::Vec3<float>* attenuationOut) const override {
        const float index = hitInfo.frontFaceSurface ? refractionIndex_ : 1.0f / refractionIndex_;
        math::Vec3<float> dir = math::Reflect(rayIn.getDirection(), hitInfo.normal);
        if (reflectance(rayIn.getDirection(), hitInfo.normal, 1.0f / index) < RndFloat()) dir = math::Refract(rayIn.getDirection(), hitInfo.normal, index, true);
        if (attenuationOut != nullptr) *attenuationOut = {1.0f, 1.0f, 1.0f};
        return math::Ray(hitInfo.point, dir);
    }
    bool describe(rt_material* out) const override { out->type = RT_MAT_DIELECTRIC; out->refraction_index = refractionIndex_; return true; }
};
} // namespace materials
