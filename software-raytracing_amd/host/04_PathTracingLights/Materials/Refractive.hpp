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
    bool describe(rt_material* out) const override { out->type = RT_MAT_DIELECTRIC; out->refraction_index = refractionIndex_; return true; }
};
} // namespace materials
