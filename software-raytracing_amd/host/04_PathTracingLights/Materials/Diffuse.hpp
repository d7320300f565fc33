// materials::Diffuse (04): scatters always, never emits; attenuation = albedo * cos (Materials/Diffuse.hpp:64-83).
#pragma once
#include "../Utils.h"
namespace materials {
class Diffuse : public Material {
    math::Vec3<float> albedo_;
public:
    explicit Diffuse(const math::Vec3<float>& albedo = {1.0f, 1.0f, 1.0f}) : albedo_(albedo) {}
    bool isScatters(const math::Ray&, const HitInfo&) const override { return true; }
    bool isEmits(const math::Ray&, const HitInfo&) const override { return false; }
    math::Vec3<float> emittedColor() const override { return {0.0f, 0.0f, 0.0f}; }
    // a direction of the hemisphere around the normal (uniform in solid angle), attenuation = albedo * cos
    math::Ray scatteredRay(const math::Ray&, const HitInfo& hitInfo, math::Vec3<float>* attenuationOut) const override {
        math::Ray scattered(hitInfo.point, RndHemisphereVec3(hitInfo.normal));
        if (attenuationOut != nullptr) *attenuationOut = albedo_ * std::max(math::Dot(scattered.getDirection(), hitInfo.normal), 0.0f);
        return scattered;
    }
    bool describe(rt_material* out) const override {
        out->type = RT_MAT_DIFFUSE; out->albedo[0] = albedo_.x; out->albedo[1] = albedo_.y; out->albedo[2] = albedo_.z;
        return true;
    }
};
} // namespace materials
