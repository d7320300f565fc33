// materials::Metal (04): mirror direction, jittered inside a 60*roughness degree cone (Materials/Metal.hpp:67-89).
#pragma once
#include "../Utils.h"
namespace materials {
class Metal : public Material {
    math::Vec3<float> albedo_;
    float roughness_;
public:
    explicit Metal(const math::Vec3<float>& albedo = {0.5f, 0.5f, 0.5f}, float roughness = 0.0f) : albedo_(albedo), roughness_(roughness) {}
    bool isScatters(const math::Ray&, const HitInfo&) const override { return true; }
    bool isEmits(const math::Ray&, const HitInfo&) const override { return false; }
    math::Vec3<float> emittedColor() const override { return {0.0f, 0.0f, 0.0f}; }
    // the mirror direction, jittered inside a cone of 60*roughness degrees; attenuation = albedo * cos to the normal
    math::Ray scatteredRay(const math::Ray& rayIn, const HitInfo& hitInfo, math::Vec3<float>* attenuationOut) const override {
        math::Ray scattered(hitInfo.point, RndHemisphereVec3(math::Reflect(rayIn.getDirection(), hitInfo.normal), 60.0f * roughness_));
        if (attenuationOut != nullptr) *attenuationOut = albedo_ * std::max(math::Dot(scattered.getDirection(), hitInfo.normal), 0.0f);
        return scattered;
    }
    bool describe(rt_material* out) const override {
        out->type = RT_MAT_METAL; out->albedo[0] = albedo_.x; out->albedo[1] = albedo_.y; out->albedo[2] = albedo_.z; out->roughness = roughness_;
        return true;
    }
};
} // namespace materials
