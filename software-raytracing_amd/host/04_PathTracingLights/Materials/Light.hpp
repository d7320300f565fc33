// materials::Light (04): never scatters; emits its colour from the FRONT face only (Materials/Light.hpp:31-56).
#pragma once
#include "../Utils.h"
namespace materials {
class Light : public Material {
    math::Vec3<float> color_;
public:
    explicit Light(const math::Vec3<float>& albedo = {1.0f, 1.0f, 1.0f}) : color_(albedo) {}
    bool isScatters(const math::Ray&, const HitInfo&) const override { return false; }
    bool isEmits(const math::Ray&, const HitInfo& hitInfo) const override { return hitInfo.frontFaceSurface; }
    math::Vec3<float> emittedColor() const override { return color_; }
    math::Ray scatteredRay(const math::Ray&, const HitInfo&, math::Vec3<float>*) const override { return {}; }     // a light scatters nothing
    bool describe(rt_material* out) const override {
        out->type = RT_MAT_LIGHT; out->albedo[0] = color_.x; out->albedo[1] = color_.y; out->albedo[2] = color_.z;
        return true;
    }
};
} // namespace materials
