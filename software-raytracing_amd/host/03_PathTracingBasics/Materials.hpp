// Materials.hpp (03) - MaterialDiffuse / MaterialMetal / MaterialDielectric: constructors, getters and setters of
// Sources/03_PathTracingBasics/Materials.hpp:8-266; the scatter rules themselves run in k_path<3>.
#pragma once
#include "Utils.h"

class MaterialDiffuse : public Material {
    math::Vec3<float> albedo_;
public:
    explicit MaterialDiffuse(const math::Vec3<float>& albedo = {0.5f, 0.5f, 0.5f}) : albedo_(albedo) {}
    void setAlbedo(const math::Vec3<float>& albedo) { albedo_ = albedo; }
    const math::Vec3<float>& getAlbedo() const { return albedo_; }
    bool describe(rt_material* out) const override {
        out->type = RT_MAT_DIFFUSE; out->albedo[0] = albedo_.x; out->albedo[1] = albedo_.y; out->albedo[2] = albedo_.z;
        return true;
    }
};

class MaterialMetal : public Material {
    math::Vec3<float> albedo_;
    float roughness_;
public:
    explicit MaterialMetal(const math::Vec3<float>& albedo = {0.5f, 0.5f, 0.5f}, float roughness = 0.0f) : albedo_(albedo), roughness_(roughness) {}
    void setAlbedo(const math::Vec3<float>& albedo) { albedo_ = albedo; }
    const math::Vec3<float>& getAlbedo() const { return albedo_; }
    void setRoughness(float roughness) { roughness_ = roughness; }
    float getRoughness() const { return roughness_; }
    bool describe(rt_material* out) const override {
        out->type = RT_MAT_METAL; out->albedo[0] = albedo_.x; out->albedo[1] = albedo_.y; out->albedo[2] = albedo_.z; out->roughness = roughness_;
        return true;
    }
};

class MaterialDielectric : public Material {
    float refractionIndex_;
public:
    explicit MaterialDielectric(const float& refractionIndex = 1.0f) : refractionIndex_(refractionIndex) {}
    void setRefractionIndex(const float& refractionIndex) { refractionIndex_ = refractionIndex; }
    float getRefractionIndex() const { return refractionIndex_; }
    bool describe(rt_material* out) const override { out->type = RT_MAT_DIELECTRIC; out->refraction_index = refractionIndex_; return true; }
};
