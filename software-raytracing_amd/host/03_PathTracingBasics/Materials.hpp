// Materials.hpp (03) - MaterialDiffuse / MaterialMetal / MaterialDielectric: constructors, getters and setters of
// Sources/03_PathTracingBasics/Materials.hpp:8-266.  Render() runs the scatter rules in k_path<3>; scatter() is their host
// form for callers of the reference interface (Diffuse :50-75, Metal :142-169, Dielectric :188-195 and :233-265).
#pragma once
#include <algorithm>
#include "Utils.h"

namespace material_detail {
inline bool emit(const HitInfo& hit, const math::Vec3<float>& dir, const math::Vec3<float>& att, math::Vec3<float>* attenuationOut, math::Ray* scatteredRayOut) {
    if (scatteredRayOut != nullptr) { scatteredRayOut->setOrigin(hit.point); scatteredRayOut->setDirection(dir); }
    if (attenuationOut != nullptr) *attenuationOut = att;
    return true;
}
}

class MaterialDiffuse : public Material {
    math::Vec3<float> albedo_;
public:
    explicit MaterialDiffuse(const math::Vec3<float>& albedo = {0.5f, 0.5f, 0.5f}) : albedo_(albedo) {}
    void setAlbedo(const math::Vec3<float>& albedo) { albedo_ = albedo; }
    const math::Vec3<float>& getAlbedo() const { return albedo_; }
    // any direction of the hemisphere around the normal, attenuation = albedo (no cosine term, as the reference)
    bool scatter(const math::Ray&, const HitInfo& hitInfo, math::Vec3<float>* attenuationOut, math::Ray* scatteredRayOut) const override {
        return material_detail::emit(hitInfo, RndHemisphereVec(hitInfo.normal), albedo_, attenuationOut, scatteredRayOut);
    }
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
    // mirror direction, jittered inside a cone of 60*roughness degrees when rough; never rejected below the surface
    bool scatter(const math::Ray& rayIn, const HitInfo& hitInfo, math::Vec3<float>* attenuationOut, math::Ray* scatteredRayOut) const override {
        math::Vec3<float> dir = math::Reflect(rayIn.getDirection(), hitInfo.normal);
        if (roughness_ > 0.0f) dir = RndHemisphereVec(dir, 60.0f * roughness_);
        return material_detail::emit(hitInfo, dir, albedo_, attenuationOut, scatteredRayOut);
    }
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
    // Schlick's approximation of the reflected share for the angle of incidence and the index ratio
    static float reflectance(const math::Vec3<float>& v, const math::Vec3<float>& normal, const float& refractionRatio) {
        const float cosine = std::max(math::Dot(-v, normal), 0.0f);
        float r0 = (1.0f - refractionRatio) / (1.0f + refractionRatio);
        r0 = r0 * r0;
        return r0 + (1.0f - r0) * powf(1.0f - cosine, 5);
    }
    // reflect, or - with probability 1 - reflectance - refract (total internal reflection reflects); no attenuation
    bool scatter(const math::Ray& rayIn, const HitInfo& hitInfo, math::Vec3<float>* attenuationOut, math::Ray* scatteredRayOut) const override {
        const float index = hitInfo.frontFaceSurface ? refractionIndex_ : 1.0f / refractionIndex_;
        math::Vec3<float> dir = math::Reflect(rayIn.getDirection(), hitInfo.normal);
        if (reflectance(rayIn.getDirection(), hitInfo.normal, 1.0f / index) < RndFloat()) dir = math::Refract(rayIn.getDirection(), hitInfo.normal, index, true);
        return material_detail::emit(hitInfo, dir, {1.0f, 1.0f, 1.0f}, attenuationOut, scatteredRayOut);
    }
    bool describe(rt_material* out) const override { out->type = RT_MAT_DIELECTRIC; out->refraction_index = refractionIndex_; return true; }
};
