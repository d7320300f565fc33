// Plane (01/02): the constructor normalises the normal (Sources/01_Basic/Plane.hpp:26-27); two-sided; the reported
// normal is the stored one whatever side is hit (:43-52).
#pragma once
#include "Types.h"

class Plane : public SceneElement {
    math::Vec3<float> position_;
    math::Vec3<float> normal_{0.0f, 1.0f, 0.0f};
public:
    Plane() = default;
    Plane(const Material& material, const math::Vec3<float>& position, const math::Vec3<float>& normal) : SceneElement(material), position_(position), normal_(math::Normalize(normal)) {}
    bool intersectsRay(const math::Ray& ray, float tMin, float tMax, float* tOut, math::Vec3<float>* normalOut) const override {
        float t = 0;
        if (!ray.intersectsPlane(normal_, position_, tMin, tMax, &t)) return false;
        if (tOut) *tOut = t;
        if (normalOut) *normalOut = normal_;
        return true;
    }
    bool describe(rt_prim* out) const override {
        out->type = RT_PRIM_PLANE;
        out->f[0] = position_.x; out->f[1] = position_.y; out->f[2] = position_.z;
        out->f[3] = normal_.x; out->f[4] = normal_.y; out->f[5] = normal_.z;
        return true;
    }
};
