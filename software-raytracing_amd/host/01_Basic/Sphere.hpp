// Sphere (01/02): constructor and intersectsRay as Sources/01_Basic/Sphere.hpp:26-52 (normal NOT face-forwarded).
#pragma once
#include "Types.h"

class Sphere : public SceneElement {
    math::Vec3<float> position_;
    float radius_ = 1.0f;
public:
    Sphere() = default;
    Sphere(const Material& material, const math::Vec3<float>& position, const float& radius) : SceneElement(material), position_(position), radius_(radius) {}
    bool intersectsRay(const math::Ray& ray, float tMin, float tMax, float* tOut, math::Vec3<float>* normalOut) const override {
        float t = 0;
        if (!ray.intersectsSphere(position_, radius_, tMin, tMax, &t)) return false;
        if (tOut) *tOut = t;
        if (normalOut) *normalOut = math::Normalize((ray.getOrigin() + (ray.getDirection() * t)) - position_);
        return true;
    }
    bool describe(rt_prim* out) const override {
        out->type = RT_PRIM_SPHERE;
        out->f[0] = position_.x; out->f[1] = position_.y; out->f[2] = position_.z; out->f[3] = radius_;
        return true;
    }
};
