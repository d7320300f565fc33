// scene::Sphere (04): (material, position, radius, inverted), Sources/04_PathTracingLights/Scene/Sphere.hpp:31-75.
#pragma once
#include "../Utils.h"

namespace scene {
class Sphere : public Hittable {
    math::Vec3<float> position_;
    float radius_ = 1.0f;
    bool inverted_ = false;
public:
    Sphere() = default;
    Sphere(const std::shared_ptr<materials::Material>& materialPtr, const math::Vec3<float>& position, const float& radius, bool inverted = false)
        : Hittable(materialPtr), position_(position), radius_(radius), inverted_(inverted) {}
    bool intersectsRay(const math::Ray& ray, float tMin, float tMax, HitInfo* hitInfo) const override {
        float t = 0;
        if (!ray.intersectsSphere(position_, radius_, tMin, tMax, &t)) return false;
        if (hitInfo) {
            hitInfo->t = t;
            hitInfo->point = ray.getOrigin() + (ray.getDirection() * t);
            hitInfo->normal = math::Normalize(hitInfo->point - position_);
            if (inverted_) hitInfo->normal = -hitInfo->normal;
            hitInfo->materialPtr = materialPtr_;
            detail::face_forward(ray.getDirection(), hitInfo);
        }
        return true;
    }
    bool describe(rt_prim* out) const override {
        out->type = RT_PRIM_SPHERE; out->flags = inverted_ ? RT_PRIM_FLAG_FLIP_NORMALS : 0u;
        out->f[0] = position_.x; out->f[1] = position_.y; out->f[2] = position_.z; out->f[3] = radius_;
        return true;
    }
};
} // namespace scene
