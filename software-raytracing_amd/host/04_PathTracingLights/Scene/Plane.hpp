// scene::Plane (04): (material, position, normal) with the normal normalised by the constructor,
// Sources/04_PathTracingLights/Scene/Plane.hpp:28-70.
#pragma once
#include "../Utils.h"

namespace scene {
class Plane : public Hittable {
    math::Vec3<float> position_;
    math::Vec3<float> normal_{0.0f, 1.0f, 0.0f};
public:
    Plane() = default;
    Plane(const std::shared_ptr<materials::Material>& materialPtr, const math::Vec3<float>& position, const math::Vec3<float>& normal)
        : Hittable(materialPtr), position_(position), normal_(math::Normalize(normal)) {}
    bool intersectsRay(const math::Ray& ray, float tMin, float tMax, HitInfo* hitInfo) const override {
        float t = 0;
        if (!ray.intersectsPlane(normal_, position_, tMin, tMax, &t)) return false;
        if (hitInfo) {
            hitInfo->t = t;
            hitInfo->point = ray.getOrigin() + (ray.getDirection() * t);
            hitInfo->normal = normal_;
            hitInfo->materialPtr = materialPtr_;
            detail::face_forward(ray.getDirection(), hitInfo);
        }
        return true;
    }
    bool describe(rt_prim* out) const override {
        out->type = RT_PRIM_PLANE;
        out->f[0] = position_.x; out->f[1] = position_.y; out->f[2] = position_.z;
        out->f[3] = normal_.x; out->f[4] = normal_.y; out->f[5] = normal_.z;
        return true;
    }
};
} // namespace scene
