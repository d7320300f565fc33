// scene::Rectangle (04): (material, centre, sizes, Euler orientation in degrees); the ray is taken into object space
// with GetRotationMat(-orientation) and tested against the XY rectangle at z = 0, local normal +Z
// (Sources/04_PathTracingLights/Scene/Rectangle.hpp:31-97).
#pragma once
#include "../Utils.h"

namespace scene {
class Rectangle : public Hittable {
    math::Vec3<float> position_;
    math::Vec3<float> orientation_;
    math::Vec2<float> sizes_{1.0f, 1.0f};
public:
    Rectangle() = default;
    Rectangle(const std::shared_ptr<materials::Material>& materialPtr, const math::Vec3<float>& position,
              const math::Vec2<float>& sizes = {1.0f, 1.0f}, const math::Vec3<float>& orientation = {0.0f, 0.0f, 0.0f})
        : Hittable(materialPtr), position_(position), orientation_(orientation), sizes_(sizes) {}
    bool intersectsRay(const math::Ray& ray, float tMin, float tMax, HitInfo* hitInfo) const override {
        const auto inv = math::GetRotationMat(-orientation_);
        const math::Ray local(inv * (ray.getOrigin() - position_), inv * ray.getDirection());
        const float hw = sizes_.x / 2.0f, hh = sizes_.y / 2.0f;
        float t = 0;
        if (!local.intersectsAARectangleXy(0.0f, -hw, hw, -hh, hh, tMin, tMax, &t)) return false;
        if (hitInfo) {
            hitInfo->t = t;
            hitInfo->point = ray.getOrigin() + (ray.getDirection() * t);
            hitInfo->normal = {0.0f, 0.0f, 1.0f};
            hitInfo->materialPtr = materialPtr_;
            detail::face_forward(local.getDirection(), hitInfo);
            hitInfo->normal = math::GetRotationMat(orientation_) * hitInfo->normal;
        }
        return true;
    }
    bool describe(rt_prim* out) const override {
        out->type = RT_PRIM_RECT;
        out->f[0] = position_.x; out->f[1] = position_.y; out->f[2] = position_.z;
        out->f[3] = sizes_.x; out->f[4] = sizes_.y;
        out->f[5] = orientation_.x; out->f[6] = orientation_.y; out->f[7] = orientation_.z;
        return true;
    }
};
} // namespace scene
