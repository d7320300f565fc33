// scene::Box (04): (material, centre, sizes, Euler orientation in degrees, flipped), six axis-aligned rectangles in the
// object frame tested in a fixed order against a shrinking tMax (Sources/04_PathTracingLights/Scene/Box.hpp:34-158).
// Kept as written there: the two Z faces accept y up to +length/2 instead of +height/2 (:88, :95).
#pragma once
#include "../Utils.h"

namespace scene {
class Box : public Hittable {
    math::Vec3<float> position_;
    math::Vec3<float> orientation_;
    math::Vec3<float> sizes_{1.0f, 1.0f, 1.0f};
    bool flipNormals_ = false;
public:
    Box() = default;
    Box(const std::shared_ptr<materials::Material>& materialPtr, const math::Vec3<float>& position, const math::Vec3<float>& sizes = {1.0f, 1.0f, 1.0f},
        const math::Vec3<float>& orientation = {0.0f, 0.0f, 0.0f}, bool flipped = false)
        : Hittable(materialPtr), position_(position), orientation_(orientation), sizes_(sizes), flipNormals_(flipped) {}
    bool intersectsRay(const math::Ray& ray, float tMin, float tMax, HitInfo* hitInfo) const override {
        const auto inv = math::GetRotationMat(-orientation_);
        const math::Ray local(inv * (ray.getOrigin() - position_), inv * ray.getDirection());
        const float hw = sizes_.x / 2.0f, hh = sizes_.y / 2.0f, hl = sizes_.z / 2.0f;
        float t = 0, closest = tMax;
        bool any = false;
        math::Vec3<float> n;
        auto take = [&](bool hit, const math::Vec3<float>& faceNormal) { if (hit) { any = true; closest = t; n = faceNormal; } };
        take(local.intersectsAARectangleXy(-hl, -hw, hw, -hh, hl, tMin, closest, &t), {0.0f, 0.0f, -1.0f});
        take(local.intersectsAARectangleXy(hl, -hw, hw, -hh, hl, tMin, closest, &t), {0.0f, 0.0f, 1.0f});
        take(local.intersectsAARectangleYz(-hw, -hh, hh, -hl, hl, tMin, closest, &t), {-1.0f, 0.0f, 0.0f});
        take(local.intersectsAARectangleYz(hw, -hh, hh, -hl, hl, tMin, closest, &t), {1.0f, 0.0f, 0.0f});
        take(local.intersectsAARectangleXz(-hh, -hw, hw, -hl, hl, tMin, closest, &t), {0.0f, -1.0f, 0.0f});
        take(local.intersectsAARectangleXz(hh, -hw, hw, -hl, hl, tMin, closest, &t), {0.0f, 1.0f, 0.0f});
        if (any && hitInfo) {
            hitInfo->t = t;
            hitInfo->point = ray.getOrigin() + (ray.getDirection() * t);
            hitInfo->normal = flipNormals_ ? -n : n;
            hitInfo->materialPtr = materialPtr_;
            detail::face_forward(local.getDirection(), hitInfo);
            hitInfo->normal = math::GetRotationMat(orientation_) * hitInfo->normal;
        }
        return any;
    }
    bool describe(rt_prim* out) const override {
        out->type = RT_PRIM_BOX; out->flags = flipNormals_ ? RT_PRIM_FLAG_FLIP_NORMALS : 0u;
        out->f[0] = position_.x; out->f[1] = position_.y; out->f[2] = position_.z;
        out->f[3] = sizes_.x; out->f[4] = sizes_.y; out->f[5] = sizes_.z;
        out->f[6] = orientation_.x; out->f[7] = orientation_.y; out->f[8] = orientation_.z;
        return true;
    }
};
} // namespace scene
