// scene::Triangle (04 interface): a triangle element on math::Ray::intersectsTriangle (Sources/Common/Ray.hpp:150-212).
// The reference defines the ray/triangle routine but no scene element that wraps it; this is that element, written against
// the reference's Hittable interface so that meshes (BASELINE config 5's height field) can be built from the host side.
// Semantics are the routine's: geometric normal Normalize(Cross(v2 - v0, v1 - v0)), inclusive edges, two-sided.
#pragma once
#include "../Utils.h"

namespace scene {
class Triangle : public Hittable {
    math::Vec3<float> v0_, v1_, v2_;
public:
    Triangle() = default;
    Triangle(const std::shared_ptr<materials::Material>& materialPtr, const math::Vec3<float>& v0, const math::Vec3<float>& v1, const math::Vec3<float>& v2)
        : Hittable(materialPtr), v0_(v0), v1_(v1), v2_(v2) {}
    bool intersectsRay(const math::Ray& ray, float tMin, float tMax, HitInfo* hitInfo) const override {
        float t = 0;
        if (!ray.intersectsTriangle(v0_, v1_, v2_, tMin, tMax, &t, nullptr)) return false;
        if (hitInfo) {
            hitInfo->t = t;
            hitInfo->point = ray.getOrigin() + (ray.getDirection() * t);
            hitInfo->normal = math::Normalize(math::Cross(v2_ - v0_, v1_ - v0_));
            hitInfo->materialPtr = materialPtr_;
            detail::face_forward(ray.getDirection(), hitInfo);
        }
        return true;
    }
    bool describe(rt_prim* out) const override {
        out->type = RT_PRIM_TRIANGLE;
        out->f[0] = v0_.x; out->f[1] = v0_.y; out->f[2] = v0_.z;
        out->f[3] = v1_.x; out->f[4] = v1_.y; out->f[5] = v1_.z;
        out->f[6] = v2_.x; out->f[7] = v2_.y; out->f[8] = v2_.z;
        return true;
    }
};
} // namespace scene
