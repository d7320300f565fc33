// Triangle (01/02 interface): a triangle element on math::Ray::intersectsTriangle (Sources/Common/Ray.hpp:150-212), which
// the reference defines but never wraps in a scene element.  Like Plane (01_Basic/Plane.hpp:43-52) it reports its stored
// geometric normal, Normalize(Cross(v2 - v0, v1 - v0)), whatever side is hit.
#pragma once
#include "Types.h"

class Triangle : public SceneElement {
    math::Vec3<float> v0_, v1_, v2_;
public:
    Triangle() = default;
    Triangle(const Material& material, const math::Vec3<float>& v0, const math::Vec3<float>& v1, const math::Vec3<float>& v2) : SceneElement(material), v0_(v0), v1_(v1), v2_(v2) {}
    bool intersectsRay(const math::Ray& ray, float tMin, float tMax, float* tOut, math::Vec3<float>* normalOut) const override {
        float t = 0;
        if (!ray.intersectsTriangle(v0_, v1_, v2_, tMin, tMax, &t, nullptr)) return false;
        if (tOut) *tOut = t;
        if (normalOut) *normalOut = math::Normalize(math::Cross(v2_ - v0_, v1_ - v0_));
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
