// Sphere (03): (position, radius, material, inverted) as Sources/03_PathTracingBasics/Sphere.hpp:29-75: the normal
// is Normalize(p - c), negated when 'inverted', then turned against the ray with frontFaceSurface recording which side.
#pragma once
#include "Utils.h"

class Sphere : public HittableElement {
    math::Vec3<float> position_;
    float radius_ = 1.0f;
    bool inverted_ = false;
public:
    Sphere() = default;
    Sphere(const math::Vec3<float>& position, const float& radius, const std::shared_ptr<Material>& materialPtr, bool inverted = false)
        : HittableElement(materialPtr), position_(position), radius_(radius), inverted_(inverted) {}
    bool intersectsRay(const math::Ray& ray, float tMin, float tMax, HitInfo* hitInfo) const override {
        float t = 0;
        if (!ray.intersectsSphere(position_, radius_, tMin, tMax, &t)) return false;
        if (hitInfo) {
            hitInfo->t = t;
            hitInfo->point = ray.getOrigin() + (ray.getDirection() * t);
            hitInfo->normal = math::Normalize(hitInfo->point - position_);
            if (inverted_) hitInfo->normal = -hitInfo->normal;
            hitInfo->frontFaceSurface = true;
            hitInfo->materialPtr = materialPtr_;
            if (math::Dot(-ray.getDirection(), hitInfo->normal) < 0.0f) { hitInfo->normal = -hitInfo->normal; hitInfo->frontFaceSurface = false; }
        }
        return true;
    }
    bool describe(rt_prim* out) const override {
        out->type = RT_PRIM_SPHERE; out->flags = inverted_ ? RT_PRIM_FLAG_FLIP_NORMALS : 0u;
        out->f[0] = position_.x; out->f[1] = position_.y; out->f[2] = position_.z; out->f[3] = radius_;
        return true;
    }
};
