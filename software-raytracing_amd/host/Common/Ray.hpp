// Ray.hpp - math::Ray with the reference's public surface (Sources/Common/Ray.hpp:11-352): origin, direction
// (normalised by the constructor and by setDirection), weight, and the analytic tests.  On the GPU path these tests
// run inside the kernels; the host versions exist so code that calls them directly keeps compiling.
#pragma once
#include "Math.hpp"

namespace math {
class Ray {
    Vec3<float> origin_, direction_{0.0f, 0.0f, -1.0f};
    float weight_ = 1.0f;
public:
    Ray() = default;
    Ray(const Vec3<float>& origin, const Vec3<float>& direction, const float& weight = 1.0f) : origin_(origin), direction_(Normalize(direction)), weight_(weight) {}
    void setOrigin(const Vec3<float>& o) { origin_ = o; }
    const Vec3<float>& getOrigin() const { return origin_; }
    void setDirection(const Vec3<float>& d, bool normalize = true) { direction_ = normalize ? Normalize(d) : d; }
    const Vec3<float>& getDirection() const { return direction_; }
    void setWeight(float w) { weight_ = w; }
    const float& getWeight() const { return weight_; }

    // a root below tMin is replaced by tMax before the minimum is taken (reference :127-134)
    bool intersectsSphere(const Vec3<float>& c, float radius, float tMin, float tMax, float* tOut) const {
        const Vec3<float> oc = origin_ - c;
        const float a = Dot(direction_, direction_), b = 2.0f * Dot(direction_, oc), cc = Dot(oc, oc) - radius * radius;
        const float disc = b * b - 4.0f * a * cc;
        if (disc < 0) return false;
        float t1 = (-b - std::sqrt(disc)) / (2.0f * a), t2 = (-b + std::sqrt(disc)) / (2.0f * a);
        if (t1 < tMin && t2 < tMin) return false;
        if (t1 < tMin) t1 = tMax;
        if (t2 < tMin) t2 = tMax;
        const float t = std::min(t1, t2);
        if (t > tMax) return false;
        if (tOut) *tOut = t;
        return true;
    }
    bool intersectsPlane(const Vec3<float>& n, const Vec3<float> p0, float tMin, float tMax, float* tOut) const {
        const float dn = Dot(direction_, n);
        if (dn == 0) return false;
        const float t = ((n.x * p0.x - n.x * origin_.x) + (n.y * p0.y - n.y * origin_.y) + (n.z * p0.z - n.z * origin_.z)) / dn;
        if (!(t > 0 && t >= tMin && t <= tMax)) return false;
        if (tOut) *tOut = t;
        return true;
    }
    bool intersectsAARectangleXy(float z, float xMin, float xMax, float yMin, float yMax, float tMin, float tMax, float* tOut) const {
        const float t = (z - origin_.z) / direction_.z;
        if (!(t > 0 && t >= tMin && t <= tMax)) return false;
        const float x = origin_.x + direction_.x * t, y = origin_.y + direction_.y * t;
        if (!(x >= xMin && x <= xMax && y >= yMin && y <= yMax)) return false;
        if (tOut) *tOut = t;
        return true;
    }
    bool intersectsAARectangleYz(float x, float yMin, float yMax, float zMin, float zMax, float tMin, float tMax, float* tOut) const {
        const float t = (x - origin_.x) / direction_.x;
        if (!(t > 0 && t >= tMin && t <= tMax)) return false;
        const float y = origin_.y + direction_.y * t, z = origin_.z + direction_.z * t;
        if (!(y >= yMin && y <= yMax && z >= zMin && z <= zMax)) return false;
        if (tOut) *tOut = t;
        return true;
    }
    bool intersectsAARectangleXz(float y, float xMin, float xMax, float zMin, float zMax, float tMin, float tMax, float* tOut) const {
        const float t = (y - origin_.y) / direction_.y;
        if (!(t > 0 && t >= tMin && t <= tMax)) return false;
        const float x = origin_.x + direction_.x * t, z = origin_.z + direction_.z * t;
        if (!(x >= xMin && x <= xMax && z >= zMin && z <= zMax)) return false;
        if (tOut) *tOut = t;
        return true;
    }
    // clockwise winding: normal = normalize(cross(v2-v0, v1-v0)); barycentric x goes with (v2-v0), y with (v1-v0)
    bool intersectsTriangle(const Vec3<float>& v0, const Vec3<float>& v1, const Vec3<float>& v2, float tMin, float tMax, float* tOut, Vec2<float>* bary) const {
        const Vec3<float> e1 = v1 - v0, e2 = v2 - v0, n = Normalize(Cross(e2, e1));
        float t = 0;
        if (!intersectsPlane(n, v0, tMin, tMax, &t)) return false;
        const Vec3<float> p = (origin_ + direction_ * t) - v0;
        // p = x*e2 + y*e1 + w*n  =>  (x, y, w) = Inverse([e2 e1 n]) * p
        const Vec3<float> b = Inverse(Mat3<float>(e2, e1, n)) * p;
        const float x = b.x, y = b.y;
        if (!(x >= 0.0f && y >= 0.0f && x + y <= 1.0f)) return false;
        if (bary) { bary->x = x; bary->y = y; }
        if (tOut) *tOut = t;
        return true;
    }
};
} // namespace math
