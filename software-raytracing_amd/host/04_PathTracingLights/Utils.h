// Utils.h (04) - HitInfo, materials::Material, scene::Hittable, scene::List with the reference's public surface
// (Sources/04_PathTracingLights/Utils.h:171-392) and the random helpers a caller of the materials needs, RndFloat (:17-24)
// and RndHemisphereVec3 (:132-157), on the host.  Render() never calls them: the kernels sample from a counter-based
// generator.  New: describe().
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <memory>
#include <random>
#include <stdexcept>
#include <utility>
#include <vector>
#include "../Common/Math.hpp"
#include "../Common/Ray.hpp"
#include "../Common/RtBackend.hpp"

namespace materials { class Material; }

inline float RndFloat(float min = 0.0f, float max = 1.0f) {
    static std::default_random_engine engine(static_cast<size_t>(std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count()));
    return std::uniform_real_distribution<float>(min, max)(engine);
}
// A direction within thetaMax degrees of dir, uniform in solid angle: rotation fi uniform in [0, 360), cos(theta) uniform in
// [cos(thetaMax), 1] (reference :132-157).
inline math::Vec3<float> RndHemisphereVec3(const math::Vec3<float>& dir, const float& thetaMax = 90.0f) {
    static std::default_random_engine engine(static_cast<size_t>(std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count()) + 1u);
    const auto b = math::Normalize(math::Cross(dir, dir + math::Vec3<float>(0.01f, 0.01f, 0.01f)));
    const auto c = math::Normalize(math::Cross(dir, b));
    const float fi = (std::uniform_real_distribution<float>(0.0f, 1.0f)(engine) * 360.0f) / 57.2958f;
    const float theta = std::acos(std::uniform_real_distribution<float>(std::cos(thetaMax / 57.2958f), 1.0f)(engine));
    const math::Vec3<float> around = (b * std::cos(fi)) + (c * std::sin(fi));
    return (dir * std::cos(theta)) + (around * std::sin(theta));
}

struct HitInfo {
    math::Vec3<float> point = {};
    math::Vec3<float> normal = {};
    float t = 0.0f;
    bool frontFaceSurface = true;
    std::shared_ptr<materials::Material> materialPtr = nullptr;
};

namespace materials {
class Material {
public:
    Material() = default;
    virtual ~Material() = default;
    // The reference's interface (:215-238).  Render() samples scattered rays on the GPU from describe()'s record;
    // scatteredRay() is the host form of the same rule for callers of their own.
    virtual bool isScatters(const math::Ray& rayIn, const HitInfo& hitInfo) const = 0;
    virtual bool isEmits(const math::Ray& rayIn, const HitInfo& hitInfo) const = 0;
    virtual math::Ray scatteredRay(const math::Ray& rayIn, const HitInfo& hitInfo, math::Vec3<float>* attenuationOut) const = 0;
    virtual math::Vec3<float> emittedColor() const = 0;
    virtual bool describe(rt_material* out) const = 0;
};
} // namespace materials

namespace scene {
class Hittable {
protected:
    std::shared_ptr<materials::Material> materialPtr_;
public:
    Hittable() : materialPtr_(nullptr) {}
    explicit Hittable(std::shared_ptr<materials::Material> materialPtr) : materialPtr_(std::move(materialPtr)) {}
    virtual ~Hittable() = default;
    void setMaterial(const std::shared_ptr<materials::Material>& materialPtr) { materialPtr_ = materialPtr; }
    const std::shared_ptr<materials::Material>& getMaterial() const { return materialPtr_; }
    virtual bool intersectsRay(const math::Ray& ray, float tMin, float tMax, HitInfo* hitInfo) const = 0;
    virtual bool describe(rt_prim* out) const { (void)out; return false; }
};

class List : public Hittable {
    std::vector<std::shared_ptr<Hittable>> elements_;
public:
    List() = default;
    explicit List(const std::shared_ptr<Hittable>& firstElement) { elements_.push_back(firstElement); }
    const std::vector<std::shared_ptr<Hittable>>& getElements() const { return elements_; }
    void addElement(const std::shared_ptr<Hittable>& element) { elements_.push_back(element); }
    void clear() { elements_.clear(); elements_.shrink_to_fit(); }
    // A List is itself a Hittable (:306), so lists nest: the closest hit of a nested list is the closest hit of its
    // elements in order, which is what the flat element order of the GPU scene gives (Render.hpp flattens recursively).
    // shrinking tMax: the later of two elements at the same t wins (:362-390)
    bool intersectsRay(const math::Ray& ray, float tMin, float tMax, HitInfo* hitInfo) const override {
        HitInfo hit{};
        bool any = false;
        float closest = tMax;
        for (const auto& e : elements_)
            if (e->intersectsRay(ray, tMin, closest, &hit)) { any = true; closest = hit.t; if (hitInfo) *hitInfo = hit; }
        return any;
    }
};

namespace detail {
// turn the geometric normal against the ray and record the side, as every 04 element does (e.g. Sphere.hpp:66-69)
inline void face_forward(const math::Vec3<float>& dir, HitInfo* h) {
    h->frontFaceSurface = true;
    if (math::Dot(-dir, h->normal) < 0.0f) { h->normal = -h->normal; h->frontFaceSurface = false; }
}
} // namespace detail
} // namespace scene
