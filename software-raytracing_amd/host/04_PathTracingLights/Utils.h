// Utils.h (04) - HitInfo, materials::Material, scene::Hittable, scene::List with the reference's public surface
// (Sources/04_PathTracingLights/Utils.h:171-392).  Its random helpers (:15-158) live in the kernels.  New: describe().
#pragma once
#include <memory>
#include <utility>
#include <vector>
#include "../Common/Math.hpp"
#include "../Common/Ray.hpp"
#include "../Common/RtBackend.hpp"

namespace materials { class Material; }

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
    // isScatters / isEmits / emittedColor keep their reference meaning (:215-238); scatteredRay() is sampled on the GPU.
    virtual bool isScatters(const math::Ray& rayIn, const HitInfo& hitInfo) const = 0;
    virtual bool isEmits(const math::Ray& rayIn, const HitInfo& hitInfo) const = 0;
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
