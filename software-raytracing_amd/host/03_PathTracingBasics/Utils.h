// Utils.h (03) - HitInfo, Material, HittableElement, Scene with the reference's public surface
// (Sources/03_PathTracingBasics/Utils.h:98-290).  The random helpers of that file (:20-88) have no host form here:
// sampling happens in the kernels from a counter-based generator.  New: describe() hooks for the GPU flattener.
#pragma once
#include <memory>
#include <utility>
#include <vector>
#include "../Common/Math.hpp"
#include "../Common/Ray.hpp"
#include "../Common/RtBackend.hpp"

class Material;

struct HitInfo {
    math::Vec3<float> point = {};
    math::Vec3<float> normal = {};
    float t = 0.0f;
    bool frontFaceSurface = true;
    std::shared_ptr<Material> materialPtr = nullptr;
};

class Material {
public:
    Material() = default;
    virtual ~Material() = default;
    // The reference's pure virtual scatter() (:136-140) draws from <random>; on the GPU path a material is data.
    virtual bool describe(rt_material* out) const = 0;
};

class HittableElement {
protected:
    std::shared_ptr<Material> materialPtr_;
public:
    HittableElement() : materialPtr_(nullptr) {}
    explicit HittableElement(std::shared_ptr<Material> materialPtr) : materialPtr_(std::move(materialPtr)) {}
    virtual ~HittableElement() = default;
    void setMaterial(const std::shared_ptr<Material>& materialPtr) { materialPtr_ = materialPtr; }
    const std::shared_ptr<Material>& getMaterial() const { return materialPtr_; }
    virtual bool intersectsRay(const math::Ray& ray, float tMin, float tMax, HitInfo* hitInfo) const = 0;
    virtual bool describe(rt_prim* out) const { (void)out; return false; }
};

class Scene : public HittableElement {
    std::vector<std::shared_ptr<HittableElement>> elements_;
public:
    Scene() = default;
    explicit Scene(const std::shared_ptr<HittableElement>& element) { elements_.push_back(element); }
    const std::vector<std::shared_ptr<HittableElement>>& getElements() const { return elements_; }
    void addElement(const std::shared_ptr<HittableElement>& element) { elements_.push_back(element); }
    void clear() { elements_.clear(); elements_.shrink_to_fit(); }
    // closest hit with a shrinking tMax: of two elements at the same t the LATER one wins (:255-283)
    bool intersectsRay(const math::Ray& ray, float tMin, float tMax, HitInfo* hitInfo) const override {
        HitInfo hit{};
        bool any = false;
        float closest = tMax;
        for (const auto& e : elements_)
            if (e->intersectsRay(ray, tMin, closest, &hit)) { any = true; closest = hit.t; if (hitInfo) *hitInfo = hit; }
        return any;
    }
};
