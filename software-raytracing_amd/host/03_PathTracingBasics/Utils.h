// Utils.h (03) - HitInfo, Material, HittableElement, Scene with the reference's public surface
// (Sources/03_PathTracingBasics/Utils.h:98-290) and its random helpers RndFloat / RndHemisphereVec (:20-88), so that
// code written against the reference's headers - a caller of Material::scatter(), a CPU tracer of its own - still compiles
// and runs on the host.  Render() itself never calls them: the kernels sample from a counter-based generator.
// New: describe() hooks for the GPU flattener.
#pragma once
#include <chrono>
#include <cmath>
#include <memory>
#include <random>
#include <utility>
#include <vector>
#include "../Common/Math.hpp"
#include "../Common/Ray.hpp"
#include "../Common/RtBackend.hpp"

class Material;

// Host-side random helpers (reference Utils.h:20-26, :64-88): one clock-seeded engine per function.
inline float RndFloat(float min = 0.0f, float max = 1.0f) {
    static std::mt19937 engine(static_cast<unsigned>(std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count()));
    return std::uniform_real_distribution<float>(min, max)(engine);
}
// A direction within thetaMax degrees of dir: rotation angle fi uniform in [0, 360), deviation theta uniform in ANGLE
// over [-thetaMax, thetaMax] - the reference's (non-physical) distribution, which the kernels reproduce.
inline math::Vec3<float> RndHemisphereVec(const math::Vec3<float>& dir, const float& thetaMax = 90.0f) {
    static std::mt19937 engine(static_cast<unsigned>(std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count()) + 1u);
    const auto b = math::Normalize(math::Cross(dir, dir + math::Vec3<float>(0.01f, 0.01f, 0.01f)));
    const auto c = math::Normalize(math::Cross(dir, b));
    const float fi = (std::uniform_real_distribution<float>(0.0f, 1.0f)(engine) * 360.0f) / 57.2958f;
    const float theta = (std::uniform_real_distribution<float>(-1.0f, 1.0f)(engine) * thetaMax) / 57.2958f;
    const math::Vec3<float> around = (b * std::cos(fi)) + (c * std::sin(fi));
    return (dir * std::cos(theta)) + (around * std::sin(theta));
}

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
    // The reference's interface (:136-140), on the host: one scattered ray and its attenuation for a hit; false = absorbed.
    // Render() does not call it (on the GPU path a material is data, see describe()); it serves callers of their own.
    virtual bool scatter(const math::Ray& rayIn, const HitInfo& hitInfo, math::Vec3<float>* attenuationOut, math::Ray* scatteredRayOut) const = 0;
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
