// Types.h (01/02) - Material, LightSource, SceneElement with the reference's public surface
// (Sources/01_Basic/Types.h:9-99; 02_SoftShadows/Types.h is the same file).  What is new: describe(), the hook
// the GPU flattener calls, because the reference keeps element geometry private with no getters.
#pragma once
#include "../Common/Math.hpp"
#include "../Common/Ray.hpp"
#include <memory>
#include "../Common/RtBackend.hpp"

struct Material {                        // aggregate: brace-initialised positionally in main() (01 Main.cpp:182-187)
    math::Vec3<float> albedo = {1.0f, 1.0f, 1.0f};
    float specularIntensity = 1.0f;
    float shininess = 16.0f;
    float primaryToSecondary = 1.0f;
    float reflectToRefract = 1.0f;
    float refractionEta = 1.0f;
};

struct LightSource {
    math::Vec3<float> position = {0.0f, 0.0f, 0.0f};
    math::Vec3<float> color = {1.0f, 1.0f, 1.0f};
    float radius = 0.0f;
};

class SceneElement {
protected:
    Material material_;
public:
    SceneElement() : material_({{1.0f, 1.0f, 1.0f}, 0.0f, 16.0f, 1.0f, 1.0f, 1.0f}) {}
    explicit SceneElement(const Material& material) : material_(material) {}
    virtual ~SceneElement() = default;
    void setMaterial(const Material& material) { material_ = material; }
    const Material& getMaterial() const { return material_; }
    virtual bool intersectsRay(const math::Ray& ray, float tMin, float tMax, float* tOut, math::Vec3<float>* normalOut) const = 0;
    // Fill type/flags/f[] of the flat record; return false if the element has no GPU form (Render then throws).
    virtual bool describe(rt_prim* out) const = 0;
};

namespace whitted_detail {
inline rt_material flat(const Material& m) {
    rt_material r{};
    r.type = RT_MAT_PHONG;
    r.albedo[0] = m.albedo.x; r.albedo[1] = m.albedo.y; r.albedo[2] = m.albedo.z;
    r.specular_intensity = m.specularIntensity; r.shininess = m.shininess; r.primary_to_secondary = m.primaryToSecondary;
    r.reflect_to_refract = m.reflectToRefract; r.refraction_eta = m.refractionEta;
    return r;
}
// Element order is kept (ties: the first element wins).
inline rtbackend::Flat flatten(const std::vector<std::shared_ptr<SceneElement>>& elements, const std::vector<LightSource>& lights) {
    rtbackend::Flat flat;
    for (const auto& e : elements) {
        rt_prim p{};
        if (!e || !e->describe(&p)) throw std::runtime_error("ERROR: scene element without a GPU description (describe() returned false)");
        p.material = flat.material(nullptr, whitted_detail::flat(e->getMaterial()));   // materials are by value in 01/02
        flat.prims.push_back(p);
    }
    for (const auto& l : lights) flat.lights.push_back(rt_light{{l.position.x, l.position.y, l.position.z}, {l.color.x, l.color.y, l.color.z}, l.radius, 0.0f});
    return flat;
}
inline void render(ImageBuffer<RGBQUAD>* fb, float fov, const std::vector<std::shared_ptr<SceneElement>>& elements,
                   const std::vector<LightSource>& lights, int example, uint32_t seed) {
    const rtbackend::Flat flat = flatten(elements, lights);
    rt_scene sc = flat.scene();
    rt_camera cam{fov, {0, 0, 0}, {0, 0, 0}};
    rt_whitted_params prm;
    rtbackend::check(rt_whitted_params_default(example, &prm));
    prm.seed = seed;
    rt_image img = rtbackend::image_of(fb);
    rtbackend::check(rt_render_whitted(rtbackend::context(), &sc, &cam, &prm, &img));
}
} // namespace whitted_detail
