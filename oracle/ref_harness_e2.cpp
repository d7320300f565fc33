// ref_harness_e2.cpp - the REFERENCE's 02_SoftShadows scene headers (Types.h, Sphere.hpp, Plane.hpp) and Common/*.hpp,
// included from /root/reference where they lie.  TEST INFRASTRUCTURE ONLY (see ref_harness_e1.cpp).
// 02's Render / TraceTay / RandomVectorToLightSphere live in Main.cpp, which cannot be built here (<windows.h>), so the
// golden below drives the reference's element classes (virtual intersectsRay), math:: functions and math::Ray with a loop
// that follows Main.cpp:353-376 (pixel ray), :389-531 (TraceTay) and :539-566 (light-disk sampler) step for step, with the
// same engine type the reference uses (std::default_random_engine, Main.cpp:40) and its constants (:14-15).
#include <cstring>
#include <cstdint>
#include <algorithm>
#include <limits>
#include <memory>
#include <random>
#include <vector>
#include <Math.hpp>
#include <Ray.hpp>
#include "Types.h"
#include "Sphere.hpp"
#include "Plane.hpp"

using V3 = math::Vec3<float>;
namespace golden2 {
const uint32_t kMaxDepth = 4;          // MAX_RECURSION_DEPTH
unsigned g_shadow_samples = 16;        // MAX_SHADOW_SAMPLES (a #define there; a parameter here for the sweep)
std::default_random_engine* g_engine = nullptr;
unsigned long long g_rays = 0;
using Elements = std::vector<std::shared_ptr<SceneElement>>;

V3 to_light_disk(const V3& shaded, const LightSource& light) {                          // Main.cpp:539-566
    V3 toLight = math::Normalize(light.position - shaded);
    std::uniform_real_distribution<float> rBiasDist(0.0f, 1.0f);
    std::uniform_real_distribution<float> vBiasDist(-1.0f, 1.0f);
    V3 vBias = {vBiasDist(*g_engine), vBiasDist(*g_engine), vBiasDist(*g_engine)};
    float rBias = rBiasDist(*g_engine);
    V3 randomPl = math::Normalize(math::Cross(toLight, toLight + vBias));
    V3 onDisk = light.position + randomPl * light.radius * rBias;
    return math::Normalize(onDisk - shaded);
}

bool trace(const math::Ray& ray, const Elements& elements, const std::vector<LightSource>& lights, float minDistance, float maxDistance,
           V3* out, uint32_t depth) {                                                   // Main.cpp:389-531
    bool hit = false;
    float nearest = std::numeric_limits<float>::max();
    V3 nearestNormal = {0.0f, 0.0f, 0.0f};
    uint32_t nearestIndex = 0;
    ++g_rays;
    if (depth <= kMaxDepth) {
        for (uint32_t i = 0; i < elements.size(); i++) {
            float t; V3 normal;
            if (elements[i]->intersectsRay(ray, minDistance, maxDistance, &t, &normal) && t < nearest) {
                hit = true; nearest = t; nearestNormal = normal; nearestIndex = i;
            }
        }
    }
    if (hit && out != nullptr) {
        const auto& element = elements[nearestIndex];
        V3 point = ray.getOrigin() + (ray.getDirection() * nearest);
        V3 primary = {0.0f, 0.0f, 0.0f}, secondary = {0.0f, 0.0f, 0.0f};
        if (element->getMaterial().primaryToSecondary > 0.0f) {
            V3 diffuse = {0.0f, 0.0f, 0.0f}, specular = {0.0f, 0.0f, 0.0f};
            for (const auto& light : lights) {
                V3 toLight = light.position - point;
                V3 toLightDir = math::Normalize(toLight);
                float intensity = 1.0f;
                float perSample = 1.0f / static_cast<float>(g_shadow_samples);
                for (unsigned i = 0; i < g_shadow_samples; i++) {
                    V3 rnd = to_light_disk(point, light);
                    math::Ray shadowRay(point, rnd);
                    if (trace(shadowRay, elements, {}, 0.01f, math::Length(toLight) - light.radius, nullptr, depth + 1)) intensity -= perSample;
                }
                diffuse = diffuse + (light.color * std::max(0.0f, math::Dot(toLightDir, nearestNormal)) * intensity);
                specular = specular + (light.color * std::pow(std::max(0.0f, math::Dot(math::Reflect(-toLightDir, nearestNormal), -ray.getDirection())),
                                                              element->getMaterial().shininess) * intensity);
            }
            primary = (diffuse * element->getMaterial().albedo) + (specular * element->getMaterial().specularIntensity);
        }
        if (element->getMaterial().primaryToSecondary < 1.0f) {
            V3 reflection = {0.0f, 0.0f, 0.0f}, refraction = {0.0f, 0.0f, 0.0f};
            if (element->getMaterial().reflectToRefract > 0.0f) {
                math::Ray reflected(point, math::Reflect(ray.getDirection(), nearestNormal));
                trace(reflected, elements, lights, 0.001f, maxDistance, &reflection, depth + 1);
            }
            if (element->getMaterial().reflectToRefract < 1.0f) {
                float eta = element->getMaterial().refractionEta;
                if (math::Dot(nearestNormal, -ray.getDirection()) < 0.0f) { nearestNormal = -nearestNormal; eta = 1.0f / eta; }
                math::Ray refracted(point, math::Refract(ray.getDirection(), nearestNormal, eta));
                trace(refracted, elements, lights, 0.001f, maxDistance, &refraction, depth + 1);
            }
            secondary = math::Mix(reflection, refraction, std::max(1.0f - element->getMaterial().reflectToRefract, 0.0f));
        }
        *out = math::Mix(primary, secondary, std::max(1.0f - element->getMaterial().primaryToSecondary, 0.0f));
    } else if (out != nullptr) *out = {0.2f, 0.7f, 0.8f};
    return hit;
}
}  // namespace golden2

extern "C" {
// The committed scene of 02 (Main.cpp:201-218) at w x h, rows [y0, y1): 'evals' independent evaluations of every pixel
// (each = one TraceTay with 'shadow_samples' shadow rays per light), summed.  seed: the engine's seed (the reference takes the clock).
void ref_e2_golden_rows(int w, int h, int evals, int shadow_samples, unsigned seed, int y0, int y1, double* sum, double* sumsq, unsigned long long* rays) {
    Material redMatte = {V3(1.0f, 0.0f, 0.0f), 0.0f};
    Material greenMatte = {V3(0.0f, 1.0f, 0.0f), 0.0f};
    Material whiteMatte = {V3(1.0f, 1.0f, 1.0f), 0.0f};
    Material redRubber = {V3(0.6f, 0.2f, 0.2f), 0.1f, 16.0f};
    golden2::Elements scene{};
    scene.push_back(std::make_shared<Plane>(whiteMatte, V3(0.0f, -10.0f, 0.0f), V3(0.0f, 1.0f, 0.0f)));
    scene.push_back(std::make_shared<Plane>(whiteMatte, V3(0.0f, 10.0f, 0.0f), V3(0.0f, -1.0f, 0.0f)));
    scene.push_back(std::make_shared<Plane>(whiteMatte, V3(0.0f, 0.0f, -20.0f), V3(0.0f, 0.0f, 1.0f)));
    scene.push_back(std::make_shared<Plane>(whiteMatte, V3(0.0f, 0.0f, 0.001f), V3(0.0f, 0.0f, -1.0f)));
    scene.push_back(std::make_shared<Plane>(greenMatte, V3(10.0f, 0.0f, 00.0f), V3(-1.0f, 0.0f, 0.0f)));
    scene.push_back(std::make_shared<Plane>(redMatte, V3(-10.0f, 0.0f, 00.0f), V3(1.0f, 0.0f, 0.0f)));
    scene.push_back(std::make_shared<Sphere>(redRubber, V3(0.0f, -6.5f, -15.0f), 3.0f));
    std::vector<LightSource> lights{};
    lights.push_back({V3(0.0f, 6.5f, -10.0f), V3(0.9f, 0.9f, 0.9f), 3.0f});
    std::default_random_engine engine(seed);
    golden2::g_engine = &engine; golden2::g_shadow_samples = static_cast<unsigned>(shadow_samples); golden2::g_rays = 0;
    const float fov = 90.0f, fw = static_cast<float>(w), fh = static_cast<float>(h);
    const float fovRadians = static_cast<float>(fov / (180.0f / M_PI));
    for (int j = y0; j < y1; j++) for (int i = 0; i < w; i++) {
        float x = (2.0f * (static_cast<float>(i) + 0.5f) / fw - 1.0f) * tanf(fovRadians / 2.0f) * fw / fh;
        float y = -(2.0f * (static_cast<float>(j) + 0.5f) / fh - 1.0f) * tanf(fovRadians / 2.0f);
        math::Ray ray({0.0f, 0.0f, 0.0f}, {x, y, -1.0f});
        double* ps = sum + (static_cast<size_t>(j - y0) * w + i) * 3;
        double* pq = sumsq + (static_cast<size_t>(j - y0) * w + i) * 3;
        for (int e = 0; e < evals; e++) {
            V3 c = {0.0f, 0.0f, 0.0f};
            golden2::trace(ray, scene, lights, 0.0f, 1000.0f, &c, 0);
            ps[0] += c.r; ps[1] += c.g; ps[2] += c.b;
            pq[0] += double(c.r) * c.r; pq[1] += double(c.g) * c.g; pq[2] += double(c.b) * c.b;
        }
    }
    golden2::g_engine = nullptr;
    *rays = golden2::g_rays;
}
}
