// ref_harness_e1.cpp - calls the REFERENCE's own header-only code (Common/Math.hpp, Common/Ray.hpp and the
// 01_Basic scene headers, included from /root/reference where they lie; nothing is copied) so that
// oracle/gen_golden.py can record known-answer vectors.  TEST INFRASTRUCTURE ONLY.
// Main.cpp (Render/TraceTay) is not built: it needs <windows.h>, which this image lacks.
#include <cstring>
#include <cstdint>
#include <algorithm>
#include <Math.hpp>
#include <Ray.hpp>
#include "Types.h"
#include "Sphere.hpp"
#include "Plane.hpp"

using V3 = math::Vec3<float>;
static V3 v(const float* p) { return V3(p[0], p[1], p[2]); }
static void st(float* o, const V3& a) { o[0] = a.x; o[1] = a.y; o[2] = a.z; }

extern "C" {
// Ray built WITHOUT normalising (setDirection(dir,false)) so the raw routines see exactly the given vectors.
static math::Ray raw_ray(const float* o, const float* d) { math::Ray r; r.setOrigin(v(o)); r.setDirection(v(d), false); return r; }

int ref_ray_sphere(const float* o, const float* d, const float* c, float r, float tmin, float tmax, float* t) {
    return raw_ray(o, d).intersectsSphere(v(c), r, tmin, tmax, t);
}
int ref_ray_plane(const float* o, const float* d, const float* n, const float* p0, float tmin, float tmax, float* t) {
    return raw_ray(o, d).intersectsPlane(v(n), v(p0), tmin, tmax, t);
}
int ref_ray_rect_xy(const float* o, const float* d, float z, float x0, float x1, float y0, float y1, float tmin, float tmax, float* t) {
    return raw_ray(o, d).intersectsAARectangleXy(z, x0, x1, y0, y1, tmin, tmax, t);
}
int ref_ray_triangle(const float* o, const float* d, const float* v0, const float* v1, const float* v2, float tmin, float tmax, float* t, float* bary) {
    math::Vec2<float> b(0, 0);
    int r = raw_ray(o, d).intersectsTriangle(v(v0), v(v1), v(v2), tmin, tmax, t, &b);
    if (r && bary) { bary[0] = b.x; bary[1] = b.y; }
    return r;
}
void ref_normalize(const float* a, float* out) { st(out, math::Normalize(v(a))); }
void ref_reflect(const float* a, const float* n, float* out) { st(out, math::Reflect(v(a), v(n))); }
void ref_refract(const float* a, const float* n, float eta, int tir, float* out) { st(out, math::Refract(v(a), v(n), eta, tir != 0)); }
void ref_rotation_mat(const float* deg, float* out9) { auto m = math::GetRotationMat(v(deg)); std::memcpy(out9, m.data, sizeof(float) * 9); }
void ref_mix(const float* a, const float* b, float r, float* out) { st(out, math::Mix(v(a), v(b), r)); }
float ref_clamp(float x, float lo, float hi) { return math::Clamp(x, lo, hi); }
void ref_ray_ctor(const float* o, const float* d, float* out_d) { math::Ray r(v(o), v(d)); st(out_d, r.getDirection()); }

// 01/02 scene elements: Sphere/Plane::intersectsRay -> t and (not face-forwarded) normal
int ref_e1_sphere_hit(const float* c, float radius, const float* o, const float* d, float tmin, float tmax, float* t, float* n) {
    Sphere s(Material{}, v(c), radius); V3 nn; int r = s.intersectsRay(raw_ray(o, d), tmin, tmax, t, &nn); if (r) st(n, nn); return r;
}
int ref_e1_plane_hit(const float* p, const float* normal, const float* o, const float* d, float tmin, float tmax, float* t, float* n) {
    Plane s(Material{}, v(p), v(normal)); V3 nn; int r = s.intersectsRay(raw_ray(o, d), tmin, tmax, t, &nn); if (r) st(n, nn); return r;
}
}

// ---- whole-frame golden of 01 -----------------------------------------------------------------------------------------
// 01 is deterministic, so a frame computed by the reference's own classes pins the CPU restatement BIT FOR BIT at image level.
// Render / TraceTay live in Main.cpp (not buildable here); the loop below follows them - Main.cpp:324-359 (pixel ray,
// quantisation) and :372-496 (TraceTay) - and everything it computes with is reference code: the element classes' virtual
// intersectsRay, Material / LightSource, math::Normalize / Length / Dot / Reflect / Refract / Mix / Clamp and math::Ray.
#include <limits>
#include <memory>
#include <vector>
#include <cmath>
namespace golden1 {
const uint32_t kMaxDepth = 4;      // MAX_RECURSION_DEPTH (Main.cpp:12)
using Elements = std::vector<std::shared_ptr<SceneElement>>;
unsigned long long g_rays = 0;

bool trace(const math::Ray& ray, const Elements& elements, const std::vector<LightSource>& lights, float minDistance, float maxDistance, V3* out, uint32_t depth) {
    ++g_rays;
    bool hit = false;
    float nearest = std::numeric_limits<float>::max();
    V3 normalAtHit = {0.0f, 0.0f, 0.0f};
    uint32_t index = 0;
    if (depth <= kMaxDepth)
        for (uint32_t i = 0; i < elements.size(); i++) {
            float t; V3 n;
            if (elements[i]->intersectsRay(ray, minDistance, maxDistance, &t, &n) && t < nearest) { hit = true; nearest = t; normalAtHit = n; index = i; }
        }
    if (!out) return hit;
    if (!hit) { *out = {0.2f, 0.7f, 0.8f}; return false; }
    const Material& m = elements[index]->getMaterial();
    const V3 point = ray.getOrigin() + (ray.getDirection() * nearest);
    V3 primary = {0.0f, 0.0f, 0.0f}, secondary = {0.0f, 0.0f, 0.0f};
    if (m.primaryToSecondary > 0.0f) {
        V3 diffuse = {0.0f, 0.0f, 0.0f}, specular = {0.0f, 0.0f, 0.0f};
        for (const auto& light : lights) {
            V3 toLight = light.position - point;
            V3 toLightDir = math::Normalize(toLight);
            math::Ray shadowRay(point, toLight);
            if (trace(shadowRay, elements, {}, 0.01f, math::Length(toLight), nullptr, depth + 1)) continue;      // hard shadow
            diffuse = diffuse + (light.color * std::max(0.0f, math::Dot(toLightDir, normalAtHit)));
            specular = specular + (light.color * std::pow(std::max(0.0f, math::Dot(math::Reflect(-toLightDir, normalAtHit), -ray.getDirection())), m.shininess));
        }
        primary = (diffuse * m.albedo) + (specular * m.specularIntensity);
    }
    if (m.primaryToSecondary < 1.0f) {
        V3 reflection = {0.0f, 0.0f, 0.0f}, refraction = {0.0f, 0.0f, 0.0f};
        if (m.reflectToRefract > 0.0f) {
            math::Ray reflected(point, math::Reflect(ray.getDirection(), normalAtHit));
            trace(reflected, elements, lights, 0.001f, maxDistance, &reflection, depth + 1);
        }
        if (m.reflectToRefract < 1.0f) {
            float eta = m.refractionEta;
            if (math::Dot(normalAtHit, -ray.getDirection()) < 0.0f) { normalAtHit = -normalAtHit; eta = 1.0f / eta; }
            math::Ray refracted(point, math::Refract(ray.getDirection(), normalAtHit, eta));
            trace(refracted, elements, lights, 0.001f, maxDistance, &refraction, depth + 1);
        }
        secondary = math::Mix(reflection, refraction, std::max(1.0f - m.reflectToRefract, 0.0f));
    }
    *out = math::Mix(primary, secondary, std::max(1.0f - m.primaryToSecondary, 0.0f));
    return true;
}
}  // namespace golden1

extern "C" {
// The committed scene of 01 (Main.cpp:182-203) at w x h: linear float colour per pixel (rgb, 3 floats), the B,G,R,255 bytes
// Render would store (bgra) and the number of TraceTay calls.
void ref_e1_golden_frame(int w, int h, float* rgb, unsigned char* bgra, unsigned long long* rays) {
    Material redMatte = {V3(1.0f, 0.0f, 0.0f), 0.0f};
    Material greenMatte = {V3(0.0f, 1.0f, 0.0f), 0.0f};
    Material whiteMatte = {V3(1.0f, 1.0f, 1.0f), 0.0f};
    Material redRubber = {V3(0.6f, 0.2f, 0.2f), 0.1f, 16.0f};
    Material mirror = {V3(0.1f, 0.1f, 0.1f), 1.0f, 512.0f, 0.25f, 1.0f};
    Material glass = {V3(0.1f, 0.1f, 0.1f), 1.0f, 128.0f, 0.25f, 0.0f, 0.6f};
    golden1::Elements scene{};
    scene.push_back(std::make_shared<Plane>(whiteMatte, V3(0.0f, -10.0f, 0.0f), V3(0.0f, 1.0f, 0.0f)));
    scene.push_back(std::make_shared<Plane>(whiteMatte, V3(0.0f, 10.0f, 0.0f), V3(0.0f, -1.0f, 0.0f)));
    scene.push_back(std::make_shared<Plane>(whiteMatte, V3(0.0f, 0.0f, -20.0f), V3(0.0f, 0.0f, 1.0f)));
    scene.push_back(std::make_shared<Plane>(whiteMatte, V3(0.0f, 0.0f, 0.001f), V3(0.0f, 0.0f, -1.0f)));
    scene.push_back(std::make_shared<Plane>(greenMatte, V3(10.0f, 0.0f, 00.0f), V3(-1.0f, 0.0f, 0.0f)));
    scene.push_back(std::make_shared<Plane>(redMatte, V3(-10.0f, 0.0f, 00.0f), V3(1.0f, 0.0f, 0.0f)));
    scene.push_back(std::make_shared<Sphere>(glass, V3(0.0f, -7.0f, -12.0f), 3.0f));
    scene.push_back(std::make_shared<Sphere>(redRubber, V3(7.5f, -7.0f, -15.0f), 2.0f));
    scene.push_back(std::make_shared<Sphere>(mirror, V3(-7.5f, -7.0f, -15.0f), 2.0f));
    std::vector<LightSource> lights{};
    lights.push_back({V3(0.0f, 8.5f, -10.0f), V3(0.9f, 0.9f, 0.9f)});
    golden1::g_rays = 0;
    const float fov = 90.0f, fw = static_cast<float>(w), fh = static_cast<float>(h);
    const float fovRadians = static_cast<float>(fov / (180.0f / M_PI));
    for (int j = 0; j < h; j++) for (int i = 0; i < w; i++) {
        float x = (2.0f * (static_cast<float>(i) + 0.5f) / fw - 1.0f) * tanf(fovRadians / 2.0f) * fw / fh;
        float y = -(2.0f * (static_cast<float>(j) + 0.5f) / fh - 1.0f) * tanf(fovRadians / 2.0f);
        math::Ray ray({0.0f, 0.0f, 0.0f}, {x, y, -1.0f});
        V3 c = {0.0f, 0.0f, 0.0f};
        golden1::trace(ray, scene, lights, 0.0f, 1000.0f, &c, 0);
        const size_t p = static_cast<size_t>(j) * w + i;
        rgb[3 * p] = c.r; rgb[3 * p + 1] = c.g; rgb[3 * p + 2] = c.b;
        bgra[4 * p] = static_cast<unsigned char>(math::Clamp(c.b, 0.0f, 1.0f) * 255.0f);
        bgra[4 * p + 1] = static_cast<unsigned char>(math::Clamp(c.g, 0.0f, 1.0f) * 255.0f);
        bgra[4 * p + 2] = static_cast<unsigned char>(math::Clamp(c.r, 0.0f, 1.0f) * 255.0f);
        bgra[4 * p + 3] = 255;
    }
    *rays = golden1::g_rays;
}
}
