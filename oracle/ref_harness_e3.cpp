// ref_harness_e3.cpp - the REFERENCE's 03_PathTracingBasics headers (Utils.h, Materials.hpp, Sphere.hpp),
// included from /root/reference where they lie.  TEST INFRASTRUCTURE ONLY (see ref_harness_e1.cpp).
// The reference's samplers own clock-seeded static engines, so sampler outputs are recorded as
// statistics (moments/histograms), not as exact vectors.
#include <cstring>
#include <cstdint>
#include <algorithm>
#include <Math.hpp>
#include <Ray.hpp>
#include "Utils.h"
#include "Materials.hpp"
#include "Sphere.hpp"

using V3 = math::Vec3<float>;
static V3 v(const float* p) { return V3(p[0], p[1], p[2]); }
static void st(float* o, const V3& a) { o[0] = a.x; o[1] = a.y; o[2] = a.z; }
static math::Ray raw_ray(const float* o, const float* d) { math::Ray r; r.setOrigin(v(o)); r.setDirection(v(d), false); return r; }

extern "C" {
// scene = n spheres {cx,cy,cz,r,inverted}; closest hit through Scene::intersectsRay. out8 = t, point, normal, frontFace; *which = sphere index
int ref_e3_scene_hit(const float* spheres, int n, const float* o, const float* d, float tmin, float tmax, float* out8, int* which) {
    Scene sc;
    std::vector<std::shared_ptr<Material>> mats;
    for (int i = 0; i < n; ++i) {
        mats.push_back(std::make_shared<MaterialDiffuse>(V3((float)i, 0, 0)));
        sc.addElement(std::make_shared<Sphere>(v(spheres + 5 * i), spheres[5 * i + 3], mats.back(), spheres[5 * i + 4] != 0.0f));
    }
    HitInfo h{};
    int r = sc.intersectsRay(raw_ray(o, d), tmin, tmax, &h);
    if (r) {
        out8[0] = h.t; st(out8 + 1, h.point); st(out8 + 4, h.normal); out8[7] = h.frontFaceSurface ? 1.0f : 0.0f;
        *which = (int)static_cast<MaterialDiffuse*>(h.materialPtr.get())->getAlbedo().x;
    }
    return r;
}
// n draws of RndHemisphereVec(dir, thetaMax) -> out[3n]
void ref_e3_rnd_hemisphere(const float* dir, float theta_max, int n, float* out) {
    for (int i = 0; i < n; ++i) st(out + 3 * i, RndHemisphereVec(v(dir), theta_max));
}
// n scatters of one material at one hit: kind 1 diffuse(albedo), 2 metal(albedo, rough), 3 dielectric(ri). out dirs[3n] (normalised by setDirection), att[3]
void ref_e3_scatter(int kind, const float* albedo, float param, const float* in_d, const float* point, const float* normal, int front,
                    int n, float* out_dirs, float* out_att) {
    std::shared_ptr<Material> m;
    if (kind == 1) m = std::make_shared<MaterialDiffuse>(v(albedo));
    else if (kind == 2) m = std::make_shared<MaterialMetal>(v(albedo), param);
    else m = std::make_shared<MaterialDielectric>(param);
    HitInfo h{}; h.point = v(point); h.normal = v(normal); h.frontFaceSurface = front != 0; h.t = 1.0f;
    math::Ray in = raw_ray(point, in_d);
    for (int i = 0; i < n; ++i) {
        math::Ray sr; V3 att;
        m->scatter(in, h, &att, &sr);
        st(out_dirs + 3 * i, sr.getDirection());
        if (i == 0) st(out_att, att);
    }
}
}
