// Render (03_PathTracingBasics): signature of Sources/03_PathTracingBasics/Main.cpp:77-83; the pixel/sample loops,
// TraceTay and the sqrt-gamma resolve (:310-370 -> :380-435) run in k_path<3> + k_resolve.
#pragma once
#include "Materials.hpp"
#include "Sphere.hpp"
#include "TraceTay.hpp"

inline uint32_t& RenderSeed() { static uint32_t seed = 1u; return seed; }

namespace path_detail {
template <class ElementList> inline void flatten(const ElementList& elements, rtbackend::Flat* flat) {
    for (const auto& e : elements) {
        rt_prim p{};
        if (const auto* nested = dynamic_cast<const Scene*>(e.get())) {      // a Scene inside a Scene (Utils.h:199): its elements, in place and in order
            flatten(nested->getElements(), flat);
            continue;
        }
        if (!e || !e->describe(&p)) throw std::runtime_error("ERROR: scene element without a GPU description (describe() returned false)");
        p.material = -1;                                         // element without material: absorbs (black)
        if (const auto& m = e->getMaterial()) {
            rt_material rm{};
            if (!m->describe(&rm)) throw std::runtime_error("ERROR: material without a GPU description (describe() returned false)");
            p.material = flat->material(m.get(), rm);
        }
        flat->prims.push_back(p);
    }
}
inline void render(ImageBuffer<RGBQUAD>* fb, const rtbackend::Flat& flat, int example, float fov, unsigned spp,
                   const math::Vec3<float>& pos, const math::Vec3<float>& orient, uint32_t seed) {
    rt_scene sc = flat.scene();
    rt_camera cam{fov, {pos.x, pos.y, pos.z}, {orient.x, orient.y, orient.z}};
    rt_path_params prm;
    rtbackend::check(rt_path_params_default(example, &prm));
    prm.spp = static_cast<int32_t>(spp);
    prm.seed = seed;
    rt_image img = rtbackend::image_of(fb);
    rtbackend::check(rt_render_path(rtbackend::context(), &sc, &cam, &prm, &img));
}
} // namespace path_detail

inline void Render(ImageBuffer<RGBQUAD>* imageBuffer, const Scene& scene, const float& fov, unsigned samplesPerPixel = 1,
                   math::Vec3<float> camPosition = {0.0f, 0.0f, 0.0f}, math::Vec3<float> camOrientation = {0.0f, 0.0f, 0.0f}) {
    rtbackend::Flat flat;
    path_detail::flatten(scene.getElements(), &flat);
    path_detail::render(imageBuffer, flat, 3, fov, samplesPerPixel, camPosition, camOrientation, RenderSeed());
}
