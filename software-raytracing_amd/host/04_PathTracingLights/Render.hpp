// Render (04_PathTracingLights): signature of Sources/04_PathTracingLights/Main.cpp:86-92.  The reference splits the
// pixels over 8 std::threads (:399-420); here every pixel-sample is a GPU lane of k_path<4>, followed by k_resolve.
#pragma once
#include "Materials/Diffuse.hpp"
#include "Materials/Light.hpp"
#include "Materials/Metal.hpp"
#include "Materials/Refractive.hpp"
#include "Scene/Box.hpp"
#include "Scene/Plane.hpp"
#include "Scene/Rectangle.hpp"
#include "Scene/Sphere.hpp"
#include "Scene/Triangle.hpp"
#include "TraceTay.hpp"

inline uint32_t& RenderSeed() { static uint32_t seed = 1u; return seed; }

namespace path_detail {
template <class ElementList> inline void flatten(const ElementList& elements, rtbackend::Flat* flat) {
    for (const auto& e : elements) {
        rt_prim p{};
        if (const auto* nested = dynamic_cast<const scene::List*>(e.get())) {      // a List inside a List (Utils.h:306): its elements, in place and in order
            flatten(nested->getElements(), flat);
            continue;
        }
        if (!e || !e->describe(&p)) throw std::runtime_error("ERROR: scene element without a GPU description (describe() returned false)");
        p.material = -1;                                         // no material: the path ends black (Main.cpp:485-498)
        if (const auto& m = e->getMaterial()) {
            rt_material rm{};
            if (!m->describe(&rm)) throw std::runtime_error("ERROR: material without a GPU description (describe() returned false)");
            p.material = flat->material(m.get(), rm);
        }
        flat->prims.push_back(p);
    }
}
} // namespace path_detail

inline void Render(ImageBuffer<RGBQUAD>* imageBuffer, const scene::List& scene, const float& fov, unsigned samples,
                   math::Vec3<float> viewPosition = {0.0f, 0.0f, 0.0f}, math::Vec3<float> viewOrient = {0.0f, 0.0f, 0.0f}) {
    rtbackend::Flat flat;
    path_detail::flatten(scene.getElements(), &flat);
    rt_scene sc = flat.scene();
    rt_camera cam{fov, {viewPosition.x, viewPosition.y, viewPosition.z}, {viewOrient.x, viewOrient.y, viewOrient.z}};
    rt_path_params prm;
    rtbackend::check(rt_path_params_default(4, &prm));
    prm.spp = static_cast<int32_t>(samples);
    prm.seed = RenderSeed();
    rt_image img = rtbackend::image_of(imageBuffer);
    rtbackend::check(rt_render_path(rtbackend::context(), &sc, &cam, &prm, &img));
}
