// RtBackend.hpp - what the four Render() wrappers share: the process-wide rt_ctx, error -> std::runtime_error
// (the reference's main() catches std::exception, 01_Basic/Main.cpp:230-233), and RGBQUAD for non-Windows hosts.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../../include/rt_mi355.h"
#include "ImageBuffer.hpp"

#ifndef _WINDOWS_
struct RGBQUAD { uint8_t rgbBlue, rgbGreen, rgbRed, rgbReserved; };   // wingdi.h layout: B, G, R, reserved
using BYTE = uint8_t;
#endif
static_assert(sizeof(RGBQUAD) == 4, "RGBQUAD must be 4 bytes: the kernels write B,G,R,255 straight into it");

namespace rtbackend {
// The GPUs the process-wide context spans: empty = the current device; several entries = a device group (rt_create_multi:
// sample ranges over the devices, one RCCL all-reduce, the first device resolves).  Set before the first Render() - by
// use_devices(), or by the environment: RT_MI355_DEVICES="0,1,2,3" or RT_MI355_GPUS=4 (devices 0..3).
inline std::vector<int32_t>& devices() {
    static std::vector<int32_t> d = [] {
        std::vector<int32_t> v;
        if (const char* e = std::getenv("RT_MI355_DEVICES")) {
            for (const char* p = e; *p;) { char* end = nullptr; long id = std::strtol(p, &end, 10); if (end == p) break; v.push_back(static_cast<int32_t>(id)); p = (*end == ',') ? end + 1 : end; }
        } else if (const char* e = std::getenv("RT_MI355_GPUS")) {
            for (int i = 0, n = std::atoi(e); i < n; ++i) v.push_back(i);
        }
        return v;
    }();
    return d;
}
inline void use_devices(const std::vector<int32_t>& ids) { devices() = ids; }
inline rt_ctx* context() {
    static rt_ctx* ctx = devices().size() > 1 ? rt_create_multi(devices().data(), static_cast<int32_t>(devices().size()))
                                              : rt_create(devices().empty() ? -1 : devices()[0]);
    if (!ctx) throw std::runtime_error(std::string("ERROR: no MI355X tracing backend: ") + rt_last_error(nullptr));
    return ctx;
}
inline void check(int32_t rc) { if (rc != RT_OK) throw std::runtime_error(std::string("ERROR: rt_mi355: ") + rt_last_error(context())); }
inline rt_image image_of(ImageBuffer<RGBQUAD>* fb) {
    return rt_image{reinterpret_cast<uint8_t*>(fb->getData()), nullptr, static_cast<int32_t>(fb->getWidth()), static_cast<int32_t>(fb->getHeight()), 0};
}
// Accumulates flat records; materials are shared by pointer identity so that two elements with one material object
// reference one rt_material (the reference shares materials through shared_ptr, 04 Main.cpp:181-199).
struct Flat {
    std::vector<rt_prim> prims; std::vector<rt_material> mats; std::vector<const void*> mat_keys; std::vector<rt_light> lights;
    int32_t material(const void* key, const rt_material& m) {
        if (key) for (size_t i = 0; i < mat_keys.size(); ++i) if (mat_keys[i] == key) return static_cast<int32_t>(i);
        mats.push_back(m); mat_keys.push_back(key);
        return static_cast<int32_t>(mats.size() - 1);
    }
    rt_scene scene() const {
        return rt_scene{prims.data(), static_cast<int32_t>(prims.size()), mats.data(), static_cast<int32_t>(mats.size()), lights.data(), static_cast<int32_t>(lights.size())};
    }
};
} // namespace rtbackend
