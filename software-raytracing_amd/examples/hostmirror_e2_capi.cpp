// hostmirror_e2_capi.cpp - the host-side TraceTay / RandomVectorToLightSphere of 02_SoftShadows behind a C call for
// tests/test_host_mirror.py (mean of evaluations against the reference's converged image).  Test support only; no GPU.
#include "../host/02_SoftShadows/Render.hpp"
#include "../host/01_Basic/Triangle.hpp"
#define HM_FRAME_NAME hm_e2_tracetay_frame
#include "hostmirror_whitted_frame.inc"
extern "C" void hm_e2_settings(unsigned shadow_samples, unsigned seed) { TraceTayShadowSamples() = shadow_samples; SeedShadowEngine(seed); }
// n directions toward the light's disk from one point: out[3n]
extern "C" void hm_e2_light_vectors(const float* point, const float* light_pos, float radius, int n, float* out) {
    const LightSource l{v(light_pos), {1.0f, 1.0f, 1.0f}, radius};
    for (int i = 0; i < n; ++i) { const V3 d = RandomVectorToLightSphere(v(point), l); out[3 * i] = d.x; out[3 * i + 1] = d.y; out[3 * i + 2] = d.z; }
}
