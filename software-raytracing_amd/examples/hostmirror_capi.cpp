// hostmirror_capi.cpp - C wrappers around host/Common/{Math,Ray}.hpp so that tests/test_host_mirror.py can check the
// host-side mirror bit for bit against the vectors recorded from the reference's own headers (tests/golden/kat_*.npz).
// Test support only; not part of librt_mi355.so.
#include "../host/Common/Math.hpp"
#include "../host/Common/Ray.hpp"
#include "../host/Common/Present.hpp"

using V = math::Vec3<float>;
static V v(const float* p) { return {p[0], p[1], p[2]}; }
// the recorded directions are already unit vectors: take them as given, as the generating harness did
static math::Ray ray(const float* o, const float* d) { math::Ray r; r.setOrigin(v(o)); r.setDirection(v(d), false); return r; }
static void put(const V& a, float* o) { o[0] = a.x; o[1] = a.y; o[2] = a.z; }

extern "C" {
void hm_normalize(const float* a, float* o) { put(math::Normalize(v(a)), o); }
void hm_reflect(const float* d, const float* n, float* o) { put(math::Reflect(v(d), v(n)), o); }
void hm_refract(const float* d, const float* n, float eta, int reflect_on_tir, float* o) { put(math::Refract(v(d), v(n), eta, reflect_on_tir != 0), o); }
void hm_mix(const float* a, const float* b, float r, float* o) { put(math::Mix(v(a), v(b), r), o); }
float hm_clamp(float x, float lo, float hi) { return math::Clamp(x, lo, hi); }
void hm_rotation_mat(const float* deg, float* o9) { auto m = math::GetRotationMat(v(deg)); for (int i = 0; i < 9; ++i) o9[i] = m.data[i]; }
void hm_ray_ctor(const float* d, float* o) { put(math::Ray({0, 0, 0}, v(d)).getDirection(), o); }
int hm_ray_sphere(const float* o, const float* d, const float* c, float r, float tmin, float tmax, float* t) { return ray(o, d).intersectsSphere(v(c), r, tmin, tmax, t); }
int hm_ray_plane(const float* o, const float* d, const float* n, const float* p0, float tmin, float tmax, float* t) { return ray(o, d).intersectsPlane(v(n), v(p0), tmin, tmax, t); }
int hm_ray_rect_xy(const float* o, const float* d, float z, float x0, float x1, float y0, float y1, float tmin, float tmax, float* t) {
    return ray(o, d).intersectsAARectangleXy(z, x0, x1, y0, y1, tmin, tmax, t);
}
int hm_ray_triangle(const float* o, const float* d, const float* v0, const float* v1, const float* v2, float tmin, float tmax, float* t, float* bary) {
    math::Vec2<float> b;
    int h = ray(o, d).intersectsTriangle(v(v0), v(v1), v(v2), tmin, tmax, t, &b);
    if (h) { bary[0] = b.x; bary[1] = b.y; }
    return h;
}
// Present.hpp sinks: 'times' presents of one frame whose first byte is bumped per present; 0 on success, 1 on an exception
int hm_present(const char* path, int bmp, unsigned char* pixels, int w, int h, int times) {
    try {
        if (bmp) { BmpFileSink sink(path); for (int i = 0; i < times; ++i) { pixels[0] = static_cast<unsigned char>(i); PresentFrame(pixels, w, h, sink); } return sink.framesPresented() == static_cast<unsigned>(times) ? 0 : 1; }
        RawStreamSink sink(path); for (int i = 0; i < times; ++i) { pixels[0] = static_cast<unsigned char>(i); PresentFrame(pixels, w, h, sink); }
        return 0;
    } catch (std::exception&) { return 1; }
}
}
