// Build and run (CPU only; the instrumented library goes to /tmp, not into the tree):
//   cd software-raytracing_amd && hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast-honor-pragmas -fno-slp-vectorize \
//       -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -shared -o /tmp/asan/librt_mi355.so csrc/rt_api.hip csrc/rt_kernels.hip csrc/rt_jit.hip -lhiprtc
//   clang++ -O1 -g -std=c++17 -fsanitize=address,undefined -Iinclude tools/host_asan_check.cpp -o /tmp/asan/check -L/tmp/asan -lrt_mi355 -Wl,-rpath,/tmp/asan -Wl,-rpath,/opt/rocm/lib
//   ASAN_OPTIONS=detect_leaks=0 /tmp/asan/check      (round 2: 300 scenes up to 20,000 primitives, no report; round 3: + 200 scenes built around coplanar elements, no report)
// Host-side sanitizer check of the scene compiler and the BVH builder: random scenes of every primitive type through
// rt_debug_bvh_probe (CPU only: no GPU call is made).  Built with -fsanitize=address,undefined together with an
// ASan-instrumented host build of librt_mi355.so.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>
#include "rt_mi355.h"
extern "C" int32_t rt_debug_bvh_probe(const rt_scene*, int32_t, int32_t*);
extern "C" int32_t rt_debug_group_share(int32_t, int32_t, int32_t, int32_t, int32_t*, int32_t*);
int main() {
    std::mt19937 g(7);
    std::uniform_real_distribution<float> U(-1.0f, 1.0f);
    long bad = 0, scenes = 0;
    for (int it = 0; it < 300; ++it) {
        const int n = 1 + (int)(g() % (it < 250 ? 200 : 20000));
        std::vector<rt_material> mats(4); memset(mats.data(), 0, mats.size() * sizeof(rt_material));
        mats[0].type = RT_MAT_DIFFUSE; mats[1].type = RT_MAT_METAL; mats[2].type = RT_MAT_DIELECTRIC; mats[2].refraction_index = 0.6f; mats[3].type = RT_MAT_LIGHT;
        std::vector<rt_prim> prims((size_t)n); memset(prims.data(), 0, prims.size() * sizeof(rt_prim));
        for (auto& P : prims) {
            const int kind = (int)(g() % 5);
            P.material = (int32_t)(g() % 5) - 1;
            float* f = P.f;
            for (int k = 0; k < 11; ++k) f[k] = 0.0f;
            f[0] = 20 * U(g); f[1] = 20 * U(g); f[2] = 20 * U(g);
            if (kind == 0) { P.type = RT_PRIM_SPHERE; f[3] = std::pow(10.0f, 2.5f * U(g)); P.flags = (g() % 8 == 0) ? RT_PRIM_FLAG_FLIP_NORMALS : 0; }
            else if (kind == 1) { P.type = RT_PRIM_PLANE; float nx = U(g), ny = U(g), nz = U(g), l = std::sqrt(nx * nx + ny * ny + nz * nz) + 1e-9f; if (g() % 2) { nx = 0; ny = 1; nz = 0; l = 1; } f[3] = nx / l; f[4] = ny / l; f[5] = nz / l; }
            else if (kind == 2) { P.type = RT_PRIM_RECT; f[3] = 0.01f + 5 * std::fabs(U(g)); f[4] = 0.01f + 5 * std::fabs(U(g)); f[5] = 180 * U(g); f[6] = 180 * U(g); f[7] = 180 * U(g); }
            else if (kind == 3) { P.type = RT_PRIM_BOX; f[3] = 0.01f + 5 * std::fabs(U(g)); f[4] = 0.01f + 5 * std::fabs(U(g)); f[5] = 0.01f + 5 * std::fabs(U(g)); f[6] = 90 * U(g); f[7] = 90 * U(g); f[8] = 90 * U(g); }
            else { P.type = RT_PRIM_TRIANGLE; for (int k = 3; k < 9; ++k) f[k] = f[k % 3] + 3 * U(g); if (g() % 50 == 0) { f[3] = f[0]; f[4] = f[1]; f[5] = f[2]; } }
        }
        rt_scene sc{prims.data(), n, mats.data(), (int32_t)mats.size(), nullptr, 0};
        int32_t out[5] = {0, 0, 0, 0, 0};
        for (int variant = 3; variant <= 4; ++variant) {
            const int rc = rt_debug_bvh_probe(&sc, variant, out);
            if (rc != 0) { ++bad; std::printf("scene %d variant %d: probe returned %d (%s)\n", it, variant, rc, rt_last_error(nullptr)); }
        }
        ++scenes;
    }
    // degenerate inputs
    rt_scene empty{nullptr, 0, nullptr, 0, nullptr, 0}; int32_t out[5];
    std::printf("empty scene: %d\n", rt_debug_bvh_probe(&empty, 4, out));
    rt_prim p{}; p.type = 99; rt_scene badtype{&p, 1, nullptr, 0, nullptr, 0};
    std::printf("unknown type: %d\n", rt_debug_bvh_probe(&badtype, 4, out));
    p.type = RT_PRIM_SPHERE; p.material = 5;
    std::printf("material out of range: %d\n", rt_debug_bvh_probe(&badtype, 4, out));
    for (int n = 1; n <= 9; ++n) for (int i = 0; i < n; ++i) { int32_t b, e; rt_debug_group_share(0, 1000003, i, n, &b, &e); }
    // round 3: scenes built AROUND coplanar planar elements (floor planes, boxes resting on them at random turns about the vertical, rectangles and triangles lying in
    // the floor or in each other's planes, duplicates): the detection of compile_scene (find_coplanar: outlines, separating axes, pair budget) under the sanitizers
    for (int it = 0; it < 200; ++it) {
        std::vector<rt_material> mats(2); memset(mats.data(), 0, mats.size() * sizeof(rt_material));
        mats[0].type = RT_MAT_DIFFUSE; mats[1].type = RT_MAT_DIELECTRIC; mats[1].refraction_index = 0.66f;
        const int n = 2 + (int)(g() % (it < 190 ? 40 : 3000));
        std::vector<rt_prim> prims((size_t)n); memset(prims.data(), 0, prims.size() * sizeof(rt_prim));
        const float floor_y = -2.0f;
        prims[0].type = RT_PRIM_PLANE; prims[0].f[1] = floor_y; prims[0].f[4] = 1.0f;
        for (int k = 1; k < n; ++k) {
            rt_prim& P = prims[(size_t)k]; float* f = P.f;
            P.material = (int32_t)(g() % 2);
            const float cx = 8.0f * U(g), cz = -6.0f + 5.0f * U(g), sx = 0.2f + 1.5f * std::fabs(U(g)), sy = 0.2f + 1.5f * std::fabs(U(g)), sz = 0.2f + 1.5f * std::fabs(U(g));
            switch (g() % 4) {
            case 0: P.type = RT_PRIM_BOX; f[0] = cx; f[1] = floor_y + sy / 2.0f; f[2] = cz; f[3] = sx; f[4] = sy; f[5] = sz; f[7] = (g() % 3) ? 90.0f * U(g) : 0.0f; break;
            case 1: P.type = RT_PRIM_RECT; f[0] = cx; f[1] = floor_y; f[2] = cz; f[3] = sx; f[4] = sz; f[5] = 90.0f; f[6] = (g() % 2) ? 45.0f * U(g) : 0.0f; break;
            case 2: P.type = RT_PRIM_TRIANGLE; f[0] = cx; f[1] = floor_y; f[2] = cz; f[3] = cx + sx; f[4] = floor_y; f[5] = cz; f[6] = cx; f[7] = floor_y; f[8] = cz - sz; break;
            default: prims[(size_t)k] = prims[(size_t)(1 + g() % (unsigned)k) % (size_t)k == 0 ? 0 : (size_t)(g() % (unsigned)k)]; break;      // a duplicate of an earlier element
            }
        }
        rt_scene sc{}; sc.prims = prims.data(); sc.n_prims = n; sc.materials = mats.data(); sc.n_materials = 2;
        for (int variant = 3; variant <= 4; ++variant) {
            int32_t out[5] = {0, 0, 0, 0, 0};
            const int rc = rt_debug_bvh_probe(&sc, variant, out);
            ++scenes;
            if (rc != 0) { ++bad; std::printf("coplanar scene %d variant %d: probe returned %d (%s)\n", it, variant, rc, rt_last_error(nullptr)); }
        }
    }
    std::printf("%ld scenes, %ld probe failures\n", scenes, bad);
    return bad ? 1 : 0;
}
