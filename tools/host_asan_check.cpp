// Build and run (CPU only; the instrumented library goes to /tmp, not into the tree):
//   cd software-raytracing_amd && hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -fno-slp-vectorize \
//       -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -shared -o /tmp/asan/librt_mi355.so csrc/rt_api.hip csrc/rt_kernels.hip csrc/rt_jit.hip -lhiprtc
//   clang++ -O1 -g -std=c++17 -fsanitize=address,undefined -Iinclude tools/host_asan_check.cpp -o /tmp/asan/check -L/tmp/asan -lrt_mi355 -Wl,-rpath,/tmp/asan -Wl,-rpath,/opt/rocm/lib
//   ASAN_OPTIONS=detect_leaks=0 /tmp/asan/check      (round 2: 300 scenes up to 20,000 primitives, no report)
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
    std::printf("%ld scenes, %ld probe failures\n", scenes, bad);
    return bad ? 1 : 0;
}
