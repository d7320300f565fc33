// Error distribution of the square roots the sphere test can use, in ulps against the correctly rounded result:
// v_sqrt_f32 (hardware, 1 ulp), x * v_rsq_f32(x), the refined form of rt_device.h (one Newton step), and sqrtf as hipcc
// compiles it by default.  hipcc --offload-arch=gfx950 -O3 tools/sqrt_bias.hip -o tools/sqrt_bias
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
__global__ void k(const float* x, float* a, float* b, float* c, float* d, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = x[i];
    a[i] = __builtin_amdgcn_sqrtf(v);
    float r = __builtin_amdgcn_rsqf(v);
    b[i] = v * r;
    float s = v * r;
    c[i] = fmaf(fmaf(-s, s, v), 0.5f * r, s);
    d[i] = sqrtf(v);
}
int main() {
    const int n = 1 << 24;
    std::vector<float> h(n);
    uint64_t st = 88172645463325252ULL;
    for (int i = 0; i < n; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; h[i] = 1.0f + 3.0f * (float)((st >> 40) * (1.0 / 16777216.0)); }   // [1, 4): both halves of sqrt's period
    float *x, *o[4];
    hipMalloc(&x, n * 4); for (auto& p : o) hipMalloc(&p, n * 4);
    hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(x, o[0], o[1], o[2], o[3], n);
    const char* names[4] = {"v_sqrt_f32", "x * v_rsq_f32(x)", "refined (rt_device.h)", "sqrtf (hipcc default)"};
    std::vector<float> r(n);
    for (int v = 0; v < 4; ++v) {
        hipMemcpy(r.data(), o[v], n * 4, hipMemcpyDeviceToHost);
        double sum = 0, sum2 = 0; long exact = 0, hi = 0, lo = 0; double worst = 0;
        for (int i = 0; i < n; ++i) {
            const double t = std::sqrt((double)h[i]);
            const float cr = (float)t;                               // correctly rounded (double sqrt then one rounding; no double-rounding issue at these widths for sqrt)
            int32_t ia, ib; memcpy(&ia, &r[i], 4); memcpy(&ib, &cr, 4);
            const int dulp = ia - ib;
            exact += dulp == 0; hi += dulp > 0; lo += dulp < 0;
            const double ulp = std::ldexp(1.0, std::ilogb(t) - 23);
            const double e = ((double)r[i] - t) / ulp;               // signed error against the TRUE root, in ulps
            sum += e; sum2 += e * e; if (std::fabs(e) > worst) worst = std::fabs(e);
        }
        std::printf("%-24s correctly rounded %.4f, above %.4f, below %.4f;  error vs true root: mean %+.4f ulp, rms %.4f, worst %.3f\n", names[v], (double)exact / n, (double)hi / n, (double)lo / n,
                    sum / n, std::sqrt(sum2 / n), worst);
    }
    return 0;
}
