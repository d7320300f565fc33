// clock_probe.hip - the shader clock MI355X sustains under a chip-wide FP32 FMA load: every wave runs a long chain of independent
// v_fma and reads the shader-clock counter (clock64) and the constant 100 MHz counter (wall_clock64) around it.
// Build: hipcc -O3 --offload-arch=gfx950 tools/clock_probe.hip -o tools/clock_probe ; run: tools/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k_load(float* out, long long* clk, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 0.999f, c = 1e-3f;
    long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll 8
        for (int k = 0; k < 8; ++k) {
            a0 = fmaf(a0, m, c); a1 = fmaf(a1, m, c); a2 = fmaf(a2, m, c); a3 = fmaf(a3, m, c);
            a4 = fmaf(a4, m, c); a5 = fmaf(a5, m, c); a6 = fmaf(a6, m, c); a7 = fmaf(a7, m, c);
        }
    }
    long long c1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}
int main() {
    const int blocks = 256 * 8, iters = 200000;      // 8 blocks of 4 waves per CU: 8 waves per SIMD
    float* out; long long* clk;
    hipMalloc(&out, blocks * 256 * sizeof(float)); hipMalloc(&clk, blocks * 2 * sizeof(long long));
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_load, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(blocks * 2);
        hipMemcpy(h.data(), clk, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        double sc = 0, wc = 0; for (int b = 0; b < blocks; ++b) { sc += h[2 * b]; wc += h[2 * b + 1]; }
        const double fma = (double)blocks * 256 * iters * 64.0;
        printf("rep %d: %.1f ms  shader clock %.0f MHz (clock64 / 100 MHz wall counter)  %.1f TFLOP/s FP32 (fma = 2)\n", rep, ms, sc / wc * 100.0, 2.0 * fma / (ms * 1e-3) / 1e12);
    }
    return 0;
}
