"""Throughput of the Whitted kernels: 01 at 4K and the 02 shadow-sample sweep."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
from rtamd import lib, scenes
ctx = lib.Context()
ctx.set_debug_flags(int(os.environ.get("PERF_FLAGS", "0")))   # 16: scene-specialised 02 kernel at the first render, 8: never
w, h = 3840, 2160
s, cam, prm = scenes.scene01()
g = min((ctx.render_whitted(s, cam, prm, w, h) for _ in range(3)), key=lambda g: g["kernel_ms"])
n = g["rays"] + g["shadow_rays"]
print(f"01 {w}x{h}: {g['kernel_ms']:.2f} ms  {n / g['kernel_ms'] / 1e6:.2f} Grays/s  ({g['rays']} + {g['shadow_rays']} shadow)", flush=True)
for ss in [int(x) for x in sys.argv[1:]] or [1, 16, 256]:
    s, cam, prm = scenes.scene02(ss)
    g = min((ctx.render_whitted(s, cam, prm, w, h) for _ in range(2)), key=lambda g: g["kernel_ms"])
    n = g["rays"] + g["shadow_rays"]
    fl = (g["rays"] * 189 + g["shadow_rays"] * 134) / (g["kernel_ms"] * 1e-3) / 1e12
    print(f"02 {w}x{h} shadow samples {ss} kernel {ctx.last_whitted_kernel()}: {g['kernel_ms']:.2f} ms  {n / g['kernel_ms'] / 1e6:.2f} Grays/s  {fl / 157.3 * 100:.1f}% of FP32 peak", flush=True)
