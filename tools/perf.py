"""Quick throughput probe of the path kernels (kernel HIP-event time, rays counted by the kernel)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
from rtamd import lib, scenes
ctx = lib.Context()
thr = [-1]
flags = [int(x) for x in sys.argv[1:]] or [0]
for name, (s, cam, prm), (w, h), fpr in [("04", scenes.scene04(256), (3840, 2160), 249), ("03", scenes.scene03(256), (1920, 1080), 168)]:
    for t in flags:
        ctx.set_debug_flags(t)
        best = None
        for rep in range(3):
            g = ctx.render_path(s, cam, prm, w, h)
            best = g if best is None or g["kernel_ms"] < best["kernel_ms"] else best
        r = best["rays"] / best["kernel_ms"] / 1e6
        print(f"{name} {w}x{h} spp {prm.spp} flags {t} kernel {ctx.last_path_kernel()[0]}: {best['kernel_ms']:.2f} ms  {r:.2f} Grays/s  {r*fpr/1e3/157.3*100:.1f}% of FP32 peak  ({best['rays']/best['samples']:.3f} rays/sample)", flush=True)
