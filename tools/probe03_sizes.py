"""03 at several frame sizes / sample counts: does the 1080p/256-spp figure carry fixed costs (ramp, tail)?  python3 tools/probe03_sizes.py [flags]"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
os.environ.setdefault("RT_MI355_JIT", "2")
from rtamd import lib, scenes
ctx = lib.Context()
ctx.set_debug_flags(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
for (w, h, spp) in ((1920, 1080, 64), (1920, 1080, 256), (1920, 1080, 1024), (3840, 2160, 256), (3840, 2160, 1024)):
    s, cam, prm = scenes.scene03(spp)
    for rep in range(3):
        g = ctx.render_path(s, cam, prm, w, h)
    print(f"03 {w}x{h} spp {spp} kernel {ctx.last_path_kernel()[0]}: {g['kernel_ms']:.3f} ms rays {g['rays']} ns/kray {g['kernel_ms'] * 1e6 / g['rays'] * 1e3:.3f} frac(168) {g['rays'] * 168 / (g['kernel_ms'] * 1e-3) / 157.3e12:.4f}", flush=True)
