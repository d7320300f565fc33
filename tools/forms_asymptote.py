"""Forms of the bounce loop on a long frame (03 at 4K / 1024 spp), where launch ramp, drain and tail do not count: item form (0), one lane per
pixel (512), outcome-sorted (2048), queue (256).  python3 tools/forms_asymptote.py"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
os.environ.setdefault("RT_MI355_JIT", "2")
from rtamd import lib, scenes
ctx = lib.Context()
for name, mk, w, h, spp, fpr in (("03", scenes.scene03, 3840, 2160, 1024, 168), ("03", scenes.scene03, 1920, 1080, 256, 168)):
    s, cam, prm = mk(spp)
    for flags in (0, 512, 2048, 256):
        ctx.set_debug_flags(flags)
        ms = []
        for rep in range(3):
            g = ctx.render_path(s, cam, prm, w, h); ms.append(g["kernel_ms"])
        print(f"{name} {w}x{h} spp {spp} flags {flags} form {ctx.last_path_form()}: {min(ms):.3f} ms ns/kray {min(ms) * 1e6 / g['rays'] * 1e3:.3f} frac {g['rays'] * fpr / (min(ms) * 1e-3) / 157.3e12:.4f}", flush=True)
