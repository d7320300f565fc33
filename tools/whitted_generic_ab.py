"""02, generic kernels (what a scene's FIRST render uses, before its specialised kernel exists): monolithic k_whitted_soft (flags 8|4096; spills, but evaluates the
colour bottom-up like the reference: bit-exact) vs the wavefront form (8|8192; no scratch, colour as a weighted sum: equal to ~1e-7)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
import numpy as np
from rtamd import lib, scenes
ctx = lib.Context()
def timed(fn, n=7):
    r = [fn() for _ in range(n)]; ms = sorted(x["kernel_ms"] for x in r); return r[-1], ms[0], ms[len(ms) // 2]
for w, h, ss in ((784, 561, 1), (784, 561, 16), (784, 561, 64), (3840, 2160, 16), (3840, 2160, 256)):
    s, cam, prm = scenes.scene02(ss, seed=1)
    ctx.set_debug_flags(8 | 4096); a, a_lo, a_med = timed(lambda: ctx.render_whitted(s, cam, prm, w, h)); fa = ctx.last_whitted_form()
    ctx.set_debug_flags(8 | 8192); b, b_lo, b_med = timed(lambda: ctx.render_whitted(s, cam, prm, w, h)); fb = ctx.last_whitted_form()
    ctx.set_debug_flags(0)
    rel = (np.abs(a["rgb"] - b["rgb"]) / np.maximum(np.abs(a["rgb"]), 1e-3)).max()
    print(f"02 {w}x{h} x{ss}: forms {fa}/{fb} rays equal {(a['rays'], a['shadow_rays']) == (b['rays'], b['shadow_rays'])} max rel {rel:.2e} | generic monolithic {a_lo:.4f} ms (median {a_med:.4f})  generic wavefront {b_lo:.4f} ms (median {b_med:.4f})", flush=True)
