"""01: generic ahead-of-time kernel (a) vs hiprtc build of the same body with the generic strict loops (b: RT_MI355_JIT_NO_STRICT=1) vs generated strict queries (c)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
import numpy as np
from rtamd import lib, scenes
ctx = lib.Context()
s, cam, prm = scenes.scene01()
w, h = 784, 561
ctx.set_debug_flags(8); a = ctx.render_whitted(s, cam, prm, w, h)
ctx.set_debug_flags(16)
os.environ["RT_MI355_JIT_NO_STRICT"] = "1"; b = ctx.render_whitted(s, cam, prm, w, h); kb = ctx.last_whitted_kernel(); nb = ctx.last_path_kernel()[1]
os.environ.pop("RT_MI355_JIT_NO_STRICT")
c = ctx.render_whitted(s, cam, prm, w, h); kc = ctx.last_whitted_kernel()
ctx.set_debug_flags(0)
print("kernels", kb, kc, nb[:200])
for n1, x, n2, y in (("a", a, "b", b), ("a", a, "c", c), ("b", b, "c", c)):
    d = np.abs(x["rgb"] - y["rgb"]).max(axis=2)
    print(n1, n2, "rays", x["rays"], y["rays"], "shadow", x["shadow_rays"], y["shadow_rays"], "pixels differing", int((d > 0).sum()), "max abs", float(d.max()), "bgra differing", int((x["bgra"] != y["bgra"]).any(axis=2).sum()))
    ys, xs = np.nonzero(d > 0)
    for k in range(min(5, len(ys))):
        print("   px", xs[k], ys[k], x["rgb"][ys[k], xs[k]], y["rgb"][ys[k], xs[k]])
