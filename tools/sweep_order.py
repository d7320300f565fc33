"""Item form: order in which units are handed out (RT_MI355_UNIT_ORDER: 0 row-major, 1 bottom-up, 2 strided) x unit size.  python3 tools/sweep_order.py"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
os.environ.setdefault("RT_MI355_JIT", "2")
from rtamd import lib, scenes
ctx = lib.Context()
cases = (("03", scenes.scene03, 1920, 1080, 256, 168), ("04", scenes.scene04, 3840, 2160, 256, 249), ("04", scenes.scene04, 784, 561, 256, 249), ("03", scenes.scene03, 784, 561, 256, 168))
for name, mk, w, h, spp, fpr in cases:
    s, cam, prm = mk(spp)
    for order in (0, 1, 2):
        os.environ["RT_MI355_UNIT_ORDER"] = str(order)
        for k in (None, 3, 4, 5):
            if k is None: os.environ.pop("RT_MI355_UNIT_LOG2", None)
            else: os.environ["RT_MI355_UNIT_LOG2"] = str(k)
            ms = []
            for rep in range(3):
                g = ctx.render_path(s, cam, prm, w, h); ms.append(g["kernel_ms"])
            print(f"{name} {w}x{h} spp {spp} order {order} unit_log2 {k} (used {ctx.last_path_form()}): {min(ms):.3f} ms (runs {' '.join('%.3f' % m for m in ms)}) ns/kray {min(ms) * 1e6 / g['rays'] * 1e3:.3f} frac {g['rays'] * fpr / (min(ms) * 1e-3) / 157.3e12:.4f}", flush=True)
