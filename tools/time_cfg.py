"""Kernel time of the path configurations at steady clocks: the chip is warmed with ~0.3 s of tracing first (a short kernel after an idle spell runs at
lower clocks: the first renders of a process measure up to 7 % slow), then every case is rendered 7 times: min / median.
python3 tools/time_cfg.py [case ...]   cases: 03 (1080p/256), 04q (4K/256), 04 (4K/1024), 03s / 04s (784x561/256), 03x (4K/1024); env switches (RT_MI355_UNIT_LOG2 ...) apply."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
os.environ.setdefault("RT_MI355_JIT", "2")
import numpy as np
from rtamd import lib, scenes
CASES = {"03": (scenes.scene03, 1920, 1080, 256, 168), "04q": (scenes.scene04, 3840, 2160, 256, 249), "04": (scenes.scene04, 3840, 2160, 1024, 249),
         "03s": (scenes.scene03, 784, 561, 256, 168), "04s": (scenes.scene04, 784, 561, 256, 249), "03x": (scenes.scene03, 3840, 2160, 1024, 168)}
ctx = lib.Context()
ctx.set_debug_flags(int(os.environ.get("TIME_CFG_FLAGS", "0")))


def warm():
    s, cam, prm = scenes.scene04(256)
    for _ in range(5):
        ctx.render_path(s, cam, prm, 3840, 2160)


def measure(name, reps=7):
    mk, w, h, spp, fpr = CASES[name]
    s, cam, prm = mk(spp)
    ctx.render_path(s, cam, prm, w, h); ctx.render_path(s, cam, prm, w, h)      # scene upload, specialised build
    warm()
    ms = []
    for _ in range(reps):
        g = ctx.render_path(s, cam, prm, w, h); ms.append(g["kernel_ms"])
    return g, float(np.min(ms)), float(np.median(ms)), fpr


if __name__ == "__main__":
    for name in (sys.argv[1:] or ["03", "04q", "04"]):
        g, lo, med, fpr = measure(name)
        mk, w, h, spp, _ = CASES[name]
        print(f"{name}: {w}x{h} spp {spp} form {ctx.last_path_form()} kernel {ctx.last_path_kernel()[0]}: min {lo:.3f} ms median {med:.3f} ms  ns/kray {lo * 1e6 / g['rays'] * 1e3:.3f}  "
              f"frac(min) {g['rays'] * fpr / (lo * 1e-3) / 157.3e12:.4f} frac(median) {g['rays'] * fpr / (med * 1e-3) / 157.3e12:.4f}", flush=True)
