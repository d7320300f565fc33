"""Item form: regeneration threshold (idle lanes per wave that trigger a regeneration pass) at steady clocks, thresholds interleaved twice.  python3 tools/sweep_regen_items.py"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
os.environ.setdefault("RT_MI355_JIT", "2")
import numpy as np
from rtamd import lib, scenes
ctx = lib.Context()
def warm():
    s, cam, prm = scenes.scene04(256)
    for _ in range(4): ctx.render_path(s, cam, prm, 3840, 2160)
for name, mk, w, h, spp, fpr in (("04", scenes.scene04, 3840, 2160, 256, 249), ("03", scenes.scene03, 1920, 1080, 256, 168)):
    s, cam, prm = mk(spp)
    ctx.render_path(s, cam, prm, w, h); ctx.render_path(s, cam, prm, w, h)
    for rnd in range(2):
        for thr in (1, 4, 8, 12, 16, 24, 32, 48):
            ctx.set_regen_threshold(thr)
            warm()
            ms = [ctx.render_path(s, cam, prm, w, h)["kernel_ms"] for _ in range(5)]
            print(f"{name} round {rnd} threshold {thr:2d}: min {min(ms):.3f} ms median {float(np.median(ms)):.3f} ms", flush=True)
    ctx.set_regen_threshold(-1)
