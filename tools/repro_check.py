"""Item form: the same frame from run to run?  Renders 03 and 04 three times each (fresh unit schedule every time) and compares the float accumulators bit for bit.
python3 tools/repro_check.py"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
os.environ.setdefault("RT_MI355_JIT", "2")
import numpy as np
from rtamd import lib, scenes
ctx = lib.Context()
for name, mk, w, h, spp in (("03", scenes.scene03, 1920, 1080, 64), ("04", scenes.scene04, 1920, 1080, 64), ("03 generic", scenes.scene03, 960, 540, 64)):
    s, cam, prm = mk(spp)
    ctx.set_debug_flags(8 if "generic" in name else 0)
    frames = [ctx.render_path(s, cam, prm, w, h) for _ in range(3)]
    ctx.set_debug_flags(0)
    print(name, "form", ctx.last_path_form(), "kernel", ctx.last_path_kernel()[0], "rays equal", frames[0]["rays"] == frames[1]["rays"] == frames[2]["rays"],
          "rgb bit-identical", bool(np.array_equal(frames[0]["rgb"], frames[1]["rgb"]) and np.array_equal(frames[0]["rgb"], frames[2]["rgb"])), flush=True)
