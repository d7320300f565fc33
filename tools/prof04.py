"""One 04 render at 4K/64 spp (profiling target for rocprofv3 --pmc passes)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
from rtamd import lib, scenes
ctx = lib.Context()
s, cam, prm = scenes.scene04(int(os.environ.get("SPP", "64")))
g = ctx.render_path(s, cam, prm, 3840, 2160)
print(g["kernel_ms"], g["rays"], g["samples"])
