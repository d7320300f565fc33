"""One scene, a few renders, for profiling passes: python tools/perf_one.py 03|04 [flags]  (rocprofv3 --pmc ... -- python3 tools/perf_one.py 03)"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
from rtamd import lib, scenes
ctx = lib.Context()
which = sys.argv[1] if len(sys.argv) > 1 else "03"
ctx.set_debug_flags(int(sys.argv[2]) if len(sys.argv) > 2 else 16)
(s, cam, prm), (w, h) = (scenes.scene03(256), (1920, 1080)) if which == "03" else (scenes.scene04(256), (3840, 2160))
for rep in range(3):
    g = ctx.render_path(s, cam, prm, w, h)
    print(f"{which} {w}x{h} spp {prm.spp} kernel {ctx.last_path_kernel()[0]}: {g['kernel_ms']:.2f} ms  {g['rays'] / g['kernel_ms'] / 1e6:.2f} Grays/s  rays {g['rays']}", flush=True)
