"""Wave-level event counts of the item form (experiment defines RT_STAT_* of rt_path_device.h, one per process: the define is part of the generated source).
python3 tools/pool_stats.py 04|03 ITER|PASS [threshold]"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
os.environ["RT_MI355_JIT"] = "2"; os.environ["RT_MI355_JIT_DEFS"] = "RT_STAT_" + sys.argv[2]
from rtamd import lib, scenes
ctx = lib.Context()
if len(sys.argv) > 3: ctx.set_regen_threshold(int(sys.argv[3]))
which = sys.argv[1]
(s, cam, prm), (w, h) = ((scenes.scene03(256), (1920, 1080)) if which == "03" else (scenes.scene04(256), (3840, 2160)))
r = ctx.render_path(s, cam, prm, w, h)
print(f"{which} {sys.argv[2]:6s}: {r['shadow_rays']} wave-events; rays {r['rays']} = {r['rays'] / 64:.0f} x 64; events per 64 rays {r['shadow_rays'] / (r['rays'] / 64):.4f}; paths per event {w * h * 256 / max(1, r['shadow_rays']):.2f}", flush=True)
