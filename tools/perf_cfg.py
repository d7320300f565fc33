"""One configuration of the path kernels, a few renders (for rocprofv3 passes): python3 tools/perf_cfg.py 04|03 <gen><r|q> [spp]"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
which, g = sys.argv[1], sys.argv[2]
os.environ["RT_MI355_JIT_GEN"] = g[0]; os.environ["RT_MI355_PATH"] = g[1:] or "q"; os.environ["RT_MI355_JIT"] = "2"
from rtamd import lib, scenes
ctx = lib.Context()
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 256
if len(sys.argv) > 4: os.environ["RT_MI355_JIT_SKIP"] = sys.argv[4]
if len(sys.argv) > 5: os.environ["RT_MI355_JIT_DEFS"] = sys.argv[5]
if len(sys.argv) > 6: ctx.set_regen_threshold(int(sys.argv[6]))
(s, cam, prm), (w, h), fpr = ((scenes.scene03(spp), (1920, 1080), 168) if which == "03" else (scenes.scene04(spp), (3840, 2160), 249))
for rep in range(2):
    r = ctx.render_path(s, cam, prm, w, h)
    print(f"{which} {g} {w}x{h} spp {spp} kernel {ctx.last_path_kernel()[0]}: {r['kernel_ms']:.2f} ms {r['rays'] / r['kernel_ms'] / 1e6:.1f} Grays/s frac {r['rays'] * fpr / (r['kernel_ms'] * 1e-3) / 157.3e12:.3f} rays {r['rays']} ns/kray {r['kernel_ms'] * 1e6 / r['rays'] * 1e3:.3f}", flush=True)
