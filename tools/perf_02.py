"""02 at 4K / 256 shadow samples, three renders (for rocprofv3 passes)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
os.environ["RT_MI355_JIT"] = "2"
from rtamd import lib, scenes
ctx = lib.Context()
s, cam, prm = scenes.scene02(256, seed=1)
for _ in range(3):
    r = ctx.render_whitted(s, cam, prm, 3840, 2160)
print(f"02 4K/256: kernel {ctx.last_whitted_kernel()} {r['kernel_ms']:.3f} ms {(r['rays'] + r['shadow_rays']) / r['kernel_ms'] / 1e6:.1f} Grays/s")
