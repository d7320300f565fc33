"""Regeneration-threshold sweep on the stress scene (many primitives: an iteration costs far more than the camera code)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
from rtamd import lib, scenes
ctx = lib.Context()
s, cam, prm = scenes.scene_stress(spp=4)
for thr in (1, 4, 16):
    ctx.set_regen_threshold(thr)
    g = ctx.render_path(s, cam, prm, 1920, 1080)
    print(f"stress thr {thr:2d}: {g['kernel_ms']:.1f} ms  {g['rays'] / g['kernel_ms'] / 1e3:.1f} Mrays/s", flush=True)
