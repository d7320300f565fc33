"""Regeneration-threshold sweep of k_path (idle lanes per wave that trigger a camera pass)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
from rtamd import lib, scenes
ctx = lib.Context()
for name, (s, cam, prm), (w, h) in [("04", scenes.scene04(256), (3840, 2160)), ("03", scenes.scene03(256), (1920, 1080))]:
    for thr in [int(x) for x in sys.argv[1:]] or [4, 8, 12, 16, 20, 24, 32]:
        ctx.set_regen_threshold(thr)
        best = min((ctx.render_path(s, cam, prm, w, h) for _ in range(3)), key=lambda g: g["kernel_ms"])
        print(f"{name} thr {thr:2d}: {best['kernel_ms']:.2f} ms  {best['rays'] / best['kernel_ms'] / 1e6:.2f} Grays/s", flush=True)
