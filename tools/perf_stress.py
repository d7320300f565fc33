"""Throughput of the global-memory kernel on the stress scene at a reduced frame."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
from rtamd import lib, scenes
ctx = lib.Context()
w, h, spp = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (480, 270, 4)))
t0 = time.time(); s, cam, prm = scenes.scene_stress(spp=spp); print(f"scene built in {time.time() - t0:.1f} s, {len(s.prims)} prims", flush=True)
fpr = scenes.flops_per_ray(s, 100)
for flags in (0, 2):
  ctx.set_debug_flags(flags)
  best = None
  for rep in range(2):
    g = ctx.render_path(s, cam, prm, w, h)
    best = g if best is None or g["kernel_ms"] < best["kernel_ms"] else best
  r = best["rays"] / best["kernel_ms"] / 1e6
  print(f"stress {w}x{h} spp {spp} flags {flags}: {best['kernel_ms']:.1f} ms  {best['rays']} rays  {r * 1e3:.1f} Mrays/s  {r * fpr / 1e3 / 157.3 * 100:.1f}% of FP32 peak at {fpr} flops/ray  ({best['rays'] / best['samples']:.3f} rays/sample), call {best['total_ms']:.1f} ms", flush=True)
