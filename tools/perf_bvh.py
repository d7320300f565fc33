"""Throughput of the optional BVH walk (rt_set_option(RT_OPT_ACCEL, RT_ACCEL_BVH)) on the stress scene (SURVEY 8(d) config 5
geometry): the scheduled kernel, its one-walk-per-lane A/B reference (debug flag 32) and, for scale, brute force on a smaller
frame of the same scene.  python tools/perf_bvh.py [w h spp]"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
import torch  # noqa: F401  (its HIP runtime first, see README)
from rtamd import abi, lib, scenes
ctx = lib.Context()
w, h, spp = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (1920, 1080, 16)))
s, cam, prm = scenes.scene_stress(spp=spp)
ctx.set_option(abi.RT_OPT_ACCEL, abi.RT_ACCEL_BVH)
for flags, name in ((0, "scheduled"), (32, "simple")):
    ctx.set_debug_flags(flags)
    best = None
    for rep in range(3):
        g = ctx.render_path(s, cam, prm, w, h)
        best = g if best is None or g["kernel_ms"] < best["kernel_ms"] else best
    print(f"bvh {name:9s} {w}x{h} spp {spp}: {best['kernel_ms']:.1f} ms  {best['rays']} rays  {best['rays'] / best['kernel_ms'] / 1e3:.1f} Mrays/s  ({best['rays'] / best['samples']:.3f} rays/sample), call {best['total_ms']:.1f} ms", flush=True)
ctx.set_debug_flags(0)
ctx.set_option(abi.RT_OPT_ACCEL, abi.RT_ACCEL_NONE)
bw, bh, bspp = 480, 270, 2
s2, cam2, prm2 = scenes.scene_stress(spp=bspp)
g = ctx.render_path(s2, cam2, prm2, bw, bh)
print(f"brute  {bw}x{bh} spp {bspp}: {g['kernel_ms']:.1f} ms  {g['rays']} rays  {g['rays'] / g['kernel_ms'] / 1e3:.2f} Mrays/s", flush=True)
