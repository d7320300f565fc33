"""Cycles per primitive test of the tiled kernel by primitive type (spheres only / triangles only of the stress scene)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
from rtamd import lib, scenes, abi
ctx = lib.Context()
full, cam, _ = scenes.scene_stress(spp=4)
def subset(keep):
    s = scenes.Scene(); s.materials = list(full.materials)
    s.prims = [p for p in full.prims if keep(p)]
    return s
w, h = 1920, 1080
for name, keep in (("spheres", lambda p: p.type == 0), ("triangles", lambda p: p.type == 3), ("all", lambda p: True)):
    s = subset(keep)
    prm = scenes.path_params(abi.RT_PATH_LIGHTS, 2, max_depth=0, seed=1)     # primary rays only: every lane busy, 1 ray per sample
    for flags in (0, 2):
        ctx.set_debug_flags(flags)
        g = min((ctx.render_path(s, cam, prm, w, h) for _ in range(2)), key=lambda g: g["kernel_ms"])
        tests = g["rays"] * len(s.prims)
        cyc = 1024 * 2.37e9 * g["kernel_ms"] * 1e-3 / (tests / 64)
        print(f"{name:9s} flags {flags}: {len(s.prims)} prims  {g['kernel_ms']:.1f} ms  {tests / g['kernel_ms'] / 1e9:.3f} T tests/s  {cyc:.1f} SIMD-cycles per wave-test", flush=True)
