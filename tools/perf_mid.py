"""Mid-size scenes (tens to hundreds of primitives): generic LDS kernel (flag 8) vs scene-specialised (flag 16).  python tools/perf_mid.py [n_spheres grid]"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
from rtamd import lib, scenes
ctx = lib.Context()
ns, grid = (int(x) for x in (sys.argv[1:3] if len(sys.argv) > 2 else (150, 4)))
s, cam, prm = scenes.scene_stress(n_spheres=ns, grid=grid, spp=16)
for flags in (8, 16):
    ctx.set_debug_flags(flags)
    t0 = time.time(); g = ctx.render_path(s, cam, prm, 1920, 1080); first = time.time() - t0
    best = min((ctx.render_path(s, cam, prm, 1920, 1080) for _ in range(2)), key=lambda g: g["kernel_ms"])
    print(f"{len(s.prims)} prims flags {flags} kernel {ctx.last_path_kernel()}: {best['kernel_ms']:.2f} ms  {best['rays'] / best['kernel_ms'] / 1e6:.2f} Grays/s  (first call {first:.2f} s)", flush=True)
