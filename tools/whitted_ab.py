"""Wavefront form of the Whitted kernels (walk / one thread per shadow ray / combine) vs the monolithic kernel (the default; wavefront = debug flag 8192)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
os.environ.setdefault("RT_MI355_JIT", "2")
import numpy as np
from rtamd import abi, lib, scenes
ctx = lib.Context()
def mirror_scene():
    s, cam, prm = scenes.scene01()
    half = s.phong((0.2, 0.3, 0.2), 0.8, 64.0, 0.3, 0.5, 0.7)      # reflects AND refracts: the recursion tree branches
    s.sphere(half, (0, 2, -12), 2.5)
    s.light((6, 5, -6), (0.4, 0.4, 0.5))
    return s, cam, scenes.whitted_params(6, 0)
def open_scene(samples):
    """02's sphere and floor under the sky: no walls, no ceiling - about half of the primary rays miss (no shadow rays for them)"""
    s = scenes.Scene()
    white = s.phong((1, 1, 1), 0.0); rubber = s.phong((0.6, 0.2, 0.2), 0.1, 16.0)
    s.plane(white, (0, -10, 0), (0, 1, 0)); s.sphere(rubber, (0, -6.5, -15), 3.0)
    s.light((0, 6.5, -10), (0.9, 0.9, 0.9), 3.0)
    return s, scenes.camera(90.0), scenes.whitted_params(4, samples, 1)
cases = [("open scene 4K x64", open_scene(64), (3840, 2160), 17 + 12 + 100, 29 + 45), ("open scene 784x561 x16", open_scene(16), (784, 561), 129, 74),
         ("01 784x561", scenes.scene01(), (784, 561), 223, 123), ("02 784x561 x16", scenes.scene02(16), (784, 561), 189, 134), ("02 784x561 x1", scenes.scene02(1), (784, 561), 189, 134),
         ("02 784x561 x64", scenes.scene02(64), (784, 561), 189, 134), ("branching, 2 lights, depth 6", mirror_scene(), (300, 200), 223, 123),
         ("01 4K", scenes.scene01(), (3840, 2160), 223, 123), ("02 4K x16", scenes.scene02(16), (3840, 2160), 189, 134), ("02 4K x256", scenes.scene02(256), (3840, 2160), 189, 134)]
for name, (s, cam, prm), (w, h), f_shaded, f_shadow in cases:
    res = {}
    for flags in (8192, 0):
        ctx.set_debug_flags(flags)
        for rep in range(3):
            r = ctx.render_whitted(s, cam, prm, w, h)
        res[flags] = r; r["form"] = ctx.last_whitted_form(); r["k"] = ctx.last_whitted_kernel()
    ctx.set_debug_flags(0)
    a, b = res[8192], res[0]
    rel = np.abs(a["rgb"] - b["rgb"]) / np.maximum(np.abs(b["rgb"]), 1e-3)
    fl = lambda r: (r["rays"] * f_shaded + r["shadow_rays"] * f_shadow) / (r["kernel_ms"] * 1e-3) / 157.3e12
    print(f"{name:30s} forms {a['form']}/{b['form']} kernels {a['k']}/{b['k']} rays {a['rays']}+{a['shadow_rays']} equal {(a['rays'], a['shadow_rays']) == (b['rays'], b['shadow_rays'])} "
          f"max rel {rel.max():.1e} bytes differ {(a['bgra'] != b['bgra']).mean():.1e} | wavefront {a['kernel_ms']:.3f} ms ({fl(a):.3f} of peak) monolithic {b['kernel_ms']:.3f} ms ({fl(b):.3f})", flush=True)
