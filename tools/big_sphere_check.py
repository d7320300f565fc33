"""One large sphere seen at a grazing angle (the situation of 03's ground): scene queries per sample, HIP kernels vs the oracle.
The hardware square root's bias showed here as +1.6e-3 (r = 100) and +1.4e-2 (r = 1000) before spheres of radius > 750 tMin got
the refined root (rt_device.h, PRIM_SPHERE_BIG)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from rtamd import abi, lib, scenes
import orc as oracle
ctx = lib.Context()
W, H, SPP = 196, 140, 1024
for r in (10.0, 100.0, 1000.0):
    s = scenes.Scene(); gmat = s.diffuse((0.8, 0.8, 0.0)); s.sphere(gmat, (0, -0.5 - r, -1.0), r)
    cam = scenes.camera(50.0, (-2.0, 1.5, 1.0), (-30.0, -42.0, 0.0))
    prm = scenes.path_params(abi.RT_PATH_SKY, SPP, seed=1)
    o = oracle.render_path(s, cam, prm, W, H, threads=16)
    n = W * H * SPP
    print(f"r = {r}: oracle {o['rays'] / n:.6f} rays/sample", flush=True)
    for name, fl in (("generic kernel", 8), ("specialised kernel", 16), ("BVH walk", -1)):
        if fl < 0: ctx.set_option(abi.RT_OPT_ACCEL, abi.RT_ACCEL_BVH)
        else: ctx.set_debug_flags(fl)
        g = ctx.render_path(s, cam, prm, W, H)
        ctx.set_debug_flags(0); ctx.set_option(abi.RT_OPT_ACCEL, abi.RT_ACCEL_NONE)
        print(f"   {name:24s} HIP {g['rays'] / n:.6f}  rel {g['rays'] / o['rays'] - 1:+.2e}", flush=True)
