"""The pixels behind the coplanar-surfaces deviation (a glass box resting exactly on the floor plane): 1 spp, HIP vs oracle, by depth limit."""
import os, sys
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from rtamd import abi, lib, scenes
import orc as oracle
ctx = lib.Context()
W, H = 196, 140
s = scenes.Scene()
white = s.diffuse((0.8, 0.8, 0.8)); lm = s.light_material((9, 9, 9)); glass = s.dielectric(0.66)
s.plane(white, (0, -2, 0), (0, 1, 0))
s.box(glass, (-1.5, -1.2, -4), (1.2, 1.6, 1.2), (0, 0, 0))
s.rect(lm, (0, 6, -6), (6, 6), (90, 0, 0))
cam = scenes.camera(70.0, (0, 0.5, 1.0), (-3, 0, 0))
for depth in (1, 2, 3):
    prm = scenes.path_params(abi.RT_PATH_LIGHTS, 1, max_depth=depth, seed=3)
    g = ctx.render_path(s, cam, prm, W, H); o = oracle.render_path(s, cam, prm, W, H, threads=16)
    d = np.abs(g["rgb"] - o["sum"]).max(axis=-1)
    ys, xs = np.nonzero(d > 1e-4)
    print("depth", depth, "differing pixels", len(ys), "rays", g["rays"], o["rays"])
    for y, x in list(zip(ys, xs))[:12]:
        print("   px", x, y, "HIP", g["rgb"][y, x], "oracle", o["sum"][y, x])
