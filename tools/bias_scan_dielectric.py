"""Bias scan, part 4: dielectric rectangles and boxes, HIP vs oracle (equal seeds)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from rtamd import abi, lib, scenes
import orc as oracle
ctx = lib.Context()
W, H, SPP = 196, 140, 64
def run(tag, build, depth=6, variant=abi.RT_PATH_LIGHTS):
    s = scenes.Scene()
    white = s.diffuse((0.8, 0.8, 0.8)); lm = s.light_material((9, 9, 9)); glass = s.dielectric(0.66)
    s.plane(white, (0, -2, 0), (0, 1, 0))
    build(s, glass)
    s.rect(lm, (0, 6, -6), (6, 6), (90, 0, 0))
    cam = scenes.camera(70.0, (0, 0.5, 1.0), (-3, 0, 0))
    prm = scenes.path_params(variant, SPP, max_depth=depth, seed=3)
    g = ctx.render_path(s, cam, prm, W, H); o = oracle.render_path(s, cam, prm, W, H, threads=16)
    print(f"{tag:44s} queries HIP {g['rays']} oracle {o['rays']} rel {g['rays'] / o['rays'] - 1:+.1e}  mean rel {g['rgb'].mean() / (o['sum'] / SPP).mean() - 1:+.1e}", flush=True)
run("glass cube 1.2", lambda s, m: s.box(m, (-1.5, -1.2, -4), (1.2, 1.2, 1.2), (0, 0, 0)))
run("glass box 1.2 x 1.6 x 1.2 (hole in z faces)", lambda s, m: s.box(m, (-1.5, -1.2, -4), (1.2, 1.6, 1.2), (0, 0, 0)))
run("glass box 1.2 x 1.0 x 1.6 (z faces overhang)", lambda s, m: s.box(m, (-1.5, -1.2, -4), (1.2, 1.0, 1.6), (0, 0, 0)))
run("glass rectangle facing the camera", lambda s, m: s.rect(m, (-1.5, -1.2, -4), (2, 2), (0, 0, 0)))
run("glass rectangle tilted", lambda s, m: s.rect(m, (-1.5, -1.2, -4), (2, 2), (30, 20, 0)))
run("two parallel glass rectangles (a slab)", lambda s, m: (s.rect(m, (-1.5, -1.2, -4), (2, 2), (0, 0, 0)), s.rect(m, (-1.5, -1.2, -5), (2, 2), (0, 0, 0))))
run("glass cube, flipped normals", lambda s, m: s.box(m, (-1.5, -1.2, -4), (1.2, 1.2, 1.2), (0, 0, 0), flip=True))
run("glass sphere (control)", lambda s, m: s.sphere(m, (-1.5, -1.2, -4), 0.8))
run("glass triangle", lambda s, m: s.triangle(m, (-2.5, -2, -4), (-0.5, -2, -4.5), (-1.5, 0.5, -4.2)))
run("glass cube, 03 estimator", lambda s, m: s.box(m, (-1.5, -1.2, -4), (1.2, 1.2, 1.2), (0, 0, 0)), depth=12, variant=abi.RT_PATH_SKY)
