"""Bias scan: HIP vs the oracle at equal seeds on scene families beyond the four reference scenes - scene queries per sample and
mean colour, both variants.  Paths are identical except where a rounding difference flips a decision; such flips must not prefer
a direction (tools/big_sphere_check.py is the case that did)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from rtamd import abi, lib, scenes
import orc as oracle
from conftest import nest_scene
ctx = lib.Context()
W, H = 196, 140
def cmp(tag, s, cam, prm, spp):
    g = ctx.render_path(s, cam, prm, W, H)
    o = oracle.render_path(s, cam, prm, W, H, threads=16)
    n = W * H * spp
    om = (o["sum"] / spp).mean(axis=(0, 1)); gm = g["rgb"].mean(axis=(0, 1))
    print(f"{tag:46s} queries/sample HIP {g['rays'] / n:9.5f} oracle {o['rays'] / n:9.5f} rel {g['rays'] / o['rays'] - 1:+.1e};  mean rel {np.array2string(gm / np.maximum(om, 1e-12) - 1, precision=1)}", flush=True)
def box_scene():
    s = scenes.Scene(); w = s.diffuse((0.7, 0.7, 0.7)); r = s.diffuse((0.7, 0.2, 0.2)); m = s.metal((0.8, 0.8, 0.8), 0.1); gl = s.dielectric(0.6); lm = s.light_material((12, 12, 12))
    s.plane(w, (0, -2, 0), (0, 1, 0)); s.plane(w, (0, 0, -12), (0, 0, 1)); s.plane(r, (-6, 0, 0), (1, 0, 0)); s.plane(w, (6, 0, 0), (-1, 0, 0)); s.plane(w, (0, 6, 0), (0, -1, 0))
    s.box(m, (2, -1, -7), (2, 2, 2), (0, 30, 0)); s.box(gl, (-2, -1, -6), (1.5, 2, 1.5), (10, -20, 5)); s.box(w, (0, -1.5, -9), (3, 1, 2), (0, 0, 0))
    s.triangle(r, (-4, -2, -10), (-1, -2, -11), (-2.5, 2, -10.5)); s.rect(lm, (0, 5.9, -7), (4, 4), (90, 0, 0))
    return s, scenes.camera(70.0, (0, 1.0, 2.0), (-5, 0, 0))
for variant, name, spp in ((abi.RT_PATH_LIGHTS, "04 estimator", 256), (abi.RT_PATH_SKY, "03 estimator", 256)):
    s, cam = nest_scene(); cmp(f"nest scene, {name}", s, cam, scenes.path_params(variant, spp, seed=3), spp)
    s, cam = box_scene(); cmp(f"boxes/triangle/planes, {name}", s, cam, scenes.path_params(variant, spp, seed=3), spp)
    s, cam, _ = scenes.scene_stress(n_spheres=600, grid=12, spp=spp, seed=3); cmp(f"reduced stress scene, {name}", s, cam, scenes.path_params(variant, 64, max_depth=8, seed=3), 64)
s, cam, _ = scenes.scene_stress(spp=16, seed=1); cmp("full stress scene (18,194 prims), 04", s, cam, scenes.path_params(abi.RT_PATH_LIGHTS, 16, max_depth=8, seed=1), 16)
s, cam, _ = scenes.scene03(256, seed=5, readme_variant=True); cmp("03 README scene (hollow glass)", s, cam, scenes.path_params(abi.RT_PATH_SKY, 256, seed=5), 256)
s, cam, _ = scenes.scene04(256, seed=5); cmp("04 scene at tMin 0.001", s, cam, scenes.path_params(abi.RT_PATH_LIGHTS, 256, seed=5, t_min=0.001), 256)
