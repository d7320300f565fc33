"""Bias scan, part 3: the nest scene taken apart."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from rtamd import abi, lib, scenes
import orc as oracle
ctx = lib.Context()
W, H, SPP = 196, 140, 256
def cmp(tag, s, cam, variant):
    prm = scenes.path_params(variant, SPP, seed=3)
    g = ctx.render_path(s, cam, prm, W, H)
    o = oracle.render_path(s, cam, prm, W, H, threads=16)
    n = W * H * SPP
    print(f"{tag:64s} HIP {g['rays'] / n:9.5f} oracle {o['rays'] / n:9.5f} rel {g['rays'] / o['rays'] - 1:+.1e}", flush=True)
def nest(parts):
    s = scenes.Scene()
    white = s.diffuse((0.8, 0.8, 0.8)); red = s.diffuse((0.8, 0.1, 0.1)); green = s.diffuse((0.1, 0.8, 0.1))
    glass = s.dielectric(0.66); glass2 = s.dielectric(0.8); metal = s.metal((0.9, 0.9, 0.9), 0.05); lm = s.light_material((9, 9, 9))
    s.plane(white, (0, -2, 0), (0, 1, 0))
    if "a" in parts: s.sphere(glass, (0, 0, -6), 1.0); s.sphere(red, (0, 0, -6), 0.5)
    if "b" in parts: s.sphere(glass2, (0.3, 0, -6), 1.6)
    if "c" in parts: s.sphere(glass, (2.2, 0, -6), 1.2); s.sphere(glass2, (3.2, 0.2, -6), 1.2)
    if "d" in parts: s.sphere(red, (-3, 0, -7), 1.0); s.sphere(green, (-3, 0, -7), 1.0)
    if "e" in parts: s.rect(red, (0, 3, -7), (2, 2), (90, 0, 0)); s.rect(green, (0, 3, -7), (2, 2), (90, 0, 0))
    if "f" in parts: s.triangle(red, (-6, -2, -9), (-2, -2, -11), (-4, 3, -10)); s.triangle(metal, (-6, -2, -9), (-2, -2, -11), (-4, 3, -10))
    if "g" in parts: s.box(metal, (3, -1, -9), (2, 2, 2), (10, 30, 5))
    if "h" in parts: s.box(glass, (-1.5, -1.2, -4), (1.2, 1.6, 1.2), (0, 20, 0))
    if "i" in parts: s.rect(green, (3, -1, -8), (2, 2), (0, 0, 0))
    if "j" in parts: s.sphere(white, (0.3, 0, -6), 2.4, flip=True)
    s.rect(lm, (0, 6, -6), (6, 6), (90, 0, 0))
    return s, scenes.camera(70.0, (0, 0.5, 1.0), (-3, 0, 0))
for parts, tag in (("abcdefghij", "all"), ("", "plane + light only"), ("a", "glass sphere with a red core"), ("b", "one glass sphere"), ("ab", "core + intersecting glass"), ("c", "two intersecting glass spheres"),
                   ("d", "coincident diffuse spheres"), ("e", "coincident rectangles"), ("f", "coincident triangles"), ("g", "metal box"), ("h", "glass box"), ("gi", "box + coplanar rectangle"),
                   ("j", "inverted shell"), ("bj", "glass sphere inside the inverted shell")):
    s, cam = nest(parts)
    cmp(f"04 estimator: {tag}", s, cam, abi.RT_PATH_LIGHTS)
