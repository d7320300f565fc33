"""Bias scan, part 2: the reduced stress scene taken apart (which primitives / materials make HIP and the oracle differ)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from rtamd import abi, lib, scenes
import orc as oracle
ctx = lib.Context()
if os.environ.get("FUZZ_ARITH") == "ref": ctx.set_option(abi.RT_OPT_ARITHMETIC, abi.RT_ARITH_REFERENCE)      # the reference's arithmetic for rays and spheres
W, H, SPP = 196, 140, 64
def cmp(tag, s, cam, variant):
    prm = scenes.path_params(variant, SPP, max_depth=8, seed=3)
    g = ctx.render_path(s, cam, prm, W, H)
    o = oracle.render_path(s, cam, prm, W, H, threads=16)
    n = W * H * SPP
    print(f"{tag:58s} HIP {g['rays'] / n:9.5f} oracle {o['rays'] / n:9.5f} rel {g['rays'] / o['rays'] - 1:+.1e}", flush=True)
def build(n_spheres, grid, sphere_mats="dmgd", with_plane=True, with_mesh=True, with_light=True, rmax=1.5):
    g = scenes._PCG32(0x5EED); s = scenes.Scene()
    for i in range(n_spheres):
        c = (g.u(-60, 60), g.u(-4, 40), g.u(-160, -10)); r = g.u(0.3, 1.5) * rmax / 1.5
        alb = (g.u(0.2, 0.9), g.u(0.2, 0.9), g.u(0.2, 0.9))
        k = sphere_mats[i % 4]
        m = s.metal(alb, 0.2) if k == "m" else s.dielectric(0.6) if k == "g" else s.diffuse(alb)
        s.sphere(m, c, r)
    if with_plane: s.plane(s.diffuse((0.5, 0.5, 0.5)), (0, -5, 0), (0, 1, 0))
    hf = s.diffuse((0.4, 0.6, 0.3))
    hs = [[-5.0 + 3.0 * g.u() for _ in range(grid + 1)] for _ in range(grid + 1)]
    x0, x1, z0, z1 = -60.0, 60.0, -160.0, -10.0
    if with_mesh:
        for j in range(grid):
            for i in range(grid):
                xa, xb = x0 + (x1 - x0) * i / grid, x0 + (x1 - x0) * (i + 1) / grid
                za, zb = z0 + (z1 - z0) * j / grid, z0 + (z1 - z0) * (j + 1) / grid
                p00, p10 = (xa, hs[j][i], za), (xb, hs[j][i + 1], za); p01, p11 = (xa, hs[j + 1][i], zb), (xb, hs[j + 1][i + 1], zb)
                s.triangle(hf, p00, p10, p11); s.triangle(hf, p00, p11, p01)
    if with_light: s.rect(s.light_material((15, 15, 15)), (0, 45, -85), (20, 20), (90, 0, 0))
    return s, scenes.camera(60.0, (0, 5, 20), (-10, 0, 0))
V3 = abi.RT_PATH_SKY
for tag, kw in (("all", {}), ("no mesh", dict(with_mesh=False)), ("no mesh, no plane", dict(with_mesh=False, with_plane=False)),
                ("mesh + plane only (no spheres)", dict(n=0)), ("mesh only", dict(n=0, with_plane=False)), ("plane only", dict(n=0, with_mesh=False)),
                ("spheres all diffuse, no mesh", dict(sphere_mats="dddd", with_mesh=False)), ("spheres all metal, no mesh", dict(sphere_mats="mmmm", with_mesh=False)),
                ("spheres all glass, no mesh", dict(sphere_mats="gggg", with_mesh=False)), ("diffuse spheres only", dict(sphere_mats="dddd", with_mesh=False, with_plane=False)),
                ("glass spheres only", dict(sphere_mats="gggg", with_mesh=False, with_plane=False)), ("metal spheres only", dict(sphere_mats="mmmm", with_mesh=False, with_plane=False))):
    kw = dict(kw); n = kw.pop("n", 600)
    s, cam = build(n, 12, **kw)
    cmp(f"03 estimator, {tag}", s, cam, V3)
