"""Config 5 scene: how often HIP and the oracle trace the same path per pixel, by sphere count (ill-conditioned discriminants of far small spheres)."""
import sys, os
sys.path.insert(0,'software-raytracing_amd'); sys.path.insert(0,'oracle')
import numpy as np, torch
from rtamd import lib, scenes, abi
import orc as oracle
ctx = lib.Context()
for n_sph, grid, spp in ((600,12,2),(2500,32,2),(10000,64,1),(10000,64,4)):
    s, cam, prm = scenes.scene_stress(n_spheres=n_sph, grid=grid, spp=spp, seed=1)
    w,h = 96,54
    o = oracle.render_path_px(s, cam, prm, w, h, threads=16)
    acc = torch.zeros(h, w, 4, device="cuda")
    ctx.set_debug_flags(4)
    ctx.path_accumulate_device(s, cam, prm, w, h, acc.data_ptr()); st = ctx.sync()
    ctx.set_debug_flags(0)
    px = acc[:, :, 3].cpu().numpy().astype(np.int64)
    rgb = acc[:, :, :3].cpu().numpy()
    d = np.abs(rgb - o["sum"]) <= 1e-3*np.maximum(np.abs(o["sum"]),1e-3)
    print(n_sph, grid, spp, "px equal", (px==o["px_rays"]).mean(), "rays", st["rays"], o["rays"], "color within", d.all(axis=-1).mean(), "kernel", ctx.last_path_kernel()[0], flush=True)
